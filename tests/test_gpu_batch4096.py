"""BASELINE configs[4] as a whole at its size: the single-class head (C = 1, Dense 539 -> (7,7,11);
/root/reference/faces/trainFace.py:30-63) + decode / threshold / NMS (/root/reference/faces/testFace.py:51-110) on the tiny
backbone at batch 4096, through the C-ABI.

The oracle needs ~16 ms of CPU per image, so at this size the checks are the size-independent ones: determinism, three
images against the oracle, the first / last image against the same image run alone, detections bit-exact against the
oracle fed the GPU's own scores, and guard words behind every activation tensor.  This is also the first execution of
every kernel above 2^31 input elements (the input is 2.47 G floats, the first activation 13 GB): the images are drawn on
the device (torch's Philox generator, seed 0) because a host copy would be 9.9 GB; the three oracle images are copied back.
img_size stays 448 as configs[4] says (the reference's C = 1 scripts decode with 224: SURVEY Q15)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import model_ref
from yolo_amd import params as P
from yolo_amd.engine import DeviceModel

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GUARD = 1 << 16            # elements of sentinel behind (and in front of) every activation tensor
SENTINEL = 77.0


def _guarded(ws):
    """Re-home the workspace's activation tensors between two sentinel regions each."""
    bufs = []
    for i, a in enumerate(ws['acts']):
        shape, n = tuple(a.shape), a.numel()
        ws['acts'][i] = None
        del a
        buf = torch.full((n + 2 * GUARD,), SENTINEL, dtype=torch.float32, device=DEV)
        ws['acts'][i] = buf[GUARD:GUARD + n].view(shape)
        bufs.append(buf)
    return bufs


def test_single_class_network_at_batch_4096():
    N, C = 4096, 1
    prm = P.make_random_params(seed=1, C=C, dense_init='spread', bn='random')
    model = DeviceModel(prm, C=C, device=DEV)
    assert model.units == 539
    x = torch.rand((N, 448, 448, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    assert x.numel() > 2 ** 31
    ws = model.workspace(N)
    bufs = _guarded(ws)
    torch.cuda.empty_cache()

    det = [t.clone() for t in model.detect(x, want_y=True)]
    y = ws['y'].clone()
    torch.cuda.synchronize()
    # guard words: no kernel wrote outside its output tensor (ragged tiles, padding rows, 64-bit bases)
    for i, buf in enumerate(bufs):
        n = ws['acts'][i].numel()
        assert bool((buf[:GUARD] == SENTINEL).all()) and bool((buf[GUARD + n:] == SENTINEL).all()), f'layer {i} wrote outside its output'
    assert bool(torch.isfinite(y).all())

    # determinism: fixed-order reductions, a second run gives the same bits
    det2 = model.detect(x, want_y=True)
    assert torch.equal(ws['y'], y)
    assert all(torch.equal(a, b) for a, b in zip(det, det2))

    # three images of the batch against the oracle (fp32 bar of north_star: 1e-3 relative)
    pick = [0, 2049, N - 1]
    xs = torch.stack([x[i] for i in pick]).cpu()
    y_ref = model_ref.model_ref(xs.numpy(), prm, 7, 2, C)
    np.testing.assert_allclose(y[pick].cpu().numpy(), y_ref, rtol=1e-3, atol=1e-4)

    # the first and the last image against the same image run alone (other tile shapes and K splits: f32 summation order
    # differs, hence the fp32 bar instead of bits)
    for i in (0, N - 1):
        yi = model.predict(x[i:i + 1].contiguous())[0]
        torch.testing.assert_close(yi, y[i], rtol=1e-3, atol=1e-4)

    # detections at the full batch: bit-exact against the oracle fed the GPU's own scores
    want = oracle.decode_nms(y.cpu().numpy(), 2, C, 448)
    boxes, score, cls, keep = (t.cpu().numpy() for t in det)
    assert np.array_equal(keep, want['keep'])
    assert np.array_equal(cls, want['cls'])
    assert np.array_equal(score.view(np.uint32), want['score'].view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(boxes).view(np.uint64), np.ascontiguousarray(want['boxes']).view(np.uint64))
    assert keep.sum(1).min() >= 1                 # the top-confidence box of an image is always kept
