"""GPU parity tests: every libty1.so kernel, called through the C-ABI (yolo_amd.ops), against the
CPU oracle on the same seeded inputs and against the committed reference-generated goldens.

Bars: decode+NMS bit-exact (keep lists, float64 boxes, classes, scores); float kernels within the
tolerances written next to each assert (north_star: 1e-3 relative for fp32 conv/head outputs)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import model_ref, loss_ref
from conftest import load_golden
from yolo_amd import ops, params as P
from yolo_amd.engine import DeviceModel

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bits64(a):
    return np.ascontiguousarray(a).view(np.uint64)


def assert_detections_equal(got, want):
    boxes, score, cls, keep = (t.cpu().numpy() for t in got)
    assert np.array_equal(keep, want['keep'])
    assert np.array_equal(cls, want['cls'])
    assert np.array_equal(score.view(np.uint32), want['score'].view(np.uint32))
    assert np.array_equal(bits64(boxes), bits64(want['boxes']))


# ------------------------------------------------------------------------------- decode + NMS
@pytest.mark.parametrize('name,C,img', [('nms_c20.npz', 20, 448), ('nms_c1.npz', 1, 224)])
def test_decode_nms_reference_goldens(name, C, img):
    g = load_golden(name)
    pred = g['prediction']
    got = ops.decode_nms(dev(pred), 2, C, img)
    boxes, score, cls, keep = (t.cpu().numpy() for t in got)
    for i in range(pred.shape[0]):                       # what the reference drew, in raster order
        idx = np.nonzero(keep[i])[0]
        rects = np.array([[int(v) for v in boxes[i, s]] for s in idx], np.int64).reshape(-1, 4)
        lo, hi = g['rect_off'][i], g['rect_off'][i + 1]
        assert np.array_equal(rects, g['rects'][lo:hi]), f'case {i}'
        if C == 20:
            assert np.array_equal(cls[i][idx], g['labels'][lo:hi])
        else:
            assert [str(np.float32(v)) for v in score[i][idx]] == list(g['conf_str'][lo:hi])
    assert_detections_equal(got, oracle.decode_nms(pred, 2, C, img))


def _synthetic_pred(N, C, seed):
    rng = np.random.default_rng(seed)
    D = C + 10
    p = np.empty((N, 7, 7, D), np.float32)
    if C > 1:
        z = 4 * rng.random((N, 7, 7, C))
        e = np.exp(z - z.max(-1, keepdims=True))
        p[..., :C] = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    else:
        p[..., :C] = 1.0
    p[..., C:C + 2] = rng.random((N, 7, 7, 2)).astype(np.float32)
    p[..., C + 2:] = rng.random((N, 7, 7, 8)).astype(np.float32)
    return p


@pytest.mark.parametrize('N,C,img', [(1, 20, 448), (64, 20, 448), (1027, 20, 448), (4096, 1, 448)])
def test_decode_nms_bit_exact_vs_oracle(N, C, img):
    pred = _synthetic_pred(N, C, seed=2)
    if N >= 64:                                   # stress: small dense boxes, quantised confidences (ties)
        pred[N // 2:, ..., C + 2:] *= np.array([1, 1, .15, .15, 1, 1, .15, .15], np.float32)
        pred[N // 4:N // 2, ..., C:C + 2] = np.round(pred[N // 4:N // 2, ..., C:C + 2] * 8) / 8
    got = ops.decode_nms(dev(pred), 2, C, img)
    want = oracle.decode_nms(pred, 2, C, img)
    assert_detections_equal(got, want)
    if N == 4096:                                 # size-independent properties at the full config-5 size
        boxes, score, cls, keep = (t.cpu().numpy() for t in got)
        assert keep.sum(1).min() >= 1             # the top-confidence box is always kept
        assert not boxes[keep == 0].any()         # suppressed slots are zeroed
        again = ops.decode_nms(dev(pred), 2, C, img)
        assert all(torch.equal(a, b) for a, b in zip(got, again))      # deterministic


def test_decode_nms_edge_cases():
    z = ops.decode_nms(torch.zeros((0, 7, 7, 30), device=DEV), 2, 20, 448)
    assert z[3].shape == (0, 98)
    pred = np.zeros((3, 7, 7, 30), np.float32)
    pred[1] = _synthetic_pred(1, 20, 5)[0]
    pred[1, 3, 3, 20] = np.nan                    # NaN confidence: np.max is NaN, loop never starts
    pred[2] = _synthetic_pred(1, 20, 6)[0]
    pred[2, ..., 24:26] = 0                       # zero-area boxes: 0/0 IoU never suppresses
    assert_detections_equal(ops.decode_nms(dev(pred), 2, 20, 448), oracle.decode_nms(pred, 2, 20, 448))
    # other grid / box counts the ABI accepts
    for S, B, C in [(4, 1, 3), (8, 3, 5), (5, 4, 2)]:
        rng = np.random.default_rng(S)
        p = rng.random((9, S, S, C + 5 * B)).astype(np.float32)
        assert_detections_equal(ops.decode_nms(dev(p), B, C, 320), oracle.decode_nms(p, B, C, 320))
    with pytest.raises(Exception, match='one wavefront'):
        ops.decode_nms(torch.zeros((1, 9, 9, 30), device=DEV), 2, 20, 448)


# ------------------------------------------------------------------------------- conv blocks
def _conv_case(N, H, cin, cout, pool, seed, use_ws=False, alpha=0.1):
    rng = np.random.default_rng(seed)
    x = rng.random((N, H, H, cin)).astype(np.float32)
    w = P.glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)
    gamma = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    beta = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    mean = rng.uniform(-0.2, 0.2, cout).astype(np.float32)
    var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    ref = model_ref.conv_block_ref(xt, w, gamma, beta, mean, var, pool, alpha).permute(0, 2, 3, 1).contiguous().numpy()
    scale, shift = ops.fold_bn(gamma, beta, mean, var)
    ws = None
    if use_ws:      # the K-split tail path (csrc/conv_gemm.hip plan_tail); poison the scratch to catch unwritten partials
        nbytes = ops.conv3x3_workspace_bytes(N, H, H, cin, cout, pool)
        ws = torch.full((max(nbytes, 16) // 4,), float('nan'), dtype=torch.float32, device=DEV)
    # the output sits between two guard regions: no kernel may write a byte outside the tensor (ragged tiles, padding rows)
    oshape = (N, H // 2, H // 2, cout) if pool else (N, H, H, cout)
    numel, guard = int(np.prod(oshape)), 1 << 18
    buf = torch.full((numel + 2 * guard,), 77.0, dtype=torch.float32, device=DEV)
    out = buf[guard:guard + numel].view(oshape)
    got = ops.conv3x3_bn_leaky(dev(x), dev(P.hwio_to_ohwi(w)), dev(scale), dev(shift), pool, alpha=alpha, out=out, workspace=ws)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 77.0).all()) and bool((buf[guard + numel:] == 77.0).all()), 'write outside the output tensor'
    got = got.cpu().numpy()
    return got, ref


LAYERS = [(2, 448, 3, 16, True), (2, 224, 16, 32, True), (2, 112, 32, 64, True), (3, 56, 64, 128, True),
          (3, 28, 128, 256, True), (5, 14, 256, 512, True), (5, 7, 512, 1024, False), (5, 7, 1024, 256, False),
          # shapes outside the model: odd tiles / tails / no-pool variants of the early layers
          (1, 40, 3, 16, False), (1, 36, 3, 32, True), (1, 20, 16, 32, False), (2, 10, 32, 64, True),
          (1, 6, 64, 128, False), (70, 7, 64, 64, False),
          # the other tile configurations of the dispatcher: 256x32 tiles (one B row group: guarded DMA pieces), 64-byte
          # K chunks with an odd chunk count (Cin = 48), a Cout that is not a multiple of the tile (192 = 3 x 64, 96)
          (2, 12, 32, 32, True), (3, 9, 32, 32, False), (1, 9, 48, 96, False), (2, 10, 48, 64, True), (3, 8, 96, 192, True),
          (1, 11, 160, 96, False)]


# whole tiles + K-split tail (294 tiles on 256 CUs), and a deep-K 7x7 layer that is all tail
SPLIT_LAYERS = [(24, 28, 128, 256, True), (12, 7, 1024, 256, False)]


@pytest.mark.parametrize('use_ws', [False, True])
@pytest.mark.parametrize('N,H,cin,cout,pool', LAYERS + SPLIT_LAYERS)
def test_conv_block_vs_oracle(N, H, cin, cout, pool, use_ws):
    if use_ws and cin == 3:
        pytest.skip('first layer has no split path')
    got, ref = _conv_case(N, H, cin, cout, pool, seed=cin + cout, use_ws=use_ws)
    assert got.shape == ref.shape
    # fp32 bar of north_star: 1e-3 relative.  Measured error is ~1e-6 (exact-f32 MFMA, fp32 accumulate).
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-5)
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('alpha', [0.0, 0.5, 1.0, 1.5, -0.25])
@pytest.mark.parametrize('N,H,cin,cout,pool', [(1, 36, 3, 16, True), (1, 40, 3, 16, False), (1, 20, 16, 32, True),
                                               (2, 10, 32, 64, True), (1, 6, 64, 128, False)])
def test_conv_block_other_slopes(N, H, cin, cout, pool, alpha):
    """The model's LeakyReLU slope is 0.1; the kernels take short cuts that depend on it (pool before activation needs
    alpha >= 0, max(m, alpha*m) needs 0 < alpha <= 1) and must fall back to the plain order outside those ranges."""
    got, ref = _conv_case(N, H, cin, cout, pool, seed=7 + cin, alpha=alpha)
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-5)


# ------------------------------------------------------------------------------- dense + head
@pytest.mark.parametrize('N,C', [(1, 20), (5, 20), (64, 20), (33, 1)])
def test_dense_head_vs_oracle(N, C):
    rng = np.random.default_rng(N)
    units = 49 * (C + 10)
    feats = rng.standard_normal((N, P.FEATURES)).astype(np.float32)
    w = (0.02 * rng.standard_normal((P.FEATURES, units))).astype(np.float32)
    b = rng.standard_normal(units).astype(np.float32)
    z_ref = model_ref.dense_ref(feats, w, b)
    y_ref = model_ref.head_activate_ref(z_ref, 7, 2, C)
    wt = dev(P.dense_to_device_layout(w))
    part = ops.dense_splitk(dev(feats), wt, units)
    z = part.sum(0)[:, :units].cpu().numpy() + b
    np.testing.assert_allclose(z, z_ref, rtol=1e-3, atol=2e-4)
    y = ops.head_activate(part, dev(b), 7, 2, C).cpu().numpy()
    np.testing.assert_allclose(y, y_ref, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('name,C', [('head_c20.npz', 20), ('head_c1.npz', 1)])
def test_head_activate_vs_reference_goldens(name, C):
    """ty1_head_activate_f32 against the output of the reference's own ReshapeYOLO.call (tests/golden/make_golden.py).
    Bar: 1e-3 relative (north_star); measured error is a few ulp of expf, hence the tighter figures below."""
    g = load_golden(name)
    z, want = g['z'], g['y']
    n, units = z.shape
    ldz = ops.round_up(units, 4)
    part = np.zeros((1, n, ldz), np.float32)
    part[0, :, :units] = z
    y = ops.head_activate(dev(part), None, 7, 2, C).cpu().numpy()
    np.testing.assert_allclose(y, want, rtol=1e-5, atol=1e-7)
    # the same logits delivered as three split-K partials plus a bias
    rng = np.random.default_rng(7)
    b = rng.standard_normal(units).astype(np.float32)
    parts = np.zeros((3, n, ldz), np.float32)
    parts[0, :, :units] = rng.standard_normal((n, units)).astype(np.float32)
    parts[1, :, :units] = rng.standard_normal((n, units)).astype(np.float32)
    parts[2, :, :units] = z - b - parts[0, :, :units] - parts[1, :, :units]
    y3 = ops.head_activate(dev(parts), dev(b), 7, 2, C).cpu().numpy()
    moderate = np.abs(z).max(axis=1) < 20            # (re-associating +-100 logits loses more than the bar checks)
    np.testing.assert_allclose(y3[moderate], want[moderate], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('N,C,splits', [(1, 20, 3), (16, 20, 8), (64, 20, 5), (33, 1, 1)])
def test_fused_head_decode_nms_equals_the_two_launches(N, C, splits):
    """ty1_head_decode_nms_f32 (the inference tail in one launch) against ty1_head_activate_f32 + ty1_decode_nms: the
    same arithmetic, so every output - head tensor, boxes, scores, classes, keep - must agree bit for bit; and against
    the oracle on the same logits."""
    rng = np.random.default_rng(100 + N)
    units = 49 * (C + 10)
    ldz = ops.round_up(units, 4)
    part = np.zeros((splits, N, ldz), np.float32)
    part[:, :, :units] = (2.0 * rng.standard_normal((splits, N, units)) / np.sqrt(splits)).astype(np.float32)
    bias = rng.standard_normal(units).astype(np.float32)
    pd, bd = dev(part), dev(bias)
    y2 = ops.head_activate(pd, bd, 7, 2, C)
    want = ops.decode_nms(y2, 2, C, 448)
    y1 = torch.full_like(y2, float('nan'))
    got = ops.head_decode_nms(pd, bd, 7, 2, C, 448, 0.2, y=y1)
    assert torch.equal(y1, y2)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    got_noy = ops.head_decode_nms(pd, bd, 7, 2, C, 448, 0.2)               # head tensor kept in LDS only
    assert all(torch.equal(a, b) for a, b in zip(got_noy, want))
    assert_detections_equal(got, oracle.decode_nms(y2.cpu().numpy(), 2, C, 448))
    z = part.sum(0)[:, :units] + bias
    np.testing.assert_allclose(y1.cpu().numpy(), model_ref.head_activate_ref(z, 7, 2, C), rtol=1e-3, atol=1e-5)


# ------------------------------------------------------------------------------- whole model
@pytest.mark.parametrize('N,C,bn', [(2, 20, 'random'), (1, 1, 'identity')])
def test_model_end_to_end_vs_oracle(N, C, bn):
    prm = P.make_random_params(seed=1, C=C, dense_init='spread', bn=bn)
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(0)).numpy()
    model = DeviceModel(prm, C=C, device=DEV)
    xd = dev(x)
    feats = model.backbone(xd).cpu().numpy()
    feats_ref, acts_ref = model_ref.backbone_ref(x, prm, return_all=True)
    for a_got, a_ref in zip(model.workspace(N)['acts'], acts_ref):
        np.testing.assert_allclose(a_got.cpu().numpy(), a_ref, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(a_ref).max()))
    np.testing.assert_allclose(feats, feats_ref, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(feats_ref).max()))
    y = model.predict(xd).cpu().numpy()
    y_ref = model_ref.model_ref(x, prm, 7, 2, C)
    np.testing.assert_allclose(y, y_ref, rtol=1e-3, atol=1e-4)
    # detections: bit-exact given identical scores (the oracle is fed the GPU's own y)
    got = model.detect(xd)
    assert_detections_equal(got, oracle.decode_nms(y, 2, C, 448))


def test_model_full_batch_properties():
    """BASELINE configs[1] at its full size (batch 64, fp32), where the oracle would take minutes: size-independent
    properties.  The batch-64 launch shapes are the ones bench.py times (49 * 2^k tiles, K-split tails, fix-up), which
    the small cases above do not reach."""
    N = 64
    prm = P.make_random_params(seed=1, C=20, dense_init='spread', bn='random')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(0))
    model = DeviceModel(prm, C=20, device=DEV)
    xd = x.to(DEV)
    y = model.predict(xd).clone()
    # determinism: fixed-order reductions, a second run gives the same bits
    assert torch.equal(model.predict(xd), y)
    # three images of the batch against the oracle (fp32 bar of north_star: 1e-3 relative)
    pick = [0, 31, 63]
    y_ref = model_ref.model_ref(x[pick].numpy(), prm, 7, 2, 20)
    np.testing.assert_allclose(y[pick].cpu().numpy(), y_ref, rtol=1e-3, atol=1e-4)
    # batch independence: an image's result does not depend on its neighbours or its position (the K-split tail
    # changes the f32 summation order of some tiles with the batch size, hence a tolerance instead of bits)
    for i in (0, 17, 63):
        yi = model.predict(xd[i:i + 1].contiguous())[0]
        torch.testing.assert_close(yi, y[i], rtol=1e-3, atol=1e-4)     # measured: 3e-5 absolute
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
    yp = model.predict(xd[perm.to(DEV)].contiguous())
    torch.testing.assert_close(yp, y[perm.to(DEV)], rtol=1e-3, atol=1e-4)
    # detections at full batch: bit-exact against the oracle fed the GPU's own scores
    y_np = y.cpu().numpy()
    assert_detections_equal(model.detect(xd), oracle.decode_nms(y_np, 2, 20, 448))


# ------------------------------------------------------------------------------- loss
@pytest.mark.parametrize('name,C', [('loss_c20.npz', 20), ('loss_c1.npz', 1)])
def test_yolo_loss_vs_reference_goldens(name, C):
    g = load_golden(name)
    img = int(g['img_size'])
    loss, per, grad = ops.yolo_loss_fwd_bwd(dev(g['y_true']), dev(g['y_pred']), 2, C, img)
    assert abs(loss.item() - g['loss']) <= 1e-5 * abs(g['loss'])
    np.testing.assert_allclose(per.cpu().numpy(), g['loss_per_image'], rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g['grad'], rtol=1e-4, atol=1e-6)
    # the two adversarial cells of make_loss_inputs: NaN IoU -> no responsible box; zero-width box -> NaN dL/dw (autodiff)
    assert np.array_equal(np.isnan(grad.cpu().numpy()), np.isnan(g['grad'])) and np.isnan(g['grad']).sum() == 3
    boxes = dev(g['y_pred'][..., C + 2:]).reshape(-1, 7, 7, 2, 4).contiguous()
    xy, wh = ops.yolo_head(boxes, img)
    assert np.array_equal(xy.cpu().numpy(), g['head_xy']) and np.array_equal(wh.cpu().numpy(), g['head_wh'])
    mins, maxes = xy - wh / 2, xy + wh / 2
    ious = ops.iou(mins[:, :, :, 0], maxes[:, :, :, 0], mins[:, :, :, 1], maxes[:, :, :, 1])
    np.testing.assert_allclose(ious.cpu().numpy(), g['iou_b0_b1'], rtol=1e-6, atol=0)


def test_yolo_loss_batch_vs_oracle():
    rng = np.random.default_rng(3)
    N, C = 32, 20
    y_true = np.zeros((N, 7, 7, 25), np.float32)
    for n in range(N):
        for _ in range(3):
            r, c, k = rng.integers(0, 7), rng.integers(0, 7), rng.integers(0, C)
            if y_true[n, r, c, 24] == 0:
                y_true[n, r, c, k] = 1
                y_true[n, r, c, 20:24] = rng.random(4)
                y_true[n, r, c, 24] = 1
    z = rng.standard_normal((N, 1470)).astype(np.float32)
    y_pred = model_ref.head_activate_ref(z)
    loss_ref_, per_ref, grad_ref = loss_ref.yolo_loss_and_grad_ref(y_true, y_pred)
    loss, per, grad = ops.yolo_loss_fwd_bwd(dev(y_true), dev(y_pred))
    assert abs(loss.item() - loss_ref_) <= 1e-5 * abs(loss_ref_)
    np.testing.assert_allclose(grad.cpu().numpy(), grad_ref, rtol=1e-4, atol=1e-6)
    # batch SUM: the loss of the batch equals the sum of its halves
    la, _, _ = ops.yolo_loss_fwd_bwd(dev(y_true[:16]), dev(y_pred[:16]))
    lb, _, _ = ops.yolo_loss_fwd_bwd(dev(y_true[16:]), dev(y_pred[16:]))
    assert abs(la.item() + lb.item() - loss.item()) <= 1e-5 * abs(loss.item())


@pytest.mark.parametrize('N,C,golden', [(32, 20, None), (5, 1, None), (0, 20, 'loss_c20.npz'), (0, 1, 'loss_c1.npz')])
def test_train_tail_equals_the_three_launches(N, C, golden):
    """ty1_train_tail_f32 (head activation + yolo_loss fwd/bwd + head Jacobian in one launch, y and dL/dy in LDS) against
    ty1_head_activate_f32 -> ty1_yolo_loss_fwd_bwd -> ty1_head_activate_bwd_f32: the same device functions in the same
    order, so loss, per-image loss, dz and the optional head tensor must agree bit for bit - including the NaN cells of
    the reference-run goldens (their labels; logits chosen so that the adversarial cells keep their zero-size boxes)."""
    rng = np.random.default_rng(11 + C)
    units, D = 49 * (C + 10), C + 10
    if golden:
        g = load_golden(golden)
        y_true = g['y_true'].astype(np.float32)
        N = y_true.shape[0]
    else:
        y_true = np.zeros((N, 7, 7, C + 5), np.float32)
        for n in range(N):
            for _ in range(4):
                r, c, k = rng.integers(0, 7), rng.integers(0, 7), rng.integers(0, C)
                if y_true[n, r, c, C + 4] == 0:
                    y_true[n, r, c, k] = 1
                    y_true[n, r, c, C:C + 4] = rng.random(4)
                    y_true[n, r, c, C + 4] = 1
    splits, ldz = 6, ops.round_up(units, 4)
    part = (3.0 * rng.standard_normal((splits, N, ldz))).astype(np.float32)
    part[:, :, units:] = np.nan                                  # padding columns must never be read
    if golden:                                                   # box logits far below the sigmoid's underflow: width exactly 0
        part[:, 0, 49 * C + 98 + 2] = -200.0
    bias = rng.standard_normal(units).astype(np.float32)
    pd, bd, td = dev(part), dev(bias), dev(y_true)
    y = ops.head_activate(pd, bd, 7, 2, C)
    loss, per, grad = ops.yolo_loss_fwd_bwd(td, y, 2, C, 448)
    dz = ops.head_activate_bwd(y, grad, ldz, 2, C)
    y2 = torch.empty_like(y)
    loss2, per2, dz2 = ops.train_tail(pd, bd, td, ldz, 7, 2, C, 448, y=y2)
    loss3, per3, dz3 = ops.train_tail(pd, bd, td, ldz, 7, 2, C, 448)          # without the head tensor
    same = lambda a, b: np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
    assert same(y, y2) and same(per, per2) and same(loss, loss2) and same(dz, dz2)
    assert same(per, per3) and same(loss, loss3) and same(dz, dz3)
    if golden:
        assert torch.isnan(dz).any()                             # the zero-width cell did produce its NaN


# ------------------------------------------------------------------------------- training kernels
def test_head_backward_wgrad_adam_vs_torch():
    rng = np.random.default_rng(4)
    N, C, units = 32, 20, 1470
    feats = rng.standard_normal((N, P.FEATURES)).astype(np.float32)
    z = rng.standard_normal((N, units)).astype(np.float32)
    gy = rng.standard_normal((N, 7, 7, 30)).astype(np.float32)
    # torch fp32 reference of the head Jacobian and dW = x^T dz
    zt = torch.from_numpy(z).requires_grad_(True)
    i1, i2 = 49 * C, 49 * C + 98
    y_t = torch.cat([torch.softmax(zt[:, :i1].reshape(N, 7, 7, C), -1), torch.sigmoid(zt[:, i1:i2].reshape(N, 7, 7, 2)),
                     torch.sigmoid(zt[:, i2:].reshape(N, 7, 7, 8))], -1)
    (y_t * torch.from_numpy(gy)).sum().backward()
    dz_ref = zt.grad.numpy()
    ldz = ops.round_up(units, 4)
    dz = ops.head_activate_bwd(dev(y_t.detach().numpy()), dev(gy), ldz)
    np.testing.assert_allclose(dz.cpu().numpy()[:, :units], dz_ref, rtol=1e-4, atol=1e-6)
    rows = ops.round_up(units, 32)
    dwt = torch.zeros((rows, P.FEATURES), device=DEV)
    db = torch.zeros((units,), device=DEV)
    ops.dense_wgrad(dz, dev(feats), units, dwt, db)
    dw_ref = (torch.from_numpy(dz_ref).t().double() @ torch.from_numpy(feats).double()).numpy()
    np.testing.assert_allclose(dwt.cpu().numpy()[:units], dw_ref, rtol=1e-3, atol=1e-4)
    assert not dwt[units:].any()
    np.testing.assert_allclose(db.cpu().numpy(), dz_ref.sum(0), rtol=1e-4, atol=1e-5)
    # Adam, Keras semantics, three steps against a float64 restatement
    n = 100003
    p = rng.standard_normal(n).astype(np.float32)
    pd, md, vd = dev(p.copy()[:100000]), torch.zeros(100000, device=DEV), torch.zeros(100000, device=DEV)
    pr, mr, vr = p[:100000].astype(np.float64), np.zeros(100000), np.zeros(100000)
    for t in range(1, 4):
        g = rng.standard_normal(100000).astype(np.float32)
        ops.adam_step(pd, dev(g), md, vd, t)
        mr = 0.9 * mr + 0.1 * g
        vr = 0.999 * vr + 0.001 * g.astype(np.float64) ** 2
        pr = pr - 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * mr / (np.sqrt(vr) + 1e-7)
    np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=1e-5, atol=1e-6)
