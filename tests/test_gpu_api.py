"""GPU tests of the drop-in Python entry points (the reference's module paths and names), which
sit on top of the C-ABI: Model.predict, TinyYOLOv1.detect, yolo_eval / yolo_loss / yolo_head / iou,
the NumPy twins, and the training step (single process and sharded + summed)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import model_ref, loss_ref
from conftest import load_golden
from yolo_amd import ops, params as P, parallel
from yolo_amd.engine import DeviceModel, Trainer
from yolo_amd.keras_facade import Model, Adam
from yolo_amd.models.model_tiny_yolov1 import model_tiny_YOLOv1, ReshapeYOLO
from yolo_amd.yolo.yolo import yolo_eval, yolo_loss, yolo_head, iou
from yolo_amd import tiny_yolov1 as ty

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _labels(N, C, seed):
    rng = np.random.default_rng(seed)
    y = np.zeros((N, 7, 7, C + 5), np.float32)
    for n in range(N):
        for _ in range(3):
            r, c, k = rng.integers(0, 7), rng.integers(0, 7), rng.integers(0, C)
            if y[n, r, c, C + 4] == 0:
                y[n, r, c, k] = 1
                y[n, r, c, C:C + 4] = rng.random(4)
                y[n, r, c, C + 4] = 1
    return y


def test_model_predict_and_detect_dropin():
    outputs, inputs = model_tiny_YOLOv1()
    model = Model(inputs=inputs, outputs=outputs)
    prm = P.make_random_params(seed=2, dense_init='spread', bn='random')
    model.set_params(prm)
    x = torch.rand((3, 448, 448, 3), generator=torch.Generator().manual_seed(1)).numpy().astype(np.float64)  # load_img dtype
    y = model.predict(x, batch_size=2)                     # 2 + 1: exercises batching
    assert isinstance(y, np.ndarray) and y.shape == (3, 7, 7, 30) and y.dtype == np.float32
    y_ref = model_ref.model_ref(x.astype(np.float32), prm)
    np.testing.assert_allclose(y, y_ref, rtol=1e-3, atol=1e-4)
    det = ty.TinyYOLOv1(None, None)
    det.model.set_params(prm)
    res = det.detect(x[0].astype(np.float32))
    # bit-exact keep-lists "given identical scores": the scores must come from a forward pass of the same batch
    # size as detect() uses (the conv kernels pick a K-split per problem size, which changes the f32 summation order)
    y1 = det.model.predict(x[:1].astype(np.float32), batch_size=1)
    np.testing.assert_allclose(y1, y[:1], rtol=1e-4, atol=1e-6)
    want = oracle.decode_nms(y1, 2, 20, 448)
    idx = np.nonzero(want['keep'][0])[0]
    assert np.array_equal(res[0]['slots'], idx)
    assert np.array_equal(res[0]['boxes'].view(np.uint64), want['boxes'][0, idx].view(np.uint64))
    assert res[0]['names'] == [ty.classes_name[c] for c in want['cls'][0, idx]]
    assert np.array_equal(det.predict(x[0].astype(np.float32)), y1)


def test_reshape_yolo_call_and_yolo_eval():
    rng = np.random.default_rng(0)
    z = rng.standard_normal((4, 1470)).astype(np.float32)
    y = ReshapeYOLO((7, 7, 30)).call(torch.from_numpy(z).to(DEV))
    np.testing.assert_allclose(y.cpu().numpy(), model_ref.head_activate_ref(z), rtol=1e-5, atol=1e-6)
    z1 = rng.standard_normal((4, 539)).astype(np.float32)
    y1 = ReshapeYOLO((7, 7, 11), C=1).call(torch.from_numpy(z1).to(DEV))
    np.testing.assert_allclose(y1.cpu().numpy(), model_ref.head_activate_ref(z1, 7, 2, 1), rtol=1e-5, atol=1e-6)
    g = load_golden('nms_c20.npz')
    boxes, score, cls, keep = yolo_eval(g['prediction'])
    want = oracle.decode_nms(g['prediction'])
    assert np.array_equal(keep.cpu().numpy(), want['keep'])
    assert np.array_equal(boxes.cpu().numpy().view(np.uint64), want['boxes'].view(np.uint64))


def test_numpy_twins_bit_exact():
    g = load_golden('head_np.npz')
    xy, wh = ty.YOLO_head(g['feats'])
    assert xy.dtype == np.float64 and wh.dtype == np.float32
    assert np.array_equal(xy, g['xy']) and np.array_equal(wh, g['wh'])
    xy2, wh2 = ty.YOLO_head(g['feats'], img_size=224)
    assert np.array_equal(xy2, g['xy224']) and np.array_equal(wh2, g['wh224'])
    got = ty.iou(g['a_min'], g['a_max'], g['b_min'], g['b_max'])
    assert np.array_equal(got, g['iou'])


def test_yolo_loss_autograd_dropin():
    g = load_golden('loss_c20.npz')
    yp = torch.from_numpy(g['y_pred']).to(DEV).requires_grad_(True)
    loss = yolo_loss(torch.from_numpy(g['y_true']).to(DEV), yp)
    assert loss.dim() == 0
    (2.0 * loss).backward()
    assert abs(loss.item() - g['loss']) <= 1e-5 * abs(g['loss'])
    np.testing.assert_allclose(yp.grad.cpu().numpy(), 2.0 * g['grad'], rtol=1e-4, atol=1e-6)
    g1 = load_golden('loss_c1.npz')
    l1 = yolo_loss(g1['y_true'], g1['y_pred'], img_size=224)
    assert abs(l1.item() - g1['loss']) <= 1e-5 * abs(g1['loss'])
    xy, wh = yolo_head(g['y_pred'][..., 22:].reshape(-1, 7, 7, 2, 4))
    assert np.array_equal(xy.cpu().numpy(), g['head_xy'])
    mins, maxes = xy - wh / 2, xy + wh / 2
    v = iou(mins[:, :, :, :1], maxes[:, :, :, :1], mins, maxes)          # broadcasting form
    assert v.shape == (6, 7, 7, 2)
    np.testing.assert_allclose(v[..., 1].cpu().numpy(), g['iou_b0_b1'], rtol=1e-6)


def _torch_train_reference(prm, feats, y_true, steps, C=20):
    """fp32 torch restatement of the trainable part: Dense -> head -> yolo_loss -> Adam (Keras form)."""
    w = torch.from_numpy(prm['dense_w'].copy()).requires_grad_(True)
    b = torch.from_numpy(prm['dense_b'].copy()).requires_grad_(True)
    m = [torch.zeros_like(w), torch.zeros_like(b)]
    v = [torch.zeros_like(w), torch.zeros_like(b)]
    losses = []
    f, yt = torch.from_numpy(feats), torch.from_numpy(y_true)
    n = f.shape[0]
    for t in range(1, steps + 1):
        z = f @ w + b
        y = torch.cat([torch.softmax(z[:, :49 * C].reshape(n, 7, 7, C), -1),
                       torch.sigmoid(z[:, 49 * C:49 * C + 98].reshape(n, 7, 7, 2)),
                       torch.sigmoid(z[:, 49 * C + 98:].reshape(n, 7, 7, 8))], -1)
        loss = loss_ref.yolo_loss_ref(yt, y, C)
        gw, gb = torch.autograd.grad(loss, [w, b])
        losses.append(loss.item())
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        with torch.no_grad():
            for p, g, mm, vv in ((w, gw, m[0], v[0]), (b, gb, m[1], v[1])):
                mm.mul_(0.9).add_(g, alpha=0.1)
                vv.mul_(0.999).addcmul_(g, g, value=0.001)
                p.sub_(lr_t * mm / (vv.sqrt() + 1e-7))
    return losses, w.detach().numpy(), b.detach().numpy()


def test_training_step_vs_torch_reference_and_sharded_sum():
    N = 8
    prm = P.make_random_params(seed=5, dense_init='spread', bn='random')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(3))
    y_true = _labels(N, 20, 9)
    model = DeviceModel(prm, device=DEV)
    xd, yd = x.to(DEV), torch.from_numpy(y_true).to(DEV)
    feats = model.backbone(xd).cpu().numpy().copy()
    tr = Trainer(model)
    # (a) sharded gradients summed (what the RCCL all-reduce does) == full-batch gradients.  The shards reuse the
    #     full batch's features: the conv kernels choose a K-split per problem size, so a separately run shard
    #     differs in the last f32 bits, and yolo_loss' responsible-box mask (iou >= max, yolo/yolo.py:79-82) is
    #     discontinuous in them; the sum property is a statement about the trainable part of the step.
    fd = torch.from_numpy(feats).to(DEV)
    tr.gradients_from_features(fd, yd)
    full_dw, full_db = tr.dwt.clone(), tr.db.clone()
    la = tr.gradients_from_features(fd[:5].contiguous(), yd[:5].contiguous()).clone()
    dw_a, db_a = tr.dwt.clone(), tr.db.clone()
    lb = tr.gradients_from_features(fd[5:].contiguous(), yd[5:].contiguous()).clone()
    torch.testing.assert_close(dw_a + tr.dwt, full_dw, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(db_a + tr.db, full_db, rtol=1e-4, atol=1e-6)
    # ... and the full path on separately computed shards agrees to forward-pass rounding
    l5 = tr.gradients(xd[:5].contiguous(), yd[:5].contiguous())
    assert abs(l5.item() - la.item()) <= 1e-4 * abs(la.item())
    # (b) three optimizer steps against the torch restatement
    losses = [tr.step(xd, yd).item() for _ in range(3)]
    ref_losses, w_ref, b_ref = _torch_train_reference(prm, feats, y_true, 3)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-4)
    assert abs(la.item() + lb.item() - ref_losses[0]) <= 2e-4 * ref_losses[0]
    w_got = P.dense_from_device_layout(model.dense_wt.cpu().numpy(), model.units)
    # Adam's step is lr * m / sqrt(v): for gradients at rounding-noise level it is a sign function of
    # that noise, so a handful of the 18.4 M weights may differ by up to one lr per step; everything
    # else must agree to 2e-5.
    diff = np.abs(w_got - w_ref)
    assert (diff > 2e-5 + 1e-3 * np.abs(w_ref)).mean() < 1e-5 and diff.max() <= 3 * 1.1e-3
    np.testing.assert_allclose(model.dense_b.cpu().numpy(), b_ref, rtol=1e-3, atol=2e-5)
    assert losses[2] < losses[0]                                     # it learns


def test_training_step_at_the_configured_batch_of_32():
    """BASELINE configs[3] at its size (/root/reference/train.py:21,59-60: 32 images per GPU), where the launch shapes
    differ from the small cases above (other conv tile plans, 21 Dense splits on 32 x 128 tiles, train_tail_kernel with
    32 workgroups, the wgrad reduction over 32 rows).  Size-independent properties: two overlapped steps are
    deterministic (loss and weights, bit for bit, against a second trainer built from the same parameters), dz of three
    images against the fp32 torch restatement of Dense -> head -> yolo_loss, and the shard sum 20 + 12 = full batch."""
    N, C = 32, 20
    prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(31)).to(DEV)
    y_true = _labels(N, C, 32)
    yd = torch.from_numpy(y_true).to(DEV)

    def two_steps():
        model = DeviceModel(prm, device=DEV)
        tr = Trainer(model, lr=1e-5, overlap=True)
        losses = [tr.step(x, yd, wait=False) for _ in range(2)]
        tr.join()
        torch.cuda.synchronize()
        return model, tr, torch.cat(losses).cpu(), tr.pflat.clone().cpu(), tr.m.clone().cpu()

    model, tr, losses, w, m = two_steps()
    _, _, losses2, w2, m2 = two_steps()
    assert torch.isfinite(losses).all() and torch.equal(losses, losses2) and torch.equal(w, w2) and torch.equal(m, m2)
    assert losses[1] < losses[0]                                       # it learns

    # dz of three images: GPU features (checked against the oracle elsewhere) -> torch fp32 Dense + head + yolo_loss
    feats = model.backbone(x)
    tr.gradients_from_features(feats, yd)
    dz = tr._dz[N].cpu().numpy()[:, :model.units]
    pick = [0, 13, 31]
    wt = P.dense_from_device_layout(model.dense_wt.cpu().numpy(), model.units)
    z = (torch.from_numpy(feats[pick].cpu().numpy()) @ torch.from_numpy(wt) + model.dense_b.cpu()).requires_grad_(True)
    n = len(pick)
    y = torch.cat([torch.softmax(z[:, :49 * C].reshape(n, 7, 7, C), -1), torch.sigmoid(z[:, 49 * C:49 * C + 98].reshape(n, 7, 7, 2)),
                   torch.sigmoid(z[:, 49 * C + 98:].reshape(n, 7, 7, 8))], -1)
    loss_ref.yolo_loss_ref(torch.from_numpy(y_true[pick]), y, C).backward()
    dz_ref = z.grad.numpy()
    # (fp32 bar: the logits themselves carry ~1e-5 relative rounding from the 12544-long dot products)
    np.testing.assert_allclose(dz[pick], dz_ref, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(dz_ref).max()))

    # shard sum (what the SUM all-reduce does): 20 + 12 images on the full batch's features
    full_dw, full_db, full_loss = tr.dwt.clone(), tr.db.clone(), tr.bucket.loss.clone()
    la = tr.gradients_from_features(feats[:20].contiguous(), yd[:20].contiguous()).clone()
    dw_a, db_a = tr.dwt.clone(), tr.db.clone()
    lb = tr.gradients_from_features(feats[20:].contiguous(), yd[20:].contiguous()).clone()
    torch.testing.assert_close(dw_a + tr.dwt, full_dw, rtol=1e-4, atol=1e-6 * max(1.0, full_dw.abs().max().item()))
    torch.testing.assert_close(db_a + tr.db, full_db, rtol=1e-4, atol=1e-6 * max(1.0, full_db.abs().max().item()))
    assert abs(la.item() + lb.item() - full_loss.item()) <= 1e-5 * abs(full_loss.item())


def test_reserved_cus_change_grids_not_results():
    """ty1_set_reserved_cus(k): every persistent grid is sized for (CUs - k) compute units - room for a collective's
    workgroups next to them - while work items, K-split plans and summation orders stay those of the full chip, so every
    output is bit-identical for every k: single layers of each kernel family (fp32 and bf16, with K-split tails), the
    whole detect() in both arithmetics, and three overlapped training steps with Trainer.reserve_cus."""
    from test_gpu_parity import dev
    rng = np.random.default_rng(21)
    assert ops.get_reserved_cus() == 0

    def layer(N, H, cin, cout, pool, bf16):
        x = dev(rng.random((N, H, H, cin)).astype(np.float32))
        w = dev(P.hwio_to_ohwi(P.glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)))
        sc, sh = dev(rng.uniform(0.5, 1.5, cout).astype(np.float32)), dev(rng.uniform(-0.5, 0.5, cout).astype(np.float32))
        if bf16:
            need = ops.conv3x3_workspace_bytes_bf16(N, H, H, cin, cout, pool)
            if cin != 3:
                x, w = ops.cast_bf16(x), ops.cast_bf16(w)
            run = lambda: ops.conv3x3_bn_leaky_bf16(x, w, sc, sh, pool, workspace=torch.empty(max(need, 16), dtype=torch.uint8, device=DEV))
        else:
            need = ops.conv3x3_workspace_bytes(N, H, H, cin, cout, pool)
            run = lambda: ops.conv3x3_bn_leaky(x, w, sc, sh, pool, workspace=torch.empty(max(need, 16), dtype=torch.uint8, device=DEV))
        return run

    try:
        cases = [(4, 448, 3, 16, True, False), (4, 224, 16, 32, True, False), (24, 28, 128, 256, True, False), (12, 7, 1024, 256, False, False),
                 (8, 112, 32, 64, True, False), (4, 448, 3, 16, True, True), (4, 224, 16, 32, True, True), (44, 112, 32, 64, True, True),
                 (80, 56, 64, 128, True, True), (64, 28, 128, 256, True, True), (128, 7, 1024, 256, False, True), (24, 28, 128, 256, True, True)]
        for case in cases:
            run = layer(*case)
            ops.set_reserved_cus(0)
            want = run()
            for k in (1, 37, 250, 100000):
                ops.set_reserved_cus(k)
                assert ops.get_reserved_cus() == k
                assert torch.equal(run(), want), (case, k)
        prm = P.make_random_params(seed=1, dense_init='spread', bn='random')
        x = torch.rand((16, 448, 448, 3), generator=torch.Generator().manual_seed(5)).to(DEV)
        for dtype in ('f32', 'bf16'):
            model = DeviceModel(prm, device=DEV, dtype=dtype)
            ops.set_reserved_cus(0)
            want = [t.clone() for t in model.detect(x)]
            ops.set_reserved_cus(48)
            assert all(torch.equal(a, b) for a, b in zip(model.detect(x), want)), dtype
        ops.set_reserved_cus(0)
        # the trainer's use: the next batch's backbone runs on a shrunk grid while the update is in flight
        prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
        xs = [torch.rand((8, 448, 448, 3), generator=torch.Generator().manual_seed(40 + i)).to(DEV) for i in range(3)]
        ys = [torch.from_numpy(_labels(8, 20, 50 + i)).to(DEV) for i in range(3)]
        res = {}
        for k in (0, 32):
            model = DeviceModel(prm, device=DEV)
            tr = Trainer(model, lr=1e-5, overlap=True)
            tr.reserve_cus = k
            losses = [tr.step(a, b, wait=False) for a, b in zip(xs, ys)]
            tr.join()
            torch.cuda.synchronize()
            assert ops.get_reserved_cus() == 0            # the trainer restores the full chip after each backbone
            res[k] = (torch.cat(losses).cpu(), tr.pflat.clone().cpu())
        assert torch.equal(res[0][0], res[32][0]) and torch.equal(res[0][1], res[32][1])
    finally:
        ops.set_reserved_cus(0)
    with pytest.raises(Exception, match='must be >= 0'):
        ops.set_reserved_cus(-1)


def test_tail_on_the_side_stream_same_bits_and_buffers_refilled_in_place():
    """step(wait=False) with the whole trainable tail (Dense forward, loss, gradients, update) in flight on the side stream
    under the next batch's backbone against the same steps with the tail on the compute stream (Trainer.tail_on_side =
    False) and against the serial trainer: same losses, weights and moments, bit for bit, over 4 steps at 8 images -
    also when the caller refills ONE image buffer and ONE label buffer in place as soon as step() has returned (the tail
    reads a private copy of the labels; the feature buffer is handed back through an event inside the next backbone)."""
    N = 8
    prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
    xs = [torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(60 + i)).to(DEV) for i in range(4)]
    ys = [torch.from_numpy(_labels(N, 20, 70 + i)).to(DEV) for i in range(4)]

    def run(mode):
        model = DeviceModel(prm, device=DEV)
        tr = Trainer(model, lr=1e-5, overlap=(mode != 'serial'))
        tr.tail_on_side = mode in ('side', 'side, in place')
        losses = []
        if mode == 'side, in place':
            xb, yb = torch.empty_like(xs[0]), torch.empty_like(ys[0])
            for x, y in zip(xs, ys):
                xb.copy_(x)
                yb.copy_(y)
                losses.append(tr.step(xb, yb, wait=False))
                xb.fill_(0.5)                 # the caller's buffers are its own again
                yb.zero_()
        else:
            for x, y in zip(xs, ys):
                losses.append(tr.step(x, y, wait=(mode == 'serial')))
        tr.join()
        torch.cuda.synchronize()
        return torch.cat(losses).cpu(), tr.pflat.clone().cpu(), tr.m.clone().cpu(), tr.v.clone().cpu()

    want = run('serial')
    assert torch.isfinite(want[0]).all() and want[0][3] < want[0][0]
    for mode in ('compute', 'side', 'side, in place'):
        got = run(mode)
        for a, b in zip(got, want):
            assert torch.equal(a, b), mode


def test_steps_frozen_backbone_once_over_consecutive_batches():
    """Trainer.steps(x, y, batch): the frozen backbone once over k = 3 consecutive batches of 8 images, then 3 optimisation
    steps in order.  (a) the overlapped form (everything behind the backbone on the side stream, twice in a row with
    wait=False, buffers refilled in place) gives the bits of the serial trainer's steps(); (b) against 6 separate
    step() calls - whose backbones run at 8 images, with other K-split plans - the losses agree to 1e-5 relative and the
    weights to the bar of test_training_step_vs_torch_reference (Adam's step is a sign function at rounding level)."""
    b, k = 8, 3
    prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
    xs = [torch.rand((k * b, 448, 448, 3), generator=torch.Generator().manual_seed(80 + i)).to(DEV) for i in range(2)]
    ys = [torch.from_numpy(_labels(k * b, 20, 90 + i)).to(DEV) for i in range(2)]

    def run(mode):
        model = DeviceModel(prm, device=DEV)
        tr = Trainer(model, lr=1e-5, overlap=(mode == 'overlap'))
        losses = []
        if mode == 'overlap':
            xb, yb = torch.empty_like(xs[0]), torch.empty_like(ys[0])
            for x, y in zip(xs, ys):
                xb.copy_(x)
                yb.copy_(y)
                losses.append(tr.steps(xb, yb, b, wait=False))
                xb.fill_(0.5)
                yb.zero_()
        elif mode == 'serial':
            losses = [tr.steps(x, y, b) for x, y in zip(xs, ys)]
        else:
            losses = [tr.step(x[j * b:(j + 1) * b].contiguous(), y[j * b:(j + 1) * b].contiguous()) for x, y in zip(xs, ys) for j in range(k)]
        tr.join()
        torch.cuda.synchronize()
        assert tr.t == 2 * k
        return torch.cat(losses).cpu(), tr.pflat.clone().cpu(), tr.m.clone().cpu()

    serial, overlap, single = run('serial'), run('overlap'), run('single')
    assert serial[0].shape == (2 * k,) and torch.isfinite(serial[0]).all()
    for a, c in zip(serial, overlap):
        assert torch.equal(a, c)
    torch.testing.assert_close(serial[0], single[0], rtol=1e-5, atol=0)
    diff = (serial[1] - single[1]).abs()
    assert (diff > 2e-5 + 1e-3 * single[1].abs()).float().mean() < 1e-5 and diff.max() <= 2 * k * 1.1e-5
    model = DeviceModel(prm, device=DEV)
    with pytest.raises(ValueError, match='do not split'):
        Trainer(model).steps(xs[0], ys[0], 7)


def test_overlapped_step_equals_serial_step_one_rank_rccl():
    """The step with its single all-reduce + Adam on a side stream, overlapped with the next batch's frozen backbone
    (wait=False), gives the same bits as the serial step, over 3 steps, through RCCL (backend nccl) on one rank; and the
    step issues exactly ONE collective carrying [dWt | db | loss]."""
    import torch.distributed as dist
    from test_parallel_gloo import _free_port
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        N = 4
        # (glorot Dense weights and a small learning rate: with features of RMS ~7 one Adam step at lr 1e-3 moves a logit by
        #  tens of units; below -104 a sigmoid is exactly 0 in fp32, and for a box width of exactly 0 the reference's
        #  autodiff - and therefore ty1_yolo_loss_fwd_bwd - gives NaN for d sqrt(w) / dw)
        prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
        xs = [torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(20 + i)).to(DEV) for i in range(3)]
        ys = [torch.from_numpy(_labels(N, 20, 30 + i)).to(DEV) for i in range(3)]
        calls = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        results = {}
        gathers = []
        real_gather = dist.all_gather_into_tensor

        def counting_gather(out, inp, *a, **k):
            gathers.append(tuple(inp.shape))
            return real_gather(out, inp, *a, **k)
        for mode in ('serial', 'overlap', 'gather'):
            model = DeviceModel(prm, device=DEV)
            tr = Trainer(model, lr=1e-5, overlap=(mode != 'serial'))
            assert tr.bucket.flat.numel() == 1472 * 12544 + 1470 + 1
            tr.bucket.always = True          # a one-rank group would short-cut the collective: send it through RCCL
            if mode == 'gather':             # features and dz travel instead of the weight gradient (Trainer.exchange)
                tr.exchange = 'gather'
            dist.all_reduce, dist.all_gather_into_tensor = counting, counting_gather
            try:
                losses = [tr.step(x, y, wait=(mode == 'serial')) for x, y in zip(xs, ys)]
                tr.join()
                torch.cuda.synchronize()
            finally:
                dist.all_reduce, dist.all_gather_into_tensor = real, real_gather
            results[mode] = (torch.cat(losses).cpu(), model.dense_wt.clone().cpu(), model.dense_b.clone().cpu(), tr.m.clone().cpu())
        assert calls == [1472 * 12544 + 1470 + 1] * 6                    # one message per step in the two all-reduce modes, none in 'gather'
        assert gathers == [(N, 12544), (N, 1472)] * 3                    # ... which sends the features and dz of each step
        for mode in ('overlap', 'gather'):                               # (one rank: the gathered rows ARE the local rows - same bits)
            for a, b in zip(results['serial'], results[mode]):
                assert torch.isfinite(a).all() and torch.equal(a, b), mode
        assert results['serial'][0][2] < results['serial'][0][0]          # it learns
    finally:
        dist.destroy_process_group()


def test_keras_style_training_entry_points(tmp_path):
    outputs, inputs = model_tiny_YOLOv1()
    model = Model(inputs=inputs, outputs=outputs)
    model.compile(loss=yolo_loss, optimizer=Adam(lr=0.001))
    x = torch.rand((4, 448, 448, 3), generator=torch.Generator().manual_seed(4)).numpy()
    y = _labels(4, 20, 1)

    class Gen:
        def __len__(self):
            return 2

        def __getitem__(self, i):
            return x, y

    hist = model.fit_generator(Gen(), steps_per_epoch=2, epochs=2, validation_data=Gen(), validation_steps=1, verbose=0)
    assert hist['loss'][1] < hist['loss'][0] and len(hist['val_loss']) == 2
    # fit_generator pairs consecutive batches for one pass of the frozen backbone (a ragged last batch steps alone) and keeps
    # the losses on the device: the same epoch through train_on_batch, one host round trip per step, must agree
    class Ragged:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            return (x, y) if i < 2 else (x[:2], y[:2])

    ma, mb = (Model(inputs=inputs, outputs=outputs, seed=7) for _ in range(2))
    for m in (ma, mb):
        m.compile(loss=yolo_loss, optimizer=Adam(lr=1e-5))
    ha = ma.fit_generator(Ragged(), epochs=1, verbose=0)
    lb = [mb.train_on_batch(*Ragged()[i]) for i in range(3)]
    assert abs(ha['loss'][0] - sum(lb) / 3) <= 1e-5 * abs(sum(lb) / 3)
    wa, wb = ma.engine.dense_wt.cpu(), mb.engine.dense_wt.cpu()
    assert ((wa - wb).abs() > 2e-5 + 1e-3 * wb.abs()).float().mean() < 1e-5 and (wa - wb).abs().max() <= 3 * 1.1e-5
    model.save_weights(str(tmp_path / 'trained.npz'))
    m2 = Model(inputs=inputs, outputs=outputs, seed=99)
    m2.load_weights(str(tmp_path / 'trained.npz'))
    np.testing.assert_array_equal(m2.predict(x[:1]), model.predict(x[:1]))


def test_detection_records_roundtrip_on_device():
    g = load_golden('nms_c20.npz')
    det = ops.decode_nms(torch.from_numpy(g['prediction']).to(DEV))
    rec = parallel.pack_detections(*det)
    boxes, score, cls, keep = parallel.unpack_detections(parallel.gather_detections(rec))
    assert torch.equal(boxes, det[0]) and torch.equal(keep, det[3]) and torch.equal(score, det[1])
    assert torch.equal(cls[keep.bool()], det[2][keep.bool()])
    # the HIP packer (device tensors) writes the same records as the host formulation (what the gloo tests pack)
    host = parallel.pack_detections(*(t.cpu() for t in det))
    assert rec.is_cuda and torch.equal(rec.cpu(), host)
    # ragged sizes and a caller-provided output
    rng = np.random.default_rng(5)
    for n, slots in [(1, 98), (3, 49), (7, 196)]:
        b = torch.from_numpy(rng.random((n, slots, 4))).to(DEV)
        sc = torch.from_numpy(rng.random((n, slots)).astype(np.float32)).to(DEV)
        cl = torch.from_numpy(rng.integers(0, 20, (n, slots)).astype(np.int32)).to(DEV)
        kp = torch.from_numpy((rng.random((n, slots)) < 0.3).astype(np.uint8)).to(DEV)
        out = torch.full((n, slots, 6), float('nan'), dtype=torch.float64, device=DEV)
        assert ops.pack_detections(b, sc, cl, kp, out=out) is out
        assert torch.equal(out.cpu(), parallel.pack_detections(b.cpu(), sc.cpu(), cl.cpu(), kp.cpu()))


# ------------------------------------------------------------------------------- preprocessing on the GPU (8f row 2)
@pytest.mark.parametrize('h0,w0', [(448, 448), (375, 500), (37, 1000), (1080, 1920), (2, 3), (449, 447), (896, 896)])
def test_preprocess_u8_vs_oracle(h0, w0):
    """ty1_preprocess_u8 against the checker's restatement of OpenCV's 8-bit INTER_LINEAR resize (oracle/resize_ref.py)
    followed by the reference's `image / 255.` in float64 (/root/reference/utils.py:13-15): bit-exact.
    PARITY UNPINNED with cv2 itself: OpenCV is not installed and the reference holds no resized image."""
    from oracle.resize_ref import resize_linear_u8_ref, load_img_ref
    img = np.random.default_rng(h0 + w0).integers(0, 256, (h0, w0, 3), dtype=np.uint8)
    want = load_img_ref(img, (448, 448, 3)).astype(np.float32)
    d = torch.from_numpy(img).to(DEV)
    got = ops.preprocess_u8(d, (448, 448)).cpu().numpy()
    assert np.array_equal(got, want)
    got_swapped = ops.preprocess_u8(d, (448, 448), swap_rb=True).cpu().numpy()
    assert np.array_equal(got_swapped, want[..., ::-1])
    # into one slot of a batch, other output size
    batch = torch.zeros((3, 224, 224, 3), device=DEV)
    ops.preprocess_u8(d, (224, 224), out=batch[1])
    assert np.array_equal(batch[1].cpu().numpy(), (resize_linear_u8_ref(img, 224, 224) / 255.).astype(np.float32))
    assert not batch[0].any() and not batch[2].any()
    # a tiny odd output size
    assert np.array_equal(ops.preprocess_u8(d, (9, 9)).cpu().numpy(), (resize_linear_u8_ref(img, 9, 9) / 255.).astype(np.float32))


@pytest.mark.parametrize('n,h0,w0', [(5, 480, 640), (3, 448, 448), (2, 896, 896), (1, 37, 1000)])
def test_preprocess_batch_u8_equals_per_frame(n, h0, w0):
    """ty1_preprocess_batch_u8 (equal-sized frames, one launch) = ty1_preprocess_u8 frame by frame = the checker's
    restatement, bit for bit, with and without the channel swap; DeviceModel.detect_u8 = preprocess + detect."""
    from oracle.resize_ref import load_img_ref
    imgs = np.random.default_rng(n * h0 + w0).integers(0, 256, (n, h0, w0, 3), dtype=np.uint8)
    d = torch.from_numpy(imgs).to(DEV)
    for swap in (False, True):
        got = ops.preprocess_batch_u8(d, (448, 448), swap_rb=swap)
        for i in range(n):
            assert torch.equal(got[i], ops.preprocess_u8(d[i], (448, 448), swap_rb=swap))
        want = load_img_ref(imgs[0], (448, 448, 3)).astype(np.float32)
        assert np.array_equal(got[0].cpu().numpy(), want[..., ::-1] if swap else want)
    out = torch.zeros((n, 448, 448, 3), device=DEV)
    assert ops.preprocess_batch_u8(d, (448, 448), out=out) is out
    with pytest.raises(ValueError):
        ops.preprocess_batch_u8(d[0])
    if n == 5:
        model = DeviceModel(P.make_random_params(seed=3, dense_init='spread', bn='random'), device=DEV)
        a = [t.clone() for t in model.detect_u8(d)]
        b = model.detect(ops.preprocess_batch_u8(d, (448, 448)))
        assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_host_frame_pipeline_equals_detect_u8(dtype):
    """engine.HostFramePipeline (pinned host frames -> copy stream -> one-launch resize -> detect, two device buffers) gives
    the detections of DeviceModel.detect_u8 on the same frames, batch after batch, also when the caller refills its host
    buffer right after submit() returns only once the NEXT submit has been made (the copy is asynchronous)."""
    from yolo_amd.engine import HostFramePipeline
    n, H0, W0 = 6, 240, 320
    model = DeviceModel(P.make_random_params(seed=3, dense_init='spread', bn='random'), device=DEV, dtype=dtype)
    rng = np.random.default_rng(17)
    batches = [torch.from_numpy(rng.integers(0, 256, (n, H0, W0, 3), dtype=np.uint8)) for _ in range(5)]
    want = [[t.clone() for t in model.detect_u8(b.to(DEV))] for b in batches]
    pipe = HostFramePipeline(model, n, H0, W0)
    pinned = [torch.empty((n, H0, W0, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    for i, b in enumerate(batches):
        pinned[i & 1].copy_(b)                       # (submitted two batches ago; the submit() after that has returned: it is free)
        got = pipe.submit(pinned[i & 1])
        assert all(torch.equal(x, y) for x, y in zip(got, want[i])), i
    with pytest.raises(ValueError):
        pipe.submit(torch.zeros((n, H0, W0 + 1, 3), dtype=torch.uint8))


def test_detect_from_file_uses_device_preprocessing(tmp_path):
    """Regression check between the two product paths (file -> GPU preprocessing, host load_img -> upload)."""
    from PIL import Image
    from yolo_amd.tiny_yolov1 import TinyYOLOv1
    from yolo_amd.utils import load_img
    img = np.random.default_rng(9).integers(0, 256, (300, 420, 3), dtype=np.uint8)
    path = str(tmp_path / 'a.png')
    Image.fromarray(img).save(path)
    det = TinyYOLOv1(input_path=path, device=DEV)
    from_file = det.detect()
    host_img, _, _ = load_img(path, (448, 448, 3))
    from_array = det.detect(host_img)
    assert len(from_file) == 1 and np.array_equal(from_file[0]['slots'], from_array[0]['slots'])
    assert np.array_equal(from_file[0]['boxes'], from_array[0]['boxes'])


# ------------------------------------------------------------------------------- the other next rows (8f 1, 3, 4) on the GPU
def test_darknet_weights_file_into_device_model(tmp_path):
    """8f row 1: a full-size Darknet .weights file (layout /root/reference/convert.py:59-60,99-144,211-229) written by
    the test, read by position with the generated cfg, runs in the HIP model exactly like the parameters it was
    written from."""
    from yolo_amd import darknet
    prm = P.make_random_params(seed=7, dense_init='spread', bn='random')
    specs = [(ci, co, i < P.N_POOL) for i, (ci, co) in enumerate(P.CONV_CHANNELS)]
    cfg, wts = str(tmp_path / 'tiny.cfg'), str(tmp_path / 'tiny.weights')
    darknet.write_cfg(cfg, specs, 448, 1470)
    darknet.write_darknet_weights(wts, prm, specs)
    loaded, size, got_specs = darknet.load_darknet_weights(cfg, wts)
    assert size == 448 and got_specs == specs and loaded['floats_read'] == 27_099_950
    assert all(np.array_equal(loaded[k], prm[k]) for k in prm)
    x = torch.rand((2, 448, 448, 3), generator=torch.Generator().manual_seed(2)).to(DEV)
    a = DeviceModel({k: v for k, v in loaded.items() if k in prm}, device=DEV, conv_specs=got_specs, img_size=size)
    b = DeviceModel(prm, device=DEV)
    assert torch.equal(a.predict(x), b.predict(x))
    assert all(torch.equal(u, v) for u, v in zip(a.detect(x), b.detect(x)))


def test_label_encoder_output_feeds_the_loss_kernel():
    """8f row 3: y_true written by the reference's own SequenceData.read (tests/golden/labels_c20.npz) and the same
    lines encoded here go through ty1_yolo_loss_fwd_bwd and give the oracle's loss and gradient."""
    from yolo_amd.data_sequence import encode_boxes, parse_annotation_line
    g = load_golden('labels_c20.npz')
    y_true = np.stack([encode_boxes(parse_annotation_line(str(ln))[1], int(h), int(w), 7, 20)
                       for ln, (h, w) in zip(g['lines'], g['hw'])]).astype(np.float32)        # Keras casts float64 labels
    assert np.array_equal(y_true, g['label_matrix'].astype(np.float32))
    N = y_true.shape[0]
    z = np.random.default_rng(11).standard_normal((N, 1470)).astype(np.float32)
    y_pred = model_ref.head_activate_ref(z)
    loss_want, per_want, grad_want = loss_ref.yolo_loss_and_grad_ref(y_true, y_pred)
    d = lambda a: torch.from_numpy(a).to(DEV)
    loss, per, grad = ops.yolo_loss_fwd_bwd(d(y_true), d(y_pred))
    assert abs(loss.item() - loss_want) <= 1e-5 * abs(loss_want)
    np.testing.assert_allclose(per.cpu().numpy(), per_want, rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), grad_want, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name,C,img', [('nms_c20.npz', 20, 448), ('nms_c1.npz', 1, 224)])
def test_detection_consumer_emits_what_the_reference_draws(name, C, img):
    """8f row 4: ty1_decode_nms -> detections.batch_records against the cv.rectangle / cv.putText calls the reference's
    _main made on the same predictions (/root/reference/tiny_yolov1.py:142-160: int() corners + class name;
    /root/reference/faces/testFace.py:144-152: int() corners + str(confidence))."""
    from yolo_amd.detections import batch_records, VOC_CLASSES
    g = load_golden(name)
    det = ops.decode_nms(torch.from_numpy(g['prediction']).to(DEV), 2, C, img)
    recs = batch_records(det, img_size=img, names=VOC_CLASSES if C == 20 else None)
    assert len(recs) == g['prediction'].shape[0]
    for i, rl in enumerate(recs):
        lo, hi = g['rect_off'][i], g['rect_off'][i + 1]
        assert [r['corners'] for r in rl] == [tuple(int(v) for v in row) for row in g['rects'][lo:hi]], i
        if C == 20:
            assert [r['name'] for r in rl] == [ty.classes_name[k] for k in g['labels'][lo:hi]]
            assert [r['class_id'] for r in rl] == list(g['labels'][lo:hi])
        else:
            assert [str(np.float32(r['score'])) for r in rl] == list(g['conf_str'][lo:hi])
        assert all(r['box'] == tuple(float(v) for v in det[0][i, r['slot']].cpu().numpy()) for r in rl)   # no image_hw: same frame


def test_train_compile_save_reload_keeps_trained_weights(tmp_path):
    """Advisor finding: compile() after training drops the trainer; save_weights must still write the trained Dense."""
    outputs, inputs = model_tiny_YOLOv1()
    model = Model(inputs=inputs, outputs=outputs)
    model.compile(loss=yolo_loss, optimizer=Adam(lr=0.001))
    x = torch.rand((2, 448, 448, 3), generator=torch.Generator().manual_seed(5)).numpy()
    y = _labels(2, 20, 2)
    before = model.predict(x)
    model.train_on_batch(x, y)
    model.compile(loss=yolo_loss, optimizer=Adam(lr=0.0005))         # e.g. a new learning rate
    after = model.predict(x)
    assert not np.array_equal(before, after)
    model.save_weights(str(tmp_path / 't.npz'))
    m2 = Model(inputs=inputs, outputs=outputs, seed=42)
    m2.load_weights(str(tmp_path / 't.npz'))
    np.testing.assert_array_equal(m2.predict(x), after)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_detect_replay_equals_detect(dtype):
    """DeviceModel.detect_replay (one HIP-graph launch per batch) gives the bits of detect(): batch 1 (the reference's
    per-frame detect), 3 and 20 (the fused inference tail), two different batches in a row through the same graph, and a
    batch written straight into the graph's input buffer."""
    prm = P.make_random_params(seed=5, dense_init='glorot', bn='identity')
    model = DeviceModel(prm, device=DEV, dtype=dtype)
    g = torch.Generator().manual_seed(9)
    for N in (1, 3, 20):
        for rep in range(2):
            x = torch.rand((N, 448, 448, 3), generator=g).to(DEV)
            want = [t.clone() for t in model.detect(x)]
            if rep == 0:
                got = model.detect_replay(x)
            else:
                buf = model.input_buffer(N)
                buf.copy_(x)
                got = model.detect_replay(buf)
            torch.cuda.synchronize()
            assert want[3].any(), 'the stress weights keep boxes'
            for w, t in zip(want, got):
                assert w.dtype == t.dtype and torch.equal(w.view(torch.uint8), t.view(torch.uint8))
    assert len(model._graphs) == 3
