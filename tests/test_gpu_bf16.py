"""GPU parity of the bf16 conv path (BASELINE config 3: bf16 inference) against the CPU oracle.

The reference computes in fp32 (Keras default floatx), so there is no reference bf16 result: the oracle is
oracle/model_ref.conv_block_ref (fp32 torch-CPU restatement of models/model_tiny_yolov1.py:43-87) evaluated on the
SAME bf16-rounded inputs and weights, and the stated tolerance is the output rounding of the kernel:
|got - ref| <= 2^-8 |ref| + 2e-3   (bf16 keeps 8 significant bits: half an ulp is 2^-9 relative; the factor two covers
fp32 summation-order differences that move a value across a rounding boundary).  Layers that store fp32 (out_f32)
are held to the fp32 bar, 1e-3 relative.
"""
import numpy as np
import pytest
import torch

from oracle import model_ref
from yolo_amd import ops, params as P

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def _case(N, H, cin, cout, pool, out_f32, seed, use_ws):
    rng = np.random.default_rng(seed)
    x = rng.random((N, H, H, cin)).astype(np.float32)
    w = P.glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)
    gamma = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    beta = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    mean = rng.uniform(-0.2, 0.2, cout).astype(np.float32)
    var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    # the oracle sees what the MFMA sees: bf16 activations and weights (the first layer takes the fp32 image and
    # fp32 weights and rounds them itself)
    xr, wr = bf16_round(x), bf16_round(w)
    if cin != 3:
        x, w = xr, wr
    xt = torch.from_numpy(xr).permute(0, 3, 1, 2).contiguous()
    ref = model_ref.conv_block_ref(xt, wr, gamma, beta, mean, var, pool).permute(0, 2, 3, 1).contiguous().numpy()
    scale, shift = ops.fold_bn(gamma, beta, mean, var)
    in_dt = torch.float32 if cin == 3 else torch.bfloat16
    ws = None
    if use_ws:
        nbytes = ops.conv3x3_workspace_bytes_bf16(N, H, H, cin, cout, pool)
        ws = torch.full((max(nbytes, 16) // 4,), float('nan'), dtype=torch.float32, device=DEV)
    # the output sits between two guard regions: a kernel that leans on the range check of its output descriptor (ragged last
    # work items, padding rows) must not write a byte outside the tensor
    out_dt = torch.float32 if out_f32 else torch.bfloat16
    oshape = (N, H // 2, H // 2, cout) if pool else (N, H, H, cout)
    numel, guard = int(np.prod(oshape)), 1 << 20
    buf = torch.full((numel + 2 * guard,), 77.0, dtype=out_dt, device=DEV)
    out = buf[guard:guard + numel].view(oshape)
    got = ops.conv3x3_bn_leaky_bf16(dev(x, in_dt), dev(P.hwio_to_ohwi(w), in_dt), dev(scale), dev(shift), pool,
                                    out_f32=out_f32, out=out, workspace=ws)
    torch.cuda.synchronize()
    assert got.dtype == out_dt
    assert bool((buf[:guard] == 77.0).all()) and bool((buf[guard + numel:] == 77.0).all()), 'write outside the output tensor'
    return got.float().cpu().numpy(), ref


# the eight layers of the network (last one stores fp32 for the Dense), then shapes with tails / no pool / K-split tail
CASES = [(2, 448, 3, 16, True, False), (2, 224, 16, 32, True, False), (2, 112, 32, 64, True, False),
         (3, 56, 64, 128, True, False), (3, 28, 128, 256, True, False), (5, 14, 256, 512, True, False),
         (5, 7, 512, 1024, False, False), (5, 7, 1024, 256, False, True),
         (1, 36, 3, 32, True, False), (1, 40, 3, 16, False, False), (1, 20, 16, 32, False, False), (1, 22, 16, 64, True, False),
         (2, 10, 32, 64, True, False), (1, 6, 64, 128, False, True), (70, 7, 64, 64, False, False),
         (24, 28, 128, 256, True, False), (12, 7, 1024, 256, False, True),
         (2, 10, 96, 192, True, False), (1, 9, 160, 64, False, True)]      # odd chunk counts, Cout = 3 x 64


@pytest.mark.parametrize('use_ws', [False, True])
@pytest.mark.parametrize('N,H,cin,cout,pool,out_f32', CASES)
def test_conv_block_bf16_vs_oracle(N, H, cin, cout, pool, out_f32, use_ws):
    if use_ws and cin in (3, 16):
        pytest.skip('direct kernels have no split path')
    got, ref = _case(N, H, cin, cout, pool, out_f32, seed=cin + cout, use_ws=use_ws)
    assert got.shape == ref.shape
    if out_f32:
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-4)
    else:
        err = np.abs(got - ref)
        assert (err <= 2.0 ** -8 * np.abs(ref) + 2e-3).all(), float(err.max())
        # and typically exact to the rounding: at least 99 % of the outputs are the nearest bf16 of the oracle value
        assert (got == bf16_round(ref)).mean() > 0.99


# batches large enough for the halo kernels (csrc/conv_halo.hip; the dispatcher takes them once a layer has at least four
# work items per CU): weights resident (Cin = 32), streamed weights with one and with two K chunks and channel tiles
HALO_CASES = [(44, 112, 32, 64, True), (80, 56, 64, 128, True), (40, 56, 128, 256, True),
              (64, 28, 128, 256, True), (128, 14, 256, 512, True),      # 14x14 patches: two and four K chunks and channel tiles
              (126, 7, 512, 1024, False),
              # the smallest batches the dispatcher still sends to these kernels (one round of the chip, or a bit less)
              (16, 112, 32, 64, True), (16, 56, 64, 128, True), (32, 28, 128, 256, True), (64, 14, 256, 512, True)]      # no pool, four whole 7x7 images per item; the last item holds two images


def _halo_launches(fn):
    """Run fn() with every library launch bracketed; -> (result, names of the halo-kernel launches it made)."""
    with ops.KernelTimer() as kt:
        res = fn()
        torch.cuda.synchronize()
        names = [k for k in kt.table() if k.startswith('conv_halo_bf16')]
    return res, names


@pytest.mark.parametrize('N,H,cin,cout,pool', HALO_CASES)
def test_conv_block_bf16_halo_vs_oracle(N, H, cin, cout, pool):
    (got, ref), halo = _halo_launches(lambda: _case(N, H, cin, cout, pool, False, seed=3 * cin + cout, use_ws=False))
    if not halo:      # the dispatch thresholds follow the CU count: on another chip the case may leave the halo path
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        pytest.skip(f'this device ({cus} CUs) sends N={N} {H}x{H}x{cin}->{cout} to the implicit-GEMM kernel: halo path not exercised')
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 2e-3).all(), float(err.max())
    assert (got == bf16_round(ref)).mean() > 0.99


@pytest.mark.parametrize('N,cin,cout,slices', [(128, 1024, 256, 4), (126, 1024, 256, 4), (64, 512, 1024, 2), (32, 512, 1024, 4),
                                               (62, 512, 1024, 2)])
def test_conv_block_bf16_halo_k_slices_vs_oracle(N, cin, cout, slices):
    """7x7 layers with fewer tiles than CUs (7x7x1024->256 at the batch of configs[2]: 64 tiles for 256 CUs): the halo kernel
    runs `slices` K slices per tile into the (NaN-poisoned) workspace and a second launch sums them; 126 / 62 images leave
    the last tile ragged."""
    assert ops.conv3x3_workspace_bytes_bf16(N, 7, 7, cin, cout, False) == ((N + 3) // 4) * (cout // 128) * slices * 256 * 128 * 4
    for use_ws in (True, False):          # without the workspace the layer falls back to the implicit GEMM
        (got, ref), halo = _halo_launches(lambda: _case(N, 7, cin, cout, False, False, seed=11, use_ws=use_ws))
        assert (halo == ['conv_halo_bf16_4x7x7_ksplit']) == use_ws, halo      # the path each leg means to test did run
        err = np.abs(got - ref)
        assert (err <= 2.0 ** -8 * np.abs(ref) + 2e-3).all(), float(err.max())
        assert (got == bf16_round(ref)).mean() > 0.99


# ------------------------------------------------------------------------------- first two blocks in one launch
def _fused_case(N, H, W, seed, neg_scale=True):
    """-> (got f32 array, oracle, two-launch result, operands): ty1_conv3x3_x2_c3_bf16 against the oracle's two conv blocks with the
    bf16 network's rounding points (image, weights and the tensor between the blocks rounded to bf16)."""
    rng = np.random.default_rng(seed)
    x = rng.random((N, H, W, 3)).astype(np.float32)
    blocks = []
    for cin, cout in ((3, 16), (16, 32)):
        w = P.glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)
        gamma = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        if neg_scale:
            gamma[::3] *= -1                     # the kernel folds sign(scale) into its weights: exercise both signs
        beta, mean = rng.uniform(-0.5, 0.5, cout).astype(np.float32), rng.uniform(-0.2, 0.2, cout).astype(np.float32)
        var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        blocks.append((w, gamma, beta, mean, var))
    t = torch.from_numpy(bf16_round(x)).permute(0, 3, 1, 2).contiguous()
    for w, gamma, beta, mean, var in blocks:
        t = torch.from_numpy(bf16_round(model_ref.conv_block_ref(t, bf16_round(w), gamma, beta, mean, var, True).numpy()))
    ref = t.permute(0, 2, 3, 1).contiguous().numpy()
    (w0, *bn0), (w1, *bn1) = blocks
    sc0, sh0 = ops.fold_bn(*bn0)
    sc1, sh1 = ops.fold_bn(*bn1)
    ops_ = dict(x=dev(x), w0=dev(P.hwio_to_ohwi(w0)), sc0=dev(sc0), sh0=dev(sh0), w1=dev(P.hwio_to_ohwi(w1), torch.bfloat16),
                sc1=dev(sc1), sh1=dev(sh1))
    oshape = (N, H // 4, W // 4, 32)
    numel, guard = int(np.prod(oshape)), 1 << 18
    buf = torch.full((numel + 2 * guard,), 77.0, dtype=torch.bfloat16, device=DEV)
    out = buf[guard:guard + numel].view(oshape)
    got = ops.conv3x3_x2_c3_bf16(ops_['x'], ops_['w0'], ops_['sc0'], ops_['sh0'], ops_['w1'], ops_['sc1'], ops_['sh1'], out=out)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 77.0).all()) and bool((buf[guard + numel:] == 77.0).all()), 'write outside the output tensor'
    mid = ops.conv3x3_bn_leaky_bf16(ops_['x'], ops_['w0'], ops_['sc0'], ops_['sh0'], True)
    two = ops.conv3x3_bn_leaky_bf16(mid, ops_['w1'], ops_['sc1'], ops_['sh1'], True)
    return got.float().cpu().numpy(), ref, two.float().cpu().numpy(), ops_


def _check_fused(got, ref, two):
    # block-1 outputs inherit the rounding flips of block 0 (an fp32 sum in another order moves a few block-0 values to the
    # neighbouring bf16), so the bar is the bf16 bar of one layer with a doubled absolute term; the share of outputs that
    # are exactly the oracle's bf16 stays above 97 % [measured 98.5-99.5 %].  Against the two separate launches: the same bar.
    for want in (ref, two):
        err = np.abs(got - want)
        assert (err <= 2.0 ** -7 * np.abs(want) + 4e-3).all(), float(err.max())
        assert (got == want).mean() > 0.97, float((got == want).mean())


@pytest.mark.parametrize('N,H,W', [(2, 448, 448), (1, 64, 128), (1, 40, 72), (3, 100, 52), (2, 4, 4), (1, 68, 132), (5, 32, 64)])
def test_fused_first_two_blocks_vs_oracle(N, H, W):
    """Whole patches (448 x 448 = 14 x 7 of them per image), one patch exactly (64 x 128), ragged last patches in both
    directions (40 x 72: 20 x 36 block-1 pixels on 16 x 32 patches; 100 x 52; 68 x 132: one row / column of patches that
    holds four pixels), and an image smaller than a patch."""
    got, ref, two, _ = _fused_case(N, H, W, seed=H + W)
    assert got.shape == ref.shape == (N, H // 4, W // 4, 32)
    _check_fused(got, ref, two)


def test_fused_first_two_blocks_full_batch_properties():
    """configs[2]'s batch (128 images per GPU): determinism, the first / a middle / the last image against the oracle,
    every image against the two separate launches."""
    N = 128
    prm = P.make_random_params(seed=3, dense_init='glorot', bn='random')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(2))
    xd = x.to(DEV)
    sc0, sh0 = ops.fold_bn(prm['bn0_gamma'], prm['bn0_beta'], prm['bn0_mean'], prm['bn0_var'])
    sc1, sh1 = ops.fold_bn(prm['bn1_gamma'], prm['bn1_beta'], prm['bn1_mean'], prm['bn1_var'])
    w0, w1 = dev(P.hwio_to_ohwi(prm['conv0_w'])), dev(P.hwio_to_ohwi(prm['conv1_w']), torch.bfloat16)
    args = (xd, w0, dev(sc0), dev(sh0), w1, dev(sc1), dev(sh1))
    got = ops.conv3x3_x2_c3_bf16(*args)
    assert torch.equal(ops.conv3x3_x2_c3_bf16(*args), got)
    two = ops.conv3x3_bn_leaky_bf16(ops.conv3x3_bn_leaky_bf16(xd, w0, args[2], args[3], True), w1, args[5], args[6], True)
    pick = [0, 77, 127]
    t = torch.from_numpy(bf16_round(x[pick].numpy())).permute(0, 3, 1, 2).contiguous()
    for i in range(2):
        t = torch.from_numpy(bf16_round(model_ref.conv_block_ref(t, bf16_round(prm[f'conv{i}_w']), prm[f'bn{i}_gamma'], prm[f'bn{i}_beta'],
                                                                 prm[f'bn{i}_mean'], prm[f'bn{i}_var'], True).numpy()))
    ref = t.permute(0, 2, 3, 1).contiguous().numpy()
    g = got.float().cpu().numpy()
    _check_fused(g[pick], ref, two.float().cpu().numpy()[pick])
    err = (got.float() - two.float()).abs()
    assert bool((err <= 2.0 ** -7 * two.float().abs() + 4e-3).all())


def test_fused_first_two_blocks_rejects_what_it_does_not_cover():
    x = torch.zeros((1, 6, 8, 3), device=DEV)
    w0, w1 = torch.zeros((16, 3, 3, 3), device=DEV), torch.zeros((32, 3, 3, 16), dtype=torch.bfloat16, device=DEV)
    s16, s32 = torch.ones(16, device=DEV), torch.ones(32, device=DEV)
    with pytest.raises(Exception, match='multiples of 4'):
        ops.conv3x3_x2_c3_bf16(x, w0, s16, s16, w1, s32, s32)
    x = torch.zeros((1, 8, 8, 3), device=DEV)
    with pytest.raises(Exception, match='alpha'):
        ops.conv3x3_x2_c3_bf16(x, w0, s16, s16, w1, s32, s32, alpha=0.0)
    assert ops.conv3x3_x2_c3_bf16(torch.zeros((0, 8, 8, 3), device=DEV), w0, s16, s16, w1, s32, s32).shape == (0, 2, 2, 32)


@pytest.mark.parametrize('N', [1, 33, 64, 70, 128, 130])
def test_dense_splitk_bf16_vs_oracle(N):
    """ty1_dense_splitk_bf16 at the batch sizes of its tile shapes (32 x 128 up to 32 rows, 64 x 64 above; ragged last row
    tile at 33 / 70 / 130) against the oracle with the same rounding of features and weights; fp32 accumulation in a
    different order: 2e-3 of the logits' RMS."""
    rng = np.random.default_rng(N)
    units = 1470
    feats = rng.standard_normal((N, P.FEATURES)).astype(np.float32)
    w = (0.02 * rng.standard_normal((P.FEATURES, units))).astype(np.float32)
    z_ref = model_ref.dense_ref(feats, w, np.zeros(units, np.float32), bf16=True)
    part = ops.dense_splitk(ops.cast_bf16(dev(feats)), ops.cast_bf16(dev(P.dense_to_device_layout(w))), units)
    assert part.shape[0] == ops.pick_splits(P.FEATURES, N, units, chunk=64)
    z = part.sum(0)[:, :units].cpu().numpy()
    assert np.abs(z - z_ref).max() < 2e-3 * np.sqrt((z_ref ** 2).mean())


def test_cast_bf16_round_to_nearest_even():
    x = torch.randn(100003, device=DEV) * 37.0
    assert torch.equal(ops.cast_bf16(x), x.to(torch.bfloat16))


def test_bf16_rejects_what_it_does_not_support():
    x = torch.zeros((1, 8, 8, 48), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((64, 3, 3, 48), dtype=torch.bfloat16, device=DEV)
    s = torch.ones(64, device=DEV)
    with pytest.raises(Exception, match='multiple of 32'):
        ops.conv3x3_bn_leaky_bf16(x, w, s, s, False)
    x = torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((64, 3, 3, 64), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(Exception, match='pool == 0'):
        ops.conv3x3_bn_leaky_bf16(x, w, s, s, True, out_f32=True)
    with pytest.raises(Exception, match='alpha'):
        ops.conv3x3_bn_leaky_bf16(x, w, s, s, False, alpha=-0.1)


# ------------------------------------------------------------------------------- whole model in bf16
@pytest.mark.parametrize('dense_init', ['spread', 'glorot'])
def test_model_bf16_end_to_end(dense_init):
    """Whole network with dtype='bf16' against (a) the oracle restating the same bf16 rounding points and (b) the
    fp32 oracle of the reference.  Stated bf16 tolerances (measured values are printed with -s):
      per-layer activations: within 2 % of the layer's largest magnitude of (a)   [measured 0.2-0.7 %];
      Dense logits: within 3 % of their RMS of (a), 6 % of (b)          [measured 0.8-1.0 % and 2.4-2.5 %];
      head outputs (probabilities, logit RMS 9 with glorot and 39 with 'spread' Dense weights under random BN):
      within 0.08 of (b) for glorot, 0.25 for 'spread'                  [measured 0.034 and 0.11].
    Detections: bit-exact given the GPU's own scores."""
    import oracle
    from yolo_amd.engine import DeviceModel
    N = 2
    prm = P.make_random_params(seed=1, C=20, dense_init=dense_init, bn='random')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(0)).numpy()
    model = DeviceModel(prm, C=20, device=DEV, dtype='bf16')
    xd = dev(x)
    y = model.predict(xd).cpu().numpy()
    ws = model.workspace(N)
    z = ws['part'].sum(0)[:, :model.units].cpu().numpy() + prm['dense_b']
    feats_a, acts_a = model_ref.backbone_ref(x, prm, return_all=True, bf16=True)
    for i, (a_got, a_ref) in enumerate(zip(ws['acts'], acts_a)):
        if i == 0 and model.fused01:
            continue                      # blocks 0 and 1 run as one launch: the tensor between them never reaches HBM
        g = a_got.float().cpu().numpy()
        err = np.abs(g - a_ref).max() / max(1.0, np.abs(a_ref).max())
        print(f'layer {i}: max err / max |ref| = {err:.2e}')
        assert err < 2e-2
    z_a = model_ref.dense_ref(feats_a, prm['dense_w'], prm['dense_b'], bf16=True)
    z_b = model_ref.dense_ref(model_ref.backbone_ref(x, prm), prm['dense_w'], prm['dense_b'])
    rms = float(np.sqrt((z_b ** 2).mean()))
    ea, eb = np.abs(z - z_a).max() / rms, np.abs(z - z_b).max() / rms
    print(f'logit rms {rms:.3f}: max err vs bf16 oracle {ea:.3e} rms, vs fp32 oracle {eb:.3e} rms')
    assert ea < 3e-2 and eb < 6e-2
    y_b = model_ref.model_ref(x, prm, 7, 2, 20)
    print('head outputs vs fp32 oracle', np.abs(y - y_b).max())
    assert np.abs(y - y_b).max() < (0.08 if dense_init == 'glorot' else 0.25)
    boxes, score, cls, keep = (t.cpu().numpy() for t in model.detect(xd))
    want = oracle.decode_nms(y, 2, 20, 448)
    assert np.array_equal(keep, want['keep']) and np.array_equal(cls, want['cls'])
    assert np.array_equal(boxes.view(np.uint64), np.ascontiguousarray(want['boxes']).view(np.uint64))


def test_model_bf16_full_batch_properties():
    """BASELINE configs[2] at its per-GPU size (128 images, bf16), where the oracle would take minutes: size-independent
    properties.  The batch-128 launch shapes (other tile configurations than at N <= 70, K-split tails, fix-ups, the
    bf16 Dense at 128 rows) are the ones the 8-GPU bench runs per rank and that the small cases above do not reach."""
    import oracle
    from yolo_amd.engine import DeviceModel
    N = 128
    prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='random')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(0))
    model = DeviceModel(prm, C=20, device=DEV, dtype='bf16')
    xd = x.to(DEV)
    y = model.predict(xd).clone()
    ws = model.workspace(N)
    z = (ws['part'].sum(0)[:, :model.units] + model.dense_b).clone()
    # determinism: fixed-order reductions, a second run gives the same bits
    assert torch.equal(model.predict(xd), y)
    # three images of the batch against the oracle that rounds where the build rounds (stated bf16 tolerance, as in
    # test_model_bf16_end_to_end: Dense logits within 3 % of their RMS; head outputs within 0.08 of the fp32 oracle)
    pick = [0, 63, 127]
    xs = x[pick].numpy()
    feats_a = model_ref.backbone_ref(xs, prm, bf16=True)
    z_a = model_ref.dense_ref(feats_a, prm['dense_w'], prm['dense_b'], bf16=True)
    rms = float(np.sqrt((z_a ** 2).mean()))
    err = np.abs(z[pick].cpu().numpy() - z_a).max() / rms
    print(f'N=128 bf16: logits rms {rms:.3f}, max err {err:.3e} rms')
    assert err < 3e-2
    y_b = model_ref.model_ref(xs, prm, 7, 2, 20)
    assert np.abs(y[pick].cpu().numpy() - y_b).max() < 0.08
    # batch independence: position in the batch and batch size only change fp32 summation order (K-split plans) and,
    # through that, a bf16 rounding here and there: the same tolerance as against the oracle
    for i in (0, 17, 127):
        model.predict(xd[i:i + 1].contiguous())
        zi = model.workspace(1)['part'].sum(0)[0, :model.units] + model.dense_b
        assert (zi - z[i]).abs().max().item() / rms < 3e-2
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(DEV)
    yp = model.predict(xd[perm].contiguous())
    assert (yp - y[perm]).abs().max().item() < 0.02
    # detections at full batch: bit-exact against the oracle fed the GPU's own scores
    y2 = model.predict(xd).cpu().numpy()
    boxes, score, cls, keep = (t.cpu().numpy() for t in model.detect(xd))
    want = oracle.decode_nms(y2, 2, 20, 448)
    assert np.array_equal(keep, want['keep']) and np.array_equal(cls, want['cls'])
    assert np.array_equal(score.view(np.uint32), want['score'].view(np.uint32))
    assert np.array_equal(boxes.view(np.uint64), np.ascontiguousarray(want['boxes']).view(np.uint64))


@pytest.mark.parametrize('N,units', [(1, 1470), (64, 1470), (128, 539), (33, 100)])
def test_dense_bf16_vs_oracle(N, units):
    """ty1_dense_splitk_bf16 on bf16 operands against the fp32 matmul of the same rounded values: the products of
    two bf16 numbers are exact in fp32, so only the summation order differs (1e-3 relative of the row scale)."""
    rng = np.random.default_rng(N + units)
    K = 12544
    x = bf16_round(rng.random((N, K)).astype(np.float32))
    w = bf16_round((rng.standard_normal((K, units)) * 0.02).astype(np.float32))
    wt = P.dense_to_device_layout(w)
    part = ops.dense_splitk(dev(x, torch.bfloat16), dev(wt, torch.bfloat16), units)
    z = part.sum(0)[:, :units].cpu().numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64)
    np.testing.assert_allclose(z, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
