"""The model oracle (oracle/model_ref.py) has no reference-side pin: TensorFlow is absent and the
reference ships no weights or fixtures for the model ("parity unpinned").
These tests cross-check the torch-CPU restatement against an independent NumPy restatement of
the same reference lines (/root/reference/models/model_tiny_yolov1.py:14-99) written with plain
array arithmetic in float64."""
import numpy as np
import pytest

from oracle import model_ref


def _block_np(x_hwc, w, gamma, beta, mean, var, pool):
    """Conv2D 3x3 same no-bias -> BatchNormalization (eps 1e-3) -> LeakyReLU(0.1) -> MaxPool 2x2/2."""
    y = model_ref.conv3x3_naive_np(x_hwc, w)
    y = (y - mean) / np.sqrt(var.astype(np.float64) + 1e-3) * gamma + beta
    y = np.where(y > 0, y, 0.1 * y)
    if pool:
        H, W, Cc = y.shape
        y = y.reshape(H // 2, 2, W // 2, 2, Cc).max(axis=(1, 3))
    return y


@pytest.mark.parametrize('cin,cout,pool,hw', [(3, 16, True, 12), (16, 32, True, 8), (32, 16, False, 7)])
def test_conv_block_against_naive_numpy(cin, cout, pool, hw):
    import torch
    rng = np.random.default_rng(cin * 100 + cout)
    x = rng.standard_normal((2, hw, hw, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.2).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    mean, var = rng.standard_normal(cout).astype(np.float32), rng.uniform(0.2, 2.0, cout).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    got = model_ref.conv_block_ref(xt, w, gamma, beta, mean, var, pool).permute(0, 2, 3, 1).numpy()
    for n in range(2):
        want = _block_np(x[n].astype(np.float64), w, gamma, beta, mean, var, pool)
        np.testing.assert_allclose(got[n], want, rtol=1e-4, atol=1e-5)


def test_head_activate_against_numpy_planar_split():
    """ReshapeYOLO.call: [0:980]->(7,7,20) softmax, [980:1078]->(7,7,2) sigmoid, rest->(7,7,8) sigmoid."""
    rng = np.random.default_rng(5)
    for S, B, C in ((7, 2, 20), (7, 2, 1)):
        z = rng.standard_normal((3, S * S * (C + 5 * B))).astype(np.float32) * 3
        got = model_ref.head_activate_ref(z, S, B, C)
        i1, i2 = S * S * C, S * S * (C + B)
        cls = z[:, :i1].reshape(3, S, S, C).astype(np.float64)
        e = np.exp(cls - cls.max(-1, keepdims=True))
        cls = e / e.sum(-1, keepdims=True)
        sig = lambda a: 1.0 / (1.0 + np.exp(-a.astype(np.float64)))
        want = np.concatenate([cls, sig(z[:, i1:i2].reshape(3, S, S, B)), sig(z[:, i2:].reshape(3, S, S, 4 * B))], -1)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
        assert got.shape == (3, S, S, C + 5 * B)


@pytest.mark.parametrize('name,C', [('head_c20.npz', 20), ('head_c1.npz', 1)])
def test_head_activate_matches_reference_goldens(name, C):
    """PINNED: tests/golden/head_c*.npz hold the output of the reference's own ReshapeYOLO.call
    (/root/reference/models/model_tiny_yolov1.py:14-34, C=1 twin /root/reference/faces/trainFace.py:38-58) run by
    tests/golden/make_golden.py on seeded Dense outputs, including saturating logits."""
    from conftest import load_golden
    g = load_golden(name)
    assert int(g['C']) == C
    got = model_ref.head_activate_ref(g['z'], 7, 2, C)
    assert got.shape == g['y'].shape == (g['z'].shape[0], 7, 7, C + 10)
    np.testing.assert_allclose(got, g['y'], rtol=1e-6, atol=1e-30)
    assert np.isfinite(g['y']).all()                       # the extremes did not overflow the reference's softmax
    if C > 1:
        np.testing.assert_allclose(g['y'][..., :C].sum(-1), 1.0, rtol=1e-5)
    else:
        assert (g['y'][..., 0] == 1.0).all()               # softmax over one class


def test_flatten_is_hwc_and_dense_is_x_at_w():
    """Flatten of a channels-last (7,7,256) tensor is HWC order; Dense is x @ W + b."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 5)).astype(np.float32)
    w = rng.standard_normal((5, 4)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    np.testing.assert_allclose(model_ref.dense_ref(x, w, b), x.astype(np.float64) @ w + b, rtol=1e-5, atol=1e-6)


def test_bf16_restatement_rounds_where_the_build_rounds():
    """backbone_ref(bf16=True) states the build's bf16 network: image, conv weights and the activations between
    layers are bf16 values (8 significant bits) and the result stays close
    to the fp32 restatement."""
    import torch
    from yolo_amd import params as P
    prm = P.make_random_params(seed=3, dense_init='glorot', bn='random')
    x = np.random.default_rng(0).random((1, 64, 64, 3)).astype(np.float32)       # a small image: 64 -> 1x1 after 6 pools
    f32, acts32 = model_ref.backbone_ref(x, prm, return_all=True)
    b16, acts16 = model_ref.backbone_ref(x, prm, return_all=True, bf16=True)

    def is_bf16(a):
        return np.array_equal(torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy(), a)
    assert all(is_bf16(a) for a in acts16) and not is_bf16(acts32[-1])
    for a32, a16 in zip(acts32, acts16):
        assert np.abs(a16 - a32).max() <= 3e-2 * max(1.0, np.abs(a32).max())
    assert b16.shape == f32.shape
