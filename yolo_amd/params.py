"""Host-side parameter handling: Keras-layout parameter dicts (NumPy) and their device layouts.

Keras layouts (what /root/reference/models/model_tiny_yolov1.py creates and convert.py writes):
  conv{i}_w (3,3,Cin,Cout) HWIO, bn{i}_{gamma,beta,mean,var} (Cout,), dense_w (12544, units), dense_b (units,)
Device layouts (what libty1.so consumes): conv OHWI (Cout,3,3,Cin), BN folded to scale/shift with
eps 1e-3, Dense weight transposed and row-padded: (round_up(units,32), K).
"""
import numpy as np

CONV_CHANNELS = [(3, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 256)]
N_POOL = 6            # max-pool after the first six conv blocks (model_tiny_yolov1.py:47-77)
FEATURES = 7 * 7 * 256
BN_EPS = 1e-3
LEAKY_ALPHA = 0.1


def glorot_uniform(rng, shape, fan_in, fan_out):
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def make_random_params(seed=1, S=7, B=2, C=20, dense_init='glorot', bn='identity'):
    """Synthetic weights.  dense_init: 'glorot' (Keras default; near-uniform scores, the
    NMS stress distribution) or 'spread' (N(0,.05) weights, N(0,1) bias: logits spread out).
    bn: 'identity' (Keras init) or 'random' (exercises the fold)."""
    rng = np.random.default_rng(seed)
    p = {}
    for i, (cin, cout) in enumerate(CONV_CHANNELS):
        p[f'conv{i}_w'] = glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)
        if bn == 'identity':
            p[f'bn{i}_gamma'] = np.ones(cout, np.float32)
            p[f'bn{i}_beta'] = np.zeros(cout, np.float32)
            p[f'bn{i}_mean'] = np.zeros(cout, np.float32)
            p[f'bn{i}_var'] = np.ones(cout, np.float32)
        else:
            # keep activations O(1) through 8 layers: glorot conv shrinks the signal, gamma restores it
            p[f'bn{i}_gamma'] = rng.uniform(1.5, 2.5, cout).astype(np.float32)
            p[f'bn{i}_beta'] = rng.uniform(-0.2, 0.2, cout).astype(np.float32)
            p[f'bn{i}_mean'] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
            p[f'bn{i}_var'] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    units = S * S * (C + 5 * B)
    if dense_init == 'glorot':
        p['dense_w'] = glorot_uniform(rng, (FEATURES, units), FEATURES, units)
        p['dense_b'] = np.zeros(units, np.float32)
    else:
        p['dense_w'] = (0.05 * rng.standard_normal((FEATURES, units))).astype(np.float32)
        p['dense_b'] = rng.standard_normal(units).astype(np.float32)
    return p


def hwio_to_ohwi(w):
    return np.ascontiguousarray(np.transpose(w, (3, 0, 1, 2)))


def dense_to_device_layout(w, row_pad=32):
    """(K, units) Keras kernel -> (round_up(units,row_pad), K) transposed, zero rows appended."""
    K, units = w.shape
    rows = (units + row_pad - 1) // row_pad * row_pad
    wt = np.zeros((rows, K), np.float32)
    wt[:units] = w.T
    return wt


def dense_from_device_layout(wt, units):
    return np.ascontiguousarray(wt[:units].T)
