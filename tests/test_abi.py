"""CPU: the C-ABI library loads and exports every symbol include/ty1.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT
from yolo_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, 'include', 'ty1.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ty1_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 16
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f'{n} declared in ty1.h but not exported by libty1.so'
    bound = set(_lib.SIGNATURES) | {'ty1_last_error'}
    assert set(names) == bound, set(names) ^ bound


def test_version_and_error_channel():
    L = _lib.lib()
    assert L.ty1_version() == 1
    # argument validation happens before any GPU work: a bad call fails with a message, no crash
    rc = L.ty1_decode_nms(None, 1, 9, 2, 20, 448, 0.2, None, None, None, None, None)
    assert rc != 0 and b'exceeds one wavefront' in L.ty1_last_error()
    rc = L.ty1_conv3x3_bn_leaky_f32(None, None, None, None, None, 1, 8, 8, 32, 32, 1, 0.1, None)
    assert rc != 0 and b'null pointer' in L.ty1_last_error()
    rc = L.ty1_conv3x3_bn_leaky_f32(None, None, None, None, None, 1, 7, 7, 32, 32, 1, 0.1, None)
    assert rc != 0 and b'pool needs even' in L.ty1_last_error()


def test_conv_workspace_query_is_host_only():
    """The K-split plan is pure host arithmetic: sizes for the model's layers at batch 64 (256 CUs assumed
    when no GPU is visible), 0 where the tile count already divides evenly or for the Cin=3 layer."""
    L = _lib.lib()
    n = ctypes.c_size_t(123)
    assert L.ty1_conv3x3_workspace_bytes(64, 448, 448, 3, 16, 1, ctypes.addressof(n)) == 0 and n.value == 0
    assert L.ty1_conv3x3_workspace_bytes(64, 224, 224, 16, 32, 1, ctypes.addressof(n)) == 0 and n.value == 0   # 12544 tiles
    assert L.ty1_conv3x3_workspace_bytes(64, 28, 28, 128, 256, 1, ctypes.addressof(n)) == 0
    assert 0 < n.value <= 64 << 20 and n.value % (128 * 128 * 4) == 0                                          # 784 tiles of 128x128
    assert L.ty1_conv3x3_workspace_bytes(64, 7, 7, 1024, 256, 0, ctypes.addressof(n)) == 0 and 0 < n.value <= 64 << 20
    assert L.ty1_conv3x3_workspace_bytes(64, 7, 7, 1024, 250, 0, ctypes.addressof(n)) != 0                      # Cout % 16


def test_fold_bn_host_helper():
    from yolo_amd import ops
    rng = np.random.default_rng(0)
    g, b, m = (rng.standard_normal(16).astype(np.float32) for _ in range(3))
    v = rng.random(16).astype(np.float32) + 0.1
    scale, shift = ops.fold_bn(g, b, m, v, 1e-3)
    np.testing.assert_allclose(scale, g / np.sqrt(v + np.float32(1e-3)), rtol=1e-6)
    np.testing.assert_allclose(shift, b - m * scale, rtol=1e-5, atol=1e-6)
