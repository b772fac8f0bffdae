"""GPU: the entry point that allocates nothing (ty1_conv3x3_bn_leaky_static_f32, include/ty1.h) against the existing
entry point it is a variant of (ty1_conv3x3_bn_leaky_ws_f32 with a caller-provided workspace): same kernels, same
K-split plan, so the outputs must agree bit for bit.  Two calls in a row share the library's static scratch array, the
second with a smaller input: stale partial tiles of the first call must not leak into it.  Collected last on purpose
(pytest -x at round end stops at the first failure)."""
import ctypes

import numpy as np
import pytest
import torch

from yolo_amd import _lib, ops, params as P

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _problem(N, H, cin, cout, seed):
    rng = np.random.default_rng(seed)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    x = d(rng.standard_normal((N, H, H, cin)).astype(np.float32))
    w = d(P.hwio_to_ohwi(P.glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)))
    scale = d(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = d(rng.uniform(-0.5, 0.5, cout).astype(np.float32))
    return x, w, scale, shift


def _both(N, H, cin, cout, pool, seed):
    x, w, scale, shift = _problem(N, H, cin, cout, seed)
    need = ops.conv3x3_workspace_bytes(N, H, H, cin, cout, pool)
    assert 0 < need <= ops.static_scratch_bytes(), need          # the case must exercise the scratch buffer
    ws = torch.full((need // 4,), float('nan'), dtype=torch.float32, device=DEV)
    want = ops.conv3x3_bn_leaky(x, w, scale, shift, pool, workspace=ws)
    oshape = want.shape
    got = torch.full(oshape, float('nan'), dtype=torch.float32, device=DEV)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    ops.conv3x3_bn_leaky_static(x, w, scale, shift, pool, 0.1, got)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    return got, want, free0 - free1


def test_static_scratch_entry_point_matches_workspace_entry_point():
    assert ops.static_scratch_bytes() == 16 << 20
    # first call: 294 tiles of 128x128 on 256 CUs -> 38 tail tiles cut along K (pooled epilogue in the fix-up)
    got, want, _ = _both(24, 28, 128, 256, True, seed=1)
    assert torch.isfinite(want).all() and torch.equal(got, want)
    # second call right after it, smaller input, other tile shape (64x64, unpooled, all tail): shares the same array
    got2, want2, leaked = _both(12, 7, 1024, 256, False, seed=2)
    assert torch.isfinite(want2).all() and torch.equal(got2, want2)
    assert leaked == 0                                   # device memory in use did not move across the call
    # and the first problem once more: results do not depend on what the array held
    got3, want3, leaked = _both(24, 28, 128, 256, True, seed=1)
    assert torch.equal(got3, want) and leaked == 0
    # layers that never split (first layer, Cin = 16) go through the same entry point without touching the array
    x, w, scale, shift = _problem(2, 32, 3, 16, 3)
    a = ops.conv3x3_bn_leaky(x, w, scale, shift, True)
    b = ops.conv3x3_bn_leaky_static(x, w, scale, shift, True, 0.1, torch.empty_like(a))
    assert torch.equal(a, b)


def test_static_scratch_entry_point_under_graph_capture():
    """The callers this entry point is for: after ty1_static_scratch_init() and one eager call (which loads the kernel and
    runs the one-time occupancy query of its tile configuration), the call is captured in a HIP graph and replayed on new
    input; the replays give the eager call's bits."""
    ops.static_scratch_init()
    N, H, cin, cout, pool = 12, 7, 1024, 256, False
    x, w, scale, shift = _problem(N, H, cin, cout, seed=5)
    assert ops.conv3x3_workspace_bytes(N, H, H, cin, cout, pool) > 0
    out = torch.empty((N, H, H, cout), dtype=torch.float32, device=DEV)
    eager = ops.conv3x3_bn_leaky_static(x, w, scale, shift, pool, 0.1, out).clone()
    torch.cuda.synchronize()
    out.fill_(float('nan'))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.conv3x3_bn_leaky_static(x, w, scale, shift, pool, 0.1, out)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    x2, _, _, _ = _problem(N, H, cin, cout, seed=6)
    want2 = ops.conv3x3_bn_leaky_static(x2, w, scale, shift, pool, 0.1, torch.empty_like(out))
    x.copy_(x2)                      # the graph reads the buffers it was captured with
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want2)


def test_static_scratch_entry_point_rejects_what_does_not_fit():
    # the network's last layer at batch 128 wants 20 MB of partial tiles (every layer at batch 64 fits)
    assert ops.conv3x3_workspace_bytes(128, 7, 7, 1024, 256, False) > ops.static_scratch_bytes()
    assert ops.conv3x3_workspace_bytes(64, 7, 7, 1024, 256, False) <= ops.static_scratch_bytes()
    L = _lib.lib()
    y = torch.zeros((1,), device=DEV)
    # the size check comes before any pointer is used or any kernel is launched
    rc = L.ty1_conv3x3_bn_leaky_static_f32(y.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr(), 128, 7, 7, 1024, 256, 0,
                                           ctypes.c_float(0.1), None)
    msg = L.ty1_last_error().decode()
    assert rc != 0 and 'static buffer holds 16777216' in msg and 'ty1_conv3x3_bn_leaky_ws_f32' in msg
    torch.cuda.synchronize()
