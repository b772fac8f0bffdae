#!/usr/bin/env python3
"""Developer tool: per-workgroup timeline of one conv launch (start / prologue / loop / end stamps, CU placement)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='64,16,512,1024,0')
ap.add_argument('--persist', action='store_true', help='stamps written by conv_persist_kernel')
ap.add_argument('--direct', action='store_true', help='stamps written by the first-layer persistent kernel (TY1_DEV_STAMPS build)')
ap.add_argument('--halo', action='store_true', help='stamps written by conv_halo_bf16_kernel (bf16 layer, dev build): where wave 0 of every workgroup spends its cycles')
ap.add_argument('--launches', type=int, default=30)
args = ap.parse_args()
n, hh, cin, cout, pool = (int(v) for v in args.shape.split(','))
dev = 'cuda:0'
if args.halo:
    xb = torch.rand((n, hh, hh, cin), device=dev).to(torch.bfloat16)
    wb = (torch.randn((cout, 3, 3, cin), device=dev) * 0.05).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    out = ops.conv3x3_bn_leaky_bf16(xb, wb, sc, sh, bool(pool))
    cap = 1 << 12
    buf = torch.zeros((cap, 8), dtype=torch.int64, device=dev)
    for _ in range(200):
        ops.conv3x3_bn_leaky_bf16(xb, wb, sc, sh, bool(pool), out=out)
    torch.cuda.synchronize()
    _lib.call('ty1_debug_stamp_buffer', buf.data_ptr(), cap)
    for _ in range(args.launches):
        ops.conv3x3_bn_leaky_bf16(xb, wb, sc, sh, bool(pool), out=out)
    torch.cuda.synchronize()
    _lib.call('ty1_debug_stamp_buffer', None, 0)
    b = buf.cpu().numpy()
    b = b[b[:, 0] != 0]
    t0 = b[:, 0].min()
    st, end = (b[:, 0] - t0) / 100.0, (b[:, 3] - t0) / 100.0
    ns = np.maximum(b[:, 7], 1).astype(np.float64)
    tot = b[:, 1] + b[:, 2] + b[:, 4]
    nw = 8 if len(b) % 8 == 0 and len(b) >= 2048 else 4
    print(f'waves {len(b)} ({nw} per workgroup)  span {end.max():.1f} us  wave time {np.mean(end - st):.1f} us  stages/wave {ns.mean():.0f}')
    for w in range(nw):
        r = b[w::nw]; n_ = ns[w::nw]; t_ = tot[w::nw]
        print(f'wave {w}: per stage (shader cycles): issue of fetch + MFMA steps {np.mean(r[:, 1] / n_):6.0f}  vmcnt wait {np.mean(r[:, 2] / n_):5.0f}  '
              f'lgkmcnt + barrier {np.mean(r[:, 4] / n_):5.0f}  sum {np.mean(t_ / n_):6.0f}   epilogues {np.mean(r[:, 5]):6.0f}')
    print(f'shader clock (stamped cycles / wave time) {np.mean((tot + b[:, 5]) / np.maximum(end - st, 1e-9)) / 1e3:.3f} GHz')
    sys.exit(0)
x = torch.rand((n, hh, hh, cin), device=dev)
w = torch.randn((cout, 3, 3, cin), device=dev) * 0.05
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
nws = ops.conv3x3_workspace_bytes(n, hh, hh, cin, cout, pool)
ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
out = ops.conv3x3_bn_leaky(x, w, sc, sh, bool(pool), workspace=ws)
for _ in range(3):
    ops.conv3x3_bn_leaky(x, w, sc, sh, bool(pool), out=out, workspace=ws)
cap = 1 << 16
buf = torch.zeros((cap, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
_lib.call('ty1_debug_stamp_buffer', buf.data_ptr(), cap)
for _ in range(args.launches):                      # back to back: the last launch's stamps survive (sustained clocks)
    ops.conv3x3_bn_leaky(x, w, sc, sh, bool(pool), out=out, workspace=ws)
torch.cuda.synchronize()
_lib.call('ty1_debug_stamp_buffer', None, 0)
ball = buf.cpu().numpy()
b = ball[:4096]                      # block rows; rows 4096... hold the per-item clock trace of blocks 0-7
b = b[b[:, 0] != 0]
nb = len(b)
t0 = b[:, 0].min()
if args.direct:
    st, end = (b[:, 0] - t0) / 100.0, (b[:, 3] - t0) / 100.0
    ni = np.maximum(b[:, 7], 1).astype(np.float64)
    sto, bar = (b[:, 6] & 0xFFFFFFFF), (b[:, 6] >> 32)
    print(f'persistent blocks {len(b)}  span {end.max():.1f} us  patches/block min {ni.min():.0f} max {ni.max():.0f}')
    print(f'per patch (s_memtime ticks): fetch-issue {np.mean(b[:, 2] / ni):.0f}  mfma+epilogue {np.mean(b[:, 1] / ni):.0f}  '
          f'lds-store {np.mean(sto / ni):.0f}  barrier {np.mean(bar / ni):.0f}   block time {np.mean(end - st):.1f} us -> {np.mean((end - st) / ni):.2f} us per patch')
    print('end-time histogram:', np.histogram(end, bins=8)[0].tolist(), f'first {end.min():.1f} last {end.max():.1f}')
    print('start-time histogram:', np.histogram(st, bins=8)[0].tolist(), f'first {st.min():.1f} last {st.max():.1f}')
    sys.exit(0)
if args.persist:
    st, end = (b[:, 0] - t0) / 100.0, (b[:, 3] - t0) / 100.0
    ni = np.maximum(b[:, 7], 1).astype(np.float64)
    print(f'persistent blocks {len(b)}  span {end.max():.1f} us  items/block min {ni.min():.0f} max {ni.max():.0f}')
    print(f'per item (shader cycles): loop {np.mean(b[:, 1] / ni):.0f}  epilogue {np.mean(b[:, 2] / ni):.0f}  rows {np.mean(b[:, 6] / ni):.0f}'
          f'   block time {np.mean(end - st):.1f} us -> {np.mean((end - st) / ni):.2f} us per item')
    print(f'shader clock while a block runs (stamped cycles / block time): {np.mean((b[:, 1] + b[:, 2] + b[:, 6]) / np.maximum(end - st, 1e-9)) / 1e3:.3f} GHz')
    tr = ball[4096:4096 + 64]
    tr = tr[tr[:, 1] != 0]
    if len(tr) > 2:
        ck = np.diff(tr[:, 0]) / np.diff(tr[:, 1]) * 0.1
        print('block 0: item end times (us)', np.round((tr[1:, 1] - t0) / 100.0).astype(int).tolist())
        print('block 0: shader clock per item (GHz)', np.round(ck, 2).tolist())
    print('end-time histogram:', np.histogram(end, bins=8)[0].tolist(), f'first {end.min():.1f} last {end.max():.1f}')
    hw = b[:, 4]; xcc = b[:, 5] & 0xF
    cuid = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)
    ids = np.nonzero(ball[:4096, 0] != 0)[0]
    groups = {}
    for bi, c in zip(ids, cuid):
        groups.setdefault(int(c), []).append(int(bi))
    diffs = [g[1] - g[0] for g in groups.values() if len(g) == 2]
    print('CUs', len(groups), 'blocks per CU', np.bincount([len(g) for g in groups.values()]).tolist(),
          'index distance of CU partners:', dict(zip(*[x.tolist() for x in np.unique(diffs, return_counts=True)])))
    print('xcc of blocks 0..15:', xcc[:16].tolist())
    print('mean end time per XCD (us):', [round(float(end[xcc == k].mean()), 1) for k in range(8)])
    cu_end = {c: max(end[ids.tolist().index(bi)] for bi in g) for c, g in groups.items()}
    by_n = {}
    for c, g in groups.items():
        by_n.setdefault(len(g), []).append(cu_end[c])
    print('CU end time by resident blocks:', {k: (len(v), round(float(np.mean(v)), 1), round(float(np.min(v)), 1), round(float(np.max(v)), 1)) for k, v in sorted(by_n.items())})
    cyc = (b[:, 1] + b[:, 2] + b[:, 6])
    print('stamped cycles per block: mean %.0f min %.0f max %.0f; block time mean %.1f min %.1f max %.1f us' % (cyc.mean(), cyc.min(), cyc.max(), (end - st).mean(), (end - st).min(), (end - st).max()))
    sys.exit(0)
sys.exit('pass --persist (conv_persist_kernel layers) or --direct (first layer, TY1_DEV_STAMPS build)')
