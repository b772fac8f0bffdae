#!/usr/bin/env python3
"""Developer tool: time every conv layer of the model alone (batch N) and print TFLOP/s per layer."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import ops, params as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--layers', default='0,1,2,3,4,5,6,7')
ap.add_argument('--prof', action='store_true', help='also print the per-kernel HIP-event table')
ap.add_argument('--no-ws', action='store_true', help='run without the tail-split workspace')
ap.add_argument('--bf16', action='store_true', help='the bf16 network layers (ty1_conv3x3_bn_leaky_bf16)')
ap.add_argument('--shapes', default='', help='extra cases "N,H,cin,cout,pool;..."')
args = ap.parse_args()
dev = 'cuda:0'
if args.prof:
    kt = ops.KernelTimer().__enter__()
h = 448
tot = 0.0
for i, (cin, cout) in enumerate(P.CONV_CHANNELS):
    pool = i < P.N_POOL
    if str(i) in args.layers.split(','):
        x = torch.rand((args.batch, h, h, cin), device=dev)
        w = torch.randn((cout, 3, 3, cin), device=dev) * 0.05
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        if args.bf16:
            if cin != 3:
                x, w = x.to(torch.bfloat16), w.to(torch.bfloat16)
            ws = None if args.no_ws else torch.empty(max(16, ops.conv3x3_workspace_bytes_bf16(args.batch, h, h, cin, cout, pool)), dtype=torch.uint8, device=dev)
            run = lambda out=None: ops.conv3x3_bn_leaky_bf16(x, w, sc, sh, pool, out_f32=(i == 7), out=out, workspace=ws)
        else:
            ws = None if args.no_ws else torch.empty(max(16, ops.conv3x3_workspace_bytes(args.batch, h, h, cin, cout, pool)), dtype=torch.uint8, device=dev)
            run = lambda out=None: ops.conv3x3_bn_leaky(x, w, sc, sh, pool, out=out, workspace=ws)
        out = run()
        for _ in range(3):
            run(out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run(out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        fl = 2.0 * args.batch * h * h * 9 * cin * cout
        tot += us
        print(f'L{i} {h}x{h} {cin}->{cout} pool={int(pool)}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s', flush=True)
    if pool:
        h //= 2
for spec in [t for t in args.shapes.split(';') if t]:
    n, hh, cin, cout, pool = (int(v) for v in spec.split(','))
    x = torch.rand((n, hh, hh, cin), device=dev)
    w = torch.randn((cout, 3, 3, cin), device=dev) * 0.05
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    ws = None if args.no_ws else torch.empty(max(16, ops.conv3x3_workspace_bytes(n, hh, hh, cin, cout, pool)), dtype=torch.uint8, device=dev)
    out = ops.conv3x3_bn_leaky(x, w, sc, sh, bool(pool), workspace=ws)
    for _ in range(3):
        ops.conv3x3_bn_leaky(x, w, sc, sh, bool(pool), out=out, workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.conv3x3_bn_leaky(x, w, sc, sh, bool(pool), out=out, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    fl = 2.0 * n * hh * hh * 9 * cin * cout
    print(f'shape {spec}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s', flush=True)
if tot:
    print(f'total {tot:.1f} us -> {args.batch / tot * 1e6:.0f} img/s (conv only)')
if args.prof:
    for k, v in sorted(ops.KernelTimer.table().items()):
        print(f"  {k:44s} n={v['launches']:3d} avg {1e3 * v['ms'] / max(v['launches'], 1):8.1f} us")
