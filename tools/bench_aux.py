#!/usr/bin/env python3
"""Auxiliary measurement for BASELINE.json configs[4] (the other configs are bench.py lines: default = configs[1],
`--dtype bf16` = configs[2], `--mode train` = configs[3]):
  --nms      C=1 head, batch 4096, dense candidates: decode + threshold + NMS stress
"""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--nms', action='store_true')
ap.add_argument('--steps', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda', 0)

if args.nms:
    N, C = 4096, 1
    rng = np.random.default_rng(2)
    p = np.empty((N, 7, 7, 11), np.float32)
    p[..., 0] = 1.0
    p[..., 1:] = rng.random((N, 7, 7, 10)).astype(np.float32)
    pd = torch.from_numpy(p).to(dev)
    out = ops.decode_nms(pd, 2, C, 448)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        ops.decode_nms(pd, 2, C, 448, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.steps
    nbytes = N * (49 * 11 * 4 + 98 * (32 + 4 + 4 + 1))
    print(json.dumps({'config': 'configs[4] C=1 batch 4096 decode+NMS', 'us': round(us, 1), 'images_per_sec': round(N / us * 1e6),
                      'kept_per_image': float(out[3].sum().item()) / N, 'algo_GBps': round(nbytes / us / 1e3, 1)}))
