#!/usr/bin/env python3
"""dense_wgrad_kernel alone against the number of rows (images) it reduces over: 32 = one rank's batch, 64 ... 512 = what a
rank reduces over in Trainer.exchange = 'gather' with 2 ... 16 ranks.  TY1_LIB selects the library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_amd import ops
dev = torch.device('cuda', 0)
K, units, ldz = 12544, 1470, 1472
dwt = torch.zeros((1472, K), device=dev)
db = torch.zeros((units,), device=dev)
for N in (32, 64, 128, 256, 512):
    dz = torch.randn((N, ldz), device=dev)
    x = torch.randn((N, K), device=dev)
    for _ in range(10):
        ops.dense_wgrad(dz, x, units, dwt, db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.dense_wgrad(dz, x, units, dwt, db)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    ref = dz[:, :units].double().t() @ x.double()
    err = (dwt[:units].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f'rows {N:4d}: {us:7.1f} us   {2.0 * N * units * K / us / 1e6:6.1f} TFLOP/s   rel err {err:.1e}')
