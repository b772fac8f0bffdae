#!/usr/bin/env python3
"""Developer check: serial vs overlapped training step, state compared after every step."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import params as P
from yolo_amd.engine import DeviceModel, Trainer
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_api import _labels
DEV = 'cuda:0'
N = 4
prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
xs = [torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(20 + i)).to(DEV) for i in range(3)]
ys = [torch.from_numpy(_labels(N, 20, 30 + i)).to(DEV) for i in range(3)]
def run(overlap, wait, sync_each):
    model = DeviceModel(prm, device=DEV)
    tr = Trainer(model, overlap=overlap)
    snaps = []
    for x, y in zip(xs, ys):
        l = tr.step(x, y, wait=wait)
        if sync_each:
            tr.join(); torch.cuda.synchronize()
            snaps.append((l.clone(), tr.pflat.clone(), tr.m.clone(), tr.v.clone(), tr.bucket.flat.clone()))
    tr.join(); torch.cuda.synchronize()
    snaps.append((l.clone(), tr.pflat.clone(), tr.m.clone(), tr.v.clone(), tr.bucket.flat.clone()))
    return snaps
def cmp(a, b, tag):
    for i, (sa, sb) in enumerate(zip(a, b)):
        print(tag, 'snap', i, [(int((u != v).sum().item()), float((u - v).abs().max().item())) for u, v in zip(sa, sb)])
s1 = run(False, True, True); s2 = run(False, True, True)
cmp(s1, s2, 'serial/serial')
o1 = run(True, True, True)
cmp(s1, o1, 'serial/overlap(wait,sync)')
o2 = run(True, False, True)
cmp(s1, o2, 'serial/overlap(nowait,sync)')
o3 = run(True, False, False)
cmp(s1[-1:], o3, 'serial/overlap(nowait,nosync)')
s3 = run(False, True, False)
cmp(s1[-1:], s3, 'serial/serial(nosync)')
