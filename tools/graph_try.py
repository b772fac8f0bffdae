#!/usr/bin/env python3
"""Developer experiment: the whole detect step as one HIP graph (torch.cuda.CUDAGraph capture of the library launches)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import params as P
from yolo_amd.engine import DeviceModel
dev = torch.device('cuda', 0)
N = int(os.environ.get("N", "64"))
prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity')
model = DeviceModel(prm, C=20, device=dev)
x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(3):
    det = model.detect(x)
torch.cuda.synchronize()
def timeit(fn, steps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
print('eager  ms/step', [round(timeit(lambda: model.detect(x)), 4) for _ in range(3)])
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model.detect(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = model.detect(x)
torch.cuda.synchronize()
ref = [t.clone() for t in model.detect(x)]
g.replay(); torch.cuda.synchronize()
print('graph == eager:', all(torch.equal(a, b) for a, b in zip(out, ref)))
print('graph  ms/step', [round(timeit(g.replay), 4) for _ in range(3)])
print('eager  ms/step', [round(timeit(lambda: model.detect(x)), 4) for _ in range(3)])
