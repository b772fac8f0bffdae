#!/usr/bin/env python3
"""The training tail's kernels alone at 32 images (Dense forward, train_tail, wgrad, Adam): us per launch, HIP events around
200 back-to-back launches.  TY1_LIB selects the library (tools/build_head_lib.sh for the old source)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from yolo_amd import ops, params as P
from yolo_amd.engine import DeviceModel, Trainer
dev = torch.device('cuda', 0)
N = int(os.environ.get('N', '32'))
prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity')
model = DeviceModel(prm, device=dev)
x = bench.synthetic_images(N, 100, dev).to(dev)
yd = torch.from_numpy(bench.make_labels(N, 20, 3)).to(dev)
tr = Trainer(model, overlap=False)
feats = model.backbone(x).clone()
ws = model.workspace(N)
tr.step(x, yd)
dz = tr._dz[N]
def t(name, fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:12} {e0.elapsed_time(e1) / n * 1e3:7.2f} us')
for rep in range(2):
    t('dense', lambda: ops.dense_splitk(feats, model.dense_wt, model.units, part=ws['part']))
    t('train_tail', lambda: ops.train_tail(ws['part'], model.dense_b, yd, tr.ldz, dz=dz, loss_out=tr.bucket.loss))
    t('wgrad', lambda: ops.dense_wgrad(dz, feats, model.units, tr.dwt, tr.db))
    t('adam', lambda: tr._adam())
