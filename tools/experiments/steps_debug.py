import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from yolo_amd import params as P
from yolo_amd.engine import DeviceModel, Trainer
from test_gpu_api import _labels
DEV = 'cuda:0'
b, k = 8, 3
prm = P.make_random_params(seed=6, dense_init='glorot', bn='random')
xs = [torch.rand((k * b, 448, 448, 3), generator=torch.Generator().manual_seed(80 + i)).to(DEV) for i in range(2)]
ys = [torch.from_numpy(_labels(k * b, 20, 90 + i)).to(DEV) for i in range(2)]
def run(mode, inplace=False, ual=2, sync=False):
    model = DeviceModel(prm, device=DEV)
    tr = Trainer(model, lr=1e-5, overlap=(mode == 'overlap'))
    tr.update_after_layer = ual
    losses = []
    for x, y in zip(xs, ys):
        if inplace:
            xb, yb = x.clone(), y.clone()
            losses.append(tr.steps(xb, yb, b, wait=(mode != 'overlap')))
            xb.fill_(0.5); yb.zero_()
        else:
            losses.append(tr.steps(x, y, b, wait=(mode != 'overlap')))
        if sync:
            torch.cuda.synchronize()
    tr.join(); torch.cuda.synchronize()
    return torch.cat(losses).cpu(), tr.pflat.clone().cpu(), tr.m.clone().cpu()
ref = run('serial')
ref2 = run('serial')
print('serial deterministic:', [torch.equal(a, c) for a, c in zip(ref, ref2)])
for name, kw in [('overlap', {}), ('overlap inplace', dict(inplace=True)), ('overlap ual None', dict(ual=None)), ('overlap sync', dict(sync=True)), ('overlap', {})]:
    got = run('overlap', **kw)
    print(name, [torch.equal(a, c) for a, c in zip(ref, got)], 'losses', (ref[0] - got[0]).tolist(), 'n weights differ', int((ref[1] != got[1]).sum()))
