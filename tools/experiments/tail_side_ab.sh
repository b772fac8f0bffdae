# developer tool (one GPU call): training step with the trainable tail on the side stream (default) against the round-2 order
# (tail on the compute stream), alternating in one process
set -e
python - <<'PY'
import statistics, sys, time, torch
sys.path.insert(0, '.')
import bench
from yolo_amd import params as P
from yolo_amd.engine import DeviceModel, Trainer
dev = torch.device('cuda', 0)
N = 32
prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity')
model = DeviceModel(prm, device=dev)
x = bench.synthetic_images(N, 100, dev).to(dev)
yd = torch.from_numpy(bench.make_labels(N, 20, 3)).to(dev)
tr = Trainer(model, overlap=True)
def run(side, ual, steps=200):
    tr.tail_on_side, tr.update_after_layer = side, ual
    for _ in range(10):
        tr.step(x, yd, wait=False)
    tr.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, yd, wait=False)
    tr.join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for _ in range(300):
    tr.step(x, yd, wait=False)
tr.join(); torch.cuda.synchronize()
cases = [(False, 4), (True, 4), (True, 2), (True, 1), (True, 0), (True, 3)]
res = {c: [] for c in cases}
for _ in range(4):
    for c in cases:
        res[c].append(run(*c))
def backbone_only(steps=200):
    ws = model.workspace(N)
    for _ in range(10):
        model.backbone(x, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.backbone(x, ws)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
tr.join(); torch.cuda.synchronize()
print('backbone alone:', [round(backbone_only(), 4) for _ in range(3)], 'ms')
for c in cases:
    m = statistics.median(res[c])
    print(f'tail_on_side={c[0]!s:5} update_after_layer={c[1]!s:4}: {m:.4f} ms  {N / m * 1e3:.0f} images/s   {[round(v, 4) for v in res[c]]}')
PY
