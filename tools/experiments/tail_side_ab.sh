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
xg = {g: bench.synthetic_images(g * N, 100, dev).to(dev) for g in (2, 4)}
yg = {g: torch.from_numpy(bench.make_labels(g * N, 20, 3)).to(dev) for g in (2, 4)}
def run_group(g, steps=200):
    tr.tail_on_side, tr.update_after_layer = True, 2
    xx, yy = (x, yd) if g == 1 else (xg[g], yg[g])
    for _ in range(6):
        tr.steps(xx, yy, N, wait=False)
    tr.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // g):
        tr.steps(xx, yy, N, wait=False)
    tr.join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (steps // g * g) * 1e3
for g in (1, 2, 4, 1, 2, 4):
    m = run_group(g)
    print(f'backbone once per {g} batches of {N}: {m:.4f} ms per optimisation step  {N / m * 1e3:.0f} images/s')
PY
