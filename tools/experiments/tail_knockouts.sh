# developer tool (one GPU call): what each kernel of the side stream costs the training step (32 images) when it runs under the
# next batch's backbone: the step with launches knocked out one after the other (results are wrong, times are what is asked)
set -e
python - <<'PY'
import statistics, sys, time, torch
sys.path.insert(0, '.')
import bench
from yolo_amd import ops, params as P
from yolo_amd.engine import DeviceModel, Trainer
dev = torch.device('cuda', 0)
N = 32
prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity')
model = DeviceModel(prm, device=dev)
x = bench.synthetic_images(N, 100, dev).to(dev)
yd = torch.from_numpy(bench.make_labels(N, 20, 3)).to(dev)
tr = Trainer(model, overlap=True)
tr.update_after_layer = 2
real = dict(adam=ops.adam_step, wgrad=ops.dense_wgrad, tail=ops.train_tail, dense=ops.dense_splitk)
part = model.workspace(N)['part']
def set_ko(ko):
    ops.adam_step = (lambda *a, **k: None) if 'adam' in ko else real['adam']
    ops.dense_wgrad = (lambda *a, **k: None) if 'wgrad' in ko else real['wgrad']
    ops.train_tail = (lambda *a, **k: None) if 'tail' in ko else real['tail']
    ops.dense_splitk = (lambda *a, **k: part) if 'dense' in ko else real['dense']
def run(ko, steps=200):
    set_ko(ko)
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
cases = [(), ('adam',), ('wgrad',), ('dense',), ('tail',), ('adam', 'wgrad'), ('adam', 'wgrad', 'dense'), ('adam', 'wgrad', 'dense', 'tail')]
res = {c: [] for c in cases}
for _ in range(3):
    for c in cases:
        res[c].append(run(c))
# the plumbing that is left when all four are gone: label copy, loss clone, events
set_ko(('adam', 'wgrad', 'dense', 'tail'))
ws = model.workspace(N)
def plumbing(name, steps=200):
    for _ in range(10):
        tr.step(x, yd, wait=False)
    tr.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, yd, wait=False)
    tr.join(); torch.cuda.synchronize()
    print(f'all four gone, {name:40}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms')
plumbing('as is')
real_back, real_gate = tr._take_features_back, tr._launch_deferred_adam
def ungated():
    if tr._adam_deferred:
        tr._adam_deferred = False
        with torch.cuda.stream(tr.side):
            tr._adam()
            tr._updated.record(tr.side)
tr._take_features_back = lambda: None
plumbing('no wait for the features (gate stays)')
tr._take_features_back, tr._launch_deferred_adam = real_back, ungated
plumbing('no gate (wait stays)')
tr._take_features_back = lambda: None
plumbing('no wait for the features, no gate')
tr._take_features_back, tr._launch_deferred_adam = real_back, real_gate
for ual in (0, 1, 3, 4, 5, 6, 2):
    tr.update_after_layer = ual
    plumbing(f'as is, Adam gate after layer {ual}')
real_clone = torch.Tensor.clone
real_copy = torch.Tensor.copy_
torch.Tensor.copy_ = lambda self, other, *a, **k: self
plumbing('no label copy')
tr._take_features_back, tr._launch_deferred_adam = (lambda: None), ungated
plumbing('no label copy, no wait, no gate')
torch.Tensor.clone = lambda self, *a, **k: self
tr.bucket.loss.record_stream = lambda s: None
plumbing('... and no loss clone')
torch.Tensor.clone, torch.Tensor.copy_ = real_clone, real_copy
tr._take_features_back, tr._launch_deferred_adam = real_back, real_gate
t0 = time.perf_counter()
for _ in range(200):
    model.backbone(x, ws)
torch.cuda.synchronize()
print(f'backbone alone: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms')
base = statistics.median(res[()])
for c in cases:
    m = statistics.median(res[c])
    print(f'without {"+".join(c) or "nothing":24}: {m:.4f} ms  ({(base - m) * 1e3:+.1f} us)  {[round(v, 4) for v in res[c]]}')
PY
