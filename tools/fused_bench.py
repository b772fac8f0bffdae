#!/usr/bin/env python3
"""Times the first two conv blocks of the bf16 network at configs[2]'s batch inside one process, alternating: one launch
(ty1_conv3x3_x2_c3_bf16) against the two separate launches.  The ratio is comparable across boxes; absolute times are not."""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import ops, params as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--iters', type=int, default=200)
ap.add_argument('--rounds', type=int, default=4)
args = ap.parse_args()
dev = torch.device('cuda', 0)
N = args.batch
prm = P.make_random_params(seed=1, dense_init='glorot', bn='identity')
x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(0)).to(dev)
d = lambda a: torch.from_numpy(a).to(dev)
sc0, sh0 = ops.fold_bn(prm['bn0_gamma'], prm['bn0_beta'], prm['bn0_mean'], prm['bn0_var'])
sc1, sh1 = ops.fold_bn(prm['bn1_gamma'], prm['bn1_beta'], prm['bn1_mean'], prm['bn1_var'])
w0 = d(P.hwio_to_ohwi(prm['conv0_w']))
w1 = ops.cast_bf16(d(P.hwio_to_ohwi(prm['conv1_w'])))
sc0, sh0, sc1, sh1 = d(sc0), d(sh0), d(sc1), d(sh1)
mid = torch.empty((N, 224, 224, 16), dtype=torch.bfloat16, device=dev)
out = torch.empty((N, 112, 112, 32), dtype=torch.bfloat16, device=dev)


def fused():
    ops.conv3x3_x2_c3_bf16(x, w0, sc0, sh0, w1, sc1, sh1, out=out)


def two():
    ops.conv3x3_bn_leaky_bf16(x, w0, sc0, sh0, True, out=mid)
    ops.conv3x3_bn_leaky_bf16(mid, w1, sc1, sh1, True, out=out)


def timeit(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.iters


for _ in range(300):
    two()
torch.cuda.synchronize()
res = {'fused': [], 'two': []}
for _ in range(args.rounds):
    res['fused'].append(timeit(fused))
    res['two'].append(timeit(two))
f, t = statistics.median(res['fused']), statistics.median(res['two'])
print(json.dumps({'batch': N, 'fused_us': round(f, 1), 'two_launches_us': round(t, 1), 'ratio': round(f / t, 3),
                  'fused_all': [round(v, 1) for v in res['fused']], 'two_all': [round(v, 1) for v in res['two']]}))
