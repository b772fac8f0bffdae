#!/usr/bin/env python3
"""Step time of configs[3] (training, 32 images) against k = CUs left free by the persistent conv grids
(Trainer.reserve_cus -> ty1_set_reserved_cus) while the step's exchange is in flight.

One GPU has no peer, so the all-reduce is stood in for by what it costs the chip locally: a pair of 74 MB device-to-device
copies of the gradient bucket on the side stream (read + write of the message, twice: reduce-scatter + all-gather each
touch every byte once), issued where the collective would be.  Prints one table; all settings run inside one process,
alternating, so they share clocks and temperature.

  python tools/reserve_cus_sweep.py [--steps 200] [--rounds 3] [--ks 0,8,16,32,64]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yolo_amd import params as P  # noqa: E402
from yolo_amd.engine import DeviceModel, Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--ks', default='0,8,16,32,64')
ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)

N = args.batch
prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity')
model = DeviceModel(prm, device=dev)
x = bench.synthetic_images(N, 100, dev).to(dev)
yd = torch.from_numpy(bench.make_labels(N, 20, 3)).to(dev)
tr = Trainer(model, overlap=True)
stage = torch.empty_like(tr.bucket.flat)


class CopyPair:
    """Stands in for dist.all_reduce(bucket.flat, async_op=True): same stream position, same bytes, no peer."""

    def wait(self):
        pass


def fake_allreduce(group=None, async_op=False):
    stage.copy_(tr.bucket.flat, non_blocking=True)
    tr.bucket.flat.copy_(stage, non_blocking=True)
    return CopyPair()


def run(k, with_copy, steps):
    tr.reserve_cus = k
    tr.bucket.allreduce = fake_allreduce if with_copy else (lambda group=None, async_op=False: None)
    for _ in range(10):
        tr.step(x, yd, wait=False)
    tr.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, yd, wait=False)
    tr.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ks = [int(v) for v in args.ks.split(',')]
for _ in range(300):                      # clocks
    tr.step(x, yd, wait=False)
tr.join()
torch.cuda.synchronize()
res = {(k, c): [] for k in ks for c in (False, True)}
for _ in range(args.rounds):
    for k in ks:
        for c in (False, True):
            res[(k, c)].append(run(k, c, args.steps))
rows = []
for k in ks:
    a, b = statistics.median(res[(k, False)]), statistics.median(res[(k, True)])
    rows.append({'reserved_cus': k, 'ms_per_step_no_exchange': round(a, 4), 'ms_per_step_with_copy_pair': round(b, 4),
                 'exposed_ms': round(b - a, 4), 'images_per_sec_with_copy_pair': round(N / b * 1e3, 1)})
print(json.dumps({'batch': N, 'steps': args.steps, 'rounds': args.rounds, 'message_bytes': tr.bucket.nbytes(), 'rows': rows}))
for r in rows:
    print(f"k={r['reserved_cus']:4d}  no exchange {r['ms_per_step_no_exchange']:.4f} ms   with copy pair {r['ms_per_step_with_copy_pair']:.4f} ms   "
          f"exposed {r['exposed_ms']:+.4f} ms   {r['images_per_sec_with_copy_pair']:.0f} images/s", file=sys.stderr)
