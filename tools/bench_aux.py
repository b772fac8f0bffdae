#!/usr/bin/env python3
"""Auxiliary measurements for the other BASELINE.json configs (not the contract line of bench.py):
  --train    configs[3]: one VOC-20 training step (frozen backbone fwd + yolo_loss fwd/bwd + Dense wgrad + SUM
             all-reduce + Adam), batch 32 per GPU; launch with torch.distributed.run for N > 1
  --bf16     configs[2]: bf16 inference incl. decode+NMS, 128 images per GPU (1024 over 8), weak scaling, no collective
  --nms      configs[4]: decode + threshold + NMS stress, C=1 head, batch 4096, dense candidates
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import ops, params as P  # noqa: E402
from yolo_amd.engine import DeviceModel, Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--train', action='store_true')
ap.add_argument('--nms', action='store_true')
ap.add_argument('--bf16', action='store_true')
ap.add_argument('--bf16-batch', type=int, default=128)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0')); lr = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(lr)
dev = torch.device('cuda', lr)
group = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=dev)

if args.train:
    N, C = args.batch, 20
    prm = P.make_random_params(seed=1, C=C, dense_init='glorot', bn='identity')
    model = DeviceModel(prm, C=C, device=dev)
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(rank)).to(dev)
    rng = np.random.default_rng(3 + rank)
    y = np.zeros((N, 7, 7, 25), np.float32)
    for n in range(N):                                   # 3 objects per image, first writer wins (data_sequence.py:79-85)
        for _ in range(3):
            r, c, k = rng.integers(7), rng.integers(7), rng.integers(20)
            if y[n, r, c, 24] == 0:
                y[n, r, c, k] = 1; y[n, r, c, 20:24] = rng.random(4); y[n, r, c, 24] = 1
    yd = torch.from_numpy(y).to(dev)
    tr = Trainer(model)
    for _ in range(3):
        tr.step(x, yd)
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = tr.step(x, yd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        table = kt.table()
    if rank == 0:
        print(json.dumps({'config': 'configs[3] training step', 'n_gpus': world, 'batch_per_gpu': N, 'ms_per_step': round(1e3 * dt / args.steps, 3),
                          'images_per_sec': round(world * N * args.steps / dt, 1), 'loss': float(loss.item()),
                          'kernels_us': {k: round(1e3 * v['ms'] / max(v['launches'], 1), 1) for k, v in sorted(table.items(), key=lambda kv: -kv[1]['ms'])}}))
if args.bf16:
    N, C = args.bf16_batch, 20
    prm = P.make_random_params(seed=1, C=C, dense_init='glorot', bn='identity')
    model = DeviceModel(prm, C=C, device=dev, dtype='bf16')
    x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(rank)).to(dev)
    for _ in range(3):
        det = model.detect(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        det = model.detect(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with ops.KernelTimer() as kt:
        for _ in range(5):
            model.detect(x)
        torch.cuda.synchronize()
        table = kt.table()
    if rank == 0:
        print(json.dumps({'config': 'configs[2] bf16 inference incl. decode+NMS', 'n_gpus': world, 'batch_per_gpu': N,
                          'ms_per_step': round(1e3 * dt / args.steps, 3), 'images_per_sec': round(world * N * args.steps / dt, 1),
                          'kept_per_image': float(det[3].sum().item()) / N,
                          'kernels_us_tflops': {k: [round(1e3 * v['ms'] / max(v['launches'], 1), 1), round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1)]
                                                for k, v in sorted(table.items(), key=lambda kv: -kv[1]['ms'])}}))
if args.nms:
    N, C = 4096, 1
    rng = np.random.default_rng(2)
    p = np.empty((N, 7, 7, 11), np.float32)
    p[..., 0] = 1.0
    p[..., 1:] = rng.random((N, 7, 7, 10)).astype(np.float32)
    pd = torch.from_numpy(p).to(dev)
    out = ops.decode_nms(pd, 2, C, 448)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        ops.decode_nms(pd, 2, C, 448, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.steps
    nbytes = N * (49 * 11 * 4 + 98 * (32 + 4 + 4 + 1))
    print(json.dumps({'config': 'configs[4] C=1 batch 4096 decode+NMS', 'us': round(us, 1), 'images_per_sec': round(N / us * 1e6),
                      'kept_per_image': float(out[3].sum().item()) / N, 'algo_GBps': round(nbytes / us / 1e3, 1)}))
