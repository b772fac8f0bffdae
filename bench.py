#!/usr/bin/env python3
"""Headline benchmark: tiny-YOLOv1 448x448 inference (backbone + Dense + head + decode + NMS).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic images already resident in HBM.
At N=1 the workload is BASELINE.json configs[1]: batch 64, fp32, one MI355X.  For N>1 every rank
runs the same per-GPU batch (weak scaling: the path shards by image with no data-path collective)
and the fixed-size detection records are all-gathered (configs[2]'s "detection gather").
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_HBM_GBS = 8000.0

# libty1 event-timer name -> kernel symbol as rocprofv3 --kernel-trace prints it
_CFG = {'128x128': '2, 2, 2, 2, 32', '128x64': '2, 1, 2, 2, 32', '64x64': '1, 1, 2, 2, 32', '256x32': '2, 1, 4, 1, 32',
        'bk16_128x64': '2, 1, 2, 2, 16', 'bk16_256x32': '2, 1, 4, 1, 16'}
ROCPROF_SYMBOL = {'conv3x3_c3_pool': 'conv3x3_c3p_kernel<true, 1, false>', 'conv3x3_c16_pool': 'conv3x3_c16_kernel<true, 1>',
                  'dense_f32_64x64': 'dense_gemm_f32_kernel<1, 1, 2, 2, 4>',
                  'dense_f32_32x128': 'dense_gemm_f32_kernel<1, 1, 1, 4, 4>',
                  'decode_nms': 'decode_nms_kernel<2>', 'head_activate': 'head_activate_kernel'}
for _k, _v in _CFG.items():
    ROCPROF_SYMBOL[f'conv_f32_{_k}_pool'] = f'conv_persist_kernel<{_v}, true, OpF32, false>'
    ROCPROF_SYMBOL[f'conv_f32_{_k}'] = f'conv_persist_kernel<{_v}, false, OpF32, false>'


def cpu_baseline(prm, x_cpu, seconds, C=20):
    """The oracle (CPU restatement of the reference Keras path + NumPy post-processing) timed on the
    host cores of this box: a reported baseline, not the target.  Bounded sample: the synthetic batch is
    walked in chunks of 8 images, again and again, until `seconds` of CPU work have been timed."""
    import oracle
    from oracle import model_ref
    threads = torch.get_num_threads()
    x = x_cpu.numpy()
    bs = 8
    model_ref.model_ref(x[:2], prm, 7, 2, C)                       # warm-up
    t0 = time.perf_counter()
    done = kept = 0
    while time.perf_counter() - t0 < seconds:
        i = done % max(x.shape[0] - bs + 1, 1)
        y = model_ref.model_ref(x[i:i + bs], prm, 7, 2, C)
        kept += int(oracle.decode_nms(y, 2, C, 448)['keep'].sum())
        done += min(bs, x.shape[0] - i)
    dt = time.perf_counter() - t0
    return {'value': done / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{done} images ({dt:.1f} s of CPU work) drawn in chunks of {bs} from the same synthetic batch: torch-CPU fp32 '
                      f'restatement of the reference Keras model + C restatement of its NumPy decode/NMS'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU per step (configs[1]: 64)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU-baseline sample: seconds of host work to time')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--settle-ms', type=float, default=200.0, help='untimed steps after the warm-up until the clocks have settled')
    ap.add_argument('--force-gather', action='store_true', help='developer check: run the N>1 code path (process group, detection gather) on one rank')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    multi = world > 1 or args.force_gather
    if multi:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)

    from yolo_amd import ops, params as P
    from yolo_amd.engine import DeviceModel

    N, C = args.batch, 20
    prm = P.make_random_params(seed=1, C=C, dense_init='glorot', bn='identity')   # SURVEY 8d stress distribution
    model = DeviceModel(prm, C=C, device=dev)
    x_cpu = torch.rand((N, 448, 448, 3), generator=torch.Generator(device='cpu').manual_seed(0))
    x = x_cpu.to(dev)
    ws = model.workspace(N)
    gathered = None
    if multi:
        slots = 98
        rec = torch.empty((N, slots, 6), dtype=torch.float64, device=dev)          # box(4) score cls*keep
        gathered = torch.empty((world * N, slots, 6), dtype=torch.float64, device=dev)

    def step():
        boxes, score, cls, keep = model.detect(x)
        if multi:                                      # detection gather: fixed-size records, one collective
            ops.pack_detections(boxes, score, cls, keep, out=rec)
            dist.all_gather_into_tensor(gathered, rec)
        return keep

    def sync_all():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    # The shader clock of this chip climbs for tens of milliseconds after it leaves idle ("Clocks"): the
    # same kernels run 10-12 % slower in the first few ms.  The metric is steady-state throughput, so untimed steps
    # continue until the GPU has been busy for `--settle-ms` (default 200 ms) before anything is measured; the count is
    # reported as `settle_steps`.  The timed region below is still exactly K steps.
    settle_steps = 0
    if multi:
        # every rank must issue the same number of collectives: a fixed count (~1.6 ms per step), not a local clock
        settle_steps = int(args.settle_ms / 1.6)
        for _ in range(settle_steps):
            step()
    else:
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms and settle_steps < 2000:
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
            settle_steps += 10
    sync_all()
    # ---- survey pass (untimed): every library launch brackets itself with HIP events on its own stream; this gives
    #      the per-kernel table and names the dominant kernel.  Bracketing EVERY launch costs ~5 % of a step, so the
    #      timed region below keeps the brackets on the dominant kernel's launches only.
    with ops.KernelTimer() as kt:
        for _ in range(args.steps):
            step()
        sync_all()
        table = kt.table()
    conv_names = {}
    for k, v in table.items():
        if k.startswith('conv') and v['ms'] > 0:
            conv_names[k.split('@')[0]] = conv_names.get(k.split('@')[0], 0.0) + v['ms']
    dom_prefix = max(conv_names, key=conv_names.get) + '@'
    # ---- timed region: exactly K steps; the dominant kernel's launches are timed live with HIP events
    with ops.KernelTimer(only=dom_prefix) as kt:
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            keep = step()
        sync_all()
        elapsed = time.perf_counter() - t0
        live = kt.table()
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kept_per_image = float(keep.sum().item()) / N

    if rank == 0:
        # un-instrumented repeat (no event records) to show what the instrumentation costs
        sync_all() if world == 1 else None
        plain = None
        if world == 1:
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize(dev)
            plain = time.perf_counter() - t1
        # per-layer entries are named "<kernel>@<H>x<Cin>><Cout>"; the roofline object is quoted per kernel
        # symbol (what rocprofv3 --stats aggregates), so fold the layers of one instantiation together
        conv = {}
        for k, v in table.items():
            if k.startswith('conv') and v['ms'] > 0:
                e = conv.setdefault(k.split('@')[0], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                for f in e:
                    e[f] += v[f]
        dom_name = dom_prefix[:-1]
        dom = dict(launches=0, ms=0.0, flops=0.0, bytes=0.0)        # the dominant kernel, timed inside the timed region
        for k, v in live.items():
            if k.startswith(dom_prefix):
                for f in dom:
                    dom[f] += v[f]
        achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        kernels = {}
        gpu_ms = sum(v['ms'] for v in table.values())
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
            kernels[k] = {'launches': v['launches'], 'avg_us': round(1e3 * v['ms'] / max(v['launches'], 1), 2),
                          'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else None,
                          'algo_gbs': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None,
                          'share': round(v['ms'] / gpu_ms, 3)}
        conv_flops = sum(v['flops'] for k, v in table.items() if k.startswith(('conv', 'dense')))
        conv_ms = sum(v['ms'] for k, v in table.items() if k.startswith(('conv', 'dense')))
        # HBM traffic of the dominant kernel: measured in separate rocprofv3 --pmc passes of this same command
        # (tools/profile.sh), committed under profiles/; averaged over that symbol's launches in one step
        traffic, traffic_src = None, None
        try:
            pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            sym = (ROCPROF_SYMBOL.get(dom_name) or '').replace('(anonymous namespace)::', '')
            if sym in pj['kernels']:
                traffic = round(pj['kernels'][sym]['hbm_read_bytes'] + pj['kernels'][sym]['hbm_write_bytes'])
                traffic_src = pj['source']
        except Exception:
            pass
        out = {
            'metric': 'images/sec tiny-YOLOv1 448x448 inference incl. decode+NMS',
            'value': round(world * N * args.steps / elapsed, 1),
            'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'settle_steps': settle_steps,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'configs[1]: tiny-YOLOv1 448x448 batch={N}/GPU fp32 backbone+Dense+head+decode+NMS, '
                                   'random-init weights (glorot, identity BN: dense-candidate NMS stress), '
                                   'torch.rand images resident in HBM',
                       'global_batch': world * N, 'per_gpu_batch': N,
                       'sharding': 'batch split across ranks, no data-path collective'
                                   + ('; all-gather of fixed-size detection records inside the step' if multi else ''),
                       'kept_boxes_per_image': round(kept_per_image, 2)},
            'roofline': {'bound': 'mfma', 'kernel': dom_name, 'rocprof_symbol': ROCPROF_SYMBOL.get(dom_name),
                         'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': traffic_src,
                         'launches': dom['launches'], 'avg_launch_us': round(1e3 * dom['ms'] / dom['launches'], 2),
                         'algo_flops_per_launch': dom['flops'] / dom['launches'],
                         'all_matmul_kernels_tflops': round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                         'whole_step_frac_of_peak': round(
                             (3.216e9 * world * N * args.steps / elapsed) / (world * PEAK_F32_MFMA_TFLOPS * 1e12), 4)},
            'kernels': kernels,
            'kernels_note': 'per-kernel table from an untimed pass of the same K steps with every launch bracketed by HIP '
                            'events; inside the timed region only the dominant kernel is bracketed (roofline.achieved)',
        }
        if plain is not None:
            out['repeat_without_events_images_per_sec'] = round(N * args.steps / plain, 1)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(prm, x_cpu, args.cpu_seconds, C)
            cb['value'] = round(cb['value'], 2)
            out['cpu_baseline'] = cb
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
