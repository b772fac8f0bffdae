#!/usr/bin/env python3
"""Headline benchmark: tiny-YOLOv1 448x448 inference (backbone + Dense + head + decode + NMS).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic images already resident in HBM.
The contract line (ONE JSON line on rank 0) is BASELINE.json configs[1]: batch 64 per GPU, fp32; for N > 1
every rank runs the same per-GPU batch (weak scaling: the path shards by image with no data-path collective) and the
fixed-size detection records are all-gathered every step (asynchronously: the collective completes under the next step).

The same line carries, under "also", the two BASELINE.json configs that are multi-GPU by definition, measured in the
same process right after the headline (skip with --no-also):
  configs[2]  bf16 inference, 128 images per GPU (1024 over 8), detection gather inside the step when N > 1
  configs[3]  VOC-20 training step, 32 images per GPU: frozen backbone fwd + Dense + yolo_loss fwd/bwd + wgrad +
              gradient exchange + Adam (everything behind the backbone on a side stream under the next backbone pass;
              one pass per --train-group = 2 batches, the one-pass-per-batch figure beside it).  Exchange for N > 1:
              --exchange gather (default: all-gather of features + dz, 1.8 MB per rank, every rank computes the group's
              weight gradient) or allreduce (ONE SUM all-reduce of the 73.9 MB [dWt | db | loss] bucket)
On one GPU two more legs follow: configs[4] (C = 1, 4096 images, whole network) and configs[1] fed from pinned HOST frames
(uint8 480x640, copy stream + one-launch GPU resize: the PCIe-inclusive rate, never the headline value).
Either of configs[2] / configs[3] can be made the headline instead: --mode infer --dtype bf16, or --mode train.
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense matrix peaks (no sparsity) and HBM3E
PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}
PEAK_HBM_GBS = 8000.0
GFLOP_PER_IMAGE = 3.216          # 2 x 1,608,015,360 MACs

# libty1 event-timer family -> kernel symbol as rocprofv3 --kernel-trace prints it (profiles/pmc_traffic*.json keys)
_CFG = {'128x128': '2, 2, 2, 2, 32', '128x64': '2, 1, 2, 2, 32', '64x64': '1, 1, 2, 2, 32', '256x32': '2, 1, 4, 1, 32',
        'bk16_128x64': '2, 1, 2, 2, 16', 'bk16_256x32': '2, 1, 4, 1, 16', '256x128': '2, 2, 4, 2, 32', '256x64': '2, 1, 4, 2, 32'}
ROCPROF_SYMBOL = {'conv01_fused_bf16': 'conv01_fused_bf16_kernel', 'conv3x3_c3_pool': 'conv3x3_c3p_kernel<true, 1, false>', 'conv3x3_c16_pool': 'conv3x3_c16_kernel<true, 1>',
                  'conv3x3_c3_pool_bf16out': 'conv3x3_c3p_kernel<true, 1, true>',
                  'conv3x3_c16_bf16_pool': 'conv3x3_c16_bf16_kernel<true, 1>',
                  'conv_halo_bf16_c32': 'conv_halo_bf16_kernel<32, 64, 4, 1, 16, 16, true, 1, 4, 4, true, 4, false>',
                  'conv_halo_bf16_8x28': 'conv_halo_bf16_kernel<64, 128, 4, 2, 8, 28, false, 1, 4, 4, true, 8, false>',
                  'conv_halo_bf16_14x14': 'conv_halo_bf16_kernel<64, 128, 4, 2, 14, 14, false, 1, 4, 4, true, 8, false>',
                  'conv_halo_bf16_4x7x7': 'conv_halo_bf16_kernel<64, 128, 4, 2, 7, 7, false, 4, 4, 4, true, 8, false>',
                  'conv_halo_bf16_4x7x7_ksplit': 'conv_halo_bf16_kernel<64, 128, 4, 2, 7, 7, false, 4, 4, 4, true, 8, true>',
                  'dense_f32_64x64': 'dense_gemm_f32_kernel<1, 1, 2, 2, 4>',
                  'dense_bf16_64x64': 'dense_gemm_f32_kernel<1, 1, 2, 2, 2>',
                  'dense_f32_32x128': 'dense_gemm_f32_kernel<1, 1, 1, 4, 4>',
                  'decode_nms': 'decode_nms_kernel<2>', 'head_activate': 'head_activate_kernel'}
for _k, _v in _CFG.items():
    ROCPROF_SYMBOL[f'conv_f32_{_k}_pool'] = f'conv_persist_kernel<{_v}, true, OpF32, false>'
    ROCPROF_SYMBOL[f'conv_f32_{_k}'] = f'conv_persist_kernel<{_v}, false, OpF32, false>'
    ROCPROF_SYMBOL[f'conv_bf16_{_k}_pool'] = f'conv_persist_kernel<{_v}, true, OpBF16, true>'
    ROCPROF_SYMBOL[f'conv_bf16_{_k}'] = f'conv_persist_kernel<{_v}, false, OpBF16, true>'


class Ctx:
    """Process-wide state: rank / world, device, the process group when there is one."""

    def __init__(self, args):
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if self.world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
        # one rank per GPU; the modulo only matters for the rehearsal of N ranks on fewer cards (--backend gloo)
        self.dev_index = self.local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device('cuda', self.dev_index)
        self.backend = args.backend
        self.multi = self.world > 1 or args.force_gather
        self.dist = None
        self.rccl_world = 1
        if self.multi:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=self.dev, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.rccl_world = dist.get_world_size()       # what RCCL's communicator reports, not the env

    def barrier(self):
        if self.backend == 'nccl':
            self.dist.barrier(device_ids=[self.dev_index])
        else:
            self.dist.barrier()

    def sync(self):
        torch.cuda.synchronize(self.dev)
        if self.multi:
            self.barrier()
            torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if not self.multi:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.multi:
            self.barrier()
            self.dist.destroy_process_group()


def settle(ctx, step, settle_ms, est_step_ms):
    """The shader clock of this chip climbs for tens of milliseconds after it leaves idle ("Clocks"): the
    same kernels run 10-12 % slower in the first few ms.  The metric is steady-state throughput, so untimed steps
    continue until the GPU has been busy for `settle_ms` before the timed region; the count is reported."""
    n = 0
    if settle_ms <= 0:
        return 0
    if ctx.multi:       # every rank must issue the same number of collectives: a fixed count, not a local clock
        n = max(1, int(settle_ms / max(est_step_ms, 0.05)))
        for _ in range(n):
            step()
    else:
        t0 = time.perf_counter()
        burst = max(1, min(10, int(20.0 / max(est_step_ms, 0.05))))     # (a configs[4] step is ~100 ms: one at a time)
        while (time.perf_counter() - t0) * 1e3 < settle_ms and n < 4000:
            for _ in range(burst):
                step()
            torch.cuda.synchronize(ctx.dev)
            n += burst
    ctx.sync()
    return n


def timed(ctx, step, k):
    """Exactly k steps between barrier + synchronise on both sides; MAX over ranks."""
    ta = time.perf_counter()
    ctx.sync()
    # the cyclic collector stays out of the region (as timeit does): with torch imported a full collection holds the host
    # for ~40 ms, and where the allocation counter trips depends on nothing the step does (tools/experiments/default_stall_hunt.sh)
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for i in range(k):
            step()
            if i == 0:
                t1 = time.perf_counter()
        t2 = time.perf_counter()
        ctx.sync()
        t3 = time.perf_counter()
    finally:
        if gc_was_on:
            gc.enable()
    if os.environ.get('TY1_BENCH_TRACE'):
        print(f'[trace] timed({k}): sync before {1e3 * (t0 - ta):.1f} ms, first step {1e3 * (t1 - t0):.1f} ms, enqueue {1e3 * (t2 - t0):.1f} ms, drain {1e3 * (t3 - t2):.1f} ms', file=sys.stderr, flush=True)
    return ctx.max_over_ranks(t3 - t0)


def families(table):
    """Per-layer timer entries are named '<kernel>@<H>x<Cin>><Cout>'; rocprofv3 --stats aggregates per kernel symbol,
    so fold the layers of one instantiation together."""
    fam = {}
    for k, v in table.items():
        e = fam.setdefault(k.split('@')[0], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        for f in e:
            e[f] += v[f]
    return fam


def kernel_rows(table):
    gpu_ms = sum(v['ms'] for v in table.values()) or 1.0
    rows = {}
    for k, v in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
        rows[k] = {'launches': v['launches'], 'avg_us': round(1e3 * v['ms'] / max(v['launches'], 1), 2),
                   'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 and v['flops'] > 0 else None,
                   'algo_gbs': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 and v['bytes'] > 0 else None,
                   'share': round(v['ms'] / gpu_ms, 3)}
    return rows


def traffic_of(dom_name, dtype, mode):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this same command (tools/profile.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    fname = {('infer', 'f32'): 'pmc_traffic.json', ('infer', 'bf16'): 'pmc_traffic_bf16.json',
             ('train', 'f32'): 'pmc_traffic_train.json'}.get((mode, dtype))
    try:
        pj = json.load(open(os.path.join(ROOT, 'profiles', fname)))
        sym = (ROCPROF_SYMBOL.get(dom_name) or '').replace('(anonymous namespace)::', '')
        cands = [sym, sym.replace('OpBF16, true>', 'OpBF16, false>')]
        for c in cands:
            if c in pj['kernels']:
                e = pj['kernels'][c]
                return round(e['hbm_read_bytes'] + e['hbm_write_bytes']), pj['source']
    except Exception:
        pass
    return None, None


def roofline_of(dom_name, dom, dtype, mode):
    """dom: launches / ms / flops / bytes of the dominant kernel measured live inside the timed region."""
    secs = dom['ms'] * 1e-3
    ai = dom['flops'] / max(dom['bytes'], 1.0)
    ridge = PEAK_TFLOPS[dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    traffic, src = traffic_of(dom_name, dtype, mode)
    common = {'kernel': dom_name, 'rocprof_symbol': ROCPROF_SYMBOL.get(dom_name), 'traffic': traffic,
              'traffic_unit': 'HBM bytes per launch', 'traffic_source': src, 'launches': dom['launches'],
              'avg_launch_us': round(1e3 * dom['ms'] / max(dom['launches'], 1), 2),
              'algo_flops_per_launch': dom['flops'] / max(dom['launches'], 1),
              'algo_bytes_per_launch': dom['bytes'] / max(dom['launches'], 1),
              'arithmetic_intensity': round(ai, 1), 'ridge': round(ridge, 1)}
    if ai >= ridge:
        a = dom['flops'] / secs / 1e12
        return {'bound': 'mfma', 'achieved': round(a, 2), 'peak': PEAK_TFLOPS[dtype], 'unit': 'TFLOP/s',
                'frac': round(a / PEAK_TFLOPS[dtype], 4), **common}
    a = dom['bytes'] / secs / 1e9
    return {'bound': 'hbm', 'achieved': round(a, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(a / PEAK_HBM_GBS, 4), **common}


def measure(ctx, step, args, dtype, mode, est_step_ms, detailed=True):
    """Warm-up, cold figure, settle, survey pass, timed region (dominant kernel bracketed live), plain repeat."""
    from yolo_amd import ops
    for _ in range(args.warmup):
        step()
    cold = timed(ctx, step, args.steps)                     # straight after the W warm-up steps: clocks still ramping
    settle_steps = settle(ctx, step, args.settle_ms, est_step_ms)
    table, dom_name = {}, None
    if detailed:
        # survey pass (untimed): every library launch brackets itself with HIP events on its own stream -> per-kernel
        # table, names the dominant kernel.  Bracketing EVERY launch costs ~5 % of a step, so the timed region below
        # keeps the brackets on the dominant kernel's launches only.
        tr_ = os.environ.get('TY1_BENCH_TRACE')
        tm = [time.perf_counter()]
        with ops.KernelTimer() as kt:
            for _ in range(args.steps):
                step()
            tm.append(time.perf_counter())
            ctx.sync()
            tm.append(time.perf_counter())
            table = kt.table()
            tm.append(time.perf_counter())
        if tr_:
            print(f'[trace] survey pass: enqueue {1e3 * (tm[1] - tm[0]):.1f} ms, drain {1e3 * (tm[2] - tm[1]):.1f} ms, table {1e3 * (tm[3] - tm[2]):.1f} ms', file=sys.stderr, flush=True)
        fam = families(table)
        dom_name = max(fam, key=lambda k: fam[k]['ms'])
    prefix = None
    if dom_name is not None:
        prefix = dom_name + '@' if any(k.startswith(dom_name + '@') for k in table) else dom_name
    # ... and of those only every `every`-th launch: two event records cost the stream 5-7 us (0.7 % of the fp32 step at two
    # launches per step, 3 % of the bf16 step).  `every` is coprime with the family's launch count, so a family of several
    # layers is sampled evenly.
    every = 1
    if dom_name is not None:
        import math
        n_fam = fam[dom_name]['launches']
        every = next((e for e in (3, 5, 7, 11, 13) if math.gcd(e, n_fam) == 1 and n_fam // e >= 10), 1)
    with ops.KernelTimer(only=prefix or 'none', every=every) as kt:
        elapsed = timed(ctx, step, args.steps)              # ---- THE timed region: exactly K steps
    # un-instrumented repeat, back to back with the timed region (host-side table work comes after it)
    plain = timed(ctx, step, args.steps)
    live = kt.table()
    out = {'elapsed': elapsed, 'plain': plain, 'cold': cold, 'settle_steps': settle_steps, 'table': table}
    if dom_name is not None:
        dom = dict(launches=0, ms=0.0, flops=0.0, bytes=0.0)
        for k, v in live.items():
            if k.split('@')[0] == dom_name:
                for f in dom:
                    dom[f] += v[f]
        if dom['launches'] and dom['ms'] > 0:
            out['roofline'] = roofline_of(dom_name, dom, dtype, mode)
            out['roofline']['launches_timed'] = f'1 launch in {every} of this kernel inside the timed region' if every > 1 else 'all inside the timed region'
            mm = {k: v for k, v in table.items() if v['flops'] > 0 and v['ms'] > 0}
            out['roofline']['all_matmul_kernels_tflops'] = round(
                sum(v['flops'] for v in mm.values()) / (sum(v['ms'] for v in mm.values()) * 1e-3) / 1e12, 2)
    return out


def synthetic_images(n, seed, dev):
    return torch.rand((n, 448, 448, 3), generator=torch.Generator(device='cpu').manual_seed(seed))


def nms_only(ctx, N, C, steps):
    """Decode + threshold + NMS alone on predictions drawn directly (seed 2: class = 1 for C = 1, confidence and
    box ~ U(0,1): dense candidates) - the "dense-box NMS stress test" half of configs[4]."""
    from yolo_amd import ops
    rng = np.random.default_rng(2)
    p = np.empty((N, 7, 7, C + 10), np.float32)
    if C > 1:
        z = 4 * rng.random((N, 7, 7, C))
        e = np.exp(z - z.max(-1, keepdims=True))
        p[..., :C] = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    else:
        p[..., :C] = 1.0
    p[..., C:] = rng.random((N, 7, 7, 10)).astype(np.float32)
    pd = torch.from_numpy(p).to(ctx.dev)
    out = ops.decode_nms(pd, 2, C, 448)
    for _ in range(3):
        ops.decode_nms(pd, 2, C, 448, out=out)
    torch.cuda.synchronize(ctx.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ops.decode_nms(pd, 2, C, 448, out=out)
    e1.record()
    torch.cuda.synchronize(ctx.dev)
    us = e0.elapsed_time(e1) * 1e3 / steps
    nbytes = N * (49 * (C + 10) * 4 + 98 * (32 + 4 + 4 + 1))           # prediction in, fixed-size records out
    return {'us_per_launch': round(us, 1), 'images_per_sec': round(N / us * 1e6), 'kept_boxes_per_image': round(float(out[3].sum().item()) / N, 2),
            'algo_bytes_per_launch': nbytes, 'algo_GBps': round(nbytes / us / 1e3, 1), 'frac_of_hbm_peak': round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4),
            'note': 'one wavefront per image, up to 98 dependent greedy rounds: latency-bound, not bandwidth-bound'}


def run_infer(ctx, args, dtype, N, detailed=True, C=20, images_on_device=False):
    """C = 20: configs[1] / configs[2].  C = 1 (configs[4]): Dense 539 -> (7,7,11) head on the same backbone."""
    from yolo_amd import ops, params as P
    from yolo_amd.engine import DeviceModel
    detailed = detailed and not args.graph      # a replayed graph has no per-launch brackets: no kernel table / roofline
    prm = P.make_random_params(seed=1, C=C, dense_init='glorot', bn='identity')   # SURVEY 8d stress distribution
    model = DeviceModel(prm, C=C, device=ctx.dev, dtype=dtype)
    model.fused01 = not args.no_fused01
    if images_on_device:       # 4096 images are 9.9 GB: drawn on the device (torch's Philox generator), no host copy
        x_cpu = None
        x = torch.rand((N, 448, 448, 3), device=ctx.dev, generator=torch.Generator(device=ctx.dev).manual_seed(0))
    else:
        x_cpu = synthetic_images(N, 0, ctx.dev)
        x = x_cpu.to(ctx.dev)
    model.workspace(N)
    if args.graph:                     # the batch lives in the graph's input buffer (no copy per step)
        xin = model.input_buffer(N)
        xin.copy_(x)
        x = xin
    recs = gathers = None
    if ctx.multi:
        # two sets of buffers: the gather of step i runs on RCCL's stream while the compute stream is already in step i + 1
        # (nothing in the next step needs the gathered records), so a rank neither waits for the collective's latency nor,
        # step by step, for the slowest device of the group; step i + 2 waits for gather i before it refills the records
        recs = [torch.empty((N, 98, 6), dtype=torch.float64, device=ctx.dev) for _ in range(2)]          # box(4) score cls*keep
        gathers = [torch.empty((ctx.world * N, 98, 6), dtype=torch.float64, device=ctx.dev) for _ in range(2)]
    state = {'i': 0, 'work': [None, None]}

    def step():
        boxes, score, cls, keep = model.detect_replay(x) if args.graph else model.detect(x)
        if ctx.multi:                                  # detection gather: fixed-size records, one collective
            b = state['i'] & 1
            state['i'] += 1
            if state['work'][b] is not None:
                state['work'][b].wait()                # stream-ordered: the compute stream waits for the gather of two steps ago
            ops.pack_detections(boxes, score, cls, keep, out=recs[b])
            if ctx.backend == 'nccl':
                state['work'][b] = ctx.dist.all_gather_into_tensor(gathers[b], recs[b], async_op=True)
            else:                                      # rehearsal backend: staged through the host
                host = torch.empty(gathers[b].shape, dtype=gathers[b].dtype)
                ctx.dist.all_gather_into_tensor(host, recs[b].cpu())
                gathers[b].copy_(host)
            state['gathered'] = gathers[b]
        state['keep'] = keep

    est = 1.65 if dtype == 'f32' else 0.8
    m = measure(ctx, step, args, dtype, 'infer', est * N / (64 if dtype == 'f32' else 128), detailed)
    for w in state['work']:
        if w is not None:
            w.wait()
    kept = float(state['keep'].sum().item()) / N
    gathered = state.get('gathered')
    K, world = args.steps, ctx.world
    res = {
        'value': round(world * N * K / m['elapsed'], 1), 'unit': 'images/sec',
        'ms_per_step': round(1e3 * m['elapsed'] / K, 4), 'settle_steps': m['settle_steps'],
        'dtype': dtype,
        'config': {'workload': (f'configs[4]: single-class (C=1, Dense 539) tiny-YOLOv1 448x448 batch={N}/GPU fp32' if C == 1 else
                                f'configs[1]: tiny-YOLOv1 448x448 batch={N}/GPU fp32' if dtype == 'f32' else
                                f'configs[2]: tiny-YOLOv1 448x448 batch={N}/GPU bf16 (conv + Dense operands bf16, fp32 accumulate)')
                               + ' backbone+Dense+head+decode+NMS, random-init weights (glorot, identity BN: dense-candidate '
                                 'NMS stress), torch.rand images resident in HBM',
                   'global_batch': world * N, 'per_gpu_batch': N,
                   'sharding': 'batch split across ranks, no data-path collective'
                               + ('; all-gather of fixed-size detection records every step, in flight under the next step' if ctx.multi else ''),
                   'kept_boxes_per_image': round(kept, 2)},
        'repeat_without_events_images_per_sec': round(world * N * K / m['plain'], 1),
        'cold_start_images_per_sec': round(world * N * K / m['cold'], 1),
        'cold_start_note': f'the same K steps timed straight after the {args.warmup} warm-up steps, before the '
                           f'{args.settle_ms:.0f} ms settle phase (shader clock still ramping)',
        'whole_step_frac_of_mfma_peak': round(GFLOP_PER_IMAGE * 1e9 * N * K / m['elapsed'] / (PEAK_TFLOPS[dtype] * 1e12), 4),
    }
    if ctx.multi:
        res['rccl'] = {'backend': 'nccl (RCCL)' if ctx.backend == 'nccl' else ctx.backend, 'world_size_reported': ctx.rccl_world,
                       'gathered_records_shape': list(gathered.shape),
                       'gathered_kept_boxes': int((gathered[..., 5] >= 0).sum().item())}
    if 'roofline' in m:
        res['roofline'] = m['roofline']
    if detailed:
        res['kernels'] = kernel_rows(m['table'])
    return res, prm, x_cpu


def run_from_host(ctx, args, N, H0=480, W0=640):
    """configs[1] fed from HOST memory: uint8 camera-sized frames in pinned buffers -> H2D on a copy stream (two device
    buffers: the copy of batch i + 1 runs under the network of batch i) -> ty1_preprocess_batch_u8 (resize + /255, one launch)
    -> detect (engine.HostFramePipeline).  The PCIe-inclusive rate of the path; never the headline `value`."""
    from yolo_amd import params as P
    from yolo_amd.engine import DeviceModel, HostFramePipeline
    prm = P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity')
    model = DeviceModel(prm, C=20, device=ctx.dev, dtype='f32')
    gen = torch.Generator().manual_seed(7 + ctx.rank)
    host = [torch.randint(0, 256, (N, H0, W0, 3), dtype=torch.uint8, generator=gen).pin_memory() for _ in range(2)]
    pipe = HostFramePipeline(model, N, H0, W0)
    dev_u8 = pipe.frames
    state = {'i': 0}

    def step():
        state['i'] += 1
        state['keep'] = pipe.submit(host[state['i'] & 1])[3]

    nbytes = N * H0 * W0 * 3
    ctx.sync()                                          # the link alone: what the pinned-buffer copy sustains with the GPU idle
    t0 = time.perf_counter()
    for _ in range(10):
        dev_u8[0].copy_(host[0], non_blocking=True)
    ctx.sync()
    link = 10 * nbytes / (time.perf_counter() - t0) / 1e9
    for _ in range(args.warmup):
        step()
    settle(ctx, step, args.settle_ms, 1.7 * N / 64)
    K = args.steps
    el = timed(ctx, step, K)
    # a second pass with one event per step: the spread of the step time on the GPU's own clock
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ctx.sync()
    marks[0].record()
    for i in range(K):
        step()
        marks[i + 1].record()
    ctx.sync()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(K))
    return {'value': round(ctx.world * N * K / el, 1), 'unit': 'images/sec', 'ms_per_step': round(1e3 * el / K, 4), 'dtype': 'f32',
            'step_ms_p10_p50_p90_max': [round(per_step[K // 10], 3), round(per_step[K // 2], 3), round(per_step[(9 * K) // 10], 3), round(per_step[-1], 3)],
            'config': {'workload': f'configs[1] from host memory: batch={N}/GPU of uint8 {H0}x{W0}x3 frames in pinned host buffers -> H2D on a '
                                   'copy stream (double-buffered) -> resize to 448x448 + /255 on the GPU (one launch) -> backbone+Dense+head+'
                                   'decode+NMS', 'per_gpu_batch': N},
            'h2d_bytes_per_step': nbytes, 'h2d_GBps_sustained': round(nbytes * K / el / 1e9, 1), 'h2d_GBps_link_alone': round(link, 1),
            'kept_boxes_per_image': round(float(state['keep'].sum().item()) / N, 2),
            'note': 'PCIe-inclusive rate, never the headline value; h2d_GBps_link_alone = the same pinned copy with the GPU idle; the same batch as '
                    'fp32 images (2.41 MB each) would need 2.6x these bytes'}


def every_nth(n, fn):
    """-> a step function that calls fn() on every n-th call.  fn() stands for n bench steps (Trainer.steps: n optimisation
    steps behind one backbone pass), so ANY window of K consecutive calls, K a multiple of n, holds exactly K of them -
    whatever the phase the window starts in."""
    state = {'calls': 0}

    def step():
        state['calls'] += 1
        if state['calls'] % n == 0:
            fn()
    return step


def make_labels(N, C, seed):
    """SURVEY 8d training labels: 3 objects per image, uniform cell and class, first writer wins (data_sequence.py:79-85)."""
    rng = np.random.default_rng(seed)
    y = np.zeros((N, 7, 7, C + 5), np.float32)
    for n in range(N):
        for _ in range(3):
            r, c, k = rng.integers(7), rng.integers(7), rng.integers(C)
            if y[n, r, c, C + 4] == 0:
                y[n, r, c, k] = 1
                y[n, r, c, C:C + 4] = rng.random(4)
                y[n, r, c, C + 4] = 1
    return y


def run_train(ctx, args, N, detailed=True):
    from yolo_amd import params as P
    from yolo_amd.engine import DeviceModel, Trainer
    C = 20
    prm = P.make_random_params(seed=1, C=C, dense_init='glorot', bn='identity')
    model = DeviceModel(prm, C=C, device=ctx.dev)
    # G consecutive batches share ONE pass of the frozen backbone (Trainer.steps): G x N images through the conv layers at
    # once, then G optimisation steps of N images each.  A bench step stays one optimisation step: every G-th call of step()
    # enqueues G of them, so any region of K calls (K a multiple of G) holds exactly K.
    G = args.train_group if args.train_group > 0 and args.steps % max(args.train_group, 1) == 0 else 1
    x = synthetic_images(G * N, 100 + ctx.rank, ctx.dev).to(ctx.dev)
    yd = torch.from_numpy(make_labels(G * N, C, 3 + ctx.rank)).to(ctx.dev)
    tr = Trainer(model, group=None, overlap=not args.no_overlap)
    if args.update_after_layer is not None:
        tr.update_after_layer = None if args.update_after_layer < 0 else args.update_after_layer
    state = {}

    trace = os.environ.get('TY1_BENCH_TRACE')          # developer aid: one stderr line per training step (rank, index, host time)
    t_start = time.perf_counter()
    if trace and os.environ.get('TY1_BENCH_WATCHDOG'):  # ... and which stream / event is still pending when a run stalls
        import threading

        def probe():
            cur = torch.cuda.current_stream(ctx.dev)
            while True:
                time.sleep(float(os.environ['TY1_BENCH_WATCHDOG']) / 2)
                print(f'[rank {ctx.rank}] probe at {time.perf_counter() - t_start:.1f} s: step {state.get("i", 0)} compute stream idle={cur.query()} '
                      f'side stream idle={tr.side.query() if tr.side else None} feats_ready={tr._feats_ready.query()} feats_free={tr._feats_free.query()} '
                      f'updated={tr._updated.query()} pending={tr._pending} adam_deferred={tr._adam_deferred}', file=sys.stderr, flush=True)
        threading.Thread(target=probe, daemon=True).start()

    def group():
        state['loss'] = tr.steps(x, yd, N, wait=False)[-1:]         # tail, exchange + Adam stay in flight under the next backbone
    grouped = every_nth(G, group)

    def step():
        grouped()
        if trace:
            state['i'] = state.get('i', 0) + 1
            print(f'[rank {ctx.rank}] train step {state["i"]} done at {time.perf_counter() - t_start:.3f} s', file=sys.stderr, flush=True)

    tr.bucket.always = ctx.multi                       # --force-gather: the collective runs on one rank too
    tr.exchange = args.exchange                        # every rank of this bench has the same N: 'gather' is legal
    real_sync = ctx.sync
    ctx.sync = lambda: (tr.join(), real_sync())[1]
    try:
        m = measure(ctx, step, args, 'f32', 'train', 1.3 * N / 32, detailed)
        K, world = args.steps, ctx.world
        # the exchange by itself: device time of the collective (events on the side stream, short separate pass) and
        # the step time with the collective left out (what the exchange costs after overlap = "exposed")
        tr.time_collective = True
        coll_ms = []
        for _ in range(5):
            step()
            ctx.sync()
            if tr.collective_events:
                coll_ms.append(tr.collective_events[0].elapsed_time(tr.collective_events[1]))
        tr.time_collective = False
        tr.skip_collective = True
        for _ in range(3):
            step()
        no_coll = timed(ctx, step, K)
        tr.skip_collective = False
        one_per_batch = None
        if G > 1 and not args.no_train_compare:   # the same K steps with one backbone pass per batch (Trainer.step), for comparison
            x1, y1 = x[:N], yd[:N]
            step1 = lambda: tr.step(x1, y1, wait=False)
            for _ in range(10):
                step1()
            one_per_batch = timed(ctx, step1, K)
    finally:
        ctx.sync = real_sync
    coll = statistics.median(coll_ms) if coll_ms else 0.0
    active = world > 1 or ctx.multi
    if args.exchange == 'gather':
        nbytes = N * (12544 + tr.ldz) * 4
        exch = {'kind': "all-gather of each rank's features (n x 12544) and dz (n x 1472, the loss in a padding column); every rank "
                        "computes the group's weight gradient dz_all^T x_all itself (Trainer.exchange = 'gather')",
                'collectives_per_step': 2 if active else 0, 'bytes_sent_per_rank_per_step': nbytes,
                'bytes_received_per_rank_per_step': (world - 1) * nbytes,
                'window_ms': round(coll, 4), 'window': 'both all-gathers with the Dense forward and the loss between them (device time)'}
    else:
        nbytes = tr.bucket.nbytes()
        exch = {'kind': 'ONE SUM all-reduce of [dWt | db | loss] (Trainer.exchange = \'allreduce\')',
                'collectives_per_step': 1 if active else 0, 'message_bytes': nbytes, 'device_ms': round(coll, 4),
                'bus_GBps': round(2 * (world - 1) / world * nbytes / (coll * 1e-3) / 1e9, 1) if coll > 0 and world > 1 else None}
    exch.update({'ms_per_step_without_collective': round(1e3 * no_coll / K, 4),
                 'exposed_ms': round(1e3 * (m['elapsed'] - no_coll) / K, 4),
                 'share_of_step': round(max(0.0, (m['elapsed'] - no_coll) / m['elapsed']), 4)})
    res = {
        'value': round(world * N * K / m['elapsed'], 1), 'unit': 'images/sec',
        'steps_per_sec': round(K / m['elapsed'], 2),
        'ms_per_step': round(1e3 * m['elapsed'] / K, 4), 'settle_steps': m['settle_steps'], 'dtype': 'f32',
        'config': {'workload': f'configs[3]: tiny-YOLOv1 VOC-20 training step, batch={N}/GPU fp32: frozen backbone fwd + Dense + '
                               'head + yolo_loss fwd/bwd + head bwd + Dense wgrad + ONE SUM all-reduce [dWt|db|loss] + Adam; '
                               'random-init weights, torch.rand images and synthetic labels resident in HBM'
                               + (f'; the frozen backbone runs once per {G} consecutive batches ({G * N} images), every batch of {N} '
                                  'then takes its own optimisation step (Trainer.steps)' if G > 1 else ''),
                   'global_batch': world * N, 'per_gpu_batch': N, 'batches_per_backbone_pass': G,
                   'sharding': 'batch split across ranks (data parallel), gradients summed (no division: the loss is a batch sum)',
                   'overlap': 'Dense forward, loss, gradients, exchange and Adam on a side stream under the next frozen-backbone pass'
                              if not args.no_overlap else 'serial'},
        'loss': float(state['loss'].item()),
        'exchange': exch,
        'repeat_without_events_images_per_sec': round(world * N * K / m['plain'], 1),
        'cold_start_images_per_sec': round(world * N * K / m['cold'], 1),
        'whole_step_frac_of_mfma_peak': round(GFLOP_PER_IMAGE * 1e9 * N * K / m['elapsed'] / (PEAK_TFLOPS['f32'] * 1e12), 4),
    }
    if one_per_batch is not None:
        res['one_backbone_pass_per_batch'] = {'images_per_sec': round(world * N * K / one_per_batch, 1), 'ms_per_step': round(1e3 * one_per_batch / K, 4),
                                              'note': 'Trainer.step per batch, un-instrumented, same process'}
    if ctx.multi:
        res['rccl'] = {'backend': 'nccl (RCCL)' if ctx.backend == 'nccl' else ctx.backend, 'world_size_reported': ctx.rccl_world}
    if 'roofline' in m:
        res['roofline'] = m['roofline']
    if detailed:
        res['kernels'] = kernel_rows(m['table'])
    return res


def cpu_baseline(prm, x_cpu, budget_s, C=20):
    """The oracle (CPU restatement of the reference Keras path + C restatement of its NumPy decode/NMS) timed on the host
    cores of this box: a reported baseline, not the target.  Protocol of BASELINE.md 4 / SURVEY 8d: N = 1 and N = 64,
    3 warm-ups, median of >= 10 runs; the run counts shrink (and say so) only if the host is too slow for the budget."""
    import oracle
    from oracle import model_ref
    x = x_cpu.numpy()

    def one(n):
        t0 = time.perf_counter()
        y = model_ref.model_ref(x[:n], prm, 7, 2, C)
        oracle.decode_nms(y, 2, C, 448)
        return time.perf_counter() - t0

    def protocol(n, budget):
        first = one(n)
        warm = 3 if first * 13 < budget else 1
        for _ in range(warm - 1):
            one(n)
        runs = 10 if first * 13 < budget else max(3, int(budget / max(first, 1e-3)) - warm)
        ts = [one(n) for _ in range(runs)]
        return statistics.median(ts), warm, runs

    n_big = min(64, x.shape[0])
    t1, w1, r1 = protocol(1, 0.2 * budget_s)
    tb, wb, rb = protocol(n_big, 0.8 * budget_s)
    return {'value': round(n_big / tb, 2), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'host_cpu_count': os.cpu_count(),
            'n1_latency_ms': round(1e3 * t1, 2), 'n1_images_per_sec': round(1.0 / t1, 2),
            'sample': f'N={n_big}: median of {rb} runs after {wb} warm-ups ({tb:.2f} s per run); N=1: median of {r1} runs after {w1} '
                      f'warm-ups; the same synthetic batch as the GPU run; torch-CPU fp32 restatement of the reference Keras model '
                      f'+ C restatement of its NumPy decode/NMS (oracle/), not Keras itself'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--mode', choices=['infer', 'train'], default='infer')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32', help='inference arithmetic (configs[1] f32, configs[2] bf16)')
    ap.add_argument('--batch', type=int, default=0, help='images per GPU per step (default: 64 f32, 128 bf16, 32 train)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the CPU-baseline leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true', help='skip the configs[2] / configs[3] measurements under "also"')
    ap.add_argument('--no-fused01', action='store_true', help='bf16 inference: conv blocks 0 and 1 as two launches instead of one (A/B)')
    ap.add_argument('--graph', action='store_true', help='inference: replay the step as one HIP graph (DeviceModel.detect_replay)')
    ap.add_argument('--update-after-layer', type=int, default=None,
                    help='training: hold the Adam launch back until the next backbone has enqueued this layer (-1: at once; default: the trainer\'s)')
    ap.add_argument('--train-group', type=int, default=2,
                    help='training: consecutive batches that share one pass of the frozen backbone (Trainer.steps); 1 = Trainer.step per batch')
    ap.add_argument('--exchange', choices=['gather', 'allreduce'], default='gather',
                    help='training, N > 1: all-gather features + dz and compute the group\'s weight gradient on every rank (1.8 MB per rank '
                         'on the links), or ONE SUM all-reduce of the 73.9 MB gradient bucket')
    ap.add_argument('--no-train-compare', action='store_true',
                    help='training: leave out the one-backbone-pass-per-batch comparison (profiling runs: it launches the same kernels at another size)')
    ap.add_argument('--no-overlap', action='store_true', help='training: run the all-reduce + Adam serially on the compute stream')
    ap.add_argument('--settle-ms', type=float, default=200.0, help='untimed steps after the warm-up until the clocks have settled')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='collective backend; gloo only to rehearse the N-rank launch path on a box with fewer GPUs than ranks '
                         '(RCCL refuses two ranks on one device): numbers from such a run mean nothing')
    ap.add_argument('--force-gather', action='store_true',
                    help='developer check: run the N>1 code path (process group, detection gather, all-reduce) on one rank')
    args = ap.parse_args()
    # stdout carries ONE line, the JSON: what libraries write to fd 1 on the way (RCCL prints a five-line version banner when
    # its first communicator comes up) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get('TY1_BENCH_TRACE'):               # developer aid: full collections of the cyclic collector, with their duration
        gc_t = {}

        def gc_note(phase, info):
            if info.get('generation') == 2:
                if phase == 'start':
                    gc_t['t'] = time.perf_counter()
                else:
                    print(f'[trace] full gc: {1e3 * (time.perf_counter() - gc_t.get("t", 0.0)):.1f} ms, collected {info.get("collected")}', file=sys.stderr, flush=True)
        gc.callbacks.append(gc_note)
    if not os.environ.get('TY1_BENCH_NO_GC_FREEZE'):
        gc.collect()
        gc.freeze()          # everything imported so far (torch: ~10^6 objects) leaves the collector's generations for good
    if os.environ.get('TY1_BENCH_WATCHDOG'):          # developer aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['TY1_BENCH_WATCHDOG']), repeat=True, file=sys.stderr)
    ctx = Ctx(args)
    headline_is_default = args.mode == 'infer' and args.dtype == 'f32'
    prm = x_cpu = None
    if args.mode == 'train':
        head = run_train(ctx, args, args.batch or 32)
        metric = 'images/sec tiny-YOLOv1 448x448 training step (yolo_loss fwd/bwd, Dense wgrad, SUM all-reduce, Adam)'
    else:
        head, prm, x_cpu = run_infer(ctx, args, args.dtype, args.batch or (64 if args.dtype == 'f32' else 128))
        metric = 'images/sec tiny-YOLOv1 448x448 inference incl. decode+NMS'
    out = {'metric': metric, 'value': head.pop('value'), 'unit': head.pop('unit'), 'n_gpus': ctx.world, 'steps': args.steps,
           'warmup': args.warmup, 'settle_steps': head.pop('settle_steps'), 'ms_per_step': head.pop('ms_per_step'),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': head.pop('dtype'), 'data': 'synthetic'}
    out.update(head)
    if 'kernels' in out:
        out['kernels_note'] = ('per-kernel table from an untimed pass of the same K steps with every launch bracketed by HIP '
                               'events; inside the timed region only the dominant kernel is bracketed (roofline.achieved)')
    if headline_is_default and not args.no_also:
        also = {}
        sub = argparse.Namespace(**vars(args))
        sub.steps, sub.warmup, sub.settle_ms = min(args.steps, 50), min(args.warmup, 10), min(args.settle_ms, 100.0)
        big = argparse.Namespace(**vars(sub))      # a configs[4] step is 4096 images (~0.1 s): fewer of them
        big.steps, big.warmup = min(args.steps, 5), min(args.warmup, 2)

        def config4():
            r = run_infer(ctx, big, 'f32', 4096, C=1, images_on_device=True)[0]
            r['decode_nms_only'] = nms_only(ctx, 4096, 1, 20)
            return r

        legs = [('configs[2] bf16 inference', sub, lambda: run_infer(ctx, sub, 'bf16', 128)[0]),
                ('configs[3] training step', sub, lambda: run_train(ctx, sub, 32))]
        if ctx.world == 1:                         # single-GPU extras: kept out of the N > 1 scaling runs
            legs.append(('configs[4] single-class batch 4096', big, config4))
            legs.append(('configs[1] from pinned host frames', sub, lambda: run_from_host(ctx, sub, 64)))
        for key, leg_args, fn in legs:
            if ctx.multi and ctx.backend != 'nccl' and 'training' in key:
                # two ranks time-slicing ONE card: the compute stream of one rank stops inside the step's kernels while the
                # other waits for it in gloo's host all-reduce (profiles/r03_gloo_train_rehearsal_run*.log); the
                # two-rank training step itself is covered by tests/test_gpu_ddp.py
                also[key] = {'skipped': 'rehearsal backend (gloo), ranks sharing a card: the training leg stalls on the GPU (DESIGN.md 5)'}
                continue
            torch.cuda.empty_cache()
            r = fn()
            r.pop('kernels', None)
            r['steps'], r['warmup'] = leg_args.steps, leg_args.warmup
            also[key] = r
        out['also'] = also
    if ctx.rank == 0:
        if ctx.world == 1 and not args.no_cpu_baseline and prm is not None:
            out['cpu_baseline'] = cpu_baseline(prm, x_cpu, args.cpu_seconds)
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    ctx.close()


if __name__ == '__main__':
    main()
