#!/usr/bin/env python3
"""The from-host pipeline of bench.py (pinned uint8 frames -> H2D on a copy stream -> batched resize -> detect) with the time
of every step and of every copy (events), for 2 or 3 device buffers: where do 2.2-2.4 ms steps come from?"""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_amd import ops, params as P
from yolo_amd.engine import DeviceModel
dev = torch.device('cuda', 0)
N, H0, W0 = 64, int(os.environ.get('H0', 480)), int(os.environ.get('W0', 640))
model = DeviceModel(P.make_random_params(seed=1, C=20, dense_init='glorot', bn='identity'), device=dev)
gc.collect(); gc.freeze(); gc.disable()
for NB in (2, 3, 2, 3):
    host = [torch.randint(0, 256, (N, H0, W0, 3), dtype=torch.uint8).pin_memory() for _ in range(NB)]
    dbuf = [torch.empty((N, H0, W0, 3), dtype=torch.uint8, device=dev) for _ in range(NB)]
    copy = torch.cuda.Stream(dev)
    ready, free = [torch.cuda.Event() for _ in range(NB)], [torch.cuda.Event() for _ in range(NB)]
    cur = torch.cuda.current_stream(dev)
    K = 150
    ce = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    se = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    def step(i, rec):
        b = i % NB
        with torch.cuda.stream(copy):
            copy.wait_event(free[b])
            if rec: ce[i][0].record(copy)
            dbuf[b].copy_(host[b], non_blocking=True)
            if rec: ce[i][1].record(copy)
            ready[b].record(copy)
        cur.wait_event(ready[b])
        x = ops.preprocess_batch_u8(dbuf[b], (448, 448), out=model.input_buffer(N))
        free[b].record(cur)
        model.detect(x)
        if rec: se[i + 1].record(cur)
    for i in range(60):
        step(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    se[0].record(cur)
    for i in range(K):
        step(i, True)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K * 1e3
    raw = [se[i].elapsed_time(se[i + 1]) for i in range(K)]
    worst = sorted(range(K), key=lambda i: -raw[i])[:5]
    st = sorted(raw)
    ct = sorted(a.elapsed_time(b) for a, b in ce)
    print(f'{NB} buffers: {el:.3f} ms per step ({N / el:.1f} k images/s); step p10/p50/p90 {st[15]:.2f}/{st[75]:.2f}/{st[135]:.2f} ms; '
          f'copy p10/p50/p90 {ct[15]:.2f}/{ct[75]:.2f}/{ct[135]:.2f} ms = {N * H0 * W0 * 3 / ct[75] / 1e6:.1f} GB/s at the median; worst steps {[(i, round(raw[i], 2)) for i in worst]}')
