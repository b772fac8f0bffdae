"""Developer experiment: Dense split-K factor vs the time of Dense GEMM + what reduces its partials (one GPU call):
the fused inference tail (N >= 16), head activation + decode/NMS (N < 16), the training tail + wgrad (--train)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_amd import ops  # noqa: E402

dev = 'cuda:0'
K, units = 12544, 1470
cases = [(True, 128, 'infer'), (False, 1, 'infer'), (False, 1, 'fused'), (True, 1, 'infer'), (True, 1, 'fused'), (False, 4, 'infer'), (False, 4, 'fused'), (False, 8, 'infer'), (False, 8, 'fused'), (False, 16, 'infer'), (False, 32, 'train')]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[2] in sys.argv[1:]]
for bf, N, kind in cases:
    dt = torch.bfloat16 if bf else torch.float32
    x = (torch.rand((N, K), device=dev) * 0.1).to(dt)
    wt = (torch.randn((1472, K), device=dev) * 0.01).to(dt)
    bias = torch.zeros(units, device=dev)
    yt = torch.zeros((N, 7, 7, 25), device=dev)
    chunk = 64 if bf else ops.DENSE_K_ALIGN
    default = ops.pick_splits(K, N, units, chunk=chunk)
    for s in sorted({default, 7, 11, 14, 16, 21, 22, 28, 33, 44}):
        if s > K // chunk:
            continue
        part = torch.empty((s, N, 1472), device=dev)
        def run():
            ops.dense_splitk(x, wt, units, splits=s, part=part)
            if kind == 'train':
                ops.train_tail(part, bias, yt, 1472)
            elif N >= 16 or kind == 'fused':
                ops.head_decode_nms(part, bias, 7, 2, 20, 448, 0.2)
            else:
                ops.decode_nms(ops.head_activate(part, bias), 2, 20, 448, 0.2)
        for _ in range(20):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{'bf16' if bf else 'f32'} N={N} {kind} splits={s}{' (default)' if s == default else ''}: {e0.elapsed_time(e1) / 300 * 1e3:.1f} us", flush=True)
