import sys, torch
sys.path.insert(0, '.')
from yolo_amd import ops
d = torch.randint(0, 256, (64, 480, 640, 3), dtype=torch.uint8, device='cuda')
out = torch.empty((64, 448, 448, 3), device='cuda')
for shape in [(64, 480, 640, 3), (64, 448, 448, 3), (64, 896, 896, 3)]:
    d = torch.randint(0, 256, shape, dtype=torch.uint8, device='cuda')
    for _ in range(5): ops.preprocess_batch_u8(d, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.preprocess_batch_u8(d, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nb = d.numel() + out.numel() * 4
    print(shape, f'{us:.1f} us', f'{nb / us / 1e3:.0f} GB/s')
