import os, sys, torch
sys.path.insert(0, '/root/repo')
from yolo_amd import ops
n = 18466238
p, g, m, v = (torch.randn(n, device='cuda:0') for _ in range(4))
v.abs_()
for _ in range(5): ops.adam_step(p, g, m, v, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(50): ops.adam_step(p, g, m, v, t + 1)
e1.record(); torch.cuda.synchronize()
print(os.environ.get('TY1_ADAM_BLOCKS'), round(e0.elapsed_time(e1) * 1e3 / 50, 1), 'us')
