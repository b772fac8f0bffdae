"""Developer sweep (dev library, one GPU call): time of one bf16 halo layer for every (KA, KC) of the halo swizzle key
key = ((hp + KA * row) / RPW + KC * row) mod slots.  usage: halo_key_sweep.py N,H,cin,cout,pool"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
shape = sys.argv[1]
code = r'''
import sys, torch
sys.path.insert(0, %r)
from yolo_amd import ops
n, hh, cin, cout, pool = (int(v) for v in %r.split(','))
x = torch.rand((n, hh, hh, cin), device='cuda:0').to(torch.bfloat16)
w = (torch.randn((cout, 3, 3, cin), device='cuda:0') * 0.05).to(torch.bfloat16)
sc, sh = torch.ones(cout, device='cuda:0'), torch.zeros(cout, device='cuda:0')
out = ops.conv3x3_bn_leaky_bf16(x, w, sc, sh, bool(pool))
for _ in range(300): ops.conv3x3_bn_leaky_bf16(x, w, sc, sh, bool(pool), out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(600): ops.conv3x3_bn_leaky_bf16(x, w, sc, sh, bool(pool), out=out)
e1.record(); torch.cuda.synchronize()
print('%%.2f' %% (e0.elapsed_time(e1) / 600 * 1e3))
''' % (root, shape)
env = dict(os.environ, TY1_LIB=os.path.join(root, 'yolo_amd/lib/libty1_dev.so'))
res = {}
for ka in range(0, 8, 1):
    row = []
    for kc in range(8):
        env['TY1_HALO_KEY'] = f'{ka},{kc}'
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
        t = float(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else float('nan')
        row.append(t)
    print(f'KA={ka}: ' + ' '.join(f'{t:6.2f}' for t in row), flush=True)
env.pop('TY1_HALO_KEY')
out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
print('compiled-in key:', out.stdout.strip().splitlines()[-1])
