#!/usr/bin/env python3
"""Developer tool: print the per-kernel table of a bench.py JSON line (file argument or stdin)."""
import json, sys
lines = open(sys.argv[1]).read().splitlines() if len(sys.argv) > 1 else sys.stdin.read().splitlines()
d = json.loads([l for l in lines if l.startswith('{')][-1])
print(d['value'], d['unit'], ' ms/step', d['ms_per_step'], ' roofline', d['roofline']['kernel'], d['roofline']['frac'])
for k, v in d['kernels'].items():
    print(f"{k:45s} {v['avg_us']:8.1f} us  {v['tflops']} TF  {v['algo_gbs']} GB/s")
if 'cpu_baseline' in d:
    print('cpu', d['cpu_baseline'])
