#!/usr/bin/env python3
"""Developer tool: where does a conv layer differ from the oracle? (prints error statistics by image / row / column / channel)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_parity import _conv_case
for case in [(2, 112, 32, 64, True), (3, 56, 64, 128, True), (1, 6, 64, 128, False), (5, 7, 512, 1024, False)]:
    got, ref = _conv_case(*case, seed=case[2] + case[3])
    err = np.abs(got - ref)
    bad = err > 1e-4 * max(1.0, np.abs(ref).max())
    print(case, 'max err', err.max(), 'bad frac', bad.mean())
    if bad.any():
        print('  by image  ', bad.mean(axis=(1, 2, 3)))
        print('  by row    ', np.round(bad.mean(axis=(0, 2, 3)), 2)[:32])
        print('  by col    ', np.round(bad.mean(axis=(0, 1, 3)), 2)[:32])
        print('  by channel', np.round(bad.mean(axis=(0, 1, 2)), 2)[:64])
