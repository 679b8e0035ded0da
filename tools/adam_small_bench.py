#!/usr/bin/env python3
"""The optimiser sweep at cfg-2 sizes (0.9 M / 1.2 M / 2.1 M parameters), 40 launches per hipGraph replay: us per launch.
python tools/adam_small_bench.py   (GPU box only)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K

dev = 'cuda'
for n in (300000, 900000, 1200000, 2100000):
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    step = torch.ones(1, dtype=torch.int32, device=dev) * 1000
    halt = torch.zeros(4, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            K.adam_l2(p, g, m, v, step, lr=5e-4, weight_decay=0.05, halt=halt)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(40):
                K.adam_l2(p, g, m, v, step, lr=5e-4, weight_decay=0.05, halt=halt)
        best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / 40)
    print('n = %8d: %6.2f us per launch  (%.2f TB/s)' % (n, best, n * 28 / best / 1e6))
