#!/usr/bin/env python3
"""How does one GEMM's time scale with M at fixed N, K?  (latency-bound vs throughput-bound)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from drvae_amd import _lib
from tools.gemm_bench import time_call
lib = _lib.load()
dev = torch.device('cuda:0')
N, Kd = 1956, 600
for t in (2, 1):
    K.gemm_force_tiling(t)
    row = 't%d ' % t
    for M in (32, 64, 128, 149, 298, 447, 596, 894, 1192, 2384, 4768):
        A = torch.randn(M + 1, Kd, device=dev)[:-1]
        B = torch.randn(N + 1, Kd, device=dev)[:-1]
        Cm = torch.empty(M, N, device=dev)
        us = time_call(lambda: K.gemm(Cm, A, B, 1, 1, overread=True))
        row += ' M=%d: %.1fus %.0fTF |' % (M, us, 2.0 * M * N * Kd / us / 1e6)
    print(row)
