#!/usr/bin/env python3
"""short-K / many-row products of the whole-set evaluation: the skinny kernel against the K-split tiling (opt[5] = -1)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
for (M, N, Kd) in [(32768, 600, 100), (16384, 600, 100), (8192, 600, 100), (24576, 200, 102), (24576, 200, 200), (6144, 200, 102),
                   (16384, 200, 100), (8192, 2048, 200), (4096, 600, 100)]:
    A, B, C = mat(M, Kd), mat(N, Kd), mat(M, N)
    A._base[:, Kd:] = 0
    B._base[:, Kd:] = 0
    bias = torch.randn(N, device='cuda')
    row = '%-20s' % ('%dx%dx%d' % (M, N, Kd))
    for opt in (0, -1):
        K.gemm_set_option(5, opt)
        us = time_call(lambda: K.linear_fwd(C, A, B, bias, act0='elu', act1='elu', overread=True, kpad=True), repeats=10)
        row += ' %s %7.1f us %5.1f TF |' % ('skinny ' if opt == 0 else 'k-split', us, 2.0 * M * N * Kd / us / 1e6)
    K.gemm_set_option(5, 0)
    print(row, flush=True)
