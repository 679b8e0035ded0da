#!/usr/bin/env python3
"""The forward products of a whole-set evaluation (N1: 8192 + 2048 rows at cfg-2 size) under every tiling the library
carries: us / TF/s per tiling, first column = the dispatcher's own choice.  python tools/eval_gemm_bench.py [--lab]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call

SHAPES = [(32768, 1956, 600), (16384, 1956, 600), (12288, 800, 978), (8192, 800, 978), (32768, 600, 100), (16384, 600, 100),
          (24576, 200, 102), (24576, 200, 200), (12288, 200, 800), (8192, 200, 800), (16384, 200, 100), (8192, 100, 100),
          (4096, 1956, 600), (2048, 800, 978), (8192, 600, 100), (8192, 2048, 200), (8192, 2048, 256), (8192, 2048, 400),
          (32768, 600, 256), (32768, 600, 400)]
tilings = [0, 2, 1, 46, 40] + ([41, 43, 44] if '--lab' in sys.argv else [])
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
print('%-22s' % 'M x N x K' + ''.join('   t%-2d us / TF/s ' % t for t in tilings))
for M, N, Kd in SHAPES:
    A, B, C = mat(M, Kd), mat(N, Kd), mat(M, N)
    A._base[:, Kd:] = 0
    B._base[:, Kd:] = 0
    bias = torch.randn(N, device='cuda')
    row = '%-22s' % ('%d x %d x %d' % (M, N, Kd))
    for t in tilings:
        if K.gemm_force_tiling(t):
            row += '        n/a       '
            continue
        us = time_call(lambda: K.linear_fwd(C, A, B, bias, act0='elu', act1='elu', overread=True, kpad=True), repeats=10)
        row += '  %7.1f / %5.1f  ' % (us, 2.0 * M * N * Kd / us / 1e6)
    K.gemm_force_tiling(0)
    print(row, flush=True)
