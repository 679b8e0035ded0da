#!/usr/bin/env python3
"""The 128x256 pipe tiling with its CU's second resident workgroup started late (dv_gemm_tune.opt[5]: 0 off, else percent of the
estimated half tile): us / TF/s of plain-epilogue products per setting (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call

# (M, N, K, a_kc, b_kc)
SHAPES = [(32768, 1956, 600, 1, 1), (16384, 1956, 600, 1, 1), (8192, 1956, 600, 1, 1), (8192, 8192, 2048, 1, 1),
          (8192, 40000, 2048, 1, 1), (8192, 2048, 40000, 1, 0), (40000, 2048, 8192, 0, 0), (8192, 20000, 2048, 1, 1),
          (2048, 20000, 1536, 1, 1), (32768, 2048, 256, 1, 1)]
SET = [0, 50, 75, 100, 125, 150, 200]
if len(sys.argv) > 1:
    SET = [int(v) for v in sys.argv[1].split(',')]
print('%-26s' % 'M x N x K (layout)' + ''.join('  o5=%-4d us / TF/s ' % v for v in SET))
for M, N, Kd, akc, bkc in SHAPES:
    A = torch.randn((M, Kd) if akc else (Kd, M), device='cuda')
    B = torch.randn((N, Kd) if bkc else (Kd, N), device='cuda')
    C = torch.empty(M, N, device='cuda')
    row = '%-26s' % ('%d x %d x %d (%d%d)' % (M, N, Kd, akc, bkc))
    for v in SET:
        K.gemm_set_option(5, v)
        us = time_call(lambda: K.gemm(C, A, B, bool(akc), bool(bkc)), repeats=8)
        row += '  %8.1f / %5.1f   ' % (us, 2.0 * M * N * Kd / us / 1e6)
    K.gemm_set_option(5, 0)
    print(row, flush=True)
    del A, B, C
