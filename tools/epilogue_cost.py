#!/usr/bin/env python3
"""What a fused epilogue (bias, ELU, identity | softplus heads) costs the 128 x 256 LDS-DMA kernel next to the plain one,
per shape (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
for M, N, Kd in [(32768, 1956, 600), (12288, 800, 980), (8192, 8192, 2048), (32768, 2048, 600), (32768, 2048, 2048)]:
    A, B, C = mat(M, Kd), mat(N, Kd), mat(M, N)
    bias = torch.randn(N, device='cuda')
    row = '%-22s' % ('%d x %d x %d' % (M, N, Kd))
    for t in (40,):
        K.gemm_force_tiling(t)
        for name, fn in [('plain', lambda: K.gemm(C, A, B, True, True, overread=True)),
                         ('bias+id', lambda: K.linear_fwd(C, A, B, bias, act0='identity', act1='identity', overread=True)),
                         ('bias+elu', lambda: K.linear_fwd(C, A, B, bias, act0='elu', act1='elu', overread=True)),
                         ('id|softplus', lambda: K.linear_fwd(C, A, B, bias, split=N // 2, act0='identity', act1='softplus', shift1=1e-3, overread=True))]:
            us = time_call(fn, repeats=10)
            row += '  %s %7.1f / %5.1f' % (name, us, 2.0 * M * N * Kd / us / 1e6)
    K.gemm_force_tiling(0)
    print(row, flush=True)
