#!/usr/bin/env python3
"""Where the short-K products of the whole-set evaluation lose their time: the 128x256 pipe tiling (40) over M x N x K with
K = 600 against longer K, full / ragged N, plain / activation epilogue, and torch.mm (vendor)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
for (M, N, Kd) in [(32768, 1956, 600), (32768, 2048, 600), (32768, 1792, 600), (32768, 2048, 1200), (32768, 2048, 2400), (32768, 2048, 304),
                   (8192, 2048, 600), (16384, 1956, 600), (12288, 800, 978), (12288, 1024, 976)]:
    A, B, C = mat(M, Kd), mat(N, Kd), mat(M, N)
    bias = torch.randn(N, device='cuda')
    row = '%-20s' % ('%dx%dx%d' % (M, N, Kd))
    for t in (40, 0):
        K.gemm_force_tiling(t)
        for epi in ('plain', 'elu'):
            if epi == 'plain':
                us = time_call(lambda: K.gemm(C, A, B, True, True, overread=True, kpad=True), repeats=10)
            else:
                us = time_call(lambda: K.linear_fwd(C, A, B, bias, act0='elu', act1='elu', overread=True, kpad=True), repeats=10)
            row += ' t%d %s %7.1f us %5.1f TF |' % (t, epi, us, 2.0 * M * N * Kd / us / 1e6)
    K.gemm_force_tiling(0)
    Ac, Bc = A.contiguous(), B.contiguous()
    us = time_call(lambda: torch.mm(Ac, Bc.t(), out=C) if C.is_contiguous() else torch.mm(Ac, Bc.t()), repeats=10)
    row += ' vendor %7.1f us %5.1f TF' % (us, 2.0 * M * N * Kd / us / 1e6)
    print(row, flush=True)
