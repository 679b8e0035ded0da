#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
for (M, N, Kd) in [(32768, 600, 100), (32768, 608, 104), (32768, 640, 96), (24576, 200, 200)]:
    A, B, C = mat(M, Kd), mat(N, Kd), mat(M, N)
    A._base[:, Kd:] = 0
    B._base[:, Kd:] = 0
    bias = torch.randn(N, device='cuda')
    row = '%-20s' % ('%dx%dx%d' % (M, N, Kd))
    for opt in (0, -1):
        K.gemm_set_option(5, opt)      # (lab: -1 switched the skinny kernel of tools/lab/gemm_skinny.inc off)
        us = time_call(lambda: K.gemm(C, A, B, True, True, overread=True, kpad=True), repeats=10)
        row += ' %s plain %6.1f |' % ('sk' if opt == 0 else 'ks', us)
        us = time_call(lambda: K.linear_fwd(C, A, B, bias, act0='elu', act1='elu', overread=True, kpad=True), repeats=10)
        row += ' elu %6.1f |' % us
    K.gemm_set_option(5, 0)
    Ac, Bc = A.contiguous(), B.contiguous()
    Cc = torch.empty(M, N, device='cuda')
    row += ' vendor mm %6.1f |' % time_call(lambda: torch.mm(Ac, Bc.t(), out=Cc), repeats=10)
    row += ' elu(C) in place %6.1f | copy %6.1f' % (time_call(lambda: torch.nn.functional.elu(Cc, inplace=True), repeats=10),
                                                    time_call(lambda: Cc.copy_(Cc), repeats=10))
    print(row, flush=True)
