#!/usr/bin/env python3
"""What the fused epilogues cost the latency-bound products of a cfg-2 step (32 x 32 K-split pipe kernels): plain product
against bias + ELU forward, and plain against the activation-backward epilogue (yref load), us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
for M, N, Kd in [(224, 800, 980), (596, 600, 100), (450, 200, 104), (450, 200, 200), (300, 200, 100)]:
    x, W, out, b = mat(M, Kd), mat(N, Kd), mat(M, N), torch.randn(N, device='cuda')
    t0 = time_call(lambda: K.gemm(out, x, W, True, True, overread=True), repeats=40)
    t1 = time_call(lambda: K.linear_fwd(out, x, W, b, act0='elu', act1='elu', overread=True), repeats=40)
    dpre, dx, y = mat(M, N), mat(M, Kd), mat(M, Kd)
    t2 = time_call(lambda: K.linear_bwd_data(dx, dpre, W, overread=True), repeats=40)
    t3 = time_call(lambda: K.linear_bwd_data(dx, dpre, W, yref=y, act='elu', overread=True), repeats=40)
    print('%4d x %4d x %4d   fwd plain %5.2f  bias+elu %5.2f   |   dX plain %5.2f  * elu\'(yref) %5.2f' % (M, N, Kd, t0, t1, t2, t3), flush=True)
