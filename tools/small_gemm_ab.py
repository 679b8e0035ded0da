#!/usr/bin/env python3
"""Round 6: what the product's GEMM launches cost on the fprop blocks' shapes (450 rows; 104 -> 200 with bias + ELU, 200 -> 200
heads), re-issued back to back from a hipGraph like tools/fprop_fused_probe.hip times its bare tile kernel -- same box, same
method: is a leaner small-product kernel (fragments straight from L2, every load in flight at once) worth building?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
dev = torch.device('cuda:0')
pad = lambda r, c: torch.zeros(r, (c + 3) // 4 * 4, device=dev)[:, :c]
for (M, N, Kd, tag) in ((450, 200, 102, 'fprop L1 (K 102 padded to 104) + bias + ELU'), (450, 200, 200, 'fprop heads as a plain Linear + bias'),
                        (300, 200, 100, 'z2F heads'), (596, 600, 100, 'decoder L1 + ELU'), (224, 200, 800, 'encoder heads (long K)')):
    x, W, b, out = pad(M, Kd), pad(N, Kd), torch.randn(N, device=dev), pad(M, N)
    x.normal_(); W.normal_()
    us = time_call(lambda: K.linear_fwd(out, x, W, b, split=N, act0='elu', act1='elu', overread=True, kpad=True), repeats=40)
    us0 = time_call(lambda: K.gemm(out, x, W, True, True, overread=True, kpad=True), repeats=40)
    print('%-50s %4d x %4d x %4d: Linear + bias + ELU %.2f us | plain product %.2f us' % (tag, M, N, Kd, us, us0))
