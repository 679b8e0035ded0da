#!/usr/bin/env python3
"""Forward products of the whole-set evaluation (big M, short K) with a plain epilogue, bias + ELU and the decoder's
(identity | softplus + 1e-3) heads: what the fused epilogue costs at these shapes.  python tools/epi_bench2.py [opts]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from drvae_amd import _lib
from tools.gemm_bench import time_call

lib = _lib.load()
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else '').split(',')):
    k, v = kv.split('=')
    K.gemm_set_option(int(k), int(v))
dev = torch.device('cuda:0')
for (M, N, Kd) in [(32768, 1956, 600), (16384, 1956, 600), (8192, 978, 600), (32768, 600, 100), (12288, 800, 978), (12288, 800, 980),
                   (24576, 200, 102), (24576, 200, 200), (12288, 200, 800)]:
    ldk = (Kd + 3) // 4 * 4
    x = torch.randn(M, ldk, device=dev)[:, :Kd]
    W = (torch.randn(N, Kd, device=dev) * Kd ** -0.5)
    b = torch.randn(N, device=dev)
    q = torch.empty(M, (N + 3) // 4 * 4, device=dev)[:, :N]
    t = {}
    t['plain'] = time_call(lambda: K.gemm(q, x, W, True, True, overread=True), repeats=5)
    t['bias'] = time_call(lambda: K.linear_fwd(q, x, W, b, overread=True), repeats=5)
    t['bias+elu'] = time_call(lambda: K.linear_fwd(q, x, W, b, overread=True, act0='elu', act1='elu'), repeats=5)
    t['id|softplus'] = time_call(lambda: K.linear_fwd(q, x, W, b, overread=True, split=N // 2, act1='softplus', shift1=1e-3), repeats=5)
    gf = 2.0 * M * N * Kd / 1e6
    print('%6d x %4d x %4d: ' % (M, N, Kd) + '  '.join('%s %.1f us (%.0f TF/s)' % (k, v, gf / v) for k, v in t.items()), flush=True)
