import sys, os, torch
sys.path.insert(0, '/root/repo')
import drvae_amd.kernels as K
from drvae_amd import _lib
sys.path.insert(0, '/root/repo/tools')
from gemm_bench import time_call
lib = _lib.load(); dev = torch.device('cuda:0')
for (M, N, Kd) in [(596, 600, 1956), (596, 608, 1952), (224, 200, 800), (640, 640, 2048)]:
    for (akc, bkc) in [(1, 1), (1, 0), (0, 0)]:
        A = torch.randn((M, Kd) if akc else (Kd, M), device=dev)
        B = torch.randn((N, Kd) if bkc else (Kd, N), device=dev)
        Cm = torch.empty(M, N, device=dev)
        row = '%dx%dx%d (%d%d)' % (M, N, Kd, akc, bkc)
        for t in (1, 2):
            lib.dv_gemm_force_tiling(t)
            us = time_call(lambda: K.gemm(Cm, A, B, akc, bkc, overread=True))
            row += '  t%d %7.2f us %6.2f TF' % (t, us, 2.0 * M * N * Kd / us / 1e6)
        print(row, flush=True)
