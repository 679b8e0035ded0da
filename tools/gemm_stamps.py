#!/usr/bin/env python3
"""Per-phase cycle breakdown of the GEMM K-loop (block 0 / wave 0), tuning only."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from drvae_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
buf = torch.zeros(8, dtype=torch.int64, device=dev)
lib.dv_gemm_debug_stamps.argtypes = [ctypes.c_void_p]
for (M, N, Kd, akc, bkc, tag) in [(596, 1956, 600, 1, 1, 'fwd'), (596, 600, 1956, 1, 0, 'dX'), (1956, 600, 596, 0, 0, 'dW'),
                                  (224, 200, 800, 1, 1, 'small'), (300, 2, 200, 1, 1, 'clf')]:
    A = torch.randn((M, Kd) if akc else (Kd, M), device=dev)
    B = torch.randn((N, Kd) if bkc else (Kd, N), device=dev)
    Cm = torch.empty(M, N, device=dev)
    for t in (1, 2):
        lib.dv_gemm_force_tiling(t)
        lib.dv_gemm_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
        for _ in range(3):
            K.gemm(Cm, A, B, akc, bkc)
        torch.cuda.synchronize()
        lib.dv_gemm_debug_stamps(None)
        v = buf.tolist()
        n = max(v[5], 1)
        print('%-6s t%d iters=%3d  per-iter cycles: mfma+ldsread=%6.0f  vmwait+ldswrite=%6.0f  loadissue=%6.0f  barrier=%6.0f  loophead=%5.0f'
              % (tag, t, v[5], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n))
lib.dv_gemm_force_tiling(0)
