#!/usr/bin/env python3
"""dv_col_moments + dv_recon_finalize in isolation at the evaluation's sizes (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
dev = 'cuda'
for M, n in ((8192, 8192), (8192, 4096), (2048, 2048), (2048, 1024)):
    X = 978
    nb = K.col_moment_blocks(n)
    x, r = torch.randn(M, X, device=dev), torch.randn(M, X, device=dev)
    rows = torch.rand(M, 6, device=dev) + 0.5
    part = torch.rand(nb, 3, X, dtype=torch.float64, device=dev)
    ll = torch.rand(M, device=dev)
    out = torch.zeros(4, dtype=torch.float64, device=dev)
    sel = torch.arange(0, M, M // n, dtype=torch.int32, device=dev) if n < M else None
    a = time_call(lambda: K.col_moments(None, x, r, sel=sel, part=part), repeats=20)
    b = time_call(lambda: K.recon_finalize(out, rows, part, X, sel=sel, n=n, ll=ll), repeats=20)
    print('M=%d n=%d blocks=%d: col_moments %.1f us  recon_finalize %.1f us' % (M, n, nb, a, b))
