#!/usr/bin/env python3
"""Run ONE GEMM shape/tiling a few times (for rocprofv3 --pmc passes): gemm_one.py M N K akc bkc tiling [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402

M, N, Kd, akc, bkc, tiling = [int(v) for v in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
lib = _lib.load()
dev = torch.device('cuda:0')
A = torch.randn(((M, Kd) if akc else (Kd, M))[0] + 1, ((M, Kd) if akc else (Kd, M))[1], device=dev)[:-1]
B = torch.randn(((N, Kd) if bkc else (Kd, N))[0] + 1, ((N, Kd) if bkc else (Kd, N))[1], device=dev)[:-1]
Cm = torch.empty(M, N, device=dev)
K.gemm_force_tiling(tiling)
for _ in range(reps):
    K.gemm(Cm, A, B, akc, bkc, overread=True)
torch.cuda.synchronize()
