#!/usr/bin/env python3
"""What the vendor library (rocBLAS/hipBLASLt through torch.mm, fp32) needs for the step's products: a
reference point for the hand-written GEMM (GPU box only).  Times back-to-back launches from a hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import SHAPES, time_call  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda:0')
print('%-18s %-24s %s' % ('shape', 'MxNxK (layout)', 'torch.mm fp32: us / TF/s'))
for (M, N, Kd, akc, bkc, tag) in SHAPES + [(8192, 8192, 2048, 1, 1, 'large')]:
    A = torch.randn(M, Kd, device=dev) if akc else torch.randn(Kd, M, device=dev).t()
    B = torch.randn(N, Kd, device=dev).t() if bkc else torch.randn(Kd, N, device=dev)
    C = torch.empty(M, N, device=dev)
    t = time_call(lambda: torch.mm(A, B, out=C))
    print('%-18s %-24s %8.2f / %6.2f' % (tag, '%dx%dx%d (%d%d)' % (M, N, Kd, akc, bkc), t, 2.0 * M * N * Kd / t * 1e-6))
