#!/usr/bin/env python3
"""The chip-filling products of the wide configuration on one tiling, all three layouts:
    python tools/gemm_big.py [tiling=3] [M N K]
x W^T (11), dy W (10), dy^T x (00); TF/s over back-to-back launches from a hipGraph (tools/gemm_bench.time_call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402

tiling = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, N, Kd = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else (8192, 8192, 2048)
lib = _lib.load()
dev = torch.device('cuda:0')
K.gemm_force_tiling(tiling)
out = []
for akc, bkc in ((1, 1), (1, 0), (0, 0)):
    A = torch.randn(((M, Kd) if akc else (Kd, M))[0] + 1, ((M, Kd) if akc else (Kd, M))[1], device=dev)[:-1]
    B = torch.randn(((N, Kd) if bkc else (Kd, N))[0] + 1, ((N, Kd) if bkc else (Kd, N))[1], device=dev)[:-1]
    Cm = torch.empty(M, N, device=dev)
    us = time_call(lambda: K.gemm(Cm, A, B, akc, bkc, overread=True))
    out.append('%d%d: %.0f us %.1f TF/s' % (akc, bkc, us, 2.0 * M * N * Kd / us / 1e6))
print('%s t%d %dx%dx%d  ' % (os.environ.get('DRVAE_HIP_LIB', 'default'), tiling, M, N, Kd) + ' | '.join(out))
