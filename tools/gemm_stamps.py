#!/usr/bin/env python3
"""Phase timing INSIDE the GEMM's K loop from shader-clock stamps (lab build: tools/gemm_lab.sh 10000, then
DRVAE_HIP_LIB=build_lab/libdv_dbg10000.so python tools/gemm_stamps.py M N K akc bkc [tiling]).  Wave 0 of every
workgroup stamps: entry, loop start, then per K-tile phase: after compute (fragment reads + MFMA issue), after
stage_store (wait for the staged loads + LDS stores), after fetch (issue of the next loads), after the barrier."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402

M, N, Kd, akc, bkc = [int(v) for v in sys.argv[1:6]]
tiling = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lib = _lib.load()
dev = torch.device('cuda:0')
A = torch.randn(((M, Kd) if akc else (Kd, M))[0] + 1, ((M, Kd) if akc else (Kd, M))[1], device=dev)[:-1]
B = torch.randn(((N, Kd) if bkc else (Kd, N))[0] + 1, ((N, Kd) if bkc else (Kd, N))[1], device=dev)[:-1]
Cm = torch.empty(M, N, device=dev)
lib.dv_gemm_force_tiling(tiling)
for _ in range(5):
    K.gemm(Cm, A, B, akc, bkc, overread=True)
nwg = 8192
buf = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
p = buf.data_ptr()
lib.dv_gemm_set_option(5, (p & 0xffffffff) - (1 << 32) if (p & 0xffffffff) >= (1 << 31) else (p & 0xffffffff))
hi = p >> 32
lib.dv_gemm_set_option(6, hi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K.gemm(Cm, A, B, akc, bkc, overread=True)
e1.record()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 64)
used = t[:, 0] != 0
t = t[used]
print('kernel %.1f us (stamped build), workgroups stamped: %d' % (e0.elapsed_time(e1) * 1e3, len(t)))
# (s_memtime is a per-XCD counter with its own base: only differences inside one workgroup mean anything)
last = t.max(axis=1)
print('workgroup life (entry -> last stamp): median %d ticks, min %d, max %d' % (np.median(last - t[:, 0]), (last - t[:, 0]).min(), (last - t[:, 0]).max()))
print('prologue (entry -> loop start): median %d ticks' % np.median(t[:, 1] - t[:, 0]))
names = ['compute', 'stage_store', 'fetch', 'barrier']
n_ph = 0
for j in range(2, 61, 4):
    if (t[:, j + 3] != 0).all():
        n_ph += 1
d = {k: [] for k in names}
for ph in range(n_ph):
    prev = t[:, 1] if ph == 0 else t[:, 2 + 4 * ph - 1]
    for i, k in enumerate(names):
        cur = t[:, 2 + 4 * ph + i]
        d[k].append(np.median(cur - prev))
        prev = cur
print('phases stamped per workgroup: %d' % n_ph)
for k in names:
    print('  %-12s median ticks per phase: %s   mean %.0f' % (k, ' '.join('%5d' % v for v in d[k]), np.mean(d[k])))
tot = sum(np.mean(d[k]) for k in names)
print('  sum per K-tile phase: %.0f ticks  (8 MFMA 32x32x2 = 512 cycles of matrix pipe)' % tot)
