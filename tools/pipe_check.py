#!/usr/bin/env python3
"""The hand-pipelined GEMM tilings (drvae_amd/csrc/gemm_pipe.inc) against an fp64 reference on ragged shapes (all three
layouts, plain and fused epilogues), then timed against the register-staged 128x128 tiling and the vendor library:
    python tools/pipe_check.py --tilings 40[,41,...] [--big] [--vendor]
(GPU box only; lab tilings need DRVAE_HIP_LIB=build_lab/libdrvae_lab.so)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402

CHECK = [(128, 256, 16), (128, 256, 64), (130, 260, 100), (1000, 516, 200), (257, 1028, 36), (64, 40, 20), (513, 300, 1024)]
BIG = [(1536, 2048, 20000), (2048, 2048, 4096), (8192, 8192, 2048), (8192, 40000, 2048), (8192, 2048, 40000), (40000, 2048, 8192), (4096, 2048, 20000)]


def operands(M, N, Kd, akc, bkc, dev, pad=0):
    sa = (M, Kd) if akc else (Kd, M)
    sb = (N, Kd) if bkc else (Kd, N)
    A = torch.randn(sa[0] + 1, sa[1] + pad, device=dev)[:-1, :sa[1]]
    B = torch.randn(sb[0] + 1, sb[1] + pad, device=dev)[:-1, :sb[1]]
    return A, B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tilings', default='40')
    ap.add_argument('--big', action='store_true')
    ap.add_argument('--vendor', action='store_true')
    ap.add_argument('--maps', default='', help='comma list of tile maps (dv_gemm_set_option(0, v)) to sweep on the big shapes')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda:0')
    tilings = [int(t) for t in args.tilings.split(',')]
    bad = 0
    for t in tilings:
        if K.gemm_force_tiling(t) != 0:
            print('tiling %d: not in this library' % t)
            continue
        worst = 0.0
        for (M, N, Kd) in CHECK:
            for akc, bkc in ((1, 1), (1, 0), (0, 0)):
                for pad in (0, 4):
                    A, B = operands(M, N, Kd, akc, bkc, dev, pad)
                    ref = ((A if akc else A.t()).double() @ (B.t() if bkc else B).double())
                    Cm = torch.full((M, N), 7.0, device=dev)
                    K.gemm(Cm, A, B, akc, bkc, overread=True)
                    e = float((Cm.double() - ref).abs().max() / ref.abs().max())
                    # fused epilogue: alpha, beta, bias + activation (forward layout) / activation backward
                    C2 = torch.randn(M, N, device=dev)
                    old = C2.clone()
                    if akc and bkc:
                        b = torch.randn(N, device=dev)
                        K.gemm(C2, A, B, akc, bkc, overread=True, epi=K.EPI_FWD, bias=b, act0='elu', act1='elu')
                        ref2 = torch.nn.functional.elu(ref + b.double())
                    elif akc:        # activation backward of the layer below fused into dy W, accumulating
                        y = torch.nn.functional.elu(torch.randn(M, N, device=dev))
                        K.gemm(C2, A, B, akc, bkc, overread=True, alpha=-0.5, beta=1.0, epi=K.EPI_BWD, yref=y, act0='elu', act1='elu')
                        dact = torch.where(y > 0, torch.ones_like(y), y + 1.0).double()
                        ref2 = -0.5 * ref * dact + old.double()
                    else:
                        K.gemm(C2, A, B, akc, bkc, overread=True, alpha=-0.5, beta=1.0)
                        ref2 = -0.5 * ref + old.double()
                    e2 = float((C2.double() - ref2).abs().max() / ref2.abs().max())
                    worst = max(worst, e, e2)
                    if not (e < 2e-5 and e2 < 2e-5):
                        bad += 1
                        print('  MISMATCH t%d %dx%dx%d (%d%d) pad %d: %.2e / %.2e' % (t, M, N, Kd, akc, bkc, pad, e, e2))
        print('tiling %d: worst relative error %.2e over %d cases' % (t, worst, len(CHECK) * 6), flush=True)
    K.gemm_force_tiling(0)
    if args.big:
        maps = [int(m) for m in args.maps.split(',')] if args.maps else [-1]
        for (M, N, Kd) in BIG:
            for akc, bkc in ((1, 1), (1, 0), (0, 0)):
                A, B = operands(M, N, Kd, akc, bkc, dev)
                Cm = torch.empty(M, N, device=dev)
                row = '%dx%dx%d (%d%d):' % (M, N, Kd, akc, bkc)
                for t in [3, 1] + tilings:
                    if K.gemm_force_tiling(t) != 0:
                        continue
                    for mp in maps:
                        K.gemm_set_option(0, mp)
                        us = time_call(lambda: K.gemm(Cm, A, B, akc, bkc, overread=True), repeats=5)
                        row += '  t%d%s %.0f us %.1f TF' % (t, '' if mp < 0 else '/m%d' % mp, us, 2.0 * M * N * Kd / us / 1e6)
                K.gemm_set_option(0, -1)
                K.gemm_force_tiling(0)
                if args.vendor:
                    Am = A if akc else A.t()
                    Bm = B.t() if bkc else B
                    us = time_call(lambda: torch.mm(Am, Bm, out=Cm), repeats=5)
                    row += '  | vendor %.0f us %.1f TF' % (us, 2.0 * M * N * Kd / us / 1e6)
                print(row, flush=True)
                del A, B, Cm
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
