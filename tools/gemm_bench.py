#!/usr/bin/env python3
"""Micro-benchmark of the fp32-MFMA GEMM family on the shapes of the cfg-2 train step.
Times `repeats` back-to-back launches from a hipGraph with HIP events (GPU box only)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402

SHAPES = [  # (M, N, K, a_kc, b_kc, tag)
    (596, 1956, 600, 1, 1, 'decx heads fwd'), (596, 600, 1956, 1, 0, 'decx heads dX'),
    (1956, 600, 596, 0, 0, 'decx heads dW'), (224, 800, 978, 1, 1, 'enc L1 fwd'), (800, 978, 224, 0, 0, 'enc L1 dW'),
    (224, 200, 800, 1, 1, 'enc heads fwd'), (596, 600, 100, 1, 1, 'decx L1 fwd'), (600, 100, 596, 0, 0, 'decx L1 dW'),
    (450, 200, 102, 1, 1, 'fp L1 fwd'), (200, 102, 450, 0, 0, 'fp L1 dW'), (450, 102, 200, 1, 0, 'fp L1 dX'),
    (300, 2, 200, 1, 1, 'clf fwd'), (596, 600, 1956, 1, 1, 'long-K fwd layout'),
]


def time_call(fn, repeats=20):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(repeats):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / repeats)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tilings', default='0,1,2,3')
    ap.add_argument('--opts', default='', help='comma list key=value for dv_gemm_set_option')
    ap.add_argument('--check', action='store_true', help='compare every result with torch (max abs error printed)')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda:0')
    for kv in filter(None, args.opts.split(',')):
        k, v = kv.split('=')
        K.gemm_set_option(int(k), int(v))
    tilings = [int(t) for t in args.tilings.split(',')]
    print('%-18s %-22s' % ('shape', 'MxNxK (layout)') + ''.join('  t%d: us / TF/s   ' % t for t in tilings))
    for (M, N, Kd, akc, bkc, tag) in SHAPES:
        A = torch.randn(((M, Kd) if akc else (Kd, M))[0] + 1, ((M, Kd) if akc else (Kd, M))[1], device=dev)[:-1]
        B = torch.randn(((N, Kd) if bkc else (Kd, N))[0] + 1, ((N, Kd) if bkc else (Kd, N))[1], device=dev)[:-1]
        Cm = torch.empty(M, N, device=dev)
        row = '%-18s %-22s' % (tag, '%dx%dx%d (%d%d)' % (M, N, Kd, akc, bkc))
        for t in tilings:
            K.gemm_force_tiling(t)
            us = time_call(lambda: K.gemm(Cm, A, B, akc, bkc, overread=True))
            row += '  %7.2f / %6.2f   ' % (us, 2.0 * M * N * Kd / us / 1e6)
            if args.check:
                ref = (A if akc else A.t()).double() @ (B.t() if bkc else B).double()
                row += '[%.1e] ' % float((Cm.double() - ref).abs().max() / ref.abs().max())
        K.gemm_force_tiling(0)
        print(row, flush=True)


if __name__ == '__main__':
    main()
