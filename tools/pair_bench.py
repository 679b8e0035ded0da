#!/usr/bin/env python3
"""dv_gemm_pair (dW = dy^T x with the bias gradient, dX = dy W with the activation backward) at the cfg-2 layer sizes:
    python tools/pair_bench.py [key=value,... for dv_gemm_set_option]      (GPU box only)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402
from tests import kernel_ref as R  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402


def main():
    lib = _lib.load()
    for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else '').split(',')):
        k, v = kv.split('=')
        K.gemm_set_option(int(k), int(v))
    dev = torch.device('cuda:0')
    for (M, N, Kd, tag) in [(596, 1956, 600, 'decoder heads'), (224, 800, 980, 'encoder L1 (x rows padded to 980)'),
                            (596, 600, 100, 'decoder L1'), (224, 200, 800, 'encoder heads'), (450, 200, 104, 'fprop L1')]:
        x, W = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev) * Kd ** -0.5
        dpre, yprev = torch.randn(M, N, device=dev), torch.randn(M, Kd, device=dev)
        dW, db, dx = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev), torch.empty(M, Kd, device=dev)
        rW, rb, rx = torch.empty_like(dW), torch.empty_like(db), torch.empty_like(dx)
        f = lambda: K.linear_bwd_pair(dW, db, dx, dpre, x, W, yref=yprev, act='elu', overread=True)
        f()
        R.linear_bwd_pair(rW, rb, rx, dpre, x, W, yref=yprev, act='elu')
        torch.cuda.synchronize()
        err = max(float((dW - rW).abs().max() / rW.abs().max()), float((db - rb).abs().max() / rb.abs().max()),
                  float((dx - rx).abs().max() / rx.abs().max()))
        us = time_call(f)
        print('%-36s dW %dx%dx%d || dX %dx%dx%d: %.2f us  (%.1f TF/s; max rel err %.1e)'
              % (tag, N, Kd, M, M, Kd, N, us, 4.0 * M * N * Kd / us / 1e6, err), flush=True)


if __name__ == '__main__':
    main()
