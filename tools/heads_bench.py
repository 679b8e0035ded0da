#!/usr/bin/env python3
"""Micro-benchmark of dv_gemm_heads against the unfused sequences it replaces (GPU box only):
heads GEMM + reparam_fwd, heads GEMM + nll_rows_fwdbwd, at the cfg-2 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402


def main():
    lib = _lib.load()
    if len(sys.argv) > 1:
        K.gemm_set_option(4, int(sys.argv[1]))     # 1: four-wave variant of the heads kernel
    dev = torch.device('cuda:0')
    r = lambda *s: torch.randn(*s, device=dev)
    for (M, S, Kd, tag) in [(224, 100, 800, 'enc heads + samples'), (300, 100, 100, 'z2F + sample'),
                            (596, 978, 600, 'decx heads + NLL')]:
        x, W, b = r(M, Kd), r(2 * S, Kd) * Kd ** -0.5, r(2 * S)
        q = torch.empty(M, 2 * S, device=dev)
        if 'NLL' in tag:
            xt, coef = r(M, S), r(M)
            part = torch.empty(M, K.heads_tiles(S), device=dev)
            row, dq = torch.empty(M, device=dev), torch.empty(M, 2 * S, device=dev)
            kw = dict(split=S, act1='softplus', shift1=1e-3)
            fused = lambda: K.linear_heads(dq, x, W, b, nll=dict(x=xt, coef=coef, part=part), overread=True, **kw)
            gemm = lambda: K.linear_fwd(q, x, W, b, overread=True, **kw)
            rows = lambda: K.nll_rows_fwdbwd(row, dq[:, :S], dq[:, S:], coef, xt, q[:, :S], q[:, S:], sd_act='softplus',
                                             sd_shift=1e-3)
        else:
            L = 2
            eps, z = r(L * M, S), torch.empty(L * M, S, device=dev)
            ptr = torch.arange(0, L * M + 1, L, dtype=torch.int32, device=dev)
            rws = torch.arange(L * M, dtype=torch.int32, device=dev)
            src = (torch.arange(L * M, device=dev) // L).to(torch.int32)
            kw = dict(split=S, shift1=-2.0)
            fused = lambda: K.linear_heads(q, x, W, b, sample=dict(eps=eps, out=z, n_src=M, seg_ptr=ptr, seg_rows=rws),
                                           overread=True, **kw)
            gemm = lambda: K.linear_fwd(q, x, W, b, overread=True, **kw)
            rows = lambda: K.reparam_fwd(z, q[:, :S], q[:, S:], eps, src_idx=src)
        tf, tg, tr = time_call(fused), time_call(gemm), time_call(rows)
        tb = time_call(lambda: (gemm(), rows()))
        print('%-22s M=%d heads=2x%d K=%d: fused %.2f us | gemm %.2f + rows %.2f = %.2f (back to back %.2f)'
              % (tag, M, S, Kd, tf, tg, tr, tg + tr, tb), flush=True)


if __name__ == '__main__':
    main()
