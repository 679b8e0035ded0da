#!/usr/bin/env python3
"""Where the decoder-heads forward launch spends its time beyond the K loop: the same 596 x 1956 product with a plain
epilogue, bias, bias + softplus, the fused NLL heads -- at K = 600 and at K = 8 (epilogue + launch only):
    python tools/epi_bench.py [key=value,...]           (GPU box only)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K  # noqa: E402
from drvae_amd import _lib  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402

lib = _lib.load()
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else '').split(',')):
    k, v = kv.split('=')
    K.gemm_set_option(int(k), int(v))
dev = torch.device('cuda:0')
M, S = 596, 978
for Kd in (600, 8):
    x, W, b = torch.randn(M, Kd, device=dev), torch.randn(2 * S, Kd, device=dev) * Kd ** -0.5, torch.randn(2 * S, device=dev)
    q = torch.empty(M, 2 * S, device=dev)
    xt, coef = torch.randn(M, S, device=dev), torch.randn(M, device=dev)
    part = torch.empty(M, K.heads_tiles(S), device=dev)
    t = {}
    t['plain'] = time_call(lambda: K.gemm(q, x, W, True, True, overread=True))
    t['bias'] = time_call(lambda: K.linear_fwd(q, x, W, b, overread=True))
    t['bias+elu'] = time_call(lambda: K.linear_fwd(q, x, W, b, overread=True, act0='elu', act1='elu'))
    t['bias+softplus head'] = time_call(lambda: K.linear_fwd(q, x, W, b, overread=True, split=S, act1='softplus', shift1=1e-3))
    t['heads NLL'] = time_call(lambda: K.linear_heads(q, x, W, b, nll=dict(x=xt, coef=coef, part=part), overread=True, split=S,
                                                     act1='softplus', shift1=1e-3))
    print('K=%d: ' % Kd + '  '.join('%s %.2f us' % kv for kv in t.items()), flush=True)
