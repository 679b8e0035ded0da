#!/usr/bin/env python3
"""Would the decoder heads' weight gradient (a leaf) ride better on the small paired launches behind it than beside the
heads' data gradient?  cfg-2 shapes, us per sequence (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from tools.gemm_bench import time_call
pad4 = lambda n: (n + 3) // 4 * 4
mat = lambda r, c: torch.randn(r, pad4(c), device='cuda')[:, :c]
R, X2, H, Z = 596, 1956, 600, 100
dpx, h, Wh, dWh, dbh, dh, yh = mat(R, X2), mat(R, H), mat(X2, H), mat(X2, H), torch.zeros(X2, device='cuda'), mat(R, H), mat(R, H)
zin, W1, dW1, db1, dz = mat(R, Z), mat(H, Z), mat(H, Z), torch.zeros(H, device='cuda'), mat(R, Z)
R2 = 300
dp2, z1, W2, dW2, db2, dz1 = mat(R2, 2 * Z), mat(R2, Z), mat(2 * Z, Z), mat(2 * Z, Z), torch.zeros(2 * Z, device='cuda'), mat(R2, Z)
half = X2 // 2

def now():
    K.linear_bwd_pair(dWh, dbh, dh, dpx, h, Wh, yref=yh, act='elu', overread=True)
    K.linear_bwd_pair(dW1, db1, dz, dh, zin, W1, overread=True)
    K.linear_bwd_pair(dW2, db2, dz1, dp2, z1, W2, beta_x=1.0, overread=True)

def riders():
    K.linear_bwd_data(dh, dpx, Wh, yref=yh, act='elu', overread=True)
    K.linear_bwd_pair(dWh[:half], dbh[:half], dz, dpx[:, :half], h, W1 if False else W1, overread=True) if False else None
    # rider 1: first half of the heads' dW beside the decoder L1 data gradient; L1's own small dW as a launch of its own
    K.linear_bwd_pair(dWh[:half], dbh[:half], dz, dh, zin, W1, overread=True) if False else None

def seq_new():
    K.linear_bwd_data(dh, dpx, Wh, yref=yh, act='elu', overread=True)
    _pair_mixed(dWh[:half], dbh[:half], dpx[:, :half], h, dz, dh, W1)
    K.linear_bwd_weight(dW1, dh, zin, dbias=db1, overread=True)
    _pair_mixed(dWh[half:], dbh[half:], dpx[:, half:], h, dz1, dp2, W2, beta_x=1.0)
    K.linear_bwd_weight(dW2, dp2, z1, dbias=db2, overread=True)

def _pair_mixed(dW, db, dpreW, xW, dx, dpreX, W, beta_x=0.0):
    """one paired launch whose two products belong to DIFFERENT layers: dW = dpreW^T xW  ||  dx = dpreX W"""
    import ctypes as C
    from drvae_amd import _lib
    d1 = K._gemm_desc(dW, dpreW, xW, False, False, a_colsum=db, overread=True)
    d2 = K._gemm_desc(dx, dpreX, W, True, False, beta=beta_x, overread=True)
    _lib.check(_lib.load().dv_gemm_pair(C.byref(d1), C.byref(d2), K._stream()), 'dv_gemm_pair')

parts = {
    'heads pair (dW || dX)': lambda: K.linear_bwd_pair(dWh, dbh, dh, dpx, h, Wh, yref=yh, act='elu', overread=True),
    'heads dX alone': lambda: K.linear_bwd_data(dh, dpx, Wh, yref=yh, act='elu', overread=True),
    'dec L1 pair': lambda: K.linear_bwd_pair(dW1, db1, dz, dh, zin, W1, overread=True),
    'half heads dW || dec L1 dX': lambda: _pair_mixed(dWh[:half], dbh[:half], dpx[:, :half], h, dz, dh, W1),
    'dec L1 dW alone': lambda: K.linear_bwd_weight(dW1, dh, zin, dbias=db1, overread=True),
    'z2F pair': lambda: K.linear_bwd_pair(dW2, db2, dz1, dp2, z1, W2, beta_x=1.0, overread=True),
    'half heads dW || z2F dX': lambda: _pair_mixed(dWh[half:], dbh[half:], dpx[:, half:], h, dz1, dp2, W2, beta_x=1.0),
    'z2F dW alone': lambda: K.linear_bwd_weight(dW2, dp2, z1, dbias=db2, overread=True),
    'SEQUENCE now (3 pairs)': now,
    'SEQUENCE riders (dX, 2 mixed pairs, 2 small dW)': seq_new,
}
for k, f in parts.items():
    print('%-50s %7.1f us' % (k, time_call(f, repeats=20)), flush=True)
