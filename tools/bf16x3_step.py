#!/usr/bin/env python3
"""LAB ONLY (round-5 review, item 10): the cfg-5 train step with its three chip-filling products -- decoder heads forward
(8192 x 40000 x 2048), their weight gradient (40000 x 2048 x 8192) and data gradient (8192 x 2048 x 40000) -- on split-bf16
emulation: every fp32 operand -> three bf16 terms in ONE pass (tools/lab/bf16x3_split.hip: 4 B read + 12 B written per element, laid
out as the six K-segments hi hi | hi mid | mid hi | hi lo | lo hi | mid mid), then ONE library bf16 GEMM with fp32 accumulation and
fp32 output over K' = 6 K.  dtype of this experiment: bf16 x 3 (fp32 accumulate); the product path and the headline stay fp32 MFMA.
Reports: ms per step of both forms (captured graphs), and losses / gradients of one step from identical state (norm-wise)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import drvae_amd.kernels as K  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libbf16x3.so'))
lib.bf16x3_split.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.bf16x3_split.restype = C.c_int
_bufs = {}
BIG = 1e11          # MACs from which a plain product takes the emulated path


def split(x, axis, side, tag):
    R, Cc = x.shape
    key = (tag, R, Cc, axis, side)
    if key not in _bufs:
        _bufs[key] = torch.empty((R, 6 * Cc) if axis == 1 else (6 * R, Cc), dtype=torch.bfloat16, device=x.device)
    dst = _bufs[key]
    assert x.stride(1) == 1 and x.stride(0) % 4 == 0 and Cc % 8 == 0
    rc = lib.bf16x3_split(x.data_ptr(), x.stride(0), R, Cc, dst.data_ptr(), axis, side, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return dst


real_gemm, real_pair = K.gemm, K.linear_bwd_pair
stats = {'fwd': 0, 'pair': 0}


def lab_gemm(Cm, A, B, a_kc, b_kc, **kw):
    plain = a_kc and b_kc and all(kw.get(k) is None for k in kw if k not in ('overread', 'kpad'))
    if plain and Cm.is_contiguous() and float(Cm.shape[0]) * Cm.shape[1] * A.shape[1] >= BIG:
        stats['fwd'] += 1
        torch.mm(split(A, 1, 0, 'fa'), split(B, 1, 1, 'fb').t(), out_dtype=torch.float32, out=Cm)
        return
    real_gemm(Cm, A, B, a_kc, b_kc, **kw)


def lab_pair(dW, dbias, dx, dpre, x, W, *, kscale=None, alpha=1.0, beta_x=0.0, yref=None, act=0, shift=0.0, overread=False,
             publish=None, npad=False, npad_x=False, klq=None):
    M, N = dpre.shape
    ok = (dbias is None and kscale is None and alpha == 1.0 and beta_x == 0.0 and klq is None and publish is None and dx is not None
          and dW.is_contiguous() and dx.is_contiguous() and float(M) * N * x.shape[1] >= BIG)
    if not ok:
        return real_pair(dW, dbias, dx, dpre, x, W, kscale=kscale, alpha=alpha, beta_x=beta_x, yref=yref, act=act, shift=shift,
                         overread=overread, publish=publish, npad=npad, npad_x=npad_x, klq=klq)
    stats['pair'] += 1
    # dW[N, K] = dpre^T x (sum over the M rows): segments along the rows of both
    torch.mm(split(dpre, 0, 0, 'wa').t(), split(x, 0, 1, 'wb'), out_dtype=torch.float32, out=dW)
    # dx[M, K] = (dpre W) * act'(yref) (sum over the N columns of dpre = the rows of W)
    torch.mm(split(dpre, 1, 0, 'xa'), split(W, 0, 1, 'xb'), out_dtype=torch.float32, out=dx)
    if yref is not None:
        K.act_bwd_(dx, yref, act0=act, act1=act, shift0=shift, shift1=shift)


def run(lab):
    K.gemm, K.linear_bwd_pair = (lab_gemm, lab_pair) if lab else (real_gemm, real_pair)
    dev = torch.device('cuda', 0)
    cfg, eng, arena, batch, desc = bench.build('wide', dev, 0, 1)
    eng.train_step()                         # iteration 0 (eager), identical in both runs up to the products' arithmetic
    torch.cuda.synchronize()
    l0 = eng.losses()
    g0 = arena.grad.clone()
    eng.capture()
    for _ in range(2):
        eng.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        eng.replay()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    out = dict(ms=ms, losses=l0, grad=g0, final=eng.losses(), shapes=arena)
    K.gemm, K.linear_bwd_pair = real_gemm, real_pair
    return out


def breakdown():
    """the pieces of the three products in isolation (HIP events, best of 5)"""
    dev = torch.device('cuda', 0)
    M, N, Kd = 8192, 40000, 2048
    x, W, dpre = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev), torch.randn(M, N, device=dev)
    out_f, out_w, out_x = torch.empty(M, N, device=dev), torch.empty(N, Kd, device=dev), torch.empty(M, Kd, device=dev)

    def t(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    for name, fn in (('split x      [M,K] axis 1', lambda: split(x, 1, 0, 'b1')), ('split W      [N,K] axis 1', lambda: split(W, 1, 1, 'b2')),
                     ('split dpre   [M,N] axis 0', lambda: split(dpre, 0, 0, 'b3')), ('split x      [M,K] axis 0', lambda: split(x, 0, 1, 'b4')),
                     ('split dpre   [M,N] axis 1', lambda: split(dpre, 1, 0, 'b5')), ('split W      [N,K] axis 0', lambda: split(W, 0, 1, 'b6'))):
        print('  %-28s %.3f ms' % (name, t(fn)))
    a1, b1 = split(x, 1, 0, 'b1'), split(W, 1, 1, 'b2')
    a2, b2 = split(dpre, 0, 0, 'b3'), split(x, 0, 1, 'b4')
    a3, b3 = split(dpre, 1, 0, 'b5'), split(W, 0, 1, 'b6')
    fl = 2.0 * M * N * Kd
    for name, fn in (('forward  mm([M,6K], [N,6K]^T)', lambda: torch.mm(a1, b1.t(), out_dtype=torch.float32, out=out_f)),
                     ('dW       mm([6M,N]^T, [6M,K])', lambda: torch.mm(a2.t(), b2, out_dtype=torch.float32, out=out_w)),
                     ('dW^T     mm([6M,K]^T, [6M,N])', lambda: torch.mm(b2.t(), a2, out_dtype=torch.float32)),
                     ('dX       mm([M,6N], [6N,K])', lambda: torch.mm(a3, b3, out_dtype=torch.float32, out=out_x)),
                     ('fp32 forward (product kernel)', lambda: real_gemm(out_f, x, W, True, True)),
                     ('fp32 dW', lambda: real_gemm(out_w, dpre, x, False, False)),
                     ('fp32 dX', lambda: real_gemm(out_x, dpre, W, True, False))):
        ms = t(fn)
        print('  %-32s %.3f ms = %.1f TF/s-equivalent' % (name, ms, fl / ms / 1e9))


def golden():
    """the reference-generated cfg-5 golden case (tests/golden/model_cfg5_wide.npz: 20 000 genes, 1024 rows, L = 4 -- evaluation
    losses, first-step gradients, train-step losses and post-Adam parameters) through the SAME assertions as the product's test
    (tests/test_gpu_engine.py::test_train_steps_match_reference_golden), with the three products on the lab form"""
    from tests import test_gpu_engine as T
    K.gemm, K.linear_bwd_pair = lab_gemm, lab_pair
    try:
        T.test_train_steps_match_reference_golden('cfg5_wide', torch.device('cuda', 0))
    finally:
        K.gemm, K.linear_bwd_pair = real_gemm, real_pair
    print('cfg-5 golden case on the lab form: every assertion of the product test holds (losses 1e-4, gradients / parameters 1e-4 '
          'norm-wise); emulated calls: forward %d, paired backward %d' % (stats['fwd'], stats['pair']))


if __name__ == '__main__':
    if '--golden' in sys.argv:
        golden()
        sys.exit(0)
    if '--breakdown' in sys.argv:
        breakdown()
        sys.exit(0)
    a = run(False)
    b = run(True)
    print('emulated calls per captured step (incl. warm-up passes): forward %d, paired backward %d' % (stats['fwd'], stats['pair']))
    rel = max(abs(a['losses'][k] - b['losses'][k]) / max(abs(a['losses'][k]), 1e-6) for k in a['losses'] if k != 'MMD')
    ge = float((a['grad'] - b['grad']).norm() / a['grad'].norm())
    print('cfg 5 step: fp32 MFMA %.2f ms | split-bf16 lab form %.2f ms (%.2fx)' % (a['ms'], b['ms'], a['ms'] / b['ms']))
    print('one step from identical state: largest relative difference of the loss scalars %.2e; gradient arena norm-wise %.2e'
          % (rel, ge))
    print('after the timed replays: ELBO fp32 %.4f | lab %.4f' % (a['final']['ELBO'], b['final']['ELBO']))
    fl = 3 * 2.0 * 8192 * 40000 * 2048
    print('(the three products: %.1f TFLOP of the step\'s 4.32)' % (fl / 1e12))
