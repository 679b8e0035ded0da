"""Tensor-level launchers for the C-ABI kernels (device fp32 tensors in, enqueue on the
current HIP stream, nothing returned that needs a sync).

Every function writes into caller-provided output tensors (``out=``-style), so the
callers own all memory and the sequence is hipGraph-capturable.  2-D tensors may be
row-strided views (``stride(1) == 1``); the row stride is passed as the leading
dimension.  There is no CPU path here: tensors must live on the GPU.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT, EPI_BWD, EPI_FWD, EPI_KLQ, EPI_PLAIN, GAUSS_LOGVAR, GAUSS_SIGMA, GemmDesc  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name='tensor'):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError('drvae_amd kernels need CUDA/HIP float32 tensors (%s is %s on %s); there is no CPU '
                           'fallback' % (name, t.dtype, t.device))
    if t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1 and t.size(-1) != 1:
        raise RuntimeError('%s must have unit inner stride' % name)
    return t.data_ptr()


def _i32(t, name='index'):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise RuntimeError('%s must be a contiguous CUDA int32 tensor' % name)
    return t.data_ptr()


def _ld(t):
    if t is None:
        return 0
    if t.dim() == 1:
        return t.numel()
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def _act(a):
    return ACT[a] if isinstance(a, str) else int(a)


# ------------------------------------------------------------------------------ GEMM
# Steering of the GEMM dispatcher for tests and tools: the state lives HERE, on the caller's side, and travels with every
# descriptor (``dv_gemm_desc.tune``); the library itself keeps none.
_TUNE = _lib.GemmTune()
_TUNE.opt[0] = -1


def gemm_force_tiling(t):
    """run every GEMM-family call on tiling code ``t`` (0: the dispatcher's heuristics); returns 0, or a non-zero status
    when this build of the library does not carry the tiling (lab tilings: ``python -m drvae_amd.build --lab``)"""
    if not _lib.load().dv_gemm_has_tiling(int(t)):
        return _lib.DV_ERR_UNSUPPORTED if hasattr(_lib, 'DV_ERR_UNSUPPORTED') else 4
    _TUNE.tiling = int(t)
    return 0


def gemm_set_option(key, value):
    """dispatcher option ``key`` (see ``dv_gemm_tune`` in include/drvae_hip.h)"""
    if not 0 <= int(key) < 10:
        return 1
    _TUNE.opt[int(key)] = int(value)
    return 0


def _tune_ptr():
    t = _TUNE
    if t.tiling == 0 and t.opt[0] == -1 and not any(t.opt[i] for i in range(1, 10)):
        return None
    return C.addressof(t)


def _gemm_desc(Cm, A, B, a_kc, b_kc, *, A2=None, a_kscale=None, alpha=1.0, beta=0.0, epi=EPI_PLAIN, scale=None,
               bias=None, split=None, act0=0, act1=0, shift0=0.0, shift1=0.0, resid=None, resid_cols=0, yref=None,
               a_colsum=None, colsum_beta=0.0, overread=False, publish=None, kpad=False, npad=False):
    M, N = Cm.shape
    if a_kc:
        K = A.shape[1] + (A2.shape[1] if A2 is not None else 0)
        assert A.shape[0] == M
    else:
        K = A.shape[0]
        assert A.shape[1] == M and A2 is None
    if b_kc:
        assert tuple(B.shape) == (N, K), (tuple(B.shape), N, K)
    else:
        assert tuple(B.shape) == (K, N), (tuple(B.shape), K, N)
    d = GemmDesc()
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = M, N, K, int(bool(a_kc)), int(bool(b_kc))
    d.A, d.lda = _f32(A, 'A'), _ld(A)
    if kpad and a_kc and b_kc and A2 is None and (K & 3) and min(_ld(A), _ld(B)) >= ((K + 3) & ~3) \
            and not ((A.data_ptr() | B.data_ptr()) & 15) and not ((_ld(A) | _ld(B)) & 3):
        # ``kpad``: the caller guarantees that the rows of both operands are ZERO from K up to the next multiple of 4
        # (row-padded activation buffers, the arena's row-padded weights): the product over the padded K is the same
        # number, and its operands qualify for the LDS-DMA kernels (16-B chunks along k)
        d.K = K = (K + 3) & ~3
    if npad and not b_kc and (N & 3) and min(_ld(B), _ld(Cm)) >= ((N + 3) & ~3) and not ((B.data_ptr() | Cm.data_ptr()) & 15) \
            and not ((_ld(B) | _ld(Cm)) & 3) and resid is None and scale is None and bias is None \
            and (yref is None or _ld(yref) >= ((N + 3) & ~3)):
        # ``npad``: the caller guarantees that the rows of B are ZERO from N up to the next multiple of 4 and that C's rows
        # are padded the same way: the product over the padded N writes zeros into C's pad columns (they are zero
        # anyway) and every output row ends on a 16-B store
        d.N = N = (N + 3) & ~3
    d.A2, d.lda2, d.K1 = _f32(A2, 'A2'), _ld(A2), (A.shape[1] if A2 is not None else K)
    d.a_kscale = _f32(a_kscale, 'a_kscale')
    d.B, d.ldb = _f32(B, 'B'), _ld(B)
    d.C, d.ldc = _f32(Cm, 'C'), _ld(Cm)
    d.alpha, d.beta, d.epilogue = alpha, beta, epi
    d.scale, d.bias = _f32(scale, 'scale'), _f32(bias, 'bias')
    d.split = N if split is None else split
    d.act0, d.act1, d.shift0, d.shift1 = _act(act0), _act(act1), shift0, shift1
    d.resid, d.ldr, d.resid_cols = _f32(resid, 'resid'), _ld(resid), resid_cols
    d.yref, d.ldy = _f32(yref, 'yref'), _ld(yref)
    d.a_colsum, d.colsum_beta = _f32(a_colsum, 'a_colsum'), colsum_beta
    d.flags = 3 if overread else 0
    if publish is not None:                 # (flag, counter, add): publish on kernel entry, see dv_flag_publish
        d.pub_flag, d.pub_ctr, d.pub_add = _i32(publish[0]), _i32(publish[1]), publish[2]
    d.tune = _tune_ptr()
    return d


def gemm(Cm, A, B, a_kc, b_kc, **kw):
    """C[M,N] = epilogue(alpha * Aop @ Bop) + beta*C, see ``dv_gemm`` in include/drvae_hip.h.
    ``overread``: rows of A and B may be over-read by up to 3 floats (padded / arena buffers)."""
    d = _gemm_desc(Cm, A, B, a_kc, b_kc, **kw)
    _lib.check(_lib.load().dv_gemm(C.byref(d), _stream()), 'dv_gemm')


def linear_bwd_pair(dW, dbias, dx, dpre, x, W, *, kscale=None, alpha=1.0, beta_x=0.0, yref=None, act=0, shift=0.0,
                    overread=False, publish=None, npad=False, npad_x=False, klq=None):
    """dW = dpre^T x (+ dbias) and dx = beta_x*dx + alpha*(dpre W) * act'(yref) in ONE launch when both fit
    the fused form of ``dv_gemm_pair`` (otherwise two launches).
    ``klq`` = dict(out, q, eps, coef, raw, kl_min, Z): the layer's input was [a sample z of the q rows | ...]: instead of
    dx the launch writes d/d(mu | logvar) of those rows incl. their prior term into ``out`` (M, 2Z) -- the epilogue
    DV_EPI_KLQ of include/drvae_hip.h; ``dx`` is not written (pass None)"""
    d1 = _gemm_desc(dW, dpre, x, False, False, a_colsum=dbias, overread=overread, publish=publish, npad=npad)
    if klq is not None:
        assert yref is None and kscale is None and beta_x == 0.0
        Z, out = klq['Z'], klq['out']
        assert out.shape[1] == 2 * Z and W.shape[1] >= Z and W.shape[1] <= 2 * Z
        d2 = _gemm_desc(out[:, :W.shape[1]], dpre, W, True, False, alpha=alpha, epi=EPI_KLQ, split=Z, yref=klq['q'],
                        resid=klq['eps'], bias=klq['coef'], scale=klq['raw'], shift0=klq['kl_min'], overread=overread)
    elif yref is None:
        d2 = _gemm_desc(dx, dpre, W, True, False, a_kscale=kscale, alpha=alpha, beta=beta_x, overread=overread, npad=npad_x)
    else:
        d2 = _gemm_desc(dx, dpre, W, True, False, a_kscale=kscale, alpha=alpha, beta=beta_x, epi=EPI_BWD, yref=yref,
                        act0=act, act1=act, shift0=shift, shift1=shift, overread=overread, npad=npad_x)
    _lib.check(_lib.load().dv_gemm_pair(C.byref(d1), C.byref(d2), _stream()), 'dv_gemm_pair')


def linear_fwd(out, x, W, bias=None, *, x2=None, scale=None, split=None, act0=0, act1=0, shift0=0.0, shift1=0.0,
               resid=None, resid_cols=0, overread=False, publish=None, kpad=False):
    """out = act([x|x2] W^T * scale + bias) + shift (+ resid) -- one Linear (or two heads) forward."""
    gemm(out, x, W, True, True, A2=x2, epi=EPI_FWD, scale=scale, bias=bias, split=split, act0=act0, act1=act1,
         shift0=shift0, shift1=shift1, resid=resid, resid_cols=resid_cols, overread=overread, publish=publish, kpad=kpad)


def heads_tiles(split):
    """column tiles of ``linear_heads`` (= floats per row of the NLL partial-sum buffer)"""
    return _lib.load().dv_gemm_heads_tiles(split)


def linear_heads(out, x, W, bias=None, *, split, x2=None, scale=None, act0=0, act1=0, shift0=0.0, shift1=0.0,
                 resid=None, resid_cols=0, overread=False, publish=None, sample=None, nll=None, kpad=False):
    """Dual-head Linear with the row work on both heads fused into its epilogue (``dv_gemm_heads``).
    ``sample`` = dict(eps, out, n_src[, seg_ptr, seg_rows, sub, out2, out3, out3_idx, out4, out4_ptr]): ``out`` =
    (mu | logvar) is written and the reparameterised samples of every source row leave the same launch (``out4``:
    sample row s is also copied to rows [out4_ptr[s], out4_ptr[s+1]) of out4);
    ``nll`` = dict(x, coef, part[, xidx]): ``out`` receives d/d(mu | pre-activation of std) of the Gaussian
    log-likelihood rows, ``part`` (M, heads_tiles(split)) their per-tile partial sums."""
    assert (sample is None) != (nll is None)
    d = _gemm_desc(out, x, W, True, True, A2=x2, epi=EPI_FWD, scale=scale, bias=bias, split=split, act0=act0, act1=act1,
                   shift0=shift0, shift1=shift1, resid=resid, resid_cols=resid_cols, overread=overread, publish=publish,
                   kpad=kpad)
    e = _lib.HeadsEpi()
    if sample is not None:
        g = sample.get
        e.mode = _lib.HEADS_SAMPLE
        e.seg_ptr, e.seg_rows, e.n_src = _i32(g('seg_ptr')), _i32(g('seg_rows')), sample['n_src']
        e.eps, e.lde = _f32(sample['eps'], 'eps'), _ld(sample['eps'])
        e.out, e.ldo = _f32(sample['out'], 'out'), _ld(sample['out'])
        e.sub, e.lds = _f32(g('sub'), 'sub'), _ld(g('sub'))
        e.out2, e.ldo2 = _f32(g('out2'), 'out2'), _ld(g('out2'))
        e.out3, e.ldo3, e.out3_idx = _f32(g('out3'), 'out3'), _ld(g('out3')), _i32(g('out3_idx'))
        e.out4, e.ldo4, e.out4_ptr = _f32(g('out4'), 'out4'), _ld(g('out4')), _i32(g('out4_ptr'))
    else:
        e.mode = _lib.HEADS_NLL
        e.x, e.ldx, e.xidx = _f32(nll['x'], 'x'), _ld(nll['x']), _i32(nll.get('xidx'))
        e.coef, e.part = _f32(nll['coef'], 'coef'), _f32(nll['part'], 'part')
        assert nll['part'].is_contiguous() and tuple(nll['part'].shape) == (out.shape[0], heads_tiles(split))
    _lib.check(_lib.load().dv_gemm_heads(C.byref(d), C.byref(e), _stream()), 'dv_gemm_heads')


def linear_bwd_data(dx, dpre, W, *, kscale=None, alpha=1.0, beta=0.0, yref=None, act=0, shift=0.0, overread=False,
                    npad=False):
    """dx = beta*dx + alpha*((dpre*kscale) W) * act'(yref)   (W may be a column slice view)."""
    if yref is None:
        gemm(dx, dpre, W, True, False, a_kscale=kscale, alpha=alpha, beta=beta, overread=overread, npad=npad)
    else:
        gemm(dx, dpre, W, True, False, a_kscale=kscale, alpha=alpha, beta=beta, epi=EPI_BWD, yref=yref, act0=act,
             act1=act, shift0=shift, shift1=shift, overread=overread, npad=npad)


def linear_bwd_weight(dW, dpre, x, *, beta=0.0, dbias=None, overread=False, npad=False):
    """dW = beta*dW + dpre^T x ;  dbias = beta*dbias + colsum(dpre) fused in the same launch."""
    gemm(dW, dpre, x, False, False, beta=beta, a_colsum=dbias, colsum_beta=beta, overread=overread, npad=npad)


def bn_fwd(y, x, w, b, mean, rstd, running_mean, running_var, eps=1e-5, momentum=0.1, training=True):
    """nn.BatchNorm1d forward over the rows of x, see ``dv_bn_fwd`` (mean / rstd: saved for the backward pass)"""
    M, N = x.shape
    _lib.check(_lib.load().dv_bn_fwd(_f32(x), _ld(x), M, N, _f32(w), _f32(b), eps, _f32(mean), _f32(rstd), _f32(y), _ld(y),
                                     _f32(running_mean), _f32(running_var), momentum, int(bool(training)), _stream()),
               'dv_bn_fwd')


def bn_bwd(dx, dw, db, dy, x, mean, rstd, w, training=True):
    M, N = x.shape
    _lib.check(_lib.load().dv_bn_bwd(_f32(dy), _ld(dy), _f32(x), _ld(x), _f32(mean), _f32(rstd), _f32(w), M, N, _f32(dx),
                                     _ld(dx), _f32(dw), _f32(db), int(bool(training)), _stream()), 'dv_bn_bwd')


def mask_scale(y, x, mask, scale):
    M, N = x.shape
    _lib.check(_lib.load().dv_mask_scale(_f32(x), _ld(x), _f32(mask), _ld(mask), scale, M, N, _f32(y), _ld(y), _stream()),
               'dv_mask_scale')


def colsum(out, X, beta=0.0):
    M, N = X.shape
    _lib.check(_lib.load().dv_colsum(_f32(X), _ld(X), M, N, _f32(out), beta, _stream()), 'dv_colsum')


def act_bwd_(dY, Y, *, split=None, act0=0, act1=0, shift0=0.0, shift1=0.0):
    M, N = Y.shape
    _lib.check(_lib.load().dv_act_bwd(_f32(dY), _ld(dY), _f32(Y), _ld(Y), M, N, N if split is None else split,
                                      _act(act0), _act(act1), shift0, shift1, _stream()), 'dv_act_bwd')


def wn_scale(scale, norm, W, g):
    N, K = W.shape
    _lib.check(_lib.load().dv_wn_scale(_f32(W), _ld(W), _f32(g), N, K, _f32(scale), _f32(norm), _stream()),
               'dv_wn_scale')


def wn_bwd(dW, dg, dWraw, W, g, norm, beta=0.0):
    N, K = W.shape
    _lib.check(_lib.load().dv_wn_bwd(_f32(dWraw), _ld(dWraw), _f32(W), _ld(W), _f32(g), _f32(norm), N, K, _f32(dW),
                                     _ld(dW), _f32(dg), beta, _stream()), 'dv_wn_bwd')


# --------------------------------------------------------------------------- reparam
def reparam_fwd(out, mu, sd, eps, *, mode=GAUSS_LOGVAR, src_idx=None, reps=1, sub=None, out2=None, out3=None,
                out3_idx=None):
    """out[l*n+j] = mu[q] + eps[l*n+j]*std(sd[q]), q = src_idx[j] or j; out2 = out - sub;
    out3[out3_idx[r]] = out[r] where out3_idx[r] >= 0."""
    R, Z = out.shape
    n = R // reps
    assert n * reps == R and eps.shape[0] == R
    assert _ld(mu) == _ld(sd)
    _lib.check(_lib.load().dv_reparam_fwd(_f32(mu), _f32(sd), _ld(mu), _i32(src_idx), n, reps, Z, _f32(eps),
                                          _ld(eps), mode, _f32(out), _ld(out), _f32(sub), _ld(sub), _f32(out2),
                                          _ld(out2), _f32(out3), _ld(out3), _i32(out3_idx), _stream()),
               'dv_reparam_fwd')


def reparam_bwd(dmu, dsd, dz, eps, sd, *, mode=GAUSS_LOGVAR, src_idx=None, reps=1, beta=0.0):
    R, Z = dz.shape
    n = R // reps
    assert _ld(dmu) == _ld(dsd)
    _lib.check(_lib.load().dv_reparam_bwd(_f32(dz), _ld(dz), _f32(eps), _ld(eps), _f32(sd), _ld(sd), _i32(src_idx),
                                          n, reps, Z, mode, _f32(dmu), _f32(dsd), _ld(dmu), beta, _stream()),
               'dv_reparam_bwd')


def _wait(park):
    """(flag, ctr, err[, add[, max_spins]]) -> dv_wait (None: no wait)"""
    if park is None:
        return None
    flag, ctr, err = park[:3]
    w = _lib.Wait()
    w.flag, w.ctr, w.err = _i32(flag), _i32(ctr), _i32(err)
    w.add = park[3] if len(park) > 3 else 1
    w.max_spins = park[4] if len(park) > 4 and park[4] is not None else WAIT_SPINS
    return C.byref(w)


def _publish(pub):
    """(flag, ctr[, add]) -> dv_publish (None: nothing published)"""
    if pub is None:
        return None
    w = _lib.Publish()
    w.flag, w.ctr = _i32(pub[0]), _i32(pub[1])
    w.add = pub[2] if len(pub) > 2 else 1
    return C.byref(w)


def _bump(counters):
    """up to two (counter, inc) -> dv_bump (None / empty: nothing)"""
    cs = [c for c in (counters or ()) if c[0] is not None]
    if not cs:
        return None
    assert len(cs) <= 2
    b = _lib.Bump()
    for i, (c, inc) in enumerate(cs):
        b.c[i], b.n[i], b.inc[i] = _i32(c), c.numel(), inc
    return C.byref(b)


def _halt(halt):
    """the (err, ticks) pairs of a step's device-side waits -> (pointer, number of pairs)"""
    if halt is None:
        return None, 0
    assert halt.numel() % 2 == 0
    return _i32(halt), halt.numel() // 2


def reparam_bwd_seg(dmu, dsd, dz, eps, sd, seg_ptr, seg_rows, *, mode=GAUSS_LOGVAR, extra=None, ex_ptr=None,
                    ex_rows=None, beta=0.0, bump=None, dz_add=None, park=None, prior=None):
    """CSR backward of the reparameterisation (+ row-aligned extra (dmu|dsd) rows), see dv_reparam_bwd_seg.
    ``bump``: up to two (counter, inc) the launch advances as well; ``dz_add``: a second gradient source for the first
    ``dz_add.shape[0]`` sample rows; ``park`` = (flag, ctr, err[, add[, max_spins]]): the launch parks on another chain's flag;
    ``prior`` = (coef, raw, kl_min, mu): the gradient of coef * max(KL(q_i || N(0,I)), kl_min) is added to row i (dv_prior_kl)"""
    nq, Z = seg_ptr.numel() - 1, dz.shape[1]
    assert _ld(dmu) == _ld(dsd)
    add = None
    if dz_add is not None:
        add = _lib.SegAdd(_f32(dz_add), _ld(dz_add), dz_add.shape[0])
    pk = None
    if prior is not None:
        pk = C.byref(_lib.PriorKl(coef=_f32(prior[0]), raw=_f32(prior[1]), kl_min=prior[2], mu=_f32(prior[3]), ld=_ld(prior[3])))
    _lib.check(_lib.load().dv_reparam_bwd_seg(_f32(dz), _ld(dz), _f32(eps), _ld(eps), _f32(sd), _ld(sd),
                                              _i32(seg_ptr), _i32(seg_rows), nq, Z, mode, _f32(extra), _ld(extra),
                                              _i32(ex_ptr), _i32(ex_rows), _f32(dmu), _f32(dsd), _ld(dmu), beta,
                                              _bump(bump), C.byref(add) if add is not None else None, _wait(park), pk,
                                              _stream()), 'dv_reparam_bwd_seg')


def z2f_post_bwd(dp2, dz1, dq2, dz2f, dzdec_pert, pair_slot, eps, p2, q2, coef, raw, kl_min, dz1b, L, B, Np,
                 park=None, prior=None):
    """fused backward of the z2Fz1 sample / KL(q(z2|x2)||p(z2|z1)) / residual block, see dv_z2f_post_bwd.
    ``park`` = (flag, ctr, err[, add[, max_spins]]): the launch first parks on another chain's flag.
    ``prior`` = (coef, raw) of the Np q2 rows: their prior-KL gradient (same kl_min) is added to dq2 (dv_prior_kl)."""
    Z = dp2.shape[1] // 2
    d = _lib.Z2F(dz2f=_f32(dz2f), ld_dz2f=_ld(dz2f), dzdec_pert=_f32(dzdec_pert), ld_pert=_ld(dzdec_pert),
                 pair_slot=_i32(pair_slot), eps=_f32(eps), lde=_ld(eps), p2=_f32(p2), ldp2=_ld(p2), q2=_f32(q2),
                 ldq2=_ld(q2), coef=_f32(coef), raw=_f32(raw), kl_min=kl_min, dz1b=_f32(dz1b), ld_dz1b=_ld(dz1b),
                 dp2=_f32(dp2), ld_dp2=_ld(dp2), dz1=_f32(dz1), ld_dz1=_ld(dz1), dq2=_f32(dq2), ld_dq2=_ld(dq2),
                 L=L, B=B, Np=Np, Z=Z, prior_coef=_f32(prior[0]) if prior is not None else None,
                 prior_raw=_f32(prior[1]) if prior is not None else None)
    _lib.check(_lib.load().dv_z2f_post_bwd(C.byref(d), _wait(park), _stream()), 'dv_z2f_post_bwd')


# --------------------------------------------------------------------------- KL rows
def _kl_desc(out, raw, mu_q, sd_q, mu_p=None, sd_p=None, *, prior=(0.0, 0.0), mode=GAUSS_LOGVAR, qidx=None, pidx=None,
             reps=1, free_bits=False, kl_min=0.0, add=None):
    R = out.numel()
    assert _ld(mu_q) == _ld(sd_q) and (mu_p is None or _ld(mu_p) == _ld(sd_p))
    return _lib.KlRows(mu_q=_f32(mu_q), sd_q=_f32(sd_q), ldq=_ld(mu_q), qidx=_i32(qidx), mu_p=_f32(mu_p), sd_p=_f32(sd_p),
                       ldp=_ld(mu_p), pidx=_i32(pidx), prior_mu=prior[0], prior_sd=prior[1], n=R // reps, reps=reps,
                       Z=mu_q.shape[1], mode=mode, free_bits=int(free_bits), kl_min=kl_min, raw_out=_f32(raw), out=_f32(out),
                       add=_f32(add))


def kl_rows_fwd_pair(first, second):
    """two independent sets of plain KL rows in one launch; ``first`` / ``second`` = (args, kwargs) of ``kl_rows_fwd``
    (no fused sample, no second term, no park)"""
    d1, d2 = _kl_desc(*first[0], **first[1]), _kl_desc(*second[0], **second[1])
    _lib.check(_lib.load().dv_kl_rows_fwd_pair(C.byref(d1), C.byref(d2), _stream()), 'dv_kl_rows_fwd_pair')


def kl_rows_fwd(out, raw, mu_q, sd_q, mu_p=None, sd_p=None, *, prior=(0.0, 0.0), mode=GAUSS_LOGVAR, qidx=None,
                pidx=None, reps=1, free_bits=False, kl_min=0.0, add=None, eps=None, zout=None, park=None, second=None):
    """``second`` = (mu2, sd2, raw2): a second, row-aligned term KL(N(mu2, sd2) || N(prior)) with its own free bits is
    added to ``out`` and its raw value written to ``raw2`` (same launch)"""
    R = out.numel()
    n = R // reps
    Z = mu_q.shape[1]
    assert _ld(mu_q) == _ld(sd_q) and (mu_p is None or _ld(mu_p) == _ld(sd_p))
    assert second is None or (_ld(second[0]) == _ld(second[1]) and second[0].shape[0] == R)
    d = _lib.KlRows(_f32(mu_q), _f32(sd_q), _ld(mu_q), _i32(qidx), _f32(mu_p), _f32(sd_p), _ld(mu_p), _i32(pidx),
                    prior[0], prior[1], n, reps, Z, mode, int(free_bits), kl_min, _f32(raw), _f32(out), _f32(add),
                    _f32(eps), _ld(eps), _f32(zout), _ld(zout),
                    _f32(second[0]) if second else None, _f32(second[1]) if second else None,
                    _ld(second[0]) if second else 0, second[0].shape[1] if second else 0,
                    _f32(second[2]) if second else None)
    _lib.check(_lib.load().dv_kl_rows_fwd(C.byref(d), _wait(park), _stream()), 'dv_kl_rows_fwd')


def kl_rows_bwd(dq_mu, dq_sd, dp_mu, dp_sd, coef, raw, mu_q, sd_q, mu_p=None, sd_p=None, *, prior=(0.0, 0.0),
                mode=GAUSS_LOGVAR, qidx=None, pidx=None, reps=1, free_bits=False, kl_min=0.0, beta=0.0, dz=None,
                eps=None):
    R = coef.numel()
    n = R // reps
    Z = mu_q.shape[1]
    assert _ld(dq_mu) == _ld(dq_sd) and (dp_mu is None or _ld(dp_mu) == _ld(dp_sd))
    d = _lib.KlRows(mu_q=_f32(mu_q), sd_q=_f32(sd_q), ldq=_ld(mu_q), qidx=_i32(qidx), mu_p=_f32(mu_p), sd_p=_f32(sd_p),
                    ldp=_ld(mu_p), pidx=_i32(pidx), prior_mu=prior[0], prior_sd=prior[1], n=n, reps=reps, Z=Z, mode=mode,
                    free_bits=int(free_bits), kl_min=kl_min, raw_out=_f32(raw), eps=_f32(eps), lde=_ld(eps))
    g = _lib.KlRowsGrad(coef=_f32(coef), dq_mu=_f32(dq_mu), dq_sd=_f32(dq_sd), lddq=_ld(dq_mu), dp_mu=_f32(dp_mu),
                        dp_sd=_f32(dp_sd), lddp=_ld(dp_mu), beta=beta, dz=_f32(dz), ldz=_ld(dz))
    _lib.check(_lib.load().dv_kl_rows_bwd(C.byref(d), C.byref(g), _stream()), 'dv_kl_rows_bwd')


# ------------------------------------------------------------------------- NLL rows
def nll_rows_fwd(out, x, mu, sd, *, mode=GAUSS_SIGMA, xidx=None, bias=None, sd_shift=1e-3):
    """``bias`` = (bias_mu, bias_sd): ``mu`` / ``sd`` are the heads' raw products, finished by the pass (mu + bias_mu,
    softplus(sd + bias_sd) + sd_shift): evaluation passes behind a plain heads product"""
    M, X = mu.shape
    assert _ld(mu) == _ld(sd)
    _lib.check(_lib.load().dv_gauss_nll_rows_fwd(_f32(x), _ld(x), _i32(xidx), _f32(mu), _f32(sd), _ld(mu), M, X,
                                                 mode, _f32(out), _f32(bias[0]) if bias is not None else None,
                                                 _f32(bias[1]) if bias is not None else None, sd_shift, _stream()),
               'dv_gauss_nll_rows_fwd')


def nll_rows_fwdbwd(out, dmu, dsd, coef, x, mu, sd, *, mode=GAUSS_SIGMA, xidx=None, sd_act=0, sd_shift=0.0, bias=None):
    """row log-likelihoods AND coef-weighted gradients w.r.t. (mu, pre-activation of sd) in one pass; ``bias`` =
    (bias_mu, bias_sd): ``mu`` / ``sd`` are the heads' raw products, finished here (+ bias, activation of sd, shift)"""
    M, X = mu.shape
    assert _ld(mu) == _ld(sd) and _ld(dmu) == _ld(dsd)
    _lib.check(_lib.load().dv_gauss_nll_rows_fwdbwd(_f32(coef), _f32(x), _ld(x), _i32(xidx), _f32(mu), _f32(sd),
                                                    _ld(mu), M, X, mode, _act(sd_act), sd_shift, _f32(out),
                                                    _f32(dmu), _f32(dsd), _ld(dmu),
                                                    _f32(bias[0]) if bias is not None else None,
                                                    _f32(bias[1]) if bias is not None else None, _stream()),
               'dv_gauss_nll_rows_fwdbwd')


def nll_raw_cs_shape(M, X):
    """(chunks, row_blocks) of ``nll_rows_raw_cs`` for M rows of X genes"""
    lib = _lib.load()
    return lib.dv_nll_raw_cs_chunks(X), lib.dv_nll_raw_cs_row_blocks(M)


def nll_rows_raw_cs(out_part, dmu, dsd, ws, coef, x, mu, sd, bias, *, xidx=None, sd_shift=1e-3):
    """the raw-heads NLL forward + backward pass with the heads' bias gradient folded in (``dv_gauss_nll_rows_raw_cs``):
    ``out_part`` (M, chunks) partial row log-likelihoods, ``ws`` (row_blocks, ld) per-block column sums of (dmu | dsd) --
    dsd's at the column offset dsd has behind dmu; the caller sums the blocks (``colsum``).  ``dmu = dsd = ws = coef =
    None``: forward only (evaluation)."""
    M, X = mu.shape
    chunks, rbs = nll_raw_cs_shape(M, X)
    fwd_only = dmu is None
    assert _ld(mu) == _ld(sd) and tuple(out_part.shape) == (M, chunks) and out_part.is_contiguous()
    off = 0
    if not fwd_only:
        assert _ld(dmu) == _ld(dsd) and dmu.untyped_storage().data_ptr() == dsd.untyped_storage().data_ptr()
        off = dsd.storage_offset() - dmu.storage_offset()
        assert ws.shape[0] == rbs and ws.dim() == 2
    else:
        assert dsd is None and ws is None
    d = _lib.NllRawCs(coef=_f32(coef), x=_f32(x), ldx=_ld(x), xidx=_i32(xidx), mu=_f32(mu), sd=_f32(sd), ldp=_ld(mu), M=M,
                      X=X, shift=sd_shift, out_part=_f32(out_part), chunks=chunks, dmu=_f32(dmu), dsd=_f32(dsd),
                      ldd=_ld(dmu), bias_mu=_f32(bias[0]), bias_sd=_f32(bias[1]), ws=_f32(ws),
                      ldw=ws.stride(0) if ws is not None else 0, sd_off=off, row_blocks=rbs)
    _lib.check(_lib.load().dv_gauss_nll_rows_raw_cs(C.byref(d), _stream()), 'dv_gauss_nll_rows_raw_cs')


def rec_nll_rows(out, x, v, *, kind, shift=0.0, xidx=None, coef=None, dpre=None):
    """log-likelihood rows of the Bernoulli ('binary') / Poisson decoders from the head's post-activation output
    ``v``; with ``coef`` also the gradient w.r.t. the head's pre-activation (``dv_rec_nll_rows``)"""
    M, X = v.shape
    _lib.check(_lib.load().dv_rec_nll_rows(_lib.REC_KIND[kind], shift, _f32(coef), _f32(x), _ld(x), _i32(xidx), _f32(v),
                                           _ld(v), M, X, _f32(out), _f32(dpre), _ld(dpre), _stream()), 'dv_rec_nll_rows')


def nll_rows_bwd(dmu, dsd, coef, x, mu, sd, *, mode=GAUSS_SIGMA, xidx=None, sd_act=0, sd_shift=0.0, dx=None,
                 beta=0.0):
    M, X = mu.shape
    assert _ld(mu) == _ld(sd) and _ld(dmu) == _ld(dsd)
    _lib.check(_lib.load().dv_gauss_nll_rows_bwd(_f32(coef), _f32(x), _ld(x), _i32(xidx), _f32(mu), _f32(sd),
                                                 _ld(mu), M, X, mode, _act(sd_act), sd_shift, _f32(dmu), _f32(dsd),
                                                 _ld(dmu), _f32(dx), _ld(dx), beta, _stream()),
               'dv_gauss_nll_rows_bwd')


# ----------------------------------------------------------------------- categorical
def softmax_clamp_fwd(probs, logits, sigmoid1=False):
    M, Y = probs.shape
    _lib.check(_lib.load().dv_softmax_clamp_fwd(_f32(logits), _ld(logits), M, Y, int(sigmoid1), _f32(probs),
                                                _ld(probs), _stream()), 'dv_softmax_clamp_fwd')


def softmax_clamp_bwd(dlogits, dprobs, probs, sigmoid1=False, beta=0.0):
    M, Y = probs.shape
    _lib.check(_lib.load().dv_softmax_clamp_bwd(_f32(dprobs), _ld(dprobs), _f32(probs), _ld(probs), M, Y,
                                                int(sigmoid1), _f32(dlogits), _ld(dlogits), beta, _stream()),
               'dv_softmax_clamp_bwd')


def cat_terms_fwd(probs, *, labels=None, prior=None, logp=None, kl=None, ent=None, best=None):
    M, Y = probs.shape
    _lib.check(_lib.load().dv_cat_terms_fwd(_f32(probs), _ld(probs), M, Y, _i32(labels), _f32(prior), _ld(prior),
                                            _f32(logp), _f32(kl), _ld(kl), _f32(ent), _i32(best), _stream()),
               'dv_cat_terms_fwd')


def cat_terms_bwd(dprobs, probs, *, labels=None, prior=None, c_logp=None, g_kl=None, c_ent=None, beta=0.0):
    M, Y = probs.shape
    _lib.check(_lib.load().dv_cat_terms_bwd(_f32(probs), _ld(probs), M, Y, _i32(labels), _f32(prior), _ld(prior),
                                            _f32(c_logp), _f32(g_kl), _ld(g_kl), _f32(c_ent), _f32(dprobs),
                                            _ld(dprobs), beta, _stream()), 'dv_cat_terms_bwd')


def smalln_fwd(probs, logits, a1, W, bias=None, a2=None, ymarg=None, park=None, fprop_kl=None):
    """probs = clamp(softmax([a1|a2] W^T + b)) for N <= 8 outputs (either output may be None).
    ``ymarg`` = (yl, kld, cfp, dqy, label, fp_ptr, klfp, log_prior, c_kld, c_yl): the y-marginalisation of every row
    (the arguments of ``ymarg_fwdbwd``) rides on the same launch.  ``park`` = (flag, ctr, err[, add[, max_spins]]): every
    workgroup first parks on another chain's flag (see ``flag_wait``).  ``fprop_kl`` = dict(Q, qidx, P, Q3, Z1, Z3,
    kl_min, raw1, raw3, dq, dp) with ``ymarg``: the KL rows of the fprop rows (forward in front of the
    y-marginalisation, the z1 term's backward behind it) ride on the same launch (``dv_fprop_kl``)."""
    M = a1.shape[0]
    N = W.shape[0]
    K1, K2 = a1.shape[1], (a2.shape[1] if a2 is not None else 0)
    ym = None
    if ymarg is not None:
        yl, kld, cfp, dqy, label, fp_ptr, klfp, log_prior, c_kld, c_yl = ymarg
        vec = log_prior if torch.is_tensor(log_prior) else None
        y = _lib.Ymarg()
        y.label, y.fp_ptr, y.klfp = _i32(label), _i32(fp_ptr), _f32(klfp)
        y.log_prior, y.log_prior_v = (0.0 if vec is not None else log_prior), _f32(vec)
        y.c_kld, y.c_yl, y.yl, y.kld, y.cfp = _f32(c_kld), _f32(c_yl), _f32(yl), _f32(kld), _f32(cfp)
        y.dqy, y.lddq = _f32(dqy), _ld(dqy)
        ym = C.byref(y)
    kf = None
    if fprop_kl is not None:
        assert ymarg is not None
        f = _lib.FpropKl()
        Q, P, Q3 = fprop_kl['Q'], fprop_kl['P'], fprop_kl['Q3']
        f.Z1, f.Z3, f.kl_min = fprop_kl['Z1'], fprop_kl['Z3'], fprop_kl['kl_min']
        assert Q.shape[1] == 2 * f.Z1 and P.shape[1] == 2 * f.Z1 and Q3.shape[1] == 2 * f.Z3
        f.mu_q, f.ldq, f.qidx = _f32(Q), _ld(Q), _i32(fprop_kl['qidx'])
        f.mu_p, f.ldp, f.mu3, f.ld3 = _f32(P), _ld(P), _f32(Q3), _ld(Q3)
        f.klfp, f.raw1, f.raw3 = _f32(ymarg[6]), _f32(fprop_kl['raw1']), _f32(fprop_kl['raw3'])
        f.dq, f.lddq, f.dp, f.lddp = _f32(fprop_kl['dq']), _ld(fprop_kl['dq']), _f32(fprop_kl['dp']), _ld(fprop_kl['dp'])
        kf = C.byref(f)
    _lib.check(_lib.load().dv_smalln_linear_fwd(_f32(a1), _ld(a1), K1, _f32(a2), _ld(a2), K2, _f32(W), _ld(W),
                                                _f32(bias), M, N, _f32(logits), _ld(logits), _f32(probs),
                                                _ld(probs), ym, _wait(park), kf, _stream()), 'dv_smalln_linear_fwd')


def smalln_bwd_data(dsts, dprobs, probs, W, seg=None):
    """dsts: list of (dst, col0, alpha, beta[, col1, alpha2]);
    dst = beta*dst + dlogit @ (alpha*W[:, col0:col0+w] + alpha2*W[:, col1:col1+w]), w = dst.shape[1].
    ``seg`` = (src, seg_ptr): the FIRST destination starts from the segment sum of ``src`` rows
    [seg_ptr[r], seg_ptr[r+1]) (its first w columns) instead of beta*dst."""
    M, N = dprobs.shape
    n = len(dsts)
    P = (C.c_void_p * n)(*[_f32(d[0]) for d in dsts])
    LD = (C.c_int64 * n)(*[_ld(d[0]) for d in dsts])
    C0 = (C.c_int32 * n)(*[d[1] for d in dsts])
    NC = (C.c_int32 * n)(*[d[0].shape[1] for d in dsts])
    AL = (C.c_float * n)(*[d[2] for d in dsts])
    BE = (C.c_float * n)(*[d[3] for d in dsts])
    C1 = (C.c_int32 * n)(*[(d[4] if len(d) > 4 else 0) for d in dsts])
    A2 = (C.c_float * n)(*[(d[5] if len(d) > 5 else 0.0) for d in dsts])
    _lib.check(_lib.load().dv_smalln_linear_bwd_data(_f32(dprobs), _ld(dprobs), _f32(probs), _ld(probs), _f32(W),
                                                     _ld(W), M, N, n, P, LD, C0, NC, AL, BE, C1, A2,
                                                     _f32(seg[0]) if seg else None, _ld(seg[0]) if seg else 0,
                                                     _i32(seg[1]) if seg else None, _stream()),
               'dv_smalln_linear_bwd_data')


SMALLN_WS_SPLITS = 16


def smalln_ws_numel(N, K):
    """floats of the optional row-split workspace of ``smalln_bwd_weight`` for N classes and K = K1 + K2 inputs"""
    return SMALLN_WS_SPLITS * N * (K + 1)


def smalln_bwd_weight(dW, db, dprobs, probs, a1, a2=None, beta=0.0, publish=None, ws=None):
    """``publish`` = (flag, ctr[, add]): the launch publishes on entry that everything in front of it is complete;
    ``ws`` (``smalln_ws_numel`` floats): lets many-row products (>= 1024 rows) split the rows over workgroups"""
    M, N = dprobs.shape
    K1, K2 = a1.shape[1], (a2.shape[1] if a2 is not None else 0)
    assert ws is None or (ws.is_contiguous() and ws.numel() >= smalln_ws_numel(N, K1 + K2))
    _lib.check(_lib.load().dv_smalln_linear_bwd_weight(_f32(dprobs), _ld(dprobs), _f32(probs), _ld(probs), _f32(a1),
                                                       _ld(a1), K1, _f32(a2), _ld(a2), K2, M, N, _f32(dW), _ld(dW),
                                                       _f32(db), beta, _publish(publish), _f32(ws),
                                                       SMALLN_WS_SPLITS if ws is not None else 0, _stream()),
               'dv_smalln_linear_bwd_weight')


def ymarg_fwd(yl, kld, qy, label, fp_ptr, klfp, log_prior):
    """``log_prior``: python float (uniform prior) or a device vector of Y log-probabilities"""
    R, Y = qy.shape
    vec = log_prior if torch.is_tensor(log_prior) else None
    _lib.check(_lib.load().dv_ymarg_fwd(_f32(qy), _ld(qy), _i32(label), _i32(fp_ptr), _f32(klfp),
                                        0.0 if vec is not None else log_prior, _f32(vec), R, Y, _f32(yl), _f32(kld),
                                        _stream()), 'dv_ymarg_fwd')


def ymarg_fwdbwd(yl, kld, cfp, dqy, qy, label, fp_ptr, klfp, log_prior, c_kld, c_yl):
    R, Y = qy.shape
    vec = log_prior if torch.is_tensor(log_prior) else None
    _lib.check(_lib.load().dv_ymarg_fwdbwd(_f32(qy), _ld(qy), _i32(label), _i32(fp_ptr), _f32(klfp),
                                           0.0 if vec is not None else log_prior, _f32(vec), _f32(c_kld), _f32(c_yl),
                                           R, Y, _f32(yl), _f32(kld), _f32(cfp), _f32(dqy), _ld(dqy), _stream()),
               'dv_ymarg_fwdbwd')


def ymarg_bwd(cfp, dqy, qy, label, fp_ptr, klfp, log_prior, c_kld, c_yl):
    R, Y = qy.shape
    vec = log_prior if torch.is_tensor(log_prior) else None
    _lib.check(_lib.load().dv_ymarg_bwd(_f32(qy), _ld(qy), _i32(label), _i32(fp_ptr), _f32(klfp),
                                        0.0 if vec is not None else log_prior, _f32(vec), _f32(c_kld), _f32(c_yl), R, Y,
                                        _f32(cfp), _f32(dqy), _ld(dqy), _stream()), 'dv_ymarg_bwd')


def ycont_fwd(yl, fpin_y, z3in_y, mu, ylab, has_y, eps, logvar, B, sqerr=False):
    """regression head forward, see ``dv_ycont_fwd``"""
    R, Y = mu.shape
    _lib.check(_lib.load().dv_ycont_fwd(_f32(mu), _ld(mu), _f32(ylab), _i32(has_y), _f32(eps), _ld(eps), logvar, int(sqerr), R, B,
                                        Y, _f32(yl), _f32(fpin_y), _ld(fpin_y), _f32(z3in_y), _ld(z3in_y), _stream()),
               'dv_ycont_fwd')


def ycont_bwd(dlogit, cfp, mu, ylab, has_y, logvar, c_yl, c_kld, dfpin_y, dz3in_y, B, sqerr=False):
    """``dlogit`` None: only cfp[r] = c_kld[r]; else the gradient w.r.t. the head's pre-sigmoid output"""
    R, Y = mu.shape
    _lib.check(_lib.load().dv_ycont_bwd(_f32(mu), _ld(mu), _f32(ylab), _i32(has_y), logvar, int(sqerr), _f32(c_yl),
                                        _f32(c_kld),
                                        _f32(dfpin_y), _ld(dfpin_y), _f32(dz3in_y), _ld(dz3in_y), R, B, Y,
                                        _f32(dlogit), _ld(dlogit), _f32(cfp), _stream()), 'dv_ycont_bwd')


def mmd_rff_fwd(diff, mmd2, th1, th2, c):
    """finish the random-Fourier-feature MMD^2 from the projections theta, see ``dv_mmd_rff_fwd``"""
    R = th1.shape[1]
    _lib.check(_lib.load().dv_mmd_rff_fwd(_f32(th1), _ld(th1), th1.shape[0], _f32(th2), _ld(th2), th2.shape[0], R, c,
                                          _f32(diff), _f32(mmd2), _stream()), 'dv_mmd_rff_fwd')


def mmd_rff_bwd(G, th, diff, gout, coef):
    n, R = th.shape
    _lib.check(_lib.load().dv_mmd_rff_bwd(_f32(th), _ld(th), n, R, _f32(diff), _f32(gout), coef, _f32(G), _ld(G),
                                          _stream()), 'dv_mmd_rff_bwd')


MMD_KIND = {'poly': 0, 'rbf': 1}


def _gammas(gammas):
    arr = (C.c_float * len(gammas))(*[float(g) for g in gammas])
    return arr, len(gammas)


def mmd_mix_fwd(part, G, kind, gammas, sa=None, sb=None):
    """part[i] = sum_j of the kernel mixture on the Gram matrix ``G`` (``dv_mmd_mix_fwd``); ``sa`` / ``sb``: the self
    products whose diagonals are the squared row norms (rbf)"""
    arr, nb = _gammas(gammas)
    M, N = G.shape
    _lib.check(_lib.load().dv_mmd_mix_fwd(_f32(G), _ld(G), M, N, MMD_KIND[kind], arr, nb, _f32(sa), (_ld(sa) + 1) if sa is not None else 0,
                                          _f32(sb), (_ld(sb) + 1) if sb is not None else 0, _f32(part), _stream()), 'dv_mmd_mix_fwd')


def mmd_mix_bwd(W, rs, G, kind, gammas, gout, coef, sa=None, sb=None):
    arr, nb = _gammas(gammas)
    M, N = G.shape
    _lib.check(_lib.load().dv_mmd_mix_bwd(_f32(G), _ld(G), M, N, MMD_KIND[kind], arr, nb, _f32(sa), (_ld(sa) + 1) if sa is not None else 0,
                                          _f32(sb), (_ld(sb) + 1) if sb is not None else 0, _f32(gout), coef, _f32(W), _ld(W),
                                          _f32(rs), _stream()), 'dv_mmd_mix_bwd')


def mmd_mix_combine(out4, p11, p12, p22, c11, c12, c22):
    _lib.check(_lib.load().dv_mmd_mix_combine(_f32(p11), p11.numel(), c11, _f32(p12), p12.numel(), c12, _f32(p22), p22.numel(),
                                              c22, _f32(out4), _stream()), 'dv_mmd_mix_combine')


def mmd_identity_fwd(diff, out, x1, x2):
    _lib.check(_lib.load().dv_mmd_identity_fwd(_f32(x1), _ld(x1), x1.shape[0], _f32(x2), _ld(x2), x2.shape[0], x1.shape[1],
                                               _f32(diff), _f32(out), _stream()), 'dv_mmd_identity_fwd')


def mmd_identity_bwd(dx, diff, gout, coef):
    _lib.check(_lib.load().dv_mmd_identity_bwd(_f32(diff), _f32(gout), coef, dx.shape[0], dx.shape[1], _f32(dx), _ld(dx),
                                               _stream()), 'dv_mmd_identity_bwd')


def rows_gather(out, src, idx=None, *, noise=None, sigma=0.0, onehot_cls=None, n_classes=0, width=None, park=None):
    n = out.shape[0]
    W = (src.shape[1] if src is not None else 0) if width is None else width
    _lib.check(_lib.load().dv_rows_gather(_f32(src), _ld(src), _i32(idx), n, W, _f32(noise), _ld(noise), sigma,
                                          _i32(onehot_cls), n_classes, _f32(out), _ld(out), _wait(park), _stream()),
               'dv_rows_gather')


def _masks_desc(m, B):
    """keyword arguments of ``batch_masks`` -> dv_batch_masks_desc"""
    d = dict(hx=None, hy=None, y=None, c_klz2=None, c_yl=None, w_pert=None, w_yl=None, label=None, c_klp=None, Np=None,
             one_slot=None, gcounts=None)
    d.update(m)
    assert d['gcounts'] is None or (d['gcounts'].dtype == torch.int32 and d['gcounts'].is_contiguous())
    return _lib.BatchMasks(_i32(d['hx']), _i32(d['hy']), _i32(d['y']), B if d['Np'] is None else d['Np'],
                           d['n_tot'], d['kl_rate'], d['pert_rate'], d['yl_rate'], _f32(d['beta']), _f32(d['c_nll']),
                           _f32(d['c_klz2']), _f32(d['c_yl']), _f32(d['w_recl']), _f32(d['w_pert']), _f32(d['w_yl']),
                           _i32(d['label']), _f32(d['c_klp']), _i32(d['one_slot']), _i32(d['gcounts']))


def batch_feed(xin, x1, x2, y32, table, n_batches, ctr, base, *, pair_rows=None, noise=None, sigma=0.0, has_y=None,
               L=1, label_r=None, fp_i=None, fp_lab=None, fp_slot=None, fp_cls=None, onehot=None, n_classes=0, yf=None,
               ylab=None, onehot2=None, masks=None, park=None):
    """graph-resident minibatch feed: see dv_batch_feed in include/drvae_hip.h; ``masks``: keyword arguments of
    ``batch_masks`` (minus table / ctr / base / B / L): the batch's masks written by the same launch"""
    B = table.shape[1]
    md = _masks_desc(masks, B) if masks is not None else None
    assert masks is None or masks.get('gcounts') is None or masks['gcounts'].shape[0] == n_batches
    Np = pair_rows.numel() if pair_rows is not None else 0
    assert xin.shape[0] == B + Np and table.shape[0] == n_batches and table.is_contiguous()
    Mf = fp_cls.numel() if fp_cls is not None else 0
    d = _lib.BatchFeed(x1=_f32(x1), ld1=_ld(x1), x2=_f32(x2) if Np else None, ld2=_ld(x2) if Np else 0, y=_i32(y32),
                       table=_i32(table), n_batches=n_batches, ctr=_i32(ctr), base=_i32(base), B=B,
                       pair_rows=_i32(pair_rows), Np=Np, X=xin.shape[1], noise=_f32(noise), ldn=_ld(noise), sigma=sigma,
                       xin=_f32(xin), ldo=_ld(xin), has_y=_i32(has_y), L=L, label_r=_i32(label_r), fp_i=_i32(fp_i),
                       fp_lab=_i32(fp_lab), fp_slot=_i32(fp_slot), Mf=Mf, fp_cls=_i32(fp_cls), onehot=_f32(onehot),
                       ldh=_ld(onehot), Y=n_classes, yf=_f32(yf), ylab=_f32(ylab),
                       Yc=ylab.shape[1] if ylab is not None else 0, onehot2=_f32(onehot2), ldh2=_ld(onehot2))
    _lib.check(_lib.load().dv_batch_feed(C.byref(d), C.byref(md) if md is not None else None, _wait(park), _stream()),
               'dv_batch_feed')


def batch_masks(B, L, *, table=None, n_batches=0, ctr=None, base=None, **masks):
    """per-batch coefficient / weight vectors of a batch-independent step plan from the batch's pair / label flags
    (``dv_batch_masks``); ``hx`` / ``hy`` / ``y``: int32 device arrays indexed by dataset row (``table`` given) or by
    batch row; ``beta``: 1-element device float; ``Np``: rows [0, Np) have pair slots (default: all B); ``gcounts``
    (int32, (n_batches | 1, 2)): the GLOBAL (N_pairs, N_labeled) per batch under data parallelism"""
    md = _masks_desc(masks, B)
    _lib.check(_lib.load().dv_batch_masks(C.byref(md), _i32(table), n_batches, _i32(ctr), _i32(base), B, L, _stream()),
               'dv_batch_masks')


def rows_segment_sum(dst, src, *, seg_ptr=None, seg_rows=None, w=None, n=None, dst_idx=None, beta=0.0, width=None,
                     park=None):
    if n is None:
        n = seg_ptr.numel() - 1 if seg_ptr is not None else (seg_rows.numel() if seg_rows is not None else
                                                              src.shape[0])
    W = dst.shape[1] if width is None else width
    _lib.check(_lib.load().dv_rows_segment_sum(_f32(src), _ld(src), _i32(seg_ptr), _i32(seg_rows), _f32(w), n, W,
                                               _i32(dst_idx), _f32(dst), _ld(dst), beta, _wait(park), _stream()),
               'dv_rows_segment_sum')


def weighted_sum(out, x, w=None, idx=None, scale=1.0, beta=0.0, n=None):
    if n is None:
        n = idx.numel() if idx is not None else x.numel()
    _lib.check(_lib.load().dv_weighted_sum(_f32(x), _f32(w), _i32(idx), n, scale, _f32(out), beta, _stream()),
               'dv_weighted_sum')


def recon_row_stats(out, x, r):
    """out (M,6): per-row {SSE, mean x, mean r, centred sums xx, rr, xr} (evaluation metrics)."""
    M, X = x.shape
    assert out.is_contiguous() and tuple(out.shape) == (M, 6)
    _lib.check(_lib.load().dv_recon_row_stats(_f32(x), _ld(x), _f32(r), _ld(r), M, X, _f32(out), _stream()),
               'dv_recon_row_stats')


def col_moments(out, x, r, sel=None, part=None, r_bias=None):
    """out (3,X) float64: per-column sum x, sum x^2, sum (x-r)^2 over the rows (``sel``: int32 list of the rows that
    count).  ``part`` (row_blocks, 3, X) float64: leave the per-block partials there instead (``recon_finalize`` adds
    them up); out may then be None.  ``r_bias`` (X): r is a raw heads product, r + r_bias the reconstruction."""
    M = sel.numel() if sel is not None else x.shape[0]
    X = x.shape[1]
    if part is not None:
        nb = part.shape[0]
        assert part.dtype == torch.float64 and part.is_cuda and part.is_contiguous() and tuple(part.shape[1:]) == (3, X)
    else:
        assert out.dtype == torch.float64 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == (3, X)
        nb = col_moment_blocks(M)      # row blocks: partial sums per block, added up here in a fixed order
        part = out.unsqueeze(0) if nb == 1 else torch.empty(nb, 3, X, dtype=torch.float64, device=out.device)
    _lib.check(_lib.load().dv_col_moments(_f32(x), _ld(x), _f32(r), _ld(r), M, X, part.data_ptr(), nb, _i32(sel), _f32(r_bias), _stream()),
               'dv_col_moments')
    if out is not None and part.data_ptr() != out.data_ptr():
        torch.sum(part, 0, out=out)


def col_moment_blocks(M):
    """row blocks of ``col_moments`` (16-wave workgroups): 256 rows up to 16 blocks, then 512 rows, at most 64 --
    ``recon_finalize``, one workgroup, walks every block's partials"""
    return max(1, min(64, max(min(16, (M + 255) // 256), (M + 511) // 512)))


RECON_ROWS_MAX_X = 1024


def recon_rows(rows, ll, x, mu, sd, *, bias=None, sd_shift=1e-3):
    """row statistics (M, 6) and log-likelihood rows (M; may be None) of a reconstruction in ONE pass over (x, mu, sd),
    X <= RECON_ROWS_MAX_X -- ``recon_row_stats`` + ``nll_rows_fwd``.  ``bias`` = (bias_mu, bias_sd): mu / sd are the
    heads' raw products, finished on the way."""
    M, X = x.shape
    assert tuple(mu.shape) == (M, X) and tuple(sd.shape) == (M, X) and _ld(mu) == _ld(sd)
    assert rows.is_contiguous() and tuple(rows.shape) == (M, 6)
    d = _lib.ReconRows(x=_f32(x), ldx=_ld(x), mu=_f32(mu), sd=_f32(sd), ldp=_ld(mu),
                       bias_mu=_f32(bias[0]) if bias is not None else None,
                       bias_sd=_f32(bias[1]) if bias is not None else None, sd_shift=sd_shift, M=M, X=X,
                       rows=_f32(rows), ll=_f32(ll))
    _lib.check(_lib.load().dv_recon_rows(C.byref(d), _stream()), 'dv_recon_rows')


def recon_finalize(out4, rows, part, X, *, sel=None, n=None, ll=None):
    """out4 (4 float64) = rmse, variance-weighted R^2, mean per-row Pearson r, mean log-likelihood from the partials of
    ``recon_row_stats`` / ``col_moments(part=...)`` (+ the per-row log-likelihoods): ``dv_recon_finalize``"""
    n = (sel.numel() if sel is not None else rows.shape[0]) if n is None else n
    assert out4.dtype == torch.float64 and out4.is_cuda and out4.numel() >= 4 and out4.is_contiguous()
    assert part.dtype == torch.float64 and part.is_contiguous() and rows.is_contiguous() and rows.shape[1] == 6
    _lib.check(_lib.load().dv_recon_finalize(_f32(rows), _i32(sel), n, X, part.data_ptr(), part.shape[0], _f32(ll),
                                             out4.data_ptr(), _stream()), 'dv_recon_finalize')


RANK_MAX_ROWS = 32768       # DV_RANK_MAX_ROWS


def rank_metrics(out, counts, proba, y32, *, pred32=None, sel=None, c0=1, n_cls=1, binary=True):
    """ROC-AUC / average precision per class and the accuracy (``dv_rank_metrics``): out (2 n_cls + 1 float64), counts
    (n_cls, n, 4) int32 zeros (left zeroed)"""
    n = sel.numel() if sel is not None else proba.shape[0]
    assert out.dtype == torch.float64 and out.is_cuda and out.numel() >= 2 * n_cls + 1 and out.is_contiguous()
    assert counts.dtype == torch.int32 and counts.is_contiguous() and counts.numel() >= n_cls * n * 4
    _lib.check(_lib.load().dv_rank_metrics(_f32(proba), _ld(proba), _i32(y32), _i32(pred32), _i32(sel), n, c0, n_cls,
                                           int(bool(binary)), counts.data_ptr(), out.data_ptr(), _stream()), 'dv_rank_metrics')


def loss_assemble(loss, terms, w_elbo, w_cmpl, after=None, bump=(), halt=None, accum=None):
    """terms: list of (x, w_or_None, scale, out_index); see ``dv_loss_assemble``.  ``after`` =
    (flag, counter, err, add, max_spins): park like ``flag_wait`` inside the same launch first;
    ``bump`` = up to two (counter, inc): advanced at the end of the launch;
    ``halt``: the (err, ticks) pairs of the step's waits -- any error set: the scalars come out NaN;
    ``accum`` (8 floats): running sums, ``accum += loss`` in the same launch."""
    hp, hn = _halt(halt)
    arr = (_lib.LossTerm * max(len(terms), 1))()
    for i, term in enumerate(terms):      # (x, w, scale, out[, row_len]): row_len > 1 = one weight per row of x
        x, w, scale, out = term[:4]
        arr[i].x, arr[i].w, arr[i].n, arr[i].scale, arr[i].out = _f32(x), _f32(w), x.numel(), scale, out
        arr[i].row_len = term[4] if len(term) > 4 else 1
    if after is None and bump:
        after = (None, None, None, 0, 1)      # no wait: only the counters ride on the launch
    if after is None:
        _lib.check(_lib.load().dv_loss_assemble(arr, len(terms), _f32(w_elbo), _f32(w_cmpl), _f32(loss), hp, hn,
                                                _f32(accum), _stream()), 'dv_loss_assemble')
    else:
        flag, ctr, err, add, spins = after
        _lib.check(_lib.load().dv_loss_assemble_after(
            _wait((flag, ctr, err, add, spins)) if flag is not None else None, arr, len(terms), _f32(w_elbo),
            _f32(w_cmpl), _f32(loss), _bump(bump), hp, hn, _f32(accum), _stream()), 'dv_loss_assemble_after')


def axpby(y, x, a=1.0, b=0.0):
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    _lib.check(_lib.load().dv_axpby(_f32(x), a, _f32(y), b, x.numel(), _stream()), 'dv_axpby')


# ------------------------------------------------------------------------- optimiser
def adam_l2(p, g, m, v, step_dev, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, gscale=1.0, gate=None,
            halt=None):
    """``gate`` = (flag, counter, add, err, lo, hi): elements [lo, hi) wait for the flag (``dv_adam_l2_gated``);
    ``halt``: the (err, ticks) pairs of the step's waits -- any error set: nothing is updated"""
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    hp, hn = _halt(halt)
    h = _lib.AdamHyper(lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, gscale=gscale)
    if gate is None:
        _lib.check(_lib.load().dv_adam_l2(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), C.byref(h), _i32(step_dev), hp, hn,
                                          _stream()), 'dv_adam_l2')
    else:
        flag, ctr, add, err, lo, hi = gate
        _lib.check(_lib.load().dv_adam_l2_gated(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), C.byref(h), _i32(step_dev),
                                                _wait((flag, ctr, err, add)), lo, hi, hp, hn, _stream()), 'dv_adam_l2_gated')


def adamax_l2(p, g, m, u, step_dev, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, gscale=1.0, halt=None):
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and u.is_contiguous()
    hp, hn = _halt(halt)
    h = _lib.AdamHyper(lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, gscale=gscale)
    _lib.check(_lib.load().dv_adamax_l2(_f32(p), _f32(g), _f32(m), _f32(u), p.numel(), C.byref(h), _i32(step_dev), hp, hn,
                                        _stream()), 'dv_adamax_l2')


def flag_publish(flag, ctr, add=1):
    _lib.check(_lib.load().dv_flag_publish(_i32(flag), _i32(ctr), add, _stream()), 'dv_flag_publish')


# Bound of a device-side wait, in polls (~1 us each: an agent-scope load + s_sleep).  Inside one process
# the chains are at most a step (~0.3 ms) apart; under data parallelism a chain may also sit out the skew
# between the ranks (the next step's first publish follows the previous step's gradient exchange), hence
# seconds rather than milliseconds.  A wait that still times out is reported (see ``check_sync``), never hung.
WAIT_SPINS = int(os.environ.get('DRVAE_WAIT_SPINS', '4000000'))


def flag_wait(flag, ctr, err, add=1, max_spins=None, publish=None):
    """``publish`` = (flag, counter[, add]): published on entry of the wait launch"""
    max_spins = WAIT_SPINS if max_spins is None else max_spins
    _lib.check(_lib.load().dv_flag_wait(_i32(flag), _i32(ctr), add, _i32(err), max_spins, _publish(publish), _stream()),
               'dv_flag_wait')


def counter_add(counter, inc=1):
    _lib.check(_lib.load().dv_counter_add(_i32(counter), counter.numel(), inc, _stream()), 'dv_counter_add')


def counters_add2(c1, inc1, c2, inc2, publish=None):
    """``publish`` = (flag, ctr[, add]) goes out on entry, before the counters move (ctr may be one of them)"""
    _lib.check(_lib.load().dv_counters_add2(_i32(c1), c1.numel(), inc1, _i32(c2), c2.numel(), inc2, _publish(publish),
                                            _stream()), 'dv_counters_add2')


def fill_normal_rows(arena, desc, seed, ctr_dev=None, park=None):
    """row-keyed N(0,1) draws of a train step's noise arena: ``desc`` (R,4) int32 = {offset, width, draw id,
    global row}; ``ctr_dev`` counts draw events (see ``dv_fill_normal_rows``)"""
    assert arena.is_contiguous() and desc.dim() == 2 and desc.shape[1] == 4
    _lib.check(_lib.load().dv_fill_normal_rows(_f32(arena), _i32(desc), desc.shape[0], seed, _i32(ctr_dev), _wait(park),
                                               _stream()), 'dv_fill_normal_rows')


def fill_normal(out, seed, ctr_dev=None):
    assert out.is_contiguous()
    _lib.check(_lib.load().dv_fill_normal(_f32(out), out.numel(), seed, _i32(ctr_dev), _stream()),
               'dv_fill_normal')
