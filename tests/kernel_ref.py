"""Plain-PyTorch fp32 reference of every launcher in ``drvae_amd.kernels`` (same names,
same signatures, writes into the same output tensors; works on CPU or GPU tensors).

Two uses, both test-only:
  * ``-m gpu`` tests run each HIP kernel next to its reference here;
  * CPU tests install these as a stand-in for ``drvae_amd.kernels`` (see
    ``install``) to check the host-side orchestration (autograd wrappers, the fused
    step engine's hand-written backward, data-parallel sharding) against the oracle
    without a GPU.  The product never imports this module.
"""
import math

import numpy as np

import torch
import torch.nn.functional as F

from drvae_amd._lib import ACT, EPI_BWD, EPI_FWD, EPI_PLAIN, GAUSS_LOGVAR, GAUSS_SIGMA  # noqa: F401

LOG_2PI = 1.8378770664093453
_ACT_NAME = {v: k for k, v in ACT.items()}


def _name(a):
    return a if isinstance(a, str) else _ACT_NAME[int(a)]


def act_fwd(a, x):
    a = _name(a)
    if a == 'identity':
        return x
    if a == 'elu':
        return F.elu(x)
    if a == 'softplus':
        return F.softplus(x)
    if a == 'sigmoid':
        return torch.sigmoid(x)
    if a == 'tanh':
        return torch.tanh(x)
    if a == 'relu':
        return torch.relu(x)
    if a == 'leaky_relu':
        return F.leaky_relu(x, 0.1)
    if a == 'selu':
        return F.selu(x)
    if a == 'softsign':
        return F.softsign(x)
    if a == 'cos':
        return torch.cos(x)
    raise KeyError(a)


def dact_from_y(a, y):
    a = _name(a)
    one = torch.ones_like(y)
    if a == 'identity':
        return one
    if a == 'elu':
        return torch.where(y > 0, one, y + 1)
    if a == 'softplus':
        return 1 - torch.exp(-y)
    if a == 'sigmoid':
        return y * (1 - y)
    if a == 'tanh':
        return 1 - y * y
    if a == 'relu':
        return (y > 0).float()
    if a == 'leaky_relu':
        return torch.where(y > 0, one, 0.1 * one)
    if a == 'selu':
        l_, a_ = 1.0507009873554804934193349852946, 1.6732632423543772848170429916717
        return torch.where(y > 0, l_ * one, y + l_ * a_)
    if a == 'softsign':
        return (1 - y.abs()) ** 2
    raise KeyError(a)


def _acc(out, val, beta):
    if beta != 0.0:
        out.copy_(beta * out + val)
    else:
        out.copy_(val)


def _seg(N, split, a0, a1, v0, v1, dev):
    """per-column selection helper for the dual-head epilogues"""
    first = (torch.arange(N, device=dev) < split)
    return first


def _ldr(t):
    """leading dimension as ``drvae_amd.kernels._ld`` sees it"""
    if t is None:
        return 0
    if t.dim() == 1:
        return t.numel()
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def _widen(t, cols):
    """the (rows, cols) view over the SAME rows of a row-padded buffer (what a kernel sees when it runs over padded K / N)"""
    return torch.as_strided(t, (t.shape[0], cols), (_ldr(t), 1), t.storage_offset())


def gemm(Cm, A, B, a_kc, b_kc, *, A2=None, a_kscale=None, alpha=1.0, beta=0.0, epi=EPI_PLAIN, scale=None,
         bias=None, split=None, act0=0, act1=0, shift0=0.0, shift1=0.0, resid=None, resid_cols=0, yref=None,
         a_colsum=None, colsum_beta=0.0, overread=False, publish=None, kpad=False, npad=False):
    """mirrors ``kernels._gemm_desc`` INCLUDING its padded forms (round 5, advisor): with ``kpad`` the product runs over
    K rounded up to 4 and so reads the operands' pad columns; with ``npad`` it runs over N rounded up and WRITES C's pad
    columns -- a pad that is not zero shows up in the CPU suite exactly as it would on the device"""
    if publish is not None:
        flag_publish(publish[0], publish[1], publish[2])
    M, N = Cm.shape
    K_ = (A.shape[1] + (A2.shape[1] if A2 is not None else 0)) if a_kc else A.shape[0]
    Kp, Np = (K_ + 3) & ~3, (N + 3) & ~3
    if kpad and a_kc and b_kc and A2 is None and (K_ & 3) and min(_ldr(A), _ldr(B)) >= Kp and not ((_ldr(A) | _ldr(B)) & 3):
        A, B = _widen(A, Kp), _widen(B, Kp)
    if npad and not b_kc and (N & 3) and min(_ldr(B), _ldr(Cm)) >= Np and not ((_ldr(B) | _ldr(Cm)) & 3) and resid is None \
            and scale is None and bias is None and (yref is None or _ldr(yref) >= Np):
        B, Cm = _widen(B, Np), _widen(Cm, Np)
        if yref is not None:
            yref = _widen(yref, Np)
        split = N if split is None else split
        N = Np
    Aop = (torch.cat([A, A2], 1) if A2 is not None else A) if a_kc else A.t()
    if a_kscale is not None:
        Aop = Aop * a_kscale[None, :]
    Bop = B.t() if b_kc else B
    v = alpha * (Aop @ Bop)
    split = N if split is None else split
    first = _seg(N, split, act0, act1, shift0, shift1, Cm.device)[None, :]
    if epi == EPI_FWD:
        if scale is not None:
            v = v * scale[None, :]
        if bias is not None:
            v = v + bias[None, :]
        v = torch.where(first, act_fwd(act0, v) + shift0, act_fwd(act1, v) + shift1)
        if resid is not None and resid_cols > 0:
            v = torch.cat([v[:, :resid_cols] + resid[:, :resid_cols], v[:, resid_cols:]], 1)
    elif epi == EPI_BWD:
        d = torch.where(first, dact_from_y(act0, yref - shift0), dact_from_y(act1, yref - shift1))
        v = v * d
    _acc(Cm, v, beta)
    if a_colsum is not None:
        _acc(a_colsum, Aop.sum(1), colsum_beta)


def linear_fwd(out, x, W, bias=None, *, x2=None, scale=None, split=None, act0=0, act1=0, shift0=0.0, shift1=0.0,
               resid=None, resid_cols=0, overread=False, publish=None, kpad=False):
    gemm(out, x, W, True, True, A2=x2, epi=EPI_FWD, scale=scale, bias=bias, split=split, act0=act0, act1=act1,
         shift0=shift0, shift1=shift1, resid=resid, resid_cols=resid_cols, publish=publish, kpad=kpad)


def heads_tiles(split):
    return (split + 15) // 16


def linear_heads(out, x, W, bias=None, *, split, x2=None, scale=None, act0=0, act1=0, shift0=0.0, shift1=0.0,
                 resid=None, resid_cols=0, overread=False, publish=None, sample=None, nll=None, kpad=False):
    """the unfused sequence ``dv_gemm_heads`` replaces: heads GEMM, then reparam_fwd / nll_rows_fwdbwd"""
    M, N = out.shape
    heads = torch.zeros(M, N, device=out.device) if nll is not None else out
    linear_fwd(heads, x, W, bias, x2=x2, scale=scale, split=split, act0=act0, act1=act1, shift0=shift0, shift1=shift1,
               resid=resid, resid_cols=resid_cols, publish=publish, kpad=kpad)
    mu, sd = heads[:, :split], heads[:, split:]
    if sample is not None:
        g = sample.get
        n_src, eps, dst = sample['n_src'], sample['eps'], sample['out']
        for r in range(n_src):
            rows = [r] if g('seg_ptr') is None else \
                g('seg_rows')[int(g('seg_ptr')[r]):int(g('seg_ptr')[r + 1])].long().tolist()
            for s_ in rows:
                z = eps[s_] * torch.exp(0.5 * sd[r]) + mu[r]
                dst[s_] = z
                if g('out2') is not None:
                    g('out2')[s_] = z - g('sub')[s_]
                if g('out3') is not None and int(g('out3_idx')[s_]) >= 0:
                    g('out3')[int(g('out3_idx')[s_])] = z
                if g('out4') is not None:
                    g('out4')[int(g('out4_ptr')[s_]):int(g('out4_ptr')[s_ + 1])] = z
    else:
        rows = torch.zeros(M, device=out.device)
        nll_rows_fwdbwd(rows, out[:, :split], out[:, split:], nll['coef'], nll['x'], mu, sd, mode=GAUSS_SIGMA,
                        xidx=nll.get('xidx'), sd_act=act1, sd_shift=shift1)
        nll['part'].zero_()
        nll['part'][:, 0] = rows              # (any split of a row's sum over the tiles is as good)


def linear_bwd_data(dx, dpre, W, *, kscale=None, alpha=1.0, beta=0.0, yref=None, act=0, shift=0.0, overread=False,
                    npad=False):
    if yref is None:
        gemm(dx, dpre, W, True, False, a_kscale=kscale, alpha=alpha, beta=beta, npad=npad)
    else:
        gemm(dx, dpre, W, True, False, a_kscale=kscale, alpha=alpha, beta=beta, epi=EPI_BWD, yref=yref, act0=act,
             act1=act, shift0=shift, shift1=shift, npad=npad)


def linear_bwd_weight(dW, dpre, x, *, beta=0.0, dbias=None, overread=False, npad=False):
    gemm(dW, dpre, x, False, False, beta=beta, a_colsum=dbias, colsum_beta=beta, npad=npad)


def linear_bwd_pair(dW, dbias, dx, dpre, x, W, *, kscale=None, alpha=1.0, beta_x=0.0, yref=None, act=0, shift=0.0,
                    overread=False, publish=None, npad=False, npad_x=False, klq=None):
    if publish is not None:
        flag_publish(publish[0], publish[1], publish[2])
    linear_bwd_weight(dW, dpre, x, dbias=dbias, npad=npad)
    if klq is not None:        # DV_EPI_KLQ: d/d(mu | logvar) of the sampled q rows instead of dx
        Z, q, out = klq['Z'], klq['q'], klq['out']
        v = alpha * (dpre @ W)[:, :Z]
        gm, gs = _prior_kl_grad(klq['coef'], klq['raw'], klq['kl_min'], q[:, :Z], q[:, Z:2 * Z])
        out[:, :Z] = gm + v
        out[:, Z:2 * Z] = gs + v * klq['eps'] * 0.5 * torch.exp(0.5 * q[:, Z:2 * Z])
        return
    linear_bwd_data(dx, dpre, W, kscale=kscale, alpha=alpha, beta=beta_x, yref=yref, act=act, shift=shift, npad=npad_x)


def colsum(out, X, beta=0.0):
    _acc(out, X.sum(0), beta)


def act_bwd_(dY, Y, *, split=None, act0=0, act1=0, shift0=0.0, shift1=0.0):
    N = Y.shape[1]
    first = _seg(N, N if split is None else split, 0, 0, 0, 0, Y.device)[None, :]
    dY.mul_(torch.where(first, dact_from_y(act0, Y - shift0), dact_from_y(act1, Y - shift1)))


def wn_scale(scale, norm, W, g):
    nrm = torch.norm(W, 2, 1)
    if norm is not None:
        norm.copy_(nrm)
    scale.copy_(g / nrm)


def wn_bwd(dW, dg, dWraw, W, g, norm, beta=0.0):
    dot = (W * dWraw).sum(1)
    _acc(dW, (g / norm)[:, None] * dWraw - (dot * g / norm ** 3)[:, None] * W, beta)
    _acc(dg, dot / norm, beta)


def _std(sd, mode):
    return torch.exp(0.5 * sd) if mode == GAUSS_LOGVAR else sd


def _qrows(n, reps, idx, dev):
    j = torch.arange(n * reps, device=dev) % n
    return idx.long()[j] if idx is not None else j


def reparam_fwd(out, mu, sd, eps, *, mode=GAUSS_LOGVAR, src_idx=None, reps=1, sub=None, out2=None, out3=None,
                out3_idx=None):
    R = out.shape[0]
    q = _qrows(R // reps, reps, src_idx, out.device)
    z = mu[q] + eps * _std(sd[q], mode)
    out.copy_(z)
    if out2 is not None:
        out2.copy_(z - sub)
    if out3 is not None:
        sel = out3_idx.long() >= 0
        out3[out3_idx.long()[sel]] = z[sel]


def reparam_bwd(dmu, dsd, dz, eps, sd, *, mode=GAUSS_LOGVAR, src_idx=None, reps=1, beta=0.0):
    R, Z = dz.shape
    n = R // reps
    a = dz.reshape(reps, n, Z).sum(0)
    b = (dz * eps).reshape(reps, n, Z).sum(0)
    q = src_idx.long() if src_idx is not None else torch.arange(n, device=dz.device)
    if mode == GAUSS_LOGVAR:
        b = b * 0.5 * torch.exp(0.5 * sd[q])
    if beta != 0.0:
        dmu[q] = beta * dmu[q] + a
        dsd[q] = beta * dsd[q] + b
    else:
        dmu[q] = a
        dsd[q] = b


def _prior_kl_grad(coef, raw, kl_min, mu, lv):
    c = (coef * torch.where(raw > kl_min, torch.ones_like(raw), torch.where(raw == kl_min, 0.5 * torch.ones_like(raw),
                                                                           torch.zeros_like(raw))))[:, None]
    return c * mu, c * (-0.5 * (1 - torch.exp(lv)))


def reparam_bwd_seg(dmu, dsd, dz, eps, sd, seg_ptr, seg_rows, *, mode=GAUSS_LOGVAR, extra=None, ex_ptr=None,
                    ex_rows=None, beta=0.0, bump=None, dz_add=None, park=None, prior=None):
    if park is not None:
        flag_wait(park[0], park[1], park[2], park[3] if len(park) > 3 else 1)
    for (c, inc) in (bump or ()):
        if c is not None:
            counter_add(c, inc)
    if dz_add is not None:
        dz = dz.clone()
        dz[:dz_add.shape[0]] += dz_add
    nq, Z = seg_ptr.numel() - 1, dz.shape[1]
    for i in range(nq):
        rows = seg_rows[int(seg_ptr[i]):int(seg_ptr[i + 1])].long()
        a = dz[rows].sum(0)
        b = (dz[rows] * eps[rows]).sum(0)
        if mode == GAUSS_LOGVAR:
            b = b * 0.5 * torch.exp(0.5 * sd[i])
        if extra is not None:
            er = ex_rows[int(ex_ptr[i]):int(ex_ptr[i + 1])].long()
            a = a + extra[er][:, :Z].sum(0)
            b = b + extra[er][:, Z:2 * Z].sum(0)
        dmu[i] = (beta * dmu[i] if beta != 0.0 else 0) + a
        dsd[i] = (beta * dsd[i] if beta != 0.0 else 0) + b
    if prior is not None:
        gm, gs = _prior_kl_grad(prior[0][:nq], prior[1][:nq], prior[2], prior[3][:nq], sd[:nq])
        dmu[:nq] += gm
        dsd[:nq] += gs


def z2f_post_bwd(dp2, dz1, dq2, dz2f, dzdec_pert, pair_slot, eps, p2, q2, coef, raw, kl_min, dz1b, L, B, Np,
                 park=None, prior=None):
    if park is not None:
        flag_wait(park[0], park[1], park[2], park[3] if len(park) > 3 else 1)                  # single-threaded stand-in: the producer has run already
    Z = dp2.shape[1] // 2
    dev = dp2.device
    if dz2f is None:
        dz2f = torch.zeros(L * B, Z, device=dev)
    slot = pair_slot.long() if (pair_slot is not None and Np) else torch.full((B,), -1, dtype=torch.long, device=dev)
    is_pair = slot >= 0
    if Np and dq2 is not None:
        dq2.zero_()
    for l in range(L):
        rs = slice(l * B, (l + 1) * B)
        mp, lp = p2[rs, :Z], p2[rs, Z:2 * Z]
        g = dz2f[rs].clone()
        if Np and dzdec_pert is not None:
            g[is_pair] = g[is_pair] + dzdec_pert[l * Np + slot[is_pair]]
        dmu = g.clone()
        dlv = g * eps[rs] * 0.5 * torch.exp(0.5 * lp)
        if Np:
            kr = l * Np + slot[is_pair]
            rv = raw[kr]
            c = (coef[kr] * torch.where(rv > kl_min, torch.ones_like(rv),
                                        torch.where(rv == kl_min, 0.5 * torch.ones_like(rv), torch.zeros_like(rv))))[:, None]
            mq, lq = q2[slot[is_pair], :Z], q2[slot[is_pair], Z:2 * Z]
            dm, ivp, vq = mq - mp[is_pair], torch.exp(-lp[is_pair]), torch.exp(lq)
            if dq2 is not None:
                dq2[slot[is_pair], :Z] += c * dm * ivp
                dq2[slot[is_pair], Z:2 * Z] += c * (-0.5 * (1 - vq * ivp))
            dmu[is_pair] = dmu[is_pair] - c * dm * ivp
            dlv[is_pair] = dlv[is_pair] + c * (-0.5 * (-1 + (dm * dm + vq) * ivp))
        dp2[rs, :Z] = dmu
        dp2[rs, Z:2 * Z] = dlv
        dz1[rs] = dz1[rs] + dmu + (dz1b[rs] if dz1b is not None else 0)
    if prior is not None and Np and dq2 is not None:
        gm, gs = _prior_kl_grad(prior[0][:Np], prior[1][:Np], kl_min, q2[:Np, :Z], q2[:Np, Z:2 * Z])
        dq2[:Np, :Z] += gm
        dq2[:Np, Z:2 * Z] += gs


def _kl_operands(mu_q, sd_q, mu_p, sd_p, prior, qidx, pidx, R, reps):
    q = _qrows(R // reps, reps, qidx, mu_q.device)
    mq, sq = mu_q[q], sd_q[q]
    if mu_p is None:
        mp, sp = torch.full_like(mq, prior[0]), torch.full_like(sq, prior[1])
    else:
        p = pidx.long() if pidx is not None else torch.arange(R, device=mu_q.device)
        mp, sp = mu_p[p], sd_p[p]
    return mq, sq, mp, sp


def kl_rows_fwd(out, raw, mu_q, sd_q, mu_p=None, sd_p=None, *, prior=(0.0, 0.0), mode=GAUSS_LOGVAR, qidx=None,
                pidx=None, reps=1, free_bits=False, kl_min=0.0, add=None, eps=None, zout=None, park=None, second=None):
    if park is not None:
        flag_wait(park[0], park[1], park[2], park[3] if len(park) > 3 else 1)
    R = out.numel()
    mq, sq, mp, sp = _kl_operands(mu_q, sd_q, mu_p, sd_p, prior, qidx, pidx, R, reps)
    if zout is not None:
        zout.copy_(mq + eps * _std(sq, mode))
    if mode == GAUSS_LOGVAR:
        t = 1 - sp + sq - ((mq - mp) ** 2 + sq.exp()) / sp.exp()
    else:
        t = 1 - torch.log(sp ** 2) + torch.log(sq ** 2) - ((mq - mp) ** 2 + sq ** 2) / sp ** 2
    r = -0.5 * t.sum(1)
    if raw is not None:
        raw.copy_(r)
    v = torch.clamp(r, min=kl_min) if free_bits else r
    if add is not None:
        v = v + add
    if second is not None:
        m2, s2, raw2 = second
        pm, ps = torch.full_like(m2, prior[0]), torch.full_like(s2, prior[1])
        if mode == GAUSS_LOGVAR:
            t2 = 1 - ps + s2 - ((m2 - pm) ** 2 + s2.exp()) / ps.exp()
        else:
            t2 = 1 - torch.log(ps ** 2) + torch.log(s2 ** 2) - ((m2 - pm) ** 2 + s2 ** 2) / ps ** 2
        r2 = -0.5 * t2.sum(1)
        if raw2 is not None:
            raw2.copy_(r2)
        v = v + (torch.clamp(r2, min=kl_min) if free_bits else r2)
    out.copy_(v)


def kl_rows_fwd_pair(first, second):
    kl_rows_fwd(*first[0], **first[1])
    kl_rows_fwd(*second[0], **second[1])


def kl_rows_bwd(dq_mu, dq_sd, dp_mu, dp_sd, coef, raw, mu_q, sd_q, mu_p=None, sd_p=None, *, prior=(0.0, 0.0),
                mode=GAUSS_LOGVAR, qidx=None, pidx=None, reps=1, free_bits=False, kl_min=0.0, beta=0.0, dz=None,
                eps=None):
    R = coef.numel()
    mq, sq, mp, sp = _kl_operands(mu_q, sd_q, mu_p, sd_p, prior, qidx, pidx, R, reps)
    c = coef.clone()
    if free_bits:
        c = c * torch.where(raw > kl_min, torch.ones_like(raw),
                            torch.where(raw == kl_min, 0.5 * torch.ones_like(raw), torch.zeros_like(raw)))
    c = c[:, None]
    dm = mq - mp
    if mode == GAUSS_LOGVAR:
        ivp, vq = torch.exp(-sp), torch.exp(sq)
        gmq, gsq = dm * ivp, -0.5 * (1 - vq * ivp)
        gsp = -0.5 * (-1 + (dm * dm + vq) * ivp)
    else:
        vp = sp * sp
        gmq, gsq = dm / vp, -1 / sq + sq / vp
        gsp = 1 / sp - (dm * dm + sq * sq) / (vp * sp)
    tq_mu, tq_sd = c * gmq, c * gsq
    if dz is not None:
        tq_mu = tq_mu + dz
        tq_sd = tq_sd + dz * eps * (0.5 * torch.exp(0.5 * sq) if mode == GAUSS_LOGVAR else 1.0)
    _acc(dq_mu, tq_mu, beta)
    _acc(dq_sd, tq_sd, beta)
    if dp_mu is not None:
        _acc(dp_mu, -c * gmq, beta)
        _acc(dp_sd, c * gsp, beta)


def nll_rows_fwd(out, x, mu, sd, *, mode=GAUSS_SIGMA, xidx=None, bias=None, sd_shift=1e-3):
    if bias is not None:          # raw heads: finished here
        mu = mu + bias[0]
        sd = F.softplus(sd + bias[1]) + sd_shift
    xr = x[xidx.long()] if xidx is not None else x
    if mode == GAUSS_SIGMA:
        t = LOG_2PI + torch.log(sd ** 2) + (xr - mu) ** 2 / sd ** 2
    else:
        t = LOG_2PI + sd + (xr - mu) ** 2 / sd.exp()
    out.copy_(-0.5 * t.sum(1))


def nll_rows_bwd(dmu, dsd, coef, x, mu, sd, *, mode=GAUSS_SIGMA, xidx=None, sd_act=0, sd_shift=0.0, dx=None,
                 beta=0.0):
    xr = x[xidx.long()] if xidx is not None else x
    d = xr - mu
    c = coef[:, None]
    if mode == GAUSS_SIGMA:
        gm, gs = d / sd ** 2, -1 / sd + d * d / sd ** 3
    else:
        gm, gs = d * torch.exp(-sd), -0.5 * (1 - d * d * torch.exp(-sd))
    if _name(sd_act) != 'identity':
        gs = gs * dact_from_y(sd_act, sd - sd_shift)
    _acc(dmu, c * gm, beta)
    _acc(dsd, c * gs, beta)
    if dx is not None:
        _acc(dx, -c * gm, beta)


def rec_nll_rows(out, x, v, *, kind, shift=0.0, xidx=None, coef=None, dpre=None):
    xs = x[xidx.long()] if xidx is not None else x
    if kind == 'binary':
        lo, hi = 1e-10, float(np.float32(1.0 - 1e-10))
        pc = v.clamp(lo, hi)
        out.copy_((xs * pc.log() + (1 - xs) * (1 - pc).log()).sum(1))
        g = torch.where((v > lo) & (v < hi), xs - v, torch.zeros_like(v))
    else:
        out.copy_((xs * v.log() - v - torch.lgamma(xs + 1)).sum(1))
        g = (xs / v - 1) * (1 - torch.exp(-(v - shift)))
    if coef is not None:
        dpre.copy_(coef[:, None] * g)


def nll_rows_fwdbwd(out, dmu, dsd, coef, x, mu, sd, *, mode=GAUSS_SIGMA, xidx=None, sd_act=0, sd_shift=0.0, bias=None):
    if bias is not None:          # raw heads: finished here
        mu = mu + bias[0]
        sd = act_fwd(sd_act, sd + bias[1]) + sd_shift
    nll_rows_fwd(out, x, mu, sd, mode=mode, xidx=xidx)
    nll_rows_bwd(dmu, dsd, coef, x, mu, sd, mode=mode, xidx=xidx, sd_act=sd_act, sd_shift=sd_shift)


P_MIN = 1e-10


def softmax_clamp_fwd(probs, logits, sigmoid1=False):
    if sigmoid1:
        s = torch.sigmoid(logits[:, :1])
        p = torch.cat([1 - s, s], 1)
    else:
        p = torch.softmax(logits, -1)
    probs.copy_(torch.clamp(p, min=P_MIN, max=1. - 1e-10))


def softmax_clamp_bwd(dlogits, dprobs, probs, sigmoid1=False, beta=0.0):
    g = torch.where(probs > P_MIN, dprobs, torch.zeros_like(dprobs))
    if sigmoid1:
        s = probs[:, 1:2]
        _acc(dlogits, (g[:, 1:2] - g[:, 0:1]) * s * (1 - s), beta)
    else:
        dot = (g * probs).sum(1, keepdim=True)
        _acc(dlogits, probs * (g - dot), beta)


def cat_terms_fwd(probs, *, labels=None, prior=None, logp=None, kl=None, ent=None, best=None):
    lp = probs.log()
    if logp is not None:
        logp.copy_(lp.gather(1, labels.long()[:, None])[:, 0])
    if kl is not None:
        kl.copy_(-probs * (prior.log() - lp))
    if ent is not None:
        ent.copy_(-(probs * lp).sum(1))
    if best is not None:
        best.copy_(torch.max(probs, 1)[1].to(best.dtype))


def cat_terms_bwd(dprobs, probs, *, labels=None, prior=None, c_logp=None, g_kl=None, c_ent=None, beta=0.0):
    lp = probs.log()
    v = torch.zeros_like(probs)
    if c_logp is not None:
        oh = F.one_hot(labels.long(), probs.shape[1]).to(probs.dtype)
        v = v + oh * (c_logp[:, None] / probs)
    if g_kl is not None:
        v = v + g_kl * (lp - prior.log() + 1)
    if c_ent is not None:
        v = v - c_ent[:, None] * (lp + 1)
    _acc(dprobs, v, beta)


def _dlogits(dprobs, probs):
    if probs is None:
        return dprobs
    g = torch.where(probs > P_MIN, dprobs, torch.zeros_like(dprobs))
    return probs * (g - (g * probs).sum(1, keepdim=True))


def smalln_fwd(probs, logits, a1, W, bias=None, a2=None, ymarg=None, park=None, fprop_kl=None):
    if park is not None:
        flag_wait(*park)
    if ymarg is not None:
        yl, kld, cfp, dqy, label, fp_ptr, klfp, log_prior, c_kld, c_yl = ymarg
        f = fprop_kl
        if f is not None:      # the unfused sequence the fused launch replaces
            Z1, Z3 = f['Z1'], f['Z3']
            kl_rows_fwd(klfp, f['raw1'], f['Q'][:, :Z1], f['Q'][:, Z1:], f['P'][:, :Z1], f['P'][:, Z1:], qidx=f['qidx'],
                        free_bits=True, kl_min=f['kl_min'], prior=(0.0, 0.0),
                        second=(f['Q3'][:, :Z3], f['Q3'][:, Z3:], f['raw3']))
        smalln_fwd(probs, logits, a1, W, bias, a2)
        ymarg_fwdbwd(yl, kld, cfp, dqy, probs, label, fp_ptr, klfp, log_prior, c_kld, c_yl)
        if f is not None:
            kl_rows_bwd(f['dq'][:, :Z1], f['dq'][:, Z1:], f['dp'][:, :Z1], f['dp'][:, Z1:], cfp, f['raw1'],
                        f['Q'][:, :Z1], f['Q'][:, Z1:], f['P'][:, :Z1], f['P'][:, Z1:], qidx=f['qidx'], free_bits=True,
                        kl_min=f['kl_min'])
        return
    x = torch.cat([a1, a2], 1) if a2 is not None else a1
    z = x @ W.t() + (bias if bias is not None else 0)
    if logits is not None:
        logits.copy_(z)
    if probs is not None:
        probs.copy_(torch.clamp(torch.softmax(z, -1), min=P_MIN, max=1. - 1e-10))


def smalln_bwd_data(dsts, dprobs, probs, W, seg=None):
    dl = _dlogits(dprobs, probs)
    for di, d in enumerate(dsts):
        dst, col0, alpha, beta = d[:4]
        w = dst.shape[1]
        if di == 0 and seg is not None:
            rows_segment_sum(dst, seg[0], seg_ptr=seg[1], beta=0.0, width=w)
            beta = 1.0
        v = alpha * (dl @ W[:, col0:col0 + w])
        if len(d) > 5 and d[5] != 0.0:
            v = v + d[5] * (dl @ W[:, d[4]:d[4] + w])
        _acc(dst, v, beta)


def smalln_ws_numel(N, K):
    return 16 * N * (K + 1)


def smalln_bwd_weight(dW, db, dprobs, probs, a1, a2=None, beta=0.0, publish=None, ws=None):
    if publish is not None:
        flag_publish(*publish)
    dl = _dlogits(dprobs, probs)
    x = torch.cat([a1, a2], 1) if a2 is not None else a1
    _acc(dW, dl.t() @ x, beta)
    if db is not None:
        _acc(db, dl.sum(0), beta)


def ymarg_fwd(yl, kld, qy, label, fp_ptr, klfp, log_prior):
    R, Y = qy.shape
    f0 = fp_ptr[:-1].long()
    nf = (fp_ptr[1:] - fp_ptr[:-1]).long()
    lab = nf == 1
    slot = label.long() <= -2                 # labeled row with all Y class slots materialised (universal plan)
    cls = torch.where(slot, -2 - label.long(), label.long().clamp(min=0))
    j = torch.arange(Y, device=qy.device)[None, :]
    fidx = torch.where(lab[:, None], f0[:, None].expand(R, Y), f0[:, None] + j).clamp(max=max(klfp.numel() - 1, 0))
    kf = klfp[fidx]
    lq = qy.log()
    yl.copy_(torch.where(lab | slot, lq.gather(1, cls[:, None])[:, 0], torch.zeros_like(yl)))
    marg = (qy * kf).sum(1) + (-qy * (log_prior - lq)).sum(1)
    kld.copy_(torch.where(lab, kf[:, 0], torch.where(slot, kf.gather(1, cls[:, None])[:, 0], marg)))


def ymarg_fwdbwd(yl, kld, cfp, dqy, qy, label, fp_ptr, klfp, log_prior, c_kld, c_yl):
    ymarg_fwd(yl, kld, qy, label, fp_ptr, klfp, log_prior)
    ymarg_bwd(cfp, dqy, qy, label, fp_ptr, klfp, log_prior, c_kld, c_yl)


def ymarg_bwd(cfp, dqy, qy, label, fp_ptr, klfp, log_prior, c_kld, c_yl):
    R, Y = qy.shape
    f0 = fp_ptr[:-1].long()
    nf = (fp_ptr[1:] - fp_ptr[:-1]).long()
    for r in range(R):       # small R in tests; clarity over speed
        if int(nf[r]) == 1:
            dqy[r].zero_()
            dqy[r, int(label[r])] = c_yl[r] / qy[r, int(label[r])]
            cfp[int(f0[r])] = c_kld[r]
        elif int(label[r]) <= -2:
            c = -2 - int(label[r])
            dqy[r].zero_()
            dqy[r, c] = c_yl[r] / qy[r, c]
            cfp[int(f0[r]):int(f0[r]) + Y] = 0
            cfp[int(f0[r]) + c] = c_kld[r]
        else:
            sl = slice(int(f0[r]), int(f0[r]) + Y)
            cfp[sl] = c_kld[r] * qy[r]
            dqy[r] = c_kld[r] * (klfp[sl] + qy[r].log() - log_prior + 1)


def ycont_fwd(yl, fpin_y, z3in_y, mu, ylab, has_y, eps, logvar, B, sqerr=False):
    R, Y = mu.shape
    i = torch.arange(R, device=mu.device) % B
    lab = has_y[i].bool()
    yv = torch.where(lab[:, None], ylab[i], mu + math.exp(0.5 * logvar) * eps[:, :Y])
    ll = -((ylab[i] - mu) ** 2).sum(1) if sqerr else \
        -0.5 * (LOG_2PI + logvar + (ylab[i] - mu) ** 2 / math.exp(logvar)).sum(1)
    yl.copy_(torch.where(lab, ll, torch.zeros_like(ll)))
    fpin_y[:, :Y] = yv
    z3in_y[:, :Y] = yv


def ycont_bwd(dlogit, cfp, mu, ylab, has_y, logvar, c_yl, c_kld, dfpin_y, dz3in_y, B, sqerr=False):
    R, Y = mu.shape
    if dlogit is None:
        cfp[:R] = c_kld[:R]
        return
    i = torch.arange(R, device=mu.device) % B
    lab = has_y[i].bool()
    dmu = torch.where(lab[:, None], c_yl[:, None] * (ylab[i] - mu) * (2.0 if sqerr else 1.0 / math.exp(logvar)),
                      dfpin_y[:, :Y] + dz3in_y[:, :Y])
    dlogit[:, :Y] = dmu * mu * (1 - mu)


def mmd_rff_fwd(diff, mmd2, th1, th2, c):
    d = c * (torch.cos(th1).mean(0) - torch.cos(th2).mean(0))
    diff.copy_(d)
    mmd2[0] = (d ** 2).sum()


def mmd_rff_bwd(G, th, diff, gout, coef):
    G.copy_(-coef * gout.reshape(-1)[0] * diff[None, :] * torch.sin(th))


def mmd_identity_fwd(diff, out, x1, x2):
    d = x1.mean(0) - x2.mean(0)
    diff.copy_(d)
    out[0] = (d ** 2).sum()


def mmd_identity_bwd(dx, diff, gout, coef):
    dx.copy_((coef * gout.reshape(-1)[0] * diff)[None, :].expand_as(dx))


def _mix(G, kind, gammas, sa, sb):
    """the kernel mixture on a Gram matrix and its derivative w.r.t. G (poly) / the squared distance (rbf): dv_mmd_mix_*"""
    gam = [float(g) for g in gammas]
    if kind == 'poly':
        v = sum((g * G + 1.0) ** 2 for g in gam) / len(gam)
        dv = sum(2.0 * g * (g * G + 1.0) for g in gam) / len(gam)
    else:
        d2 = (sa.diagonal()[:, None] + sb.diagonal()[None, :] - 2.0 * G).clamp_min(0.0)
        v = sum(torch.exp(-g * d2) for g in gam) / len(gam)
        dv = sum(-g * torch.exp(-g * d2) for g in gam) / len(gam)
    return v, dv


def mmd_mix_fwd(part, G, kind, gammas, sa=None, sb=None):
    part.copy_(_mix(G, kind, gammas, sa, sb)[0].sum(1))


def mmd_mix_bwd(W, rs, G, kind, gammas, gout, coef, sa=None, sb=None):
    W.copy_(coef * gout.reshape(-1)[0] * _mix(G, kind, gammas, sa, sb)[1])
    rs.copy_(W.sum(1))


def mmd_mix_combine(out4, p11, p12, p22, c11, c12, c22):
    m11, m12, m22 = p11.double().sum() / c11, p12.double().sum() / c12, p22.double().sum() / c22
    out4.copy_(torch.stack([m11 - 2.0 * m12 + m22, m11, m12, m22]).float())


def rows_gather(out, src, idx=None, *, noise=None, sigma=0.0, onehot_cls=None, n_classes=0, width=None, park=None):
    if park is not None:
        flag_wait(park[0], park[1], park[2], park[3] if len(park) > 3 else 1)
    n = out.shape[0]
    W = (src.shape[1] if src is not None else 0) if width is None else width
    if W > 0:
        v = src[idx.long()][:, :W] if idx is not None else src[:n, :W]
        if noise is not None:
            v = v + sigma * noise[:n, :W]
        out[:, :W] = v
    if onehot_cls is not None:
        out[:, W:W + n_classes] = F.one_hot(onehot_cls.long(), n_classes).to(out.dtype)


def batch_feed(xin, x1, x2, y32, table, n_batches, ctr, base, *, pair_rows=None, noise=None, sigma=0.0, has_y=None,
               L=1, label_r=None, fp_i=None, fp_lab=None, fp_slot=None, fp_cls=None, onehot=None, n_classes=0, yf=None,
               ylab=None, onehot2=None, masks=None, park=None):
    if park is not None:
        flag_wait(park[0], park[1], park[2], park[3] if len(park) > 3 else 1)
    if masks is not None:
        batch_masks(table.shape[1], L, table=table, n_batches=n_batches, ctr=ctr, base=base, **masks)
    if ylab is not None:
        bb = min(max(int(ctr[0]) - int(base[0]), 0), n_batches - 1)
        ylab.copy_(yf.reshape(-1, ylab.shape[1])[table[bb].long()])
    b = min(max(int(ctr[0]) - int(base[0]), 0), n_batches - 1)
    tb = table[b].long()
    B = tb.numel()
    xin[:B] = x1[tb]
    if pair_rows is not None and pair_rows.numel():
        xin[B:] = x2[tb[pair_rows.long()]]
    if noise is not None:
        xin += sigma * noise
    if label_r is not None:
        label_r.copy_(torch.where(has_y.bool(), y32[tb], torch.zeros_like(y32[tb])).repeat(L))
    if fp_cls is not None and fp_cls.numel():
        cls = torch.where(fp_lab.bool(), y32[tb[fp_i.long()]], fp_slot)
        fp_cls.copy_(cls)
        if onehot is not None:
            onehot.copy_(torch.nn.functional.one_hot(cls.long(), n_classes).to(onehot.dtype))
        if onehot2 is not None:
            onehot2.copy_(torch.nn.functional.one_hot(cls.long(), n_classes).to(onehot2.dtype))


def rows_segment_sum(dst, src, *, seg_ptr=None, seg_rows=None, w=None, n=None, dst_idx=None, beta=0.0, width=None,
                     park=None):
    if park is not None:
        flag_wait(park[0], park[1], park[2], park[3] if len(park) > 3 else 1)
    if n is None:
        n = seg_ptr.numel() - 1 if seg_ptr is not None else (seg_rows.numel() if seg_rows is not None else
                                                              src.shape[0])
    W = dst.shape[1] if width is None else width
    for i in range(n):
        if seg_ptr is not None:
            ts = range(int(seg_ptr[i]), int(seg_ptr[i + 1]))
        else:
            ts = [i]
        s = torch.zeros(W, device=dst.device)
        for t in ts:
            row = int(seg_rows[t]) if seg_rows is not None else t
            s = s + (float(w[t]) if w is not None else 1.0) * src[row, :W]
        di = int(dst_idx[i]) if dst_idx is not None else i
        dst[di, :W] = (beta * dst[di, :W] if beta != 0.0 else 0) + s


def weighted_sum(out, x, w=None, idx=None, scale=1.0, beta=0.0, n=None):
    if n is None:
        n = idx.numel() if idx is not None else x.numel()
    xv = x.reshape(-1)[idx.long()[:n]] if idx is not None else x.reshape(-1)[:n]
    s = (xv * w.reshape(-1)[:n]).sum() if w is not None else xv.sum()
    out.reshape(-1)[0] = (beta * out.reshape(-1)[0] if beta != 0.0 else 0) + scale * s


def recon_row_stats(out, x, r):
    mx, mr = x.mean(1, keepdim=True), r.mean(1, keepdim=True)
    out.copy_(torch.stack([((x - r) ** 2).sum(1), mx[:, 0], mr[:, 0], ((x - mx) ** 2).sum(1), ((r - mr) ** 2).sum(1),
                           ((x - mx) * (r - mr)).sum(1)], 1))


def col_moment_blocks(M):
    return max(1, min(64, max(min(16, (M + 255) // 256), (M + 511) // 512)))


def col_moments(out, x, r, sel=None, part=None, r_bias=None):
    if r_bias is not None:
        r = r + r_bias
    if sel is not None:
        x, r = x[sel.long()], r[sel.long()]
    xd, rd = x.double(), r.double()
    tot = torch.stack([xd.sum(0), (xd * xd).sum(0), ((xd - rd) ** 2).sum(0)], 0)
    if part is not None:
        part.zero_()
        part[0].copy_(tot)
    if out is not None:
        out.copy_(tot)


RECON_ROWS_MAX_X = 1024


def recon_rows(rows, ll, x, mu, sd, *, bias=None, sd_shift=1e-3):
    if bias is not None:          # raw heads: finished here
        mu = mu + bias[0]
        sd = F.softplus(sd + bias[1]) + sd_shift
    recon_row_stats(rows, x, mu)
    if ll is not None:
        nll_rows_fwd(ll, x, mu, sd, mode=GAUSS_SIGMA)


def recon_finalize(out4, rows, part, X, *, sel=None, n=None, ll=None):
    r = (rows[sel.long()] if sel is not None else rows).double()
    n = r.shape[0] if n is None else n
    cols = part.sum(0)
    out4[0] = torch.sqrt(r[:, 0].sum() / (n * X)) if n else float('nan')
    out4[1] = 1.0 - cols[2].sum() / (cols[1] - cols[0] ** 2 / n).sum()
    out4[2] = (r[:, 5] / torch.sqrt(r[:, 3] * r[:, 4])).mean() if n else float('nan')
    out4[3] = (ll[sel.long()] if sel is not None else ll).double().mean() if (ll is not None and n) else float('nan')


RANK_MAX_ROWS = 32768


def rank_metrics(out, counts, proba, y32, *, pred32=None, sel=None, c0=1, n_cls=1, binary=True):
    from drvae_amd import metrics as MET
    idx = sel.long() if sel is not None else torch.arange(proba.shape[0])
    y, pr = y32[idx], proba[idx]
    for c in range(n_cls):
        pos = (y > 0) if binary else (y == c0 + c)
        out[2 * c] = MET.roc_auc(pos, pr[:, c0 + c])
        out[2 * c + 1] = MET.average_precision(pos, pr[:, c0 + c])
    out[2 * n_cls] = float((pred32[idx] == y).float().sum() / max(len(idx), 1)) if (pred32 is not None and len(idx)) else float('nan')


def _halted(halt):
    return halt is not None and bool((halt.reshape(-1)[0::2] != 0).any())


def loss_assemble(loss, terms, w_elbo, w_cmpl, after=None, bump=(), halt=None, accum=None):
    if after is not None:
        flag_wait(after[0], after[1], after[2], after[3], after[4])
    for (c, inc) in bump:
        if c is not None:
            counter_add(c, inc)
    if after is not None and not terms:
        return
    acc = torch.zeros(8, device=loss.device)
    for term in terms:
        x, w, scale, out = term[:4]
        row_len = term[4] if len(term) > 4 else 1
        v = x.reshape(-1)
        if w is not None and row_len > 1:
            w = w.reshape(-1).repeat_interleave(row_len)
        acc[out] = acc[out] + scale * ((v * w.reshape(-1)).sum() if w is not None else v.sum())
    acc[5] = (w_elbo[:3] * acc[:3]).sum()
    acc[6] = (w_cmpl[:8] * acc[:8]).sum()
    loss[:8] = torch.full_like(acc, float('nan')) if _halted(halt) else acc
    if accum is not None:
        accum[:8] += loss[:8]


def axpby(y, x, a=1.0, b=0.0):
    y.copy_(a * x + (b * y if b != 0.0 else 0))


def adam_l2(p, g, m, v, step_dev, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, gscale=1.0, gate=None,
            halt=None):
    if gate is not None:
        flag_wait(gate[0], gate[1], gate[3], gate[2])
    if _halted(halt):
        return
    t = int(step_dev.reshape(-1)[0])
    gg = g * gscale
    if weight_decay != 0.0:
        gg = gg + weight_decay * p
    m.lerp_(gg, 1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def adamax_l2(p, g, m, u, step_dev, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, gscale=1.0, halt=None):
    if _halted(halt):
        return
    t = int(step_dev.reshape(-1)[0])
    gg = g * gscale
    if weight_decay != 0.0:
        gg = gg + weight_decay * p
    m.lerp_(gg, 1 - beta1)
    torch.maximum(u.mul_(beta2), gg.abs().add_(eps), out=u)
    p.addcdiv_(m, u, value=-(lr / (1 - beta1 ** t)))


def flag_publish(flag, ctr, add=1):
    flag[0] = int(ctr[0]) + add


def flag_wait(flag, ctr, err, add=1, max_spins=None, publish=None):
    if publish is not None:
        flag_publish(*publish)
    if int(flag[0]) < int(ctr[0]) + add:      # single-threaded stand-in: the producer must have run already
        err[0] = 1                            # (err[1], the parked-time statistic, stays 0)


def counters_add2(c1, inc1, c2, inc2, publish=None):
    if publish is not None:
        flag_publish(*publish)
    counter_add(c1, inc1)
    counter_add(c2, inc2)


def counter_add(counter, inc=1):
    if counter.numel() == 1:
        counter += inc
    else:
        v = (int(counter[1]) & 0xffffffff) << 32 | (int(counter[0]) & 0xffffffff)
        v = (v + inc) & 0xffffffffffffffff
        lo, hi = v & 0xffffffff, v >> 32
        counter[0] = lo - (1 << 32) if lo >= (1 << 31) else lo
        counter[1] = hi - (1 << 32) if hi >= (1 << 31) else hi


def batch_masks(B, L, *, n_tot, kl_rate, pert_rate, yl_rate, beta, c_nll, w_recl, hx=None, hy=None, y=None, c_klz2=None,
                c_yl=None, w_pert=None, w_yl=None, label=None, c_klp=None, table=None, n_batches=0, ctr=None, base=None,
                Np=None, one_slot=None, gcounts=None):
    Np = B if Np is None else Np
    gc = None
    if gcounts is not None:       # data parallelism: the GLOBAL (N_pairs, N_labeled) of the batch are table data
        gc = gcounts[min(max(int(ctr[0]) - int(base[0]), 0), n_batches - 1) if table is not None else 0]
    if c_klp is not None:
        s_ = table[min(max(int(ctr[0]) - int(base[0]), 0), n_batches - 1)].long() if table is not None else \
            torch.arange(B, device=c_nll.device)
        c_klp[:B] = 1.0 / n_tot
        c_klp[B:B + Np] = torch.where(hx[s_[:Np]] != 0, 1.0 / n_tot, 0.0) if hx is not None else 0.0
    if table is not None:
        b = min(max(int(ctr[0]) - int(base[0]), 0), n_batches - 1)
        src = table[b].long()
    else:
        src = torch.arange(B, device=c_nll.device)
    LB, ct, bt = L * B, 1.0 / (L * n_tot), float(beta[0]) if beta is not None else 1.0
    c_nll[:LB] = -ct
    w_recl[:LB] = ct
    if hx is not None:
        LP = L * Np
        px = (hx[src[:Np]] != 0).repeat(L)
        npair = max(float((hx[src] != 0).sum()) if gc is None else float(gc[0]), 1.0)
        c_nll[LB:LB + LP] = torch.where(px, -ct, 0.0)
        w_recl[LB:LB + LP] = torch.where(px, ct, 0.0)
        c_nll[LB + LP:LB + 2 * LP] = torch.where(px, -bt * pert_rate / (L * npair), 0.0)
        w_pert[:LP] = torch.where(px, 1.0 / (L * npair), 0.0)
        c_klz2[:LP] = torch.where(px, bt * kl_rate * ct, 0.0)
    if hy is not None:
        py = (hy[src] != 0).repeat(L)
        nlab = max(float((hy[src] != 0).sum()) if gc is None else float(gc[1]), 1.0)
        c_yl[:LB] = torch.where(py, -yl_rate / (L * nlab), 0.0)
        w_yl[:LB] = 1.0 / (L * nlab)
        label[:LB] = torch.where(py, -2 - y[src].to(label.dtype).repeat(L), torch.zeros_like(label[:LB]))
        if one_slot is not None:
            label[:LB] = torch.where(one_slot.repeat(L) != 0, y[src].to(label.dtype).repeat(L), label[:LB])


def fill_normal_rows(arena, desc, seed, ctr_dev=None, park=None):
    """Stand-in only (NOT bit-compatible with the device Philox stream), but keyed the same way: a value
    depends on (seed, step, draw id, global row, column) only."""
    if park is not None:
        flag_wait(*park)
    base = 0 if ctr_dev is None else (int(ctr_dev[1]) << 32 | (int(ctr_dev[0]) & 0xffffffff))
    g = torch.Generator(device='cpu')
    for off, width, draw, grow in desc.cpu().tolist():
        g.manual_seed((((seed * 1000003 + base) * 1000003 + draw) * 1000003 + grow) % (1 << 62))
        arena[off:off + width] = torch.randn(width, generator=g).to(arena.device)


def fill_normal(out, seed, ctr_dev=None):
    """Stand-in only (NOT bit-compatible with the device Philox stream)."""
    g = torch.Generator(device='cpu')
    base = 0 if ctr_dev is None else (int(ctr_dev[1]) << 32 | (int(ctr_dev[0]) & 0xffffffff))
    g.manual_seed((seed * 1000003 + base) % (1 << 62))
    out.copy_(torch.randn(out.shape, generator=g).to(out.device))


def nll_raw_cs_shape(M, X):
    return ((X >> 2) + 255) // 256, (M + 63) // 64


def nll_rows_raw_cs(out_part, dmu, dsd, ws, coef, x, mu, sd, bias, *, xidx=None, sd_shift=1e-3):
    """stand-in of dv_gauss_nll_rows_raw_cs: per-chunk row partials (1024 genes per chunk), per-block (64 rows) column sums;
    dmu is None: forward only"""
    M, X = mu.shape
    chunks, rbs = nll_raw_cs_shape(M, X)
    if dmu is not None:
        full = torch.zeros(M, dtype=mu.dtype, device=mu.device)
        nll_rows_fwdbwd(full, dmu, dsd, coef, x, mu, sd, mode=GAUSS_SIGMA, xidx=xidx, sd_act='softplus', sd_shift=sd_shift, bias=bias)
    xs = x[xidx.long()] if xidx is not None else x
    m = mu + bias[0]
    s_ = F.softplus(sd + bias[1]) + sd_shift
    el = -0.5 * (math.log(2 * math.pi) + 2 * torch.log(s_) + ((xs - m) / s_) ** 2)
    for c in range(chunks):
        out_part[:, c] = el[:, c * 1024:(c + 1) * 1024].sum(1)
    if dmu is not None:
        for b in range(rbs):
            ws[b, :X] = dmu[b * 64:(b + 1) * 64].sum(0)
            ws[b, X:2 * X] = dsd[b * 64:(b + 1) * 64].sum(0)


FUNCTIONS = ['mmd_identity_fwd', 'mmd_identity_bwd', 'mmd_mix_fwd', 'mmd_mix_bwd', 'mmd_mix_combine', 'smalln_ws_numel', 'col_moment_blocks', 'recon_finalize', 'rank_metrics', 'nll_raw_cs_shape', 'nll_rows_raw_cs', 'rec_nll_rows', 'batch_masks', 'fill_normal_rows', 'linear_heads', 'heads_tiles', 'adamax_l2', 'batch_feed', 'mmd_rff_fwd', 'mmd_rff_bwd', 'counters_add2', 'ycont_fwd', 'ycont_bwd', 'flag_publish', 'flag_wait', 'gemm', 'linear_fwd', 'linear_bwd_data', 'linear_bwd_weight', 'linear_bwd_pair', 'colsum', 'act_bwd_', 'wn_scale', 'wn_bwd',
             'reparam_fwd', 'reparam_bwd', 'reparam_bwd_seg', 'z2f_post_bwd', 'kl_rows_fwd', 'kl_rows_fwd_pair', 'kl_rows_bwd', 'nll_rows_fwd', 'nll_rows_bwd', 'nll_rows_fwdbwd',
             'softmax_clamp_fwd', 'softmax_clamp_bwd', 'cat_terms_fwd', 'cat_terms_bwd', 'smalln_fwd', 'smalln_bwd_data',
             'smalln_bwd_weight', 'ymarg_fwd', 'ymarg_bwd', 'ymarg_fwdbwd',
             'rows_gather', 'rows_segment_sum', 'weighted_sum', 'recon_row_stats', 'recon_rows', 'col_moments', 'loss_assemble', 'axpby', 'adam_l2', 'counter_add', 'fill_normal']


def install(monkeypatch):
    """Replace the HIP launchers by these references for one CPU test (pytest monkeypatch)."""
    import drvae_amd.kernels as K
    me = globals()
    for name in FUNCTIONS:
        monkeypatch.setattr(K, name, me[name])
