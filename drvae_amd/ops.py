"""``torch.autograd.Function`` wrappers over the HIP launchers -- the differentiable ops
the drop-in ``blocks`` / ``layers`` modules are made of.  Forward AND backward are
hand-written kernels (``drvae_amd.kernels``); autograd only routes tensors.

(The fused train step in ``drvae_amd.engine`` does not go through autograd at all.)
"""
import math

import torch

from . import kernels as K
from ._lib import GAUSS_LOGVAR, GAUSS_SIGMA  # noqa: F401

FUSED_ACTS = ('identity', 'elu', 'softplus', 'sigmoid', 'tanh', 'relu', 'leaky_relu', 'selu', 'softsign')


def _c(t):
    """fp32 tensor with unit inner stride (row-strided views pass through untouched)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1):
        return t
    return t.contiguous()


class MMDRff(torch.autograd.Function):
    """mmd^2 of two row sets under random Fourier features (src/blocks.py:40-55): projections and their
    transposes on the MFMA GEMM, cos / column means / difference / sin on row kernels"""

    @staticmethod
    def forward(ctx, x1, x2, W, b, a, c):
        x1, x2, W = x1.contiguous(), x2.contiguous(), W.contiguous()          # W: (Z, R) row-major
        R = W.shape[1]
        dev = x1.device
        scale, bias = torch.full((R,), a, device=dev), (2 * math.pi) * b
        th1, th2 = torch.empty(x1.shape[0], R, device=dev), torch.empty(x2.shape[0], R, device=dev)
        K.gemm(th1, x1, W, True, False, epi=K.EPI_FWD, scale=scale, bias=bias)
        K.gemm(th2, x2, W, True, False, epi=K.EPI_FWD, scale=scale, bias=bias)
        diff, out = torch.empty(R, device=dev), torch.empty(1, device=dev)
        K.mmd_rff_fwd(diff, out, th1, th2, c)
        ctx.save_for_backward(th1, th2, W, diff)
        ctx.ac = (a, c)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        th1, th2, W, diff = ctx.saved_tensors
        a, c = ctx.ac
        g = g.reshape(1).contiguous().float()
        grads = []
        for th, sign, need in ((th1, 1.0, ctx.needs_input_grad[0]), (th2, -1.0, ctx.needs_input_grad[1])):
            if not need:
                grads.append(None)
                continue
            G = torch.empty_like(th)
            K.mmd_rff_bwd(G, th, diff, g, sign * 2.0 * c / th.shape[0])
            dx = torch.empty(th.shape[0], W.shape[0], device=th.device)
            K.gemm(dx, G, W, True, True, alpha=a)                 # (n,R) x (Z,R)^T
            grads.append(dx)
        return grads[0], grads[1], None, None, None, None


class MMDMix(torch.autograd.Function):
    """MMD^2 of two row sets under a mixture of polynomial or RBF kernels -- ``mmd_objective(kernel='poly' | 'rbf')``
    before its square root (src/blocks.py:59-76): the three Gram products on the MFMA GEMM, the element-wise mixture, its
    means and its derivative on row kernels (``dv_mmd_mix_*``), the input gradients as GEMMs again."""

    @staticmethod
    def forward(ctx, x1, x2, kind, gammas):
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        dev, n1, n2 = x1.device, x1.shape[0], x2.shape[0]
        G11, G22, G12 = torch.empty(n1, n1, device=dev), torch.empty(n2, n2, device=dev), torch.empty(n1, n2, device=dev)
        K.gemm(G11, x1, x1, True, True)
        K.gemm(G22, x2, x2, True, True)
        K.gemm(G12, x1, x2, True, True)
        p11, p12, p22 = torch.empty(n1, device=dev), torch.empty(n1, device=dev), torch.empty(n2, device=dev)
        K.mmd_mix_fwd(p11, G11, kind, gammas, G11, G11)
        K.mmd_mix_fwd(p22, G22, kind, gammas, G22, G22)
        K.mmd_mix_fwd(p12, G12, kind, gammas, G11, G22)
        out = torch.empty(4, device=dev)
        K.mmd_mix_combine(out, p11, p12, p22, float(n1) * n1, float(n1) * n2, float(n2) * n2)
        ctx.save_for_backward(x1, x2, G11, G22, G12)
        ctx.kind, ctx.gammas = kind, tuple(float(g) for g in gammas)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x1, x2, G11, G22, G12 = ctx.saved_tensors
        kind, gammas = ctx.kind, ctx.gammas
        dev, n1, n2, Z = x1.device, x1.shape[0], x2.shape[0], x1.shape[1]
        g = g.reshape(1).contiguous().float()
        W11, W22, W12 = torch.empty_like(G11), torch.empty_like(G22), torch.empty_like(G12)
        r11, r22, r12 = torch.empty(n1, device=dev), torch.empty(n2, device=dev), torch.empty(n1, device=dev)
        # d mmd^2 / d(m11, m12, m22) = (1, -2, 1); every mean is a sum over M x N elements.  The factors of the chain rule
        # below ride on ``coef`` (W and its row sums come out scaled): no element-wise launches in between
        #   poly: W = dL/dG, G = a b^T -> da = W b, db = W^T a; a self product is symmetric: da = 2 W a
        #   rbf : W = dL/d(d2), d2_ij = |a_i|^2 + |b_j|^2 - 2 a_i.b_j -> da = 2 (rowsum(W) * a - W b),
        #         db = 2 (colsum(W) * b - W^T a); a self product counts twice: da = 4 (rowsum(W) * a - W a)
        f_self, f_cross = (2.0, 1.0) if kind == 'poly' else (4.0, 2.0)
        K.mmd_mix_bwd(W11, r11, G11, kind, gammas, g, f_self / (n1 * n1), G11, G11)
        K.mmd_mix_bwd(W22, r22, G22, kind, gammas, g, f_self / (n2 * n2), G22, G22)
        K.mmd_mix_bwd(W12, r12, G12, kind, gammas, g, f_cross * -2.0 / (n1 * n2), G11, G22)
        dx1 = torch.empty(n1, Z, device=dev) if ctx.needs_input_grad[0] else None
        dx2 = torch.empty(n2, Z, device=dev) if ctx.needs_input_grad[1] else None
        sgn = 1.0 if kind == 'poly' else -1.0
        if dx1 is not None:
            K.gemm(dx1, W11, x1, True, False, alpha=sgn)
            K.gemm(dx1, W12, x2, True, False, alpha=sgn, beta=1.0)
        if dx2 is not None:
            K.gemm(dx2, W22, x2, True, False, alpha=sgn)
            K.gemm(dx2, W12, x1, False, False, alpha=sgn, beta=1.0)
        if kind == 'rbf':
            if dx1 is not None:
                K.rows_segment_sum(dx1, x1, w=r11, n=n1, beta=1.0)
                K.rows_segment_sum(dx1, x1, w=r12, n=n1, beta=1.0)
            if dx2 is not None:
                c12 = torch.empty(n2, device=dev)
                K.colsum(c12, W12)
                K.rows_segment_sum(dx2, x2, w=r22, n=n2, beta=1.0)
                K.rows_segment_sum(dx2, x2, w=c12, n=n2, beta=1.0)
        return dx1, dx2, None, None


class MMDIdentity(torch.autograd.Function):
    """|| mean(x1, 0) - mean(x2, 0) ||^2 -- the ``identity`` kernel (src/blocks.py:37-38) on one HIP launch each way"""

    @staticmethod
    def forward(ctx, x1, x2):
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        diff, out = torch.empty(x1.shape[1], device=x1.device), torch.empty(1, device=x1.device)
        K.mmd_identity_fwd(diff, out, x1, x2)
        ctx.save_for_backward(diff)
        ctx.n = (x1.shape[0], x2.shape[0])
        return out[0]

    @staticmethod
    def backward(ctx, g):
        diff, = ctx.saved_tensors
        g = g.reshape(1).contiguous().float()
        outs = []
        for n, sign, need in ((ctx.n[0], 1.0, ctx.needs_input_grad[0]), (ctx.n[1], -1.0, ctx.needs_input_grad[1])):
            if not need:
                outs.append(None)
                continue
            dx = torch.empty(n, diff.numel(), device=diff.device)
            K.mmd_identity_bwd(dx, diff, g, sign * 2.0 / n)
            outs.append(dx)
        return outs[0], outs[1]


class _LinearAct(torch.autograd.Function):
    """y = act(scale * ([x1|x2] W^T) + b) + shift, scale = g/||W|| when g is given.
    Replaces F.linear / nn.Linear + activation module (src/blocks.py:139-151,163) and
    WeightNormLinear.forward (src/layers.py:38-40)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, g, act, shift):
        x1, x2, weight = _c(x1), _c(x2), _c(weight)
        M, N = x1.shape[0], weight.shape[0]
        out = torch.empty(M, N, device=x1.device, dtype=torch.float32)
        scale = norm = None
        if g is not None:
            scale, norm = torch.empty_like(g), torch.empty_like(g)
            K.wn_scale(scale, norm, weight, g)
        K.linear_fwd(out, x1, weight, bias, x2=x2, scale=scale, act0=act, act1=act, shift0=shift, shift1=shift)
        ctx.act, ctx.shift = act, shift
        ctx.save_for_backward(x1, x2, weight, g, scale, norm, out)
        return out

    @staticmethod
    def backward(ctx, dy):
        x1, x2, W, g, scale, norm, out = ctx.saved_tensors
        need_x1, need_x2, need_w, need_b, need_g = ctx.needs_input_grad[:5]
        dpre = dy.contiguous().clone() if ctx.act != 'identity' else _c(dy)
        if ctx.act != 'identity':
            K.act_bwd_(dpre, out, act0=ctx.act, act1=ctx.act, shift0=ctx.shift, shift1=ctx.shift)
        K1 = x1.shape[1]
        dx1 = dx2 = dW = db = dg = None
        if need_x1:
            dx1 = torch.empty_like(x1, memory_format=torch.contiguous_format)
            K.linear_bwd_data(dx1, dpre, W[:, :K1], kscale=scale)
        if x2 is not None and need_x2:
            dx2 = torch.empty_like(x2, memory_format=torch.contiguous_format)
            K.linear_bwd_data(dx2, dpre, W[:, K1:], kscale=scale)
        if need_w or need_g:
            raw = torch.empty_like(W, memory_format=torch.contiguous_format)
            if need_b and bias_ok(dpre):
                db = torch.empty(W.shape[0], device=W.device)
            K.linear_bwd_weight(raw[:, :K1], dpre, x1, dbias=db)
            if x2 is not None:
                K.linear_bwd_weight(raw[:, K1:], dpre, x2)
            if g is not None:
                dW, dg = torch.empty_like(raw), torch.empty_like(g)
                K.wn_bwd(dW, dg, raw, W, g, norm)
            else:
                dW = raw
        if need_b and db is None:
            db = torch.empty(W.shape[0], device=W.device)
            K.colsum(db, dpre)
        return dx1, dx2, dW, db, dg, None, None


def bias_ok(dpre):
    return dpre.dim() == 2 and dpre.shape[0] > 0


def linear_act(inputs, weight, bias=None, g=None, act='identity', shift=0.0):
    """Linear (+WeightNorm) + activation over the column-concatenation of ``inputs`` without
    materialising the concat for up to two sources (src/blocks.py:161)."""
    if act not in FUSED_ACTS:
        raise ValueError('activation %r is not fused; apply it outside' % (act,))
    inputs = list(inputs)
    if len(inputs) > 2:
        inputs = [inputs[0], torch.cat(inputs[1:], 1)]
    x1 = inputs[0]
    x2 = inputs[1] if len(inputs) == 2 else None
    return _LinearAct.apply(x1, x2, weight, bias, g, act, float(shift))


class _Reparam(torch.autograd.Function):
    """z = mu + eps * std   (src/blocks.py:170-174 logvar form, :208-211 sigma form)."""

    @staticmethod
    def forward(ctx, mu, sd, eps, mode):
        mu, sd, eps = _c(mu), _c(sd), _c(eps)
        if mu.stride(0) != sd.stride(0):
            mu, sd = mu.contiguous(), sd.contiguous()
        out = torch.empty(mu.shape, device=mu.device, dtype=torch.float32)
        K.reparam_fwd(out, mu, sd, eps, mode=mode)
        ctx.mode = mode
        ctx.save_for_backward(sd, eps)
        return out

    @staticmethod
    def backward(ctx, dz):
        sd, eps = ctx.saved_tensors
        dz = _c(dz)
        dq = torch.empty(dz.shape[0], 2 * dz.shape[1], device=dz.device)
        Z = dz.shape[1]
        K.reparam_bwd(dq[:, :Z], dq[:, Z:], dz, eps, sd, mode=ctx.mode)
        return dq[:, :Z], dq[:, Z:], None, None


def reparam(mu, sd, eps, mode):
    return _Reparam.apply(mu, sd, eps, mode)


def _pair(a, b):
    """two (M,Z) tensors with a common row stride"""
    a, b = _c(a), _c(b)
    if a.stride(0) != b.stride(0):
        a, b = a.contiguous(), b.contiguous()
    return a, b


class _KLRows(torch.autograd.Function):
    """per-row KL(q||p) of diagonal Gaussians (src/blocks.py:180-182, 217-220); p is either
    per-row tensors or the scalar prior of src/blocks.py:188-190 / 226-228."""

    @staticmethod
    def forward(ctx, mu_q, sd_q, mu_p, sd_p, prior, mode):
        mu_q, sd_q = _pair(mu_q, sd_q)
        if mu_p is not None:
            mu_p, sd_p = _pair(mu_p, sd_p)
        out = torch.empty(mu_q.shape[0], device=mu_q.device)
        K.kl_rows_fwd(out, None, mu_q, sd_q, mu_p, sd_p, prior=prior, mode=mode)
        ctx.prior, ctx.mode = prior, mode
        ctx.save_for_backward(mu_q, sd_q, mu_p, sd_p)
        return out

    @staticmethod
    def backward(ctx, drow):
        mu_q, sd_q, mu_p, sd_p = ctx.saved_tensors
        M, Z = mu_q.shape
        dq = torch.empty(M, 2 * Z, device=mu_q.device)
        dp = torch.empty(M, 2 * Z, device=mu_q.device) if mu_p is not None else None
        K.kl_rows_bwd(dq[:, :Z], dq[:, Z:], dp[:, :Z] if dp is not None else None,
                      dp[:, Z:] if dp is not None else None, drow.contiguous(), None, mu_q, sd_q, mu_p, sd_p,
                      prior=ctx.prior, mode=ctx.mode)
        if dp is None:
            return dq[:, :Z], dq[:, Z:], None, None, None, None
        return dq[:, :Z], dq[:, Z:], dp[:, :Z], dp[:, Z:], None, None


def kl_rows(mu_q, sd_q, mu_p, sd_p, mode):
    return _KLRows.apply(mu_q, sd_q, mu_p, sd_p, (0.0, 0.0), mode)


def kl_rows_prior(mu_q, sd_q, prior_mu, prior_sd, mode):
    return _KLRows.apply(mu_q, sd_q, None, None, (float(prior_mu), float(prior_sd)), mode)


class _NLLRows(torch.autograd.Function):
    """per-row Gaussian log-density summed over features (src/blocks.py:195-196, 233-234)."""

    @staticmethod
    def forward(ctx, x, mu, sd, mode):
        x = _c(x)
        mu, sd = _pair(mu, sd)
        out = torch.empty(mu.shape[0], device=mu.device)
        K.nll_rows_fwd(out, x, mu, sd, mode=mode)
        ctx.mode = mode
        ctx.save_for_backward(x, mu, sd)
        return out

    @staticmethod
    def backward(ctx, drow):
        x, mu, sd = ctx.saved_tensors
        M, X = mu.shape
        d = torch.empty(M, 2 * X, device=mu.device)
        dx = torch.empty(M, X, device=mu.device) if ctx.needs_input_grad[0] else None
        K.nll_rows_bwd(d[:, :X], d[:, X:], drow.contiguous(), x, mu, sd, mode=ctx.mode, dx=dx)
        return dx, d[:, :X], d[:, X:], None


def nll_rows(x, mu, sd, mode):
    return _NLLRows.apply(x, mu, sd, mode)


class _SoftmaxClamp(torch.autograd.Function):
    """clamp(softmax(logits) | cat(1-sigmoid, sigmoid), 1e-10, 1-1e-10) (src/blocks.py:446-463)."""

    @staticmethod
    def forward(ctx, logits, sigmoid1):
        logits = _c(logits)
        M = logits.shape[0]
        Y = 2 if sigmoid1 else logits.shape[1]
        probs = torch.empty(M, Y, device=logits.device)
        K.softmax_clamp_fwd(probs, logits, sigmoid1)
        ctx.sigmoid1 = sigmoid1
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        probs, = ctx.saved_tensors
        dl = torch.empty(probs.shape[0], 1 if ctx.sigmoid1 else probs.shape[1], device=probs.device)
        K.softmax_clamp_bwd(dl, _c(dprobs), probs, ctx.sigmoid1)
        return dl, None


def softmax_clamp(logits, sigmoid1=False):
    return _SoftmaxClamp.apply(logits, sigmoid1)


class _CatTerm(torch.autograd.Function):
    """which = 'logp' (needs labels), 'kl' (needs prior, elementwise) or 'ent' (per row):
    src/blocks.py:473-474, 479-480, 476-477."""

    @staticmethod
    def forward(ctx, probs, labels, prior, which):
        probs = _c(probs)
        M, Y = probs.shape
        prior = _c(prior)
        if which == 'logp':
            out = torch.empty(M, device=probs.device)
            K.cat_terms_fwd(probs, labels=labels, logp=out)
        elif which == 'kl':
            out = torch.empty(M, Y, device=probs.device)
            K.cat_terms_fwd(probs, prior=prior, kl=out)
        else:
            out = torch.empty(M, device=probs.device)
            K.cat_terms_fwd(probs, ent=out)
        ctx.which = which
        ctx.save_for_backward(probs, labels, prior)
        return out

    @staticmethod
    def backward(ctx, g):
        probs, labels, prior = ctx.saved_tensors
        dp = torch.empty_like(probs)
        g = g.contiguous()
        if ctx.which == 'logp':
            K.cat_terms_bwd(dp, probs, labels=labels, c_logp=g)
        elif ctx.which == 'kl':
            K.cat_terms_bwd(dp, probs, prior=prior, g_kl=g)
        else:
            K.cat_terms_bwd(dp, probs, c_ent=g)
        return dp, None, None, None


def cat_logp_rows(probs, labels):
    return _CatTerm.apply(probs, labels.reshape(-1).to(torch.int32).contiguous(), None, 'logp')


def cat_kl_elem(probs, prior):
    return _CatTerm.apply(probs, None, prior.expand_as(probs).contiguous(), 'kl')


def cat_entropy_rows(probs):
    return _CatTerm.apply(probs, None, None, 'ent')


def cat_most_probable(probs):
    probs = _c(probs.detach())
    best = torch.empty(probs.shape[0], dtype=torch.int32, device=probs.device)
    K.cat_terms_fwd(probs, best=best)
    return best.long()


class _BatchNorm(torch.autograd.Function):
    """nn.BatchNorm1d(affine=True) forward / backward on the HIP kernels (``dv_bn_fwd`` / ``dv_bn_bwd``); the
    running statistics are updated in place by the forward launch when training."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps):
        x = _c(x)
        M, N = x.shape
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        mean, rstd = torch.empty(N, device=x.device), torch.empty(N, device=x.device)
        K.bn_fwd(y, x, weight, bias, mean, rstd, running_mean, running_var, eps=eps, momentum=momentum, training=training)
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        N = x.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = torch.empty(N, device=x.device), torch.empty(N, device=x.device)
        K.bn_bwd(dx, dw, db, dy, x, mean, rstd, weight, training=ctx.training)
        return dx, dw, db, None, None, None, None, None


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    return _BatchNorm.apply(x, weight, bias, running_mean, running_var, training, momentum, eps)


class _MaskScale(torch.autograd.Function):
    """y = x * mask / keep: nn.Dropout with its keep mask; the backward is the same map on dy"""

    @staticmethod
    def forward(ctx, x, mask, scale):
        x, mask = _c(x), _c(mask)
        y = torch.empty_like(x)
        K.mask_scale(y, x, mask, scale)
        ctx.save_for_backward(mask)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        K.mask_scale(dx, dy, mask, ctx.scale)
        return dx, None, None


def dropout(x, mask, keep):
    return _MaskScale.apply(x, mask, 1.0 / keep)
