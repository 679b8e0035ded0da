"""Oracle, model level: fp32 CPU restatement of the ELBO loss assembly and train step
of reference ``src/DrVAE.py``, ``src/PVAE.py``, ``src/VFAE.py`` and
``src/DGMMixin.py``.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Differences in *form* (not arithmetic) from the reference:
  * parameters live in a dict keyed by the reference's ``state_dict`` names;
  * noise is explicit.  The reference draws ``eps`` from the global CPU generator in
    a data-dependent order (per data group: [x1 noise] [x2 noise] then per MC
    sample: z1, [z2], z2Fz1, then one z3 draw per labeled row / one per class for
    unlabeled rows -- DrVAE.py:404-433, 341-343, 518-524).  Here every draw is
    addressed by (kind, l, class, *global row*), see ``make_noise``; the golden
    generator replays exactly these arrays into the reference in its draw order,
    so both sides see identical noise and the result is independent of how rows
    are grouped or sharded.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import blocks_ref as B


# ----------------------------------------------------------------------------- spec
@dataclass
class ModelSpec:
    """Shapes + hyper-parameters (ctor args of DrVAE.py:45-69 / PVAE.py:46-62 /
    VFAE.py:42-61 that reach the loss)."""
    kind: str = 'drvae'                 # 'drvae' | 'pvae' | 'vfae'
    dim_x: int = 978
    dim_y: int = 2
    dim_z1: int = 100
    dim_z3: int = 100                   # DrVAE z3 / VFAE z2 (the top latent)
    h_en_z1: List[int] = field(default_factory=lambda: [800])
    h_de_z1: List[int] = field(default_factory=lambda: [200])
    h_en_z3: List[int] = field(default_factory=lambda: [200])
    h_de_x: List[int] = field(default_factory=lambda: [600])
    h_clf: List[int] = field(default_factory=list)
    nonlin: str = 'elu'
    weight_norm: bool = False           # reference hard-codes False (DrVAE.py:79)
    L: int = 2
    learning_rate: float = 5e-4
    weight_decay: float = 0.05
    add_noise_var: float = 0.01
    yloss_rate: float = 1.0
    kl_qz2pz2_rate: float = 1.0
    pertloss_rate: float = 0.05
    anneal_perturb_rate_itermax: int = 1
    anneal_perturb_rate_offset: int = 0
    clf_z1z2: bool = True
    semi_supervised: bool = True        # VFAE only (VFAE.py:54)
    kl_min: float = 2.0                 # DrVAE.py:90
    optim_alg: str = 'adam'             # 'adam' | 'adamax' (DGMMixin.py:35-38)
    prior_y: Optional[List[float]] = None   # None = 'uniform'; else class prior of length dim_y (DrVAE.py:83-85)
    clf_1sig: bool = False              # 2 classes from ONE sigmoid output (DrVAE.py:160-163)
    type_y: str = 'discrete'            # 'discrete' | 'cont' (regression head, DrVAE.py:159-169)
    # EXTENSION -- the reference crashes with use_s=True (torch.cat([z, s_raw]), DrVAE.py:438) and
    # its MMD glue cannot run (DGMMixin.py:42-66), so there is no reference output to pin these to; the maths is
    # what the code evidently intends: one_hot(s) appended to the inputs of encoder_z1 and decoder_x
    # (DrVAE.py:134-135,179-180,400-402,436-438) and, with use_MMD, minus the MMD between the latent samples of
    # each nuisance class and the rest, per data group and Monte-Carlo sample (DrVAE.py:394-398,537-540,616,623-624)
    type_rec: str = 'diag_gaussian'     # | 'binary' | 'poisson': decoders named at DrVAE.py:124-129, absent from blocks.py
    use_s: bool = False
    dim_s: int = 2
    use_MMD: bool = False
    mmd_rate: float = 1.0
    kernel_MMD: str = 'identity'
    top_name: str = ''                  # filled in __post_init__

    def __post_init__(self):
        self.top_name = 'encoder_z2' if self.kind == 'vfae' else 'encoder_z3'


def param_shapes(spec):
    """state_dict key -> shape, in the reference's construction order
    (DrVAE.py:112-183, PVAE.py:106-153, VFAE.py:103-165)."""
    out = OrderedDict()

    def lin(prefix, n_in, n_out):
        out[prefix + '.weight'] = (n_out, n_in)
        out[prefix + '.bias'] = (n_out,)
        if spec.weight_norm:
            out[prefix + '.g'] = (n_out,)

    def mlp(prefix, n_in, hidden):
        for i, h in enumerate(hidden):
            lin('%s.model.linear%d' % (prefix, i + 1), n_in, h)
            n_in = h
        return n_in

    def gauss(prefix, n_in, hidden, n_out, second='lv'):
        n = mlp(prefix + '.nnet', n_in, hidden)
        lin(prefix + '.encoder_mu.linear_mu', n, n_out)
        lin('%s.encoder_%s.linear_%s' % (prefix, second, second), n, n_out)

    X, Y, Z1, Z3 = spec.dim_x, spec.dim_y, spec.dim_z1, spec.dim_z3
    S = spec.dim_s if spec.use_s else 0
    gauss('encoder_z1', X + S, spec.h_en_z1, Z1)
    if spec.kind in ('drvae', 'pvae'):
        out['decoder_z2Fz1.W_mu'] = (Z1, Z1)
        out['decoder_z2Fz1.bias_mu'] = (Z1,)
        out['decoder_z2Fz1.encoder_lv.linear_lv.weight'] = (Z1, Z1)   # plain Linear, blocks.py:332
        out['decoder_z2Fz1.encoder_lv.linear_lv.bias'] = (Z1,)
    if spec.kind in ('drvae', 'vfae'):
        n_clf_in = 2 * Z1 if (spec.kind == 'drvae' and spec.clf_z1z2) else Z1
        if spec.type_y == 'cont':        # DiagGaussianModule(fixed_variance=0.05**2, constrain_means=True), DrVAE.py:167-169
            gauss('encoder_y', n_clf_in, spec.h_clf, Y)
        else:
            n = mlp('encoder_y.nnet', n_clf_in, spec.h_clf)
            lin('encoder_y.decoder_p.linear_p', n, 1 if spec.clf_1sig else Y)
        gauss(spec.top_name, Z1 + Y, spec.h_en_z3, Z3)
        gauss('decoder_z1', Z3 + Y, spec.h_de_z1, Z1)
    if spec.type_rec == 'diag_gaussian':
        gauss('decoder_x', Z1 + S, spec.h_de_x, X, second='sg')
    else:
        n = mlp('decoder_x.nnet', Z1 + S, spec.h_de_x)
        lin('decoder_x.decoder_p.linear_p' if spec.type_rec == 'binary' else 'decoder_x.decoder_r.linear_r', n, X)
    return out


def init_params(spec, seed=123, as_numpy=False):
    """Deterministic parameters from a frozen legacy ``RandomState`` stream, drawn in
    ``param_shapes`` order: U(+-1/sqrt(fan_in)) for Linear weights and biases (the
    nn.Linear family), U(+-1e-4) for W_mu / bias_mu (blocks.py:338-340), U(.5,1.5) for
    WeightNorm gains so that the g/||W|| scaling is exercised."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    fan_in = 1
    for k, shp in param_shapes(spec).items():
        if k.endswith('W_mu') or k.endswith('bias_mu'):
            a = rs.uniform(-1e-4, 1e-4, shp)
        elif k.endswith('.g'):
            a = rs.uniform(0.5, 1.5, shp)
        else:
            if k.endswith('.weight'):
                fan_in = shp[1]
            a = rs.uniform(-1.0, 1.0, shp) / np.sqrt(fan_in)
        out[k] = a.astype(np.float32)
    if as_numpy:
        return out
    return OrderedDict((k, torch.tensor(v, requires_grad=True)) for k, v in out.items())


# ---------------------------------------------------------------------------- noise
def make_noise(spec, n_rows, seed=7):
    """All N(0,1) draws one loss evaluation can consume, addressed by global row:
    nx1/nx2 (B,X) input noise; ez1/ez2/ez2F (L,B,Z1) reparam noise for z1, z2 (pairs),
    z2Fz1; ez3 (L,Y,B,Z3) top-latent noise (labeled rows use class slot 0)."""
    rs = np.random.RandomState(seed)
    f = lambda *s: rs.standard_normal(s).astype(np.float32)
    L, Y = spec.L, spec.dim_y
    return {
        'nx1': f(n_rows, spec.dim_x), 'nx2': f(n_rows, spec.dim_x),
        'ez1': f(L, n_rows, spec.dim_z1), 'ez2': f(L, n_rows, spec.dim_z1),
        'ez2F': f(L, n_rows, spec.dim_z1), 'ez3': f(L, Y, n_rows, spec.dim_z3),
        'ey': f(L, n_rows, Y),          # regression head only: the y sample of unlabeled rows (DrVAE.py:528-529)
    }


def slice_noise(noise, lo, hi):
    """Rows [lo,hi) of a noise container (data-parallel shard)."""
    return {k: (v[lo:hi] if k in ('nx1', 'nx2') else v[..., lo:hi, :]) for k, v in noise.items()}


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))


# ----------------------------------------------------------------- synthetic batches
def make_batch(spec, n_rows, seed=1234, group_mod=None):
    """SURVEY.md 8(d) synthetic inputs: x1~N(0,1); x2 = x1+0.1 N(0,1) for pairs and 0
    for singletons (wrap_in_DrVAEDataset zero-imputation, DrVAE.py:924); y~Bern(.5);
    groups by ``i mod 4`` -> ls,us,lp,up (drvae) or ``i mod 2`` (pvae: s,p; vfae: l,u)."""
    rs = np.random.RandomState(seed)
    x1 = rs.standard_normal((n_rows, spec.dim_x)).astype(np.float32)
    x2 = (x1 + 0.1 * rs.standard_normal((n_rows, spec.dim_x))).astype(np.float32)
    y = rs.randint(0, spec.dim_y, (n_rows, 1)).astype(np.int64)
    i = np.arange(n_rows)
    if spec.kind == 'drvae':
        has_y = (i % 2 == 0)
        has_x2 = ((i // 2) % 2 == 1)
    elif spec.kind == 'pvae':
        has_y = np.zeros(n_rows, bool)
        has_x2 = (i % 2 == 1)
    else:
        has_y = (i % 2 == 0)
        has_x2 = np.zeros(n_rows, bool)
    if spec.type_rec == 'binary':        # 0/1 data for the Bernoulli decoder, counts for the Poisson decoder
        x1, x2 = (x1 > 0).astype(np.float32), (x2 > 0.3).astype(np.float32)
    elif spec.type_rec == 'poisson':
        x1, x2 = np.floor(np.abs(3 * x1)).astype(np.float32), np.floor(np.abs(3 * x2)).astype(np.float32)
    x2 = x2 * has_x2[:, None].astype(np.float32)
    s = rs.randint(0, spec.dim_s, (n_rows, 1)).astype(np.int64) if spec.use_s else np.zeros((n_rows, 1), np.int64)
    return {'x1': x1, 'x2': x2, 's': s, 'y': y,
            'has_x2': has_x2.astype(np.int64), 'has_y': has_y.astype(np.int64)}


# ----------------------------------------------------------------------- loss pieces
class _Acc:
    """Per-row accumulators (global row index) next to the reference's scalar sums."""

    def __init__(self, n):
        self.rows = {k: torch.zeros(n) for k in ('RECL', 'KLD', 'PERT', 'YL')}
        self.sums = {k: 0. for k in ('RECL', 'KLD', 'PERT', 'YL', 'MMD')}

    def add(self, key, idx, row_values, scale):
        """sum += row_values.sum() * scale   (reference: ``X += f(...).sum() / Lf``)"""
        self.sums[key] = self.sums[key] + row_values.sum() * scale
        self.rows[key] = self.rows[key].index_add(0, idx, (row_values * scale).detach())


def _fprop(spec, p, z1, qz1, y1hot, eps3):
    """DrVAE.py:333-365 / VFAE.py:234-266: KL(q(top|z1,y)||N(0,I)) + KL(q(z1|x)||p(z1|top,y)),
    each with free bits on the per-row sum."""
    nh3, nhd = len(spec.h_en_z3), len(spec.h_de_z1)
    q3 = B.diag_gaussian([z1, y1hot], p, spec.top_name, nh3, spec.nonlin)
    z3 = B.sample_logvar(q3[0], q3[1], eps3)
    kl = B.free_bits(B.kl_logvar_prior_rows(*q3), spec.kl_min)
    pz1 = B.diag_gaussian([z3, y1hot], p, 'decoder_z1', nhd, spec.nonlin)
    return kl + B.free_bits(B.kl_logvar_rows(qz1[0], qz1[1], pz1[0], pz1[1]), spec.kl_min)


def _beta_pert(spec, iters):
    if spec.anneal_perturb_rate_itermax > 0:
        return B.anneal_coef(iters, spec.anneal_perturb_rate_itermax, spec.anneal_perturb_rate_offset)
    return 1.


def _group_losses(spec, p, acc, idx, x1, x2, y, noise, iters, training, s=None):
    """One ``_compute_losses`` call (DrVAE.py:367-543, PVAE.py:265-409, VFAE.py:268-401)
    on the rows ``idx``; ``x2``/``y`` are None for singleton / unlabeled groups."""
    L, Lf = spec.L, 1. * spec.L
    n = idx.numel()
    pair, labeled = x2 is not None, y is not None
    has_pert = spec.kind in ('drvae', 'pvae')
    has_y = spec.kind in ('drvae', 'vfae')
    nz = lambda key, *lead: _t(noise[key])[lead][idx] if lead else _t(noise[key])[idx]
    nh1, nhx, nhc = len(spec.h_en_z1), len(spec.h_de_x), len(spec.h_clf)

    x1 = x1.clone()
    if training and spec.add_noise_var > 0.:          # DrVAE.py:404-407 (N(0,1) * add_noise_var, in place)
        x1 = x1 + nz('nx1') * spec.add_noise_var
    def rec_logp(zin, x):
        """log p(x|z) rows of the data decoder (DrVAE.py:441-442)"""
        if spec.type_rec == 'binary':
            return B.bernoulli_logp_rows(x, *B.bernoulli(zin, p, 'decoder_x', nhx, spec.nonlin))
        if spec.type_rec == 'poisson':
            return B.poisson_logp_rows(x, *B.poisson(zin, p, 'decoder_x', nhx, spec.nonlin))
        return B.logp_sigma_rows(x, *B.diag_gaussian_sigma(zin, p, 'decoder_x', nhx, spec.nonlin))

    cond = []
    if spec.use_s:                                    # one_hot(s) conditions encoder_z1 and decoder_x (extension)
        cond = [B.one_hot(s, spec.dim_s)]
        sind = [(s.reshape(-1) == k) for k in range(spec.dim_s)]
    qz1 = B.diag_gaussian([x1] + cond, p, 'encoder_z1', nh1, spec.nonlin)
    if pair:
        x2 = x2.clone()
        if training and spec.add_noise_var > 0.:      # DrVAE.py:414-417
            x2 = x2 + nz('nx2') * spec.add_noise_var
        qz2 = B.diag_gaussian([x2] + cond, p, 'encoder_z1', nh1, spec.nonlin)   # same encoder, DrVAE.py:418
    cont = spec.type_y == 'cont'
    if has_y and labeled:
        y1hot = y.float().reshape(n, -1) if cont else B.one_hot(y, spec.dim_y)      # DrVAE.py:511-515

    for l in range(L):
        z1 = B.sample_logvar(qz1[0], qz1[1], nz('ez1', l))
        if pair and has_pert:
            z2 = B.sample_logvar(qz1[0], qz1[1], nz('ez2', l))   # quirk: samples from qz1 (DrVAE.py:427)
        if has_pert:
            pz2F = B.diag_gaussian_linear([z1], p, 'decoder_z2Fz1')
            z2F = B.sample_logvar(pz2F[0], pz2F[1], nz('ez2F', l))

        acc.add('RECL', idx, rec_logp([z1] + cond, x1), 1. / Lf)          # DrVAE.py:441-442
        if spec.use_s and spec.use_MMD:                                    # DrVAE.py:537-540
            acc.sums['MMD'] = acc.sums['MMD'] + B.mmd_criterion(z1, sind, spec.kernel_MMD) / Lf
            if pair and has_pert:
                acc.sums['MMD'] = acc.sums['MMD'] + B.mmd_criterion(z2, sind, spec.kernel_MMD) / Lf

        if spec.kind == 'pvae':                                            # PVAE.py:330-339
            acc.add('KLD', idx, B.free_bits(B.kl_logvar_prior_rows(*qz1), spec.kl_min), 1. / Lf)

        if pair and has_pert:
            acc.add('RECL', idx, rec_logp([z2] + cond, x2), 1. / Lf)      # DrVAE.py:451-452
            acc.add('PERT', idx, rec_logp([z2F] + cond, x2), 1. / Lf)     # DrVAE.py:459-460
            if spec.kind == 'pvae':                                        # PVAE.py:363-372
                acc.add('KLD', idx, B.free_bits(B.kl_logvar_prior_rows(*qz2), spec.kl_min), 1. / Lf)
            klz2 = B.free_bits(B.kl_logvar_rows(qz2[0], qz2[1], pz2F[0], pz2F[1]), spec.kl_min)
            # DrVAE.py:482-487: KLD += beta_pert * (kl_qz2pz2_rate * sum / Lf)
            acc.add('KLD', idx, klz2, _beta_pert(spec, iters) * spec.kl_qz2pz2_rate / Lf)

        if not has_y:
            continue
        if spec.kind == 'drvae':
            clf_in = [z1, z2F - z1] if spec.clf_z1z2 else [z2F]           # DrVAE.py:495-498
        else:
            clf_in = [z1]                                                  # VFAE.py:325
        if cont:
            qy = B.diag_gaussian(clf_in, p, 'encoder_y', nhc, spec.nonlin, constrain_means=True,
                                 fixed_variance=0.05 ** 2)
            if labeled and spec.kind == 'vfae':      # VFAE scores the regression head by squared error (VFAE.py:351)
                acc.add('YL', idx, -((y1hot - qy[0]) ** 2).sum(1), 1. / Lf)
                kld = _fprop(spec, p, z1, qz1, y1hot, nz('ez3', l, 0))
            elif labeled:
                acc.add('YL', idx, B.logp_logvar_rows(y1hot, *qy), 1. / Lf)          # DrVAE.py:506
                kld = _fprop(spec, p, z1, qz1, y1hot, nz('ez3', l, 0))
            else:       # SGVB: sample y (DrVAE.py:527-530); the log-prior term only exists for a data prior
                ys = B.sample_logvar(qy[0], qy[1], nz('ey', l))
                kld = _fprop(spec, p, z1, qz1, ys, nz('ez3', l, 0))
            acc.add('KLD', idx, kld, 1. / Lf)
            continue
        qy = B.categorical(clf_in, p, 'encoder_y', nhc, spec.nonlin, 1 if spec.clf_1sig else spec.dim_y)
        if labeled:
            acc.add('YL', idx, B.categorical_logp_rows(y, qy), 1. / Lf)   # DrVAE.py:506
            kld = _fprop(spec, p, z1, qz1, y1hot, nz('ez3', l, 0))
        else:
            kld = 0.
            for j in range(spec.dim_y):                                    # DrVAE.py:520-524
                yj = B.one_hot(torch.full((n,), j), spec.dim_y)
                kld = kld + qy[:, j] * _fprop(spec, p, z1, qz1, yj, nz('ez3', l, j))
            if spec.prior_y is None:
                prior = torch.full((n, spec.dim_y), 1. / spec.dim_y)      # 'uniform', DrVAE.py:388-389
            else:                                                          # np.ones((N,Y)) * prior_y, DrVAE.py:389
                prior = torch.from_numpy(np.ones((n, spec.dim_y)) * np.asarray(spec.prior_y)).float()
            kld = kld + B.categorical_kl_elem(qy, prior).sum(1)           # DrVAE.py:526
        acc.add('KLD', idx, kld, 1. / Lf)                                  # DrVAE.py:534


def loss_function(spec, p, batch, noise, iters=0, training=True, counts=None):
    """``loss_function`` of DrVAE.py:545-626 / PVAE.py:411-467 / VFAE.py:403-460.

    ``counts`` = (N_total, N_pairs, N_labeled) overrides the normalisers with global
    values (data-parallel shards); default: this batch's own counts.
    Returns (losses OrderedDict of 0-d tensors, per-row dict of (B,) tensors)."""
    x1, x2, y = _t(batch['x1']), _t(batch['x2']), _t(batch['y'])
    hy = _t(batch['has_y']).bool()
    hx = _t(batch['has_x2']).bool()
    n = x1.size(0)
    acc = _Acc(n)
    if spec.kind == 'drvae':      # order ls, us, lp, up  (DrVAE.py:585-608)
        groups = [(hy & ~hx, False, True), (~hy & ~hx, False, False), (hy & hx, True, True), (~hy & hx, True, False)]
    elif spec.kind == 'pvae':     # singletons, pairs       (PVAE.py:441-453)
        groups = [(~hx, False, False), (hx, True, False)]
    elif spec.semi_supervised:    # labeled, unlabeled      (VFAE.py:421-433)
        groups = [(hy, False, True), (~hy, False, False)]
    else:                         # labeled rows only       (VFAE.py:445-450)
        groups = [(hy, False, True)]
    for mask, pair, labeled in groups:
        idx = torch.nonzero(mask).view(-1)
        if idx.numel() == 0:
            continue              # reference: warnings.warn + zero dummy losses
        _group_losses(spec, p, acc, idx, x1[idx], x2[idx] if pair else None,
                      y[idx] if labeled else None, noise, iters, training,
                      s=_t(batch['s'])[idx] if spec.use_s else None)

    n_pairs, n_lab = int(hx.sum()), int(hy.sum())
    if counts is not None:
        n_tot, n_pairs, n_lab = counts
    elif spec.kind == 'vfae' and not spec.semi_supervised:
        n_tot = n_lab
    else:
        n_tot = n
    zero = torch.zeros(())
    S = {k: (v if torch.is_tensor(v) else zero) for k, v in acc.sums.items()}
    out = OrderedDict()
    out['RECL'] = S['RECL'] / n_tot
    out['KLD'] = S['KLD'] / n_tot
    if spec.kind != 'vfae':
        out['PERT'] = S['PERT'] / max(1., n_pairs)              # DrVAE.py:614
    if spec.kind != 'pvae':
        # DrVAE.py:615 divides by max(1, N_labeled); VFAE.py:443 by Nl (same when Nl>0)
        out['YL'] = S['YL'] / max(1., n_lab)
    # the reference's use_MMD glue is unreachable; with the use_s extension: DrVAE.py:616
    out['MMD'] = S['MMD'] / n_tot if (spec.use_s and spec.use_MMD) else zero
    beta = _beta_pert(spec, iters)
    if spec.kind == 'vfae':
        out['ELBO'] = out['RECL'] - out['KLD']                  # VFAE.py:453
    else:
        out['ELBO'] = out['RECL'] + beta * spec.pertloss_rate * out['PERT'] - out['KLD']   # DrVAE.py:619
    out['CMPL'] = -out['ELBO']
    if spec.kind != 'pvae':
        out['CMPL'] = out['CMPL'] - spec.yloss_rate * out['YL']  # DrVAE.py:622
    if spec.use_s and spec.use_MMD:
        out['CMPL'] = out['CMPL'] - spec.mmd_rate * out['MMD']   # DrVAE.py:623-624
    return out, acc.rows


# ----------------------------------------------------------------------- evaluation
def eval_x_reconstruction(x, x_rec, x_rec_std=None):
    """DGMMixin.py:128-156: RMSE, variance-weighted R^2 (sklearn), mean per-row Pearson r (scipy),
    mean Gaussian log-likelihood (decoder_x.logp_perx = the sigma form, blocks.py:233-234)."""
    import scipy.stats
    import sklearn.metrics
    x_np, r_np = _t(x).numpy().astype(float), _t(x_rec).numpy().astype(float)
    out = {'rmse': float(np.sqrt(((x_np - r_np) ** 2).mean())),
           'r2': float(sklearn.metrics.r2_score(x_np, r_np, multioutput='variance_weighted')),
           'pearr': float(np.mean([scipy.stats.pearsonr(x_np[i], r_np[i])[0] for i in range(x_np.shape[0])]))}
    out['ll'] = float(B.logp_sigma_rows(_t(x), _t(x_rec), _t(x_rec_std)).mean()) if x_rec_std is not None \
        else float('nan')
    return out


# ----------------------------------------------------------------------- train step
class RefTrainer:
    """``run_on_batch(train_mode=True)`` of DGMMixin.py:91-126 with the optimizer of
    DGMMixin.py:31-40: torch.optim.Adam, coupled L2 ``weight_decay`` on every parameter."""

    def __init__(self, spec, params):
        self.spec = spec
        self.params = params
        self.iters = 0
        opt = {'adam': torch.optim.Adam, 'adamax': torch.optim.Adamax}[spec.optim_alg]
        self.opt = opt(list(params.values()), lr=spec.learning_rate, weight_decay=spec.weight_decay)

    def loss(self, batch, noise, training=True, counts=None):
        return loss_function(self.spec, self.params, batch, noise, self.iters, training, counts)

    def step(self, batch, noise, counts=None):
        self.opt.zero_grad()
        losses, rows = self.loss(batch, noise, True, counts)
        losses['CMPL'].backward()
        self.opt.step()
        self.iters += 1
        return losses, rows
