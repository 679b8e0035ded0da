"""Fused ELBO train step for DrVAE / PVAE / VFAE: hand-written forward AND backward
as one static sequence of HIP kernel launches over pre-allocated buffers (no autograd,
no allocation, no host sync) -> capturable in a hipGraph and replayed per step.

Restructuring relative to the reference (arithmetic identical):
  * the 4 data groups (ls/us/lp/up, src/DrVAE.py:585-608), the L Monte-Carlo samples
    (src/DrVAE.py:421) and the per-class passes for unlabeled rows (src/DrVAE.py:520-524)
    are stacked into the GEMM M dimension: every Linear layer runs ONCE per step;
  * group membership becomes row-index lists; the loss is linear in per-row terms with
    coefficients known before the forward pass (1/(L N), beta_pert ..., src/DrVAE.py:611-624),
    so the backward starts directly from per-row coefficients;
  * both heads of a block are one GEMM with a split epilogue; bias gradients ride on the
    dW GEMM; every scatter-add is a deterministic segment sum (no atomics).

Row layout (B rows, Np pairs, L samples):
  XIN  [B + Np]            noisy x1 rows, then noisy x2 rows of the pairs    (src/DrVAE.py:404-418)
  Q    [B + Np, 2 Z1]      (mu | logvar) of q(z1|x1) and q(z2|x2)            (shared encoder, src/DrVAE.py:418)
  ZDEC [L B + 2 L Np, Z1]  z1 samples | z2 samples (from qz1: src/DrVAE.py:427) | z2Fz1 samples of pairs
  fprop rows               per (l, row): 1 row if labeled else Y rows        (src/DrVAE.py:503-526)
"""
import math
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import kernels as K
from . import tuning as T
from ._lib import GAUSS_LOGVAR, GAUSS_SIGMA
from .arena import N_LOSS, ParamArena, span
from .chain import _Chain, _Lin, _pad4
from .plan import LOSS_IDX, _Plan
from .schedule import StepSchedule, _Branch



@dataclass
class StepConfig:
    """Shapes and hyper-parameters that reach the loss (ctor args of src/DrVAE.py:45-69,
    src/PVAE.py:46-62, src/VFAE.py:42-61 plus the attributes hard-coded in their __init__)."""
    kind: str = 'drvae'
    dim_x: int = 978
    dim_y: int = 2
    dim_z1: int = 100
    dim_z3: int = 100
    h_en_z1: List[int] = field(default_factory=lambda: [800])
    h_de_z1: List[int] = field(default_factory=lambda: [200])
    h_en_z3: List[int] = field(default_factory=lambda: [200])
    h_de_x: List[int] = field(default_factory=lambda: [600])
    h_clf: List[int] = field(default_factory=list)
    nonlin: str = 'elu'
    weight_norm: bool = False
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
    semi_supervised: bool = True
    kl_min: float = 2.0
    optim_alg: str = 'adam'
    prior_y: Optional[Tuple[float, ...]] = None    # None = 'uniform' (src/DrVAE.py:386-389)
    clf_1sig: bool = False                          # two classes from one sigmoid output (src/DrVAE.py:160-163)
    type_y: str = 'discrete'                        # 'cont': regression head N(sigmoid(.), 0.05^2) (src/DrVAE.py:167-169)
    # EXTENSION (unreachable in the reference, which crashes at src/DrVAE.py:438): one_hot(s) appended
    # to the inputs of encoder_z1 and decoder_x (src/DrVAE.py:134-135,179-180), and with use_MMD the model-level MMD
    # penalty between the nuisance classes' latent samples (src/DrVAE.py:394-398,537-540,616,623-624)
    # type of p(x|z): 'diag_gaussian' (the only one the reference can build) | 'binary' | 'poisson' -- the Bernoulli /
    # Poisson decoders src/DrVAE.py:124-129 names and blocks.py never defines: one Linear head, log-likelihood rows
    # by dv_rec_nll_rows (EXTENSION)
    type_rec: str = 'diag_gaussian'
    use_s: bool = False
    dim_s: int = 2
    use_MMD: bool = False
    mmd_rate: float = 1.0
    kernel_MMD: str = 'rbf_fourier'

    @property
    def cont(self):
        return self.type_y == 'cont'

    @property
    def top_name(self):
        return 'encoder_z2' if self.kind == 'vfae' else 'encoder_z3'

    @property
    def has_pert(self):
        return self.kind in ('drvae', 'pvae')

    @property
    def has_y(self):
        return self.kind in ('drvae', 'vfae')


def anneal_coef(iter_num, iter_max=1000, iter_offset=0):
    """src/DGMMixin.py:77-89."""
    if iter_num - iter_offset > 0:
        return min(1., 0.01 + (iter_num - iter_offset) / (1. * iter_max))
    return 0.01


def param_shapes(cfg):
    """state_dict name -> shape in the reference's construction order (src/DrVAE.py:112-183,
    src/PVAE.py:106-153, src/VFAE.py:103-165)."""
    out = OrderedDict()

    def lin(prefix, n_in, n_out):
        out[prefix + '.weight'] = (n_out, n_in)
        out[prefix + '.bias'] = (n_out,)
        if cfg.weight_norm:
            out[prefix + '.g'] = (n_out,)

    def trunk(prefix, n_in, hidden):
        for i, h in enumerate(hidden, start=1):
            lin('%s.model.linear%d' % (prefix, i), n_in, h)
            n_in = h
        return n_in

    def gauss(prefix, n_in, hidden, n_out, second='lv'):
        n = trunk(prefix + '.nnet', n_in, hidden)
        lin(prefix + '.encoder_mu.linear_mu', n, n_out)
        lin('%s.encoder_%s.linear_%s' % (prefix, second, second), n, n_out)

    X, Y, Z1, Z3 = cfg.dim_x, cfg.dim_y, cfg.dim_z1, cfg.dim_z3
    S = cfg.dim_s if cfg.use_s else 0
    gauss('encoder_z1', X + S, cfg.h_en_z1, Z1)
    if cfg.has_pert:
        out['decoder_z2Fz1.W_mu'] = (Z1, Z1)
        out['decoder_z2Fz1.bias_mu'] = (Z1,)
        out['decoder_z2Fz1.encoder_lv.linear_lv.weight'] = (Z1, Z1)
        out['decoder_z2Fz1.encoder_lv.linear_lv.bias'] = (Z1,)
    if cfg.has_y:
        n_in = 2 * Z1 if (cfg.kind == 'drvae' and cfg.clf_z1z2) else Z1
        if cfg.cont:
            gauss('encoder_y', n_in, cfg.h_clf, Y)
        else:
            n = trunk('encoder_y.nnet', n_in, cfg.h_clf)
            lin('encoder_y.decoder_p.linear_p', n, 1 if cfg.clf_1sig else Y)
        gauss(cfg.top_name, Z1 + Y, cfg.h_en_z3, Z3)
        gauss('decoder_z1', Z3 + Y, cfg.h_de_z1, Z1)
    if cfg.type_rec == 'diag_gaussian':
        gauss('decoder_x', Z1 + S, cfg.h_de_x, X, second='sg')
    else:
        n = trunk('decoder_x.nnet', Z1 + S, cfg.h_de_x)
        lin(REC_HEAD[cfg.type_rec], n, X)
    return out


REC_HEAD = {'binary': 'decoder_x.decoder_p.linear_p', 'poisson': 'decoder_x.decoder_r.linear_r'}
REC_ACT = {'binary': ('sigmoid', 0.0), 'poisson': ('softplus', 1e-6)}
Y_LOGVAR_CONT = math.log(0.05 ** 2)      # fixed variance of the regression head (src/DrVAE.py:168)


def frozen_params(cfg):
    """parameters that exist in the state_dict but never receive a gradient (torch's Adam then skips
    them entirely -- no weight decay either): the log-variance head of the fixed-variance regression
    classifier (src/blocks.py:297-298)"""
    if cfg.has_y and cfg.cont:
        q = 'encoder_y.encoder_lv.linear_lv'
        return tuple([q + '.weight', q + '.bias'] + ([q + '.g'] if cfg.weight_norm else []))
    return ()


class FusedStep(StepSchedule):
    """Owns the arena views, the per-batch plan (index lists + buffers) and the launch
    sequence.  Typical use::

        eng = FusedStep(cfg, arena)
        eng.set_batch(x1, x2, y, has_x2, has_y)      # device tensors + host flags
        eng.set_noise(noise) | eng.draw_noise()
        eng.forward(); eng.backward(); eng.optimizer_step()
    """

    def __init__(self, cfg, arena, seed=12345, concurrent=True, row0=0):
        self.cfg, self.arena = cfg, arena
        # data parallelism: position of this rank's first row in the GLOBAL minibatch (the Philox draws are keyed
        # by global row); every rank uses the same ``seed``
        self.row0 = int(row0)
        self.dev = arena.device
        self.iters = 0                      # finished_training_iters (src/DGMMixin.py:124)
        # one batch-independent plan (and one captured graph) for ANY composition of pairs / labels in the batch:
        # group membership becomes device-side masks (see ``_Plan.universal``); costs the rows of the worst case
        self.universal = False
        self.universal_pair_slots = None      # (set_batch on the universal plan) only the first so many rows may be pairs
        self.universal_labeled_range = None   # ... and rows [a, b) are labeled for sure (one fprop row each)
        self.plan = None
        self._plans = {}                    # plans by batch structure (a handful of signatures in practice)
        self.max_plans = 8
        self.pinned_plans = set()     # keys the cache never drops (a feed's bucket plans: DeviceBatcher)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)       # Adam step (device side)
        self.loss_sum = torch.zeros(8, device=self.dev)      # running sums of the loss scalars over train steps
        self.rng_ctr = torch.zeros(2, dtype=torch.int32, device=self.dev)        # Philox counter (device side)
        self.seed = seed
        self.training = True
        self.fuse_bwd = False               # set by train_step/capture: forward is followed by backward
        # train-step scheduling of the classifier/fprop side chain (tuning 'sched'): 1 = graph fork/join per
        # pass, 3 = one graph fork/join per step, 5 (default) = two single-stream graphs (main chain / side
        # chain) launched on two streams per step and ordered ONLY by device flags (dv_flag_publish /
        # dv_flag_wait): no graph edges, no events.  (Also tried and dropped: an extra cross edge, 0.37 ms;
        # the side chain as a second ROOT of one graph with flags, 0.42 ms -- the executor starts it late.)
        self.sched = T.get('sched')
        self._rec = 'both'
        self.late_leaf = bool(T.get('late_leaf'))
        self.side_adam = bool(T.get('side_adam'))
        self.fold_join = bool(T.get('fold_join'))
        # the row work that consumes both heads of a block (the reparameterised samples; forward AND backward of
        # the reconstruction log-likelihood) leaves the heads' own GEMM launch (dv_gemm_heads)
        self.fuse_heads = bool(T.get('fuse_heads'))
        # the side chain's second wait rides on the row kernel behind it (dv_wait argument of dv_kl_rows_fwd)
        # instead of being a launch of its own
        self.fold_waits = bool(T.get('fold_waits'))
        # the side chain's tail (bit mask): 1 / 4 = its two publishes ride on the entry of the launch behind them
        # (classifier dW; the counter launch), 2 = the wait in front of the next step's Philox draws is a park of the
        # draw launch itself -- measured SLOWER (356 workgroups polling one flag: +10 us/step), hence off
        self.fold_tail = T.get('fold_tail')
        self.noise_ahead = False          # set by capture(): the side chain draws the NEXT step's noise behind the join
        self._noise_stale = True          # (then) the noise buffer does not hold the draws of the current Philox counter
        self._adam_n = None
        self._split_capture = False
        self._adam_gate = None
        self._side_graph = None
        self._flag_side = None
        self._after_decoder_bwd = None
        self.side_ctr = torch.zeros(1, dtype=torch.int32, device=self.dev)   # the side chain's own step count
        self.side_t = torch.ones(1, dtype=torch.int32, device=self.dev)      # ... + 1: the optimiser step it works on
        self.flags = torch.zeros(8, dtype=torch.int32, device=self.dev)
        self.sync_err = torch.zeros(14, dtype=torch.int32, device=self.dev)  # (error, ticks parked) x 7 wait sites
        self.add_noise = True               # `fit(add_noise=...)` flag of the reference (src/DrVAE.py:769)
        # classifier/fprop chain || decoder chain.  PVAE's side chain is one tiny KL kernel: a second stream
        # costs it far more than it hides (measured 0.19 ms single-stream vs 0.9 ms forked), so it runs serial
        # (models without a classifier have no second chain: a graph branch for the step's leaf work alone -- two KL row
        # launches, the loss scalars -- was measured: the main chain shrinks by 18 us, the fork/join costs as much;
        # PVAE cfg 1: 0.1742 -> 0.1763 ms)
        # (PVAE: its side chain would be the two KL row launches; as a graph branch next to the decoder they cost more
        # than they hide: cfg 1 0.167 -> 0.175 ms)
        self.branch = _Branch(self.dev, enabled=concurrent and cfg.has_y)
        self.concurrent = bool(concurrent)
        self.wbranch = _Branch(self.dev, enabled=concurrent and bool(T.get('wbranch')))   # measured slower on MI355X (third graph branch): off
        self._build_layers()

    # ------------------------------------------------------------------ layer table
    def _gauss(self, prefix, n_hidden, second, act_second='identity', shift_second=0.0):
        cfg, a = self.cfg, self.arena
        wn = cfg.weight_norm
        layers = []
        for i in range(1, n_hidden + 1):
            p = '%s.nnet.model.linear%d' % (prefix, i)
            layers.append(_Lin(a, p + '.weight', p + '.bias', p + '.g' if wn else None, act=cfg.nonlin))
        m = prefix + '.encoder_mu.linear_mu'
        s = '%s.encoder_%s.linear_%s' % (prefix, second, second)
        layers.append(_Lin(a, m + '.weight', m + '.bias', m + '.g' if wn else None,
                           second=(s + '.weight', s + '.bias', s + '.g' if wn else None),
                           act='identity', act1=act_second, shift1=shift_second))
        return layers

    def _build_layers(self):
        cfg, a = self.cfg, self.arena
        wn = cfg.weight_norm
        self.L_enc = self._gauss('encoder_z1', len(cfg.h_en_z1), 'lv', shift_second=-2.0)
        if cfg.type_rec == 'diag_gaussian':
            self.L_decx = self._gauss('decoder_x', len(cfg.h_de_x), 'sg', 'softplus', 1e-3)
        else:       # Bernoulli / Poisson decoder: trunk + ONE head (probabilities / rates)
            layers = []
            for i in range(1, len(cfg.h_de_x) + 1):
                q = 'decoder_x.nnet.model.linear%d' % i
                layers.append(_Lin(a, q + '.weight', q + '.bias', q + '.g' if wn else None, act=cfg.nonlin))
            q = REC_HEAD[cfg.type_rec]
            act, shift = REC_ACT[cfg.type_rec]
            layers.append(_Lin(a, q + '.weight', q + '.bias', q + '.g' if wn else None, act=act, shift0=shift))
            self.L_decx = layers
        if cfg.has_pert:
            p = 'decoder_z2Fz1.'
            self.L_z2F = [_Lin(a, p + 'W_mu', p + 'bias_mu', None,
                               second=(p + 'encoder_lv.linear_lv.weight', p + 'encoder_lv.linear_lv.bias', None),
                               act='identity', shift1=-2.0)]
        if cfg.has_y:
            layers = []
            for i in range(1, len(cfg.h_clf) + 1):
                q = 'encoder_y.nnet.model.linear%d' % i
                layers.append(_Lin(a, q + '.weight', q + '.bias', q + '.g' if wn else None, act=cfg.nonlin))
            q = 'encoder_y.encoder_mu.linear_mu' if cfg.cont else 'encoder_y.decoder_p.linear_p'
            layers.append(_Lin(a, q + '.weight', q + '.bias', q + '.g' if wn else None,
                               act='sigmoid' if cfg.cont else 'identity'))
            self.L_clf = layers
            # single Linear with <= 8 classes: dedicated wave-per-row kernels instead of MFMA tiles
            self.clf_small = (not cfg.h_clf) and cfg.dim_y <= 8 and not wn and not cfg.clf_1sig and not cfg.cont and \
                bool(T.get('clf_small'))
            self.L_top = self._gauss(cfg.top_name, len(cfg.h_en_z3), 'lv', shift_second=-2.0)
            self.L_dz1 = self._gauss('decoder_z1', len(cfg.h_de_z1), 'lv', shift_second=-2.0)

    # ------------------------------------------------------------------------- plan
    def universal_ok(self):
        cfg = self.cfg
        return not cfg.cont and not (cfg.kind == 'vfae' and not cfg.semi_supervised) and not cfg.use_s

    def set_structure_universal(self, n_rows, n_pair_slots=None, labeled_range=None, n_tot=None):
        """Select (or build) the batch-independent plan for ``n_rows`` rows.  ``n_pair_slots`` < n_rows: only the FIRST
        so many rows of a batch have pair slots (x2 row, z2 / z2Fz1 sample rows) -- for feeds that put a batch's pairs
        first and choose the plan by the batch's number of pairs (``DeviceBatcher(mode='sampler', pair_bucket=...)``):
        the decoder then runs L*B + 2*L*n_pair_slots rows instead of 3*L*B.  ``labeled_range`` = (a, b): the feed
        guarantees that rows [a, b) of every batch are labeled -- they get ONE fprop row (their class, as in a plan
        built for the batch's structure) instead of a row per class.  ``n_tot`` (data parallelism): the
        rows are one rank's slice of a GLOBAL batch of ``n_tot`` rows; N_pairs / N_labeled of every batch then are the
        global counts, handed over as data (``set_batch(counts=...)``, the feed's ``gcounts`` table) instead of being
        counted over this slice by dv_batch_masks."""
        cfg = self.cfg
        assert self.universal_ok(), 'universal plan: discrete labels, semi-supervised models'
        nps = n_rows if (n_pair_slots is None or not cfg.has_pert) else int(n_pair_slots)
        assert 0 <= nps <= n_rows
        lab = (0, 0) if (labeled_range is None or not cfg.has_y) else (int(labeled_range[0]), int(labeled_range[1]))
        if lab[1] <= lab[0]:
            lab = (0, 0)
        assert 0 <= lab[0] <= lab[1] <= n_rows
        key = ('universal', n_rows, self.row0)
        if nps != n_rows or lab != (0, 0):
            key = key + (nps,) + (lab if lab != (0, 0) else ())
        if n_tot is not None:
            key = key + (('n_tot', int(n_tot)),)
        if self.plan is None or self.plan.key != key:
            self.plan = self._plans.get(key)
            if self.plan is None:
                ar = np.arange(n_rows)
                ones, zeros = ar < nps, np.zeros(n_rows, bool)
                self.plan = self._plans[key] = _Plan(self, ar, ones if cfg.has_pert else zeros,
                                                      (ar >= lab[0]) & (ar < lab[1]),
                                                      None if n_tot is None else (int(n_tot), 0, 0), key, universal=True)
                if cfg.has_y:
                    self.plan.set_labels_host(np.zeros(n_rows, np.int64))      # class slots: static
                self._evict_plans(key)
        return self.plan

    def _evict_plans(self, keep):
        """keep the plan cache bounded (randomly composed minibatches rarely repeat a structure; whole-set
        evaluations come in a few sizes), never dropping the plan a captured graph points into"""
        for old in list(self._plans):
            if len(self._plans) <= self.max_plans:
                break
            if old != keep and old != getattr(self, '_graph_key', None) and old not in getattr(self, '_captures', {}) \
                    and old not in self.pinned_plans:
                del self._plans[old]

    def set_structure(self, has_x2, has_y, counts=None):
        """Select (or build) the plan for a batch STRUCTURE: which rows are pairs / labeled.
        Returns (plan, rows) where ``rows`` are the participating row positions."""
        cfg = self.cfg
        has_x2 = np.asarray(has_x2.cpu() if torch.is_tensor(has_x2) else has_x2).astype(bool).reshape(-1)
        has_y = np.asarray(has_y.cpu() if torch.is_tensor(has_y) else has_y).astype(bool).reshape(-1)
        if not cfg.has_pert:
            has_x2 = np.zeros_like(has_x2)
        if not cfg.has_y:
            has_y = np.zeros_like(has_y)
        rows = np.arange(len(has_y))
        if cfg.kind == 'vfae' and not cfg.semi_supervised:
            rows = rows[has_y]               # supervised-only model ignores unlabeled rows (src/VFAE.py:445-450)
        # the plan (index lists, buffers, captured graph) depends on the group STRUCTURE only; the
        # class labels of the labeled rows are data and are refreshed in place
        key = (len(rows), has_x2[rows].tobytes(), has_y[rows].tobytes(), counts, self.row0)
        if self.plan is None or self.plan.key != key:
            self.plan = self._plans.get(key)
            if self.plan is None:
                self.plan = self._plans[key] = _Plan(self, rows, has_x2[rows], has_y[rows], counts, key)
                self._evict_plans(key)
        return self.plan, rows

    def set_batch(self, x1, x2, y, has_x2, has_y, counts=None, s=None):
        """x1,x2: (B,X) device fp32; y: (B,) or (B,1) ints (host or device); has_*: host bool/int
        arrays.  ``counts`` = (N_total, N_pairs, N_labeled) GLOBAL normalisers under data
        parallelism; default: this batch's own counts (src/DrVAE.py:611-616).
        ``s``: nuisance classes of the rows (``use_s`` extension)."""
        cfg = self.cfg
        if cfg.use_s:
            assert s is not None, 'use_s: the nuisance classes of the batch are needed'
            p = self._set_batch(x1, x2, y, has_x2, has_y, counts)
            sv = np.asarray(s.cpu() if torch.is_tensor(s) else s).astype(np.int64).reshape(-1)
            p.set_s_host(sv[p.rows])
            return p
        return self._set_batch(x1, x2, y, has_x2, has_y, counts)

    def _set_batch(self, x1, x2, y, has_x2, has_y, counts=None):
        cfg = self.cfg
        if self.universal and self.universal_ok():
            hy = np.asarray(has_y.cpu() if torch.is_tensor(has_y) else has_y).reshape(-1)
            p = self.set_structure_universal(len(hy), self.universal_pair_slots, self.universal_labeled_range,
                                             n_tot=None if counts is None else counts[0])
            p.feed_active = False
            if counts is not None:       # this rank's slice of a global batch: the GLOBAL (N_pairs, N_labeled) are data
                p.gcounts_dev.copy_(torch.tensor([int(counts[1]), int(counts[2])], dtype=torch.int32))
            p.XSRC[:p.B].copy_(x1)
            i32 = lambda a: torch.as_tensor(np.asarray(a.cpu() if torch.is_tensor(a) else a).reshape(-1).astype(np.int32))
            if cfg.has_pert:
                if x2 is None:      # no second profiles at all (e.g. a whole-set evaluation of singletons)
                    assert has_x2 is None or not np.asarray(has_x2.cpu() if torch.is_tensor(has_x2) else has_x2).any(), \
                        'has_x2 marks pairs but x2 is None'
                    p.XSRC[p.B:].zero_()
                    p.hx_dev.zero_()
                else:
                    assert p.Np == p.B or not np.asarray(i32(has_x2))[p.Np:].any(), 'a pair beyond the plan\'s pair slots'
                    p.XSRC[p.B:].copy_(x2)
                    p.hx_dev.copy_(i32(has_x2))
            if cfg.has_y:
                p.hy_dev.copy_(i32(has_y))
                p.y_dev.copy_(i32(y) if y is not None else torch.zeros(p.B, dtype=torch.int32))
                if p.one_slot is not None:       # rows with one fprop row: their class columns
                    assert np.asarray(i32(has_y))[p._has_y_host.astype(bool)].all(), 'an unlabeled row in the labeled range'
                    p.set_labels_host(np.asarray(i32(y)).astype(np.int64))
            return p
        p, rows = self.set_structure(has_x2, has_y, counts)
        p.feed_active = False       # explicit data supersedes an installed epoch feed (a captured step that
        #                             gathers from the feed then refuses to replay: ``replay`` checks the source)
        n_in = len(np.asarray(has_y.cpu() if torch.is_tensor(has_y) else has_y).reshape(-1))
        sel = torch.as_tensor(rows, device=self.dev) if len(rows) != n_in else None
        p.XSRC[:p.B].copy_(x1.index_select(0, sel) if sel is not None else x1)
        if x2 is not None and cfg.has_pert:
            p.XSRC[p.B:].copy_(x2.index_select(0, sel) if sel is not None else x2)
        if cfg.has_y and cfg.cont:
            yv = np.asarray(y.cpu() if torch.is_tensor(y) else y).astype(np.float32).reshape(n_in, -1) \
                if y is not None else np.zeros((n_in, cfg.dim_y), np.float32)
            p.ylab.copy_(torch.from_numpy(np.ascontiguousarray(yv[rows])))
        elif cfg.has_y:
            yv = np.asarray(y.cpu() if torch.is_tensor(y) else y).astype(np.int64).reshape(-1) if y is not None \
                else np.zeros(n_in, np.int64)
            p.set_labels_host(yv[rows])
        return p

    # ------------------------------------------------------------------------ noise
    def set_noise(self, noise):
        """Inject explicit N(0,1) draws addressed by global row (``oracle.models_ref.make_noise``
        layout: nx1/nx2 (B,X); ez1/ez2/ez2F (L,B,Z1); ez3 (L,Y,B,Z3))."""
        p, cfg = self.plan, self.cfg
        self._noise_stale = True
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32)
        rows, L = p.rows, cfg.L
        p.EX[:p.B].copy_(t(np.asarray(noise['nx1'])[rows]))
        if p.Np:
            pr = rows[p.pair_host]
            p.EX[p.B:].copy_(t(np.asarray(noise['nx2'])[pr]))
            p.E2.copy_(t(np.asarray(noise['ez2'])[:, pr].reshape(L * p.Np, -1)))
        p.E1.copy_(t(np.asarray(noise['ez1'])[:, rows].reshape(L * p.B, -1)))
        if cfg.has_pert:
            p.E2F.copy_(t(np.asarray(noise['ez2F'])[:, rows].reshape(L * p.B, -1)))
        if cfg.has_y and p.Mf:
            ez3 = np.asarray(noise['ez3'])
            slot = p.fp_slot_host
            if p.universal:     # a labeled row's one draw (the reference's, slot 0) belongs to its TRUE class slot
                hy, yv, i = p.hy_dev.cpu().numpy().astype(bool), p.y_dev.cpu().numpy(), p.fp_i_host
                slot = np.where(hy[i] & (slot == yv[i]), 0, slot)
            p.E3.copy_(t(ez3[p.fp_l_host, slot, rows[p.fp_i_host]]))
        if cfg.has_y and cfg.cont:
            p.EY.copy_(t(np.asarray(noise['ey'])[:, rows].reshape(L * p.B, -1)))

    def draw_noise(self, bump=True):
        """Fresh on-device N(0,1) for every draw of the step (Philox, one launch).  ``bump=False``: the
        Philox counter is advanced later, together with the step counter, by ``optimizer_step`` (one
        launch less on the train step's critical path)."""
        n = 1       # the Philox counter counts draw EVENTS (row-keyed draws: see ``_Plan.noise_desc``)
        if self._rec == 'main' and self.noise_ahead:
            self._rng_pending = n         # dual-graph step: the side chain of the PREVIOUS step has drawn them
            return
        desc = self.plan.noise_desc
        if not (self.training and self.add_noise and self.cfg.add_noise_var > 0):
            # no input noise in this pass (evaluation; ``fit(add_noise=False)``): its rows -- two thirds of the arena at
            # 978 genes -- are not drawn (whole-set evaluation of 8192 rows: 40 -> 15 us); a draw is keyed by (draw id,
            # global row), so the latent draws are the same numbers either way
            desc = desc[self.plan.B + self.plan.Np:]
        K.fill_normal_rows(self.plan.noise, desc, self.seed, self.rng_ctr)
        self._noise_stale = True          # (an eager draw: a later replay must draw for its own counter first)
        if bump:
            K.counter_add(self.rng_ctr, n)
        else:
            self._rng_pending = n

    # ---------------------------------------------------------------------- forward
    @staticmethod
    def _heads_small(dpx):
        """the paired-heads launch (32 x (32+32) tiles) is for the latency-bound sizes; once the decoder heads
        alone fill the chip with 128x128 tiles many times over (wide configuration) the plain GEMM + row pass wins"""
        return ((dpx.shape[0] + 127) // 128) * ((dpx.shape[1] + 127) // 128) < 1024

    def beta_pert(self):
        cfg = self.cfg
        if cfg.anneal_perturb_rate_itermax > 0:
            return anneal_coef(self.iters, cfg.anneal_perturb_rate_itermax, cfg.anneal_perturb_rate_offset)
        return 1.

    def forward(self):
        """The forward pass as a launch sequence; this function only schedules its phases: ``_encoder_forward`` (inputs,
        q(z1|x1), q(z2|x2), perturbation function, samples), then two independent chains -- ``_decoder_forward`` (the big
        GEMMs + NLL) and ``_side_forward`` (fprop / classifier: many small launches).  How the two chains are ordered
        depends on ``_mode()``: 5 = each chain is recorded into its own graph (``_rec`` says which one is being recorded)
        and device flags order them; 3 / 1 = one graph, fork/join per step / per pass; 0 = plain evaluation."""
        cfg, p = self.cfg, self.plan
        if not self.fuse_bwd and self.dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
            self.join_side()        # an evaluation forward reads parameters the side chain's tail may still be updating
        p.set_beta(self.beta_pert())
        B, Np, L, Z1 = p.B, p.Np, cfg.L, cfg.dim_z1
        rec = self._rec
        Z1blk = p.ZDEC[:L * B]
        if rec == 'side':           # side-chain graph: the main graph launches these; only the views are needed
            Q = p.c_enc.out[-1]
            Qmu, Qlv = Q[:, :Z1], Q[:, Z1:]
        else:
            Qmu, Qlv = self._encoder_forward(rec)
        if p.DZMMD is not None and rec != 'side':
            self._mmd_penalty()
        # ---- two independent chains from here: the classifier / fprop chain (many small launches)
        # runs on a side stream next to the decoder chain (the big GEMMs)
        mode = self._mode()
        if mode == 5:
            two = cfg.has_pert                      # flag 0: z1 samples final; flag 2: z2Fz1 samples final
            if rec == 'side' and not cfg.has_y:
                return              # (PVAE: the side chain is the step's tail only, see ``_side_graph_tail``)
            if rec == 'side':
                w0 = (self.flags[0:1], self.side_ctr, self.sync_err[2:4])
                w2 = (self.flags[2:3], self.side_ctr, self.sync_err[4:6])
                # the FIRST wait is long by design (the side graph is launched first and sits out the encoder): it
                # stays a one-thread launch -- folded into the 180-workgroup gather behind it, the polling of 180
                # workgroups for ~45 us slowed the main chain by 14 % (0.204 -> 0.233 ms).  The second wait has
                # nothing left to wait for when the side chain reaches it: it rides on the KL row kernel behind it
                K.flag_wait(*w0)
                fold = self.fold_waits and not cfg.cont
                self._side_forward(Qmu, Qlv, Z1blk, (lambda: K.flag_wait(*w2)) if two else None,
                                   mid_park=w2 if (two and fold) else None)
                return
            pub = (self.flags[2:3] if two else self.flags[0:1], self.step_dev, 1)
            if self.L_decx[0].g is not None:             # WeightNorm: the chain's first launch is not the GEMM
                K.flag_publish(pub[0], pub[1], 1)
                pub = None
        else:
            if not getattr(self, '_late_fork', False):
                self.branch.fork()
            pub = None
        self._decoder_forward(pub)
        if mode == 5:
            klz2 = (self._klz2_on_main() or not cfg.has_y) and cfg.has_pert and Np
            P2 = p.c_z2F.out[-1] if klz2 else None
            z2 = ((p.KLZ2, p.KLZ2raw, Qmu, Qlv, P2[:, :Z1], P2[:, Z1:]),
                  dict(qidx=p.qz2_idx, pidx=p.pidx, reps=L, free_bits=True, kl_min=cfg.kl_min)) if klz2 else None
            if cfg.kind == 'pvae':      # (its KL rows against the prior: the main chain's own, the backward reads their raw values)
                zp = ((p.KLP, p.KLPraw, Qmu, Qlv), dict(prior=(0.0, 0.0), free_bits=True, kl_min=cfg.kl_min))
                if klz2 and T.get('kl_pair'):                # ... next to the pairs' rows: one launch for both sets
                    K.kl_rows_fwd_pair(zp, z2)
                    z2 = None
                else:
                    K.kl_rows_fwd(*zp[0], **zp[1])
            if z2 is not None:
                K.kl_rows_fwd(*z2[0], **z2[1])
            return             # main-chain graph: the side chain lives in its own graph on the side stream
        with self.branch:
            self._side_forward(Qmu, Qlv, Z1blk)
        if mode == 3:
            return             # train step, single fork/join: the side chain runs on into its backward
        self.branch.join()
        if self.fuse_bwd:
            return             # the loss scalars are assembled on the side chain of backward()
        self._loss_scalars()

    def _encoder_forward(self, rec):
        """inputs (explicit batch or the graph-resident feed, + the per-batch masks of a universal plan), q(z1|x1) / q(z2|x2) with
        their samples, the perturbation function q(z2Fz1|z1) with its samples; returns the heads' (mu, logvar) views"""
        cfg, p = self.cfg, self.plan
        B, Np, L, Z1 = p.B, p.Np, cfg.L, cfg.dim_z1
        sigma = cfg.add_noise_var if (self.training and self.add_noise and cfg.add_noise_var > 0) else 0.0
        Z1blk = p.ZDEC[:L * B]
        # ---- inputs (+ training noise N(0,1)*add_noise_var, src/DrVAE.py:404-407,414-417): one gather
        fd = p.live_feed if (self.fuse_bwd and self.training) else None
        # (dual-graph schedule) the step's first launch is where the main chain meets the PREVIOUS step's side chain: its
        # tail (its half of the optimiser sweep, the loss scalars, this step's noise, its counters) must be through --
        # flag 3, published by the tail's last launch.  The optimiser launch used to park on that flag (4.3 us per step
        # at cfg 2, 7.6 us with the sampler feed: the tail is 44-49 us of serial launches against 35 us of main-chain
        # launches behind the join); it only needs the classifier's gradient (flag 6)
        start_park = None
        if rec == 'main' and self._tail_gated():
            start_park = (self.flags[3:4], self.step_dev, self.sync_err[12:14], 0)
        masks = None
        if p.universal:
            # which rows of THIS batch are pairs / labeled -> coefficient and weight vectors, on the device (with
            # the graph-resident feed: by one more workgroup of the feed's launch)
            gc = None
            if p.global_counts:      # data parallelism: the normalisers are the global batch's counts (table data / set_batch)
                gc = getattr(fd, 'gcounts', None) if fd is not None else p.gcounts_dev
                assert gc is not None, 'this plan normalises by global counts: the feed carries none (DeviceBatcher.bind(dp=...))'
            masks = dict(n_tot=p.n_tot, gcounts=gc, kl_rate=cfg.kl_qz2pz2_rate, pert_rate=cfg.pertloss_rate,
                         yl_rate=cfg.yloss_rate, beta=p.beta_dev, c_nll=p.c_nll, w_recl=p.w_recl,
                         hx=(fd.hx32 if fd is not None else p.hx_dev) if cfg.has_pert else None,
                         hy=(fd.hy32 if fd is not None else p.hy_dev) if cfg.has_y else None,
                         y=(fd.y32 if fd is not None else p.y_dev) if cfg.has_y else None,
                         c_klz2=p.c_klz2 if cfg.has_pert else None, c_yl=p.c_yl, w_pert=p.w_pert, w_yl=p.w_yl,
                         label=p.label_r if cfg.has_y else None, c_klp=p.c_klp if cfg.kind == 'pvae' else None,
                         Np=Np if cfg.has_pert else 0, one_slot=p.one_slot)
            if fd is None:
                K.batch_masks(B, L, **masks)
        if fd is not None:
            # batch (optimiser step - epoch base) of the epoch's index table, straight from the
            # HBM-resident dataset; also refreshes the label-dependent index buffers
            # (universal plan: dv_batch_masks has the labels; its rows with ONE fprop row get their class columns here)
            lab = cfg.has_y and not cfg.cont and (not p.universal or p.one_slot is not None)
            K.batch_feed(p.XIN, fd.x1, fd.x2, fd.y32, fd.table, fd.n_batches, self.step_dev, fd.base, park=start_park,
                         pair_rows=p.pair_idx if Np else None, noise=p.EX if sigma else None, sigma=sigma,
                         has_y=p.has_y_i32 if (lab and not p.universal) else None, L=L,
                         label_r=p.label_r if (lab and not p.universal) else None,
                         fp_i=p.fp_q if lab else None, fp_lab=p.fp_lab_i32 if lab else None,
                         fp_slot=p.fp_slot_dev if lab else None, fp_cls=p.fp_cls if (lab and p.Mf) else None,
                         onehot=p.Z3IN[:, cfg.dim_z3:] if (lab and p.Mf) else None, n_classes=cfg.dim_y,
                         onehot2=p.FPIN[:, Z1:] if (lab and p.Mf) else None,
                         yf=fd.yf if (cfg.has_y and cfg.cont) else None,
                         ylab=p.ylab if (cfg.has_y and cfg.cont) else None, masks=masks)
        else:
            K.rows_gather(p.XIN, p.XSRC, p.xin_idx, noise=p.EX if sigma else None, sigma=sigma, park=start_park)
        # ---- q(z1|x1), q(z2|x2): one pass of the shared encoder; the samples (src/blocks.py:170-174) -- z1
        # for every row and z2 for the pairs, drawn from q(z1|x1), not q(z2|x2) (quirk 1, src/DrVAE.py:427) --
        # leave the heads' launch itself (``fuse_heads``) or one launch of their own
        fuse = self.fuse_heads and self._heads_small(p.DPX)
        Z1blk = p.ZDEC[:L * B]
        if fuse:
            # (... and every z1 sample is copied into the z1 columns of its fprop rows on the way out: the side
            # chain's gather is gone)
            fp4 = self._fprop_from_heads()
            Q = p.c_enc.forward(p.enc_in, heads=dict(sample=dict(
                eps=p.E12, out=p.ZDEC[:p.o3], n_src=B, seg_ptr=p.zseg_ptr, seg_rows=p.zseg_rows,
                out4=p.FPIN[:, :Z1] if fp4 else None, out4_ptr=p.fp_ptr_ext if fp4 else None)))
            Qmu, Qlv = Q[:, :Z1], Q[:, Z1:]
        else:
            Q = p.c_enc.forward(p.enc_in)
            Qmu, Qlv = Q[:, :Z1], Q[:, Z1:]
            K.reparam_fwd(p.ZDEC[:p.o3], Qmu, Qlv, p.E12, src_idx=p.z_src_idx)
        if cfg.has_pert:
            # (dual-graph schedule) entry of this launch = the z1 samples are final: lets the side
            # chain's fprop start before the perturbation function has run
            pub1 = (self.flags[0:1], self.step_dev, 1) if rec == 'main' else None
            if fuse:
                # z2Fz1 sample, the classifier input z2Fz1 - z1, and the decoder's copy for the pairs
                p.c_z2F.forward([Z1blk], resid=Z1blk, publish=pub1, heads=dict(sample=dict(
                    eps=p.E2F, out=p.Z2F, n_src=L * B, sub=Z1blk, out2=p.D, out3=p.ZDEC if Np else None,
                    out3_idx=p.pert_out_idx if Np else None)))
            else:
                P2 = p.c_z2F.forward([Z1blk], resid=Z1blk, publish=pub1)
                K.reparam_fwd(p.Z2F, P2[:, :Z1], P2[:, Z1:], p.E2F, sub=Z1blk, out2=p.D,
                              out3=p.ZDEC if Np else None, out3_idx=p.pert_out_idx if Np else None)
        return Qmu, Qlv

    def _decoder_forward(self, pub=None):
        """p(x|z): the decoder over all stacked sample rows, then the NLL over genes -- in a train step inside the heads'
        launch, together with its gradient (``dv_gemm_heads``, NLL epilogue); ``pub``: flag published on entry of the first launch"""
        cfg, p = self.cfg, self.plan
        # ---- p(x|z): decoder over all stacked samples, then the NLL over genes
        X = cfg.dim_x
        gauss = cfg.type_rec == 'diag_gaussian'
        self._nll_fused = bool(gauss and self.fuse_bwd and self.fuse_heads and self._heads_small(p.DPX))
        self._nll_cs = False
        if self._nll_fused:    # train step: the heads' launch emits d/d(mu, pre-softplus) and the row sums' partials
            p.c_decx.forward(p.dec_in, publish=pub, heads=dict(out=p.DPX, nll=dict(
                x=p.XIN, xidx=p.tgt, coef=p.c_nll, part=p.NLLP)))
            PX = None
        else:
            # chip-filling heads in a train step: the product runs with the plain epilogue, the NLL row pass behind it
            # adds the bias and applies softplus + shift on its way (wide configuration: 10.87 -> 9.97 ms for the launch)
            raw_ok = bool(gauss and T.get('raw_heads') and p.c_decx.raw_last_ok() and p.c_decx.layers[-1].act1 == 'softplus'
                          and (not self._heads_small(p.DPX) or T.get('raw_heads') == 2))     # (2: any size -- tests)
            raw = raw_ok and self.fuse_bwd
            # an EVALUATION pass over many rows (whole-set evaluation, round 5): the heads are needed for the row terms only
            # -- plain product, finished inside the row pass (32768 x 1956 x 600: 737 -> 589 us for the product, and no
            # 256 MB of finished heads written and read back)
            raw_eval = raw_ok and not self.fuse_bwd and not self.training and bool(T.get('nll_cs'))
            PX = p.c_decx.forward(p.dec_in, publish=pub, raw_last=raw or raw_eval)
            if getattr(self, '_late_fork', False):       # (chip-filling step: the side chain starts HERE, next to the row pass below)
                self.branch.fork()
        if self._nll_fused:
            raw_eval = False
        elif not gauss:        # Bernoulli / Poisson rows (+ the gradient w.r.t. the head's pre-activation in a train step)
            K.rec_nll_rows(p.NLL, p.XIN, PX, kind=cfg.type_rec, shift=REC_ACT[cfg.type_rec][1], xidx=p.tgt,
                           coef=p.c_nll if self.fuse_bwd else None, dpre=p.DPX if self.fuse_bwd else None)
        elif raw_eval and p.NLLC is not None:
            # evaluation: the row terms from the raw heads (bias + softplus + shift applied by the pass; no gradients)
            lh = p.c_decx.layers[-1]
            K.nll_rows_raw_cs(p.NLLC, None, None, None, None, p.XIN, PX[:, :X], PX[:, X:], (lh.b[:X], lh.b[X:]), xidx=p.tgt,
                              sd_shift=lh.shift1)
            self._nll_cs = True
        elif raw_eval:         # (gene counts that are no multiple of 4 -- 978: the wave-per-row pass)
            lh = p.c_decx.layers[-1]
            K.nll_rows_fwd(p.NLL, p.XIN, PX[:, :X], PX[:, X:], mode=GAUSS_SIGMA, xidx=p.tgt, bias=(lh.b[:X], lh.b[X:]),
                           sd_shift=lh.shift1)
        elif self.fuse_bwd and raw and p.NLLC is not None and T.get('nll_cs'):
            # ... and the heads' bias gradient with it (column sums of the gradients this pass writes: no second pass over them)
            lh = p.c_decx.layers[-1]
            K.nll_rows_raw_cs(p.NLLC, p.DPX[:, :X], p.DPX[:, X:], p.NLLWS, p.c_nll, p.XIN, PX[:, :X], PX[:, X:],
                              (lh.b[:X], lh.b[X:]), xidx=p.tgt, sd_shift=lh.shift1)
            K.colsum(lh.db, p.NLLWS)
            self._nll_cs = True
        elif self.fuse_bwd:    # train step: d/d(mu, pre-softplus) emitted in the same row pass
            lh = p.c_decx.layers[-1]
            K.nll_rows_fwdbwd(p.NLL, p.DPX[:, :X], p.DPX[:, X:], p.c_nll, p.XIN, PX[:, :X], PX[:, X:], mode=GAUSS_SIGMA,
                              xidx=p.tgt, sd_act='softplus', sd_shift=lh.shift1 if raw else 1e-3,
                              bias=(lh.b[:X], lh.b[X:]) if raw else None)
        else:
            K.nll_rows_fwd(p.NLL, p.XIN, PX[:, :X], PX[:, X:], mode=GAUSS_SIGMA, xidx=p.tgt)

    def _side_forward(self, Qmu, Qlv, Z1blk, mid=None, first_park=None, mid_park=None):
        """fprop first: it only needs the z1 samples, so (dual-graph schedule) it can start before
        the perturbation function has run; ``mid`` then waits for the z2Fz1 samples.  ``first_park`` / ``mid_park``
        (dual-graph schedule): the two waits ride on the launch that follows them where that is a row kernel
        with a small grid (one launch less each); else ``mid`` / a wait launch of its own."""
        cfg, p = self.cfg, self.plan
        B, Np, L, Z1 = p.B, p.Np, cfg.L, cfg.dim_z1
        if cfg.kind == 'pvae':
            K.kl_rows_fwd(p.KLP, p.KLPraw, Qmu, Qlv, prior=(0.0, 0.0), free_bits=True, kl_min=cfg.kl_min)
        if cfg.has_y and cfg.cont:
            # regression head (src/DrVAE.py:159-169,503-530): q(y|.) first -- the fprop of an unlabeled
            # row conditions on a SAMPLE of it -- then one fprop row per classifier row
            if mid is not None:
                mid()
            if cfg.has_pert and Np:
                P2 = p.c_z2F.out[-1]
                K.kl_rows_fwd(p.KLZ2, p.KLZ2raw, Qmu, Qlv, P2[:, :Z1], P2[:, Z1:], qidx=p.qz2_idx, pidx=p.pidx,
                              reps=L, free_bits=True, kl_min=cfg.kl_min)
            Z3, Y = cfg.dim_z3, cfg.dim_y
            if cfg.kind == 'drvae':
                clf_in = [Z1blk, p.D] if cfg.clf_z1z2 else [p.Z2F]
            else:
                clf_in = [Z1blk]
            QYm = p.c_clf.forward(clf_in)                         # sigmoid-constrained means
            K.rows_gather(p.FPIN[:, :Z1], Z1blk, p.fp_src)
            K.ycont_fwd(p.YLrow, p.FPIN[:, Z1:], p.Z3IN[:, Z3:], QYm, p.ylab, p.has_y_i32, p.EY, Y_LOGVAR_CONT, B,
                        sqerr=cfg.kind == 'vfae')       # VFAE scores by squared error (src/VFAE.py:351)
            Q3 = p.c_top.forward([p.FPIN])
            K.kl_rows_fwd(p.KL3, p.KL3raw, Q3[:, :Z3], Q3[:, Z3:], prior=(0.0, 0.0), free_bits=True,
                          kl_min=cfg.kl_min, eps=p.E3, zout=p.Z3IN[:, :Z3])
            PZ1 = p.c_dz1.forward([p.Z3IN])
            K.kl_rows_fwd(p.KLDrow, p.KL1raw, Qmu, Qlv, PZ1[:, :Z1], PZ1[:, Z1:], qidx=p.fp_q, free_bits=True,
                          kl_min=cfg.kl_min, add=p.KL3)
            return
        # ---- fprop over (labeled: true class | unlabeled: every class)
        # (a parked launch polls from every workgroup: small grids only -- the C side refuses more than 512)
        if first_park is not None and not (cfg.has_y and p.Mf and (p.Mf * (Z1 + cfg.dim_y) + 255) // 256 <= 256):
            K.flag_wait(*first_park)
            first_park = None
        if cfg.has_y:
            if p.Mf:
                Z3, Y = cfg.dim_z3, cfg.dim_y
                if self._fprop_from_heads():
                    if first_park is not None:
                        K.flag_wait(*first_park)
                else:
                    K.rows_gather(p.FPIN, Z1blk, p.fp_src, onehot_cls=p.fp_cls, n_classes=Y,
                                  park=(first_park[0], first_park[1], first_park[2]) if first_park is not None else None)
                # (evaluation passes over many fprop rows -- whole-set evaluation: 24576 -- take the plain product + row pass:
                # the 32 x (16 + 16) paired-heads tiles are the latency-bound sizes' kernel; 3.33 -> 3.27 ms per pair)
                if self.fuse_heads and (self.fuse_bwd or p.Mf < 8192):
                    # the z3 sample leaves the heads' launch of q(z3|z1,y); its KL term against N(0,I) (with its
                    # own free bits) is evaluated next to the z1 term below: one launch less
                    Q3 = p.c_top.forward([p.FPIN], heads=dict(sample=dict(eps=p.E3, out=p.Z3IN[:, :Z3], n_src=p.Mf)))
                    PZ1 = p.c_dz1.forward([p.Z3IN])
                    # KLFP = max(KL(q(z1|x)||p(z1|z3,y)), kl_min) + max(KL(q(z3|.)||N(0,I)), kl_min)  (src/DrVAE.py:347,358)
                    # -- in the train step inside the classifier-head launch below (``_fprop_tail``)
                    if not self._fprop_tail():
                        K.kl_rows_fwd(p.KLFP, p.KL1raw, Qmu, Qlv, PZ1[:, :Z1], PZ1[:, Z1:], qidx=p.fp_q,
                                      free_bits=True, kl_min=cfg.kl_min, prior=(0.0, 0.0),
                                      second=(Q3[:, :Z3], Q3[:, Z3:], p.KL3raw))
                else:
                    Q3 = p.c_top.forward([p.FPIN])
                    # KL(q(z3|z1,y)||N(0,I)) with free bits + the z3 sample, one row pass
                    K.kl_rows_fwd(p.KL3, p.KL3raw, Q3[:, :Z3], Q3[:, Z3:], prior=(0.0, 0.0), free_bits=True,
                                  kl_min=cfg.kl_min, eps=p.E3, zout=p.Z3IN[:, :Z3])
                    PZ1 = p.c_dz1.forward([p.Z3IN])
                    # KLFP = max(KL(q(z1|x)||p(z1|z3,y)), kl_min) + the z3 term   (src/DrVAE.py:347,358)
                    K.kl_rows_fwd(p.KLFP, p.KL1raw, Qmu, Qlv, PZ1[:, :Z1], PZ1[:, Z1:], qidx=p.fp_q, free_bits=True,
                                  kl_min=cfg.kl_min, add=p.KL3)
        # KL(q(z2|x2)||p(z2|z1)) of the pairs: both arguments come from the main chain and its consumers are the
        # main chain's z2Fz1 backward and the loss scalars -- in the dual-graph train step the MAIN chain computes
        # it (it has time to spare in front of the join, the side chain has not); then the wait for the z2Fz1
        # samples rides on the classifier launch
        klz2_here = cfg.has_pert and Np and not self._klz2_on_main()
        clf_park = (mid_park is not None and not klz2_here and cfg.has_y and self.clf_small
                    and (L * B + 3) // 4 <= 256)
        fold_mid = mid_park is not None and klz2_here and (L * Np + 3) // 4 <= 256
        if mid is not None and not fold_mid and not clf_park:
            mid()
        if klz2_here:
            P2 = p.c_z2F.out[-1]
            K.kl_rows_fwd(p.KLZ2, p.KLZ2raw, Qmu, Qlv, P2[:, :Z1], P2[:, Z1:], qidx=p.qz2_idx, pidx=p.pidx,
                          reps=L, free_bits=True, kl_min=cfg.kl_min,
                          park=(mid_park[0], mid_park[1], mid_park[2]) if fold_mid else None)
        # ---- q(y|.)
        if cfg.has_y:
            if cfg.kind == 'drvae':
                clf_in = [Z1blk, p.D] if cfg.clf_z1z2 else [p.Z2F]
            else:
                clf_in = [Z1blk]
            ym = (p.YLrow, p.KLDrow, p.CFP, p.DQY, p.label_r, p.fp_ptr, p.KLFP, p.log_prior, p.c_kld, p.c_yl)
            if self.clf_small:
                lc = self.L_clf[0]
                # train step: the y-marginalisation (forward and backward) rides on the classifier's launch
                fk = None
                if self._fprop_tail():
                    # the fprop rows' KL terms in front of the y-marginalisation and the backward of the z1 term
                    # (with the coefficients it has just produced) behind it: same launch
                    fk = dict(Q=p.c_enc.out[-1], qidx=p.fp_q, P=p.c_dz1.out[-1], Q3=p.c_top.out[-1], Z1=Z1,
                              Z3=cfg.dim_z3, kl_min=cfg.kl_min, raw1=p.KL1raw, raw3=p.KL3raw, dq=p.DQFP, dp=p.DPZ1)
                K.smalln_fwd(p.QY, None, clf_in[0], lc.W, lc.b, clf_in[1] if len(clf_in) > 1 else None,
                             ymarg=ym if self.fuse_bwd else None,
                             park=(mid_park[0], mid_park[1], mid_park[2]) if clf_park else None, fprop_kl=fk)
            else:
                K.softmax_clamp_fwd(p.QY, p.c_clf.forward(clf_in), sigmoid1=cfg.clf_1sig)
            if self.fuse_bwd and self.clf_small:
                pass
            elif self.fuse_bwd:    # train step: CFP / DQY of the backward pass come out of the same launch
                K.ymarg_fwdbwd(p.YLrow, p.KLDrow, p.CFP, p.DQY, p.QY, p.label_r, p.fp_ptr, p.KLFP, p.log_prior,
                               p.c_kld, p.c_yl)
            else:
                K.ymarg_fwd(p.YLrow, p.KLDrow, p.QY, p.label_r, p.fp_ptr, p.KLFP, p.log_prior)

    def join_side(self):
        """Order the CURRENT stream behind everything the side chain has been given so far.  After a ``replay()`` of the
        dual-graph step the side chain's tail (its half of the optimiser sweep, the loss scalars, the next step's noise) is
        awaited only by the NEXT captured step (tail gating), so whatever else touches parameters, moments or loss scalars
        on the current stream -- ``losses()``, an eager ``train_step``, an evaluation forward, ``capture()``, a checkpoint
        -- goes through here first (one event record + wait; no host sync)."""
        if getattr(self, '_side_graph', None) is not None and self.dev.type == 'cuda':
            torch.cuda.current_stream().wait_stream(self.flag_side)

    def _dual_capable(self):
        """may this model's train step run as two flag-ordered graphs?  With a classifier the side chain is the fprop /
        classifier chain and the step's tail; PVAE (no classifier) has only the tail to give it -- the decoder heads' half of
        the optimiser sweep, the loss scalars, the next step's noise (cfg 1: 20 serial launches -> 16 + a 4-launch tail)"""
        cfg = self.cfg
        if cfg.has_y:
            return self.branch.on
        return bool(cfg.kind == 'pvae' and self.concurrent and T.get('pvae_tail') and self.late_leaf and self.side_adam
                    and cfg.optim_alg == 'adam' and self.fold_join)

    def _late_ok(self):
        """the side chain carries the step's leaf work (classifier dW, heads' optimiser half, loss scalars) behind the join"""
        cfg = self.cfg
        return bool(self.late_leaf and not cfg.cont and cfg.optim_alg == 'adam'
                    and (self.clf_small if cfg.has_y else self._dual_capable()))

    def _tail_gated(self):
        """dual-graph train step (ONE pair of graphs) whose side chain runs its half of the optimiser sweep and the loss
        scalars behind the join: the optimiser launch gates on the classifier's gradient only, and the NEXT step's first
        launch waits for the tail's end"""
        cfg = self.cfg
        if not (self._mode() == 5 and self._late_ok() and not getattr(self, '_split_kind', False) and T.get('tail_gate')):
            return False
        if not self._side_adam_layout()[0]:     # (= ``side_adam`` of backward(): the tail then holds the flag-4 wait launch)
            return False
        # ... where the step's first launch is the graph-resident feed (a few dozen workgroups that can park): sampler feed
        # 0.213 -> 0.2074 ms.  With the resident batch the first launch is the input gather (856 workgroups; parked with
        # 128 it is slower by itself and waits the 4 us the optimiser launch used to wait: 0.1932 -> 0.198 ms)
        return self.plan.live_feed is not None or T.get('tail_gate') == 2

    def _side_adam_layout(self):
        """(may the side chain sweep the decoder heads' half of the arena?, first element of that half)"""
        heads = self.L_decx[-1]
        g0 = self.arena.grad.storage_offset()
        hs = min(heads.dW.storage_offset(), heads.db.storage_offset()) - g0
        ok = bool(self.side_adam and len(self.L_decx) > 1 and heads.g is None and not self.wbranch.on
                  and hs % 4 == 0 and self.arena.n_live == self.arena.n_params
                  and max(heads.dW.storage_offset() + span(heads.dW),
                          heads.db.storage_offset() + heads.db.numel()) - g0 >= self.arena.n_live - 3)
        return ok, hs

    def sync_side_counters(self):
        """the side chain's counters and its tail flag follow the step counter (after it was set from outside: a
        restored checkpoint, a capture)"""
        self.side_ctr.copy_(self.step_dev)
        self.side_t.copy_(self.step_dev + 1)
        self.flags[3:4].copy_(self.step_dev)

    def _fprop_from_heads(self):
        """the encoder heads' sample epilogue also fills the z1 columns of the fprop input (the class columns are
        written when the labels are: ``_Plan._refresh_onehot`` / ``dv_batch_feed``)"""
        cfg, p = self.cfg, self.plan
        return bool(self.fuse_heads and self._heads_small(p.DPX) and cfg.has_y and not cfg.cont and p.Mf
                    and T.get('fprop_heads'))

    def _fprop_tail(self):
        """train step: the fprop rows' KL forward and the z1 term's backward ride on the classifier-head launch
        (``dv_fprop_kl``)"""
        cfg, p = self.cfg, self.plan
        return bool(self.fuse_bwd and self.fuse_heads and self.clf_small and cfg.has_y and not cfg.cont and p.Mf
                    and T.get('fprop_tail'))

    def _klz2_on_main(self):
        """dual-graph train step: the pairs' KL(q(z2|x2)||p(z2|z1)) rows run on the main chain (tuning klz2_main=0: on
        the side chain, as in every other schedule)"""
        # (not with the batch-independent plan: its worst-case decoder rows make the main chain the longer one again,
        # the side chain parks ~13 us per step behind it -- sampler feed 0.250 -> 0.248 ms with the rows on the side chain)
        on = T.get('klz2_main')
        if on < 0:
            # (... the every-row-may-be-anything plan; the bucketed ones are close to a structure plan's rows and keep the
            # rows on the main chain: sampler feed 0.2181 -> 0.2140 ms in a same-box A/B)
            p = self.plan
            on = 0 if (p is not None and p.universal and len(p.key) <= 3) else 1
        return bool(self._mode() == 5 and not self.cfg.cont and self.cfg.has_y and on)

    def _mmd_penalty(self):
        """Model-level MMD penalty of the ``use_s`` extension (src/DrVAE.py:394-398,537-540): minus the MMD between
        the latent samples of each nuisance class and the rest, per data group and Monte-Carlo sample, on z1 and
        (pairs) z2.  A cross-row term: evaluated with the block-level MMD operators (``blocks.mmd_criterion`` -> the
        HIP MMD kernels, forward and backward) on the sample rows of the stacked decoder input; value -> the loss
        tail, gradient -> ``DZMMD``, added to d/dz behind the decoder's backward pass.  Capturable: the row lists of
        every category are plan data (``_Plan.set_s_host``), the random Fourier features come from torch's
        graph-safe device generator; a captured step is valid for the composition of nuisance classes it was
        captured with (``replay`` checks)."""
        cfg, p = self.cfg, self.plan
        if cfg.kernel_MMD in ('rbf_fourier', 'identity') and getattr(p, 'mmd_items', None) and T.get('mmd_explicit'):
            return self._mmd_penalty_launches()
        from . import blocks as blk
        with torch.enable_grad():
            z = p.ZDEC[:p.o3].detach().clone().requires_grad_(True)
            total = z.new_zeros(())
            for rows, sind, pairs in p.mmd_calls:
                total = total + blk.mmd_criterion(z.index_select(0, rows), sind, cfg.kernel_MMD, pairs=pairs) / cfg.L
            total.backward()
        p.MMDval.copy_(total.detach().reshape(1))
        # CMPL = ... - mmd_rate * MMD_sum / N_total  (src/DrVAE.py:616,623-624)
        p.DZMMD.copy_(z.grad * (-cfg.mmd_rate / p.n_tot))

    def _mmd_penalty_launches(self):
        """the same penalty WITHOUT autograd (round 6): per call and category pair an explicit launch list -- gather both sides'
        sample rows, [random Fourier features: draw W ~ N(0,1), b ~ U(0,1) in the block's order; ONE projection product over both
        sides with the scale / bias epilogue; cos / column means / difference (``dv_mmd_rff_fwd``)] or [the difference of the
        column means (``dv_mmd_identity_fwd``)], value -w sqrt(mmd^2) into ``MMDval``, the gradient of that -- times the
        penalty's factor in CMPL -- back through the same kernels and added to the rows' slots of ``DZMMD``: 13 launches per
        term instead of ~38 (cfg 4 with the penalty: 0.543 -> 0.379 ms per step, same box).  Same draws in the same order as
        the block-level path, same kernels: the two agree to rounding (``tests/test_gpu_models.py``)."""
        cfg, p = self.cfg, self.plan
        Z, rff = cfg.dim_z1, cfg.kernel_MMD == 'rbf_fourier'
        R = 500                                                  # dim_r of blocks.mmd_fourier (src/blocks.py:40)
        a, c = math.sqrt(2. / 2.) / math.sqrt(Z), math.sqrt(2. / R)     # bandwidth 2 (blocks.mmd_objective)
        z = p.ZDEC[:p.o3]
        dev = z.device
        bufs = p.__dict__.get('_mmd_bufs')
        if bufs is None:
            f = lambda *shape: torch.zeros(*shape, device=dev)
            nmax = max(it['n0'] + it['n1'] for it in p.mmd_items)
            bufs = p._mmd_bufs = dict(zz=f(nmax, _pad4(Z))[:, :Z], dz=f(nmax, _pad4(Z))[:, :Z], diff=f(R if rff else Z), m2=f(1),
                                      rs=f(1), g=f(1))
            if rff:
                bufs.update(th=f(nmax, R), G=f(nmax, R), W=f(Z, R), b=f(R), bias=f(R), scale=torch.full((R,), a, device=dev))
        B_ = bufs
        p.DZMMD.zero_()
        p.MMDval.zero_()
        fac = -cfg.mmd_rate / p.n_tot                            # CMPL = ... - mmd_rate * MMD_sum / N_total
        for it in p.mmd_items:
            n0, n1 = it['n0'], it['n1']
            n = n0 + n1
            zz, dz = B_['zz'][:n], B_['dz'][:n]
            K.rows_gather(zz, z, it['idx32'])
            if rff:
                th, G = B_['th'][:n], B_['G'][:n]
                B_['W'].normal_()
                B_['b'].uniform_()
                torch.mul(B_['b'], 2 * math.pi, out=B_['bias'])
                K.gemm(th, zz, B_['W'], True, False, epi=K.EPI_FWD, scale=B_['scale'], bias=B_['bias'])
                K.mmd_rff_fwd(B_['diff'], B_['m2'], th[:n0], th[n0:], c)
            else:
                K.mmd_identity_fwd(B_['diff'], B_['m2'], zz[:n0], zz[n0:])
            torch.rsqrt(B_['m2'], out=B_['rs'])
            p.MMDval.addcmul_(B_['m2'], B_['rs'], value=-it['w'])         # - w sqrt(mmd^2)
            torch.mul(B_['rs'], -0.5 * it['w'] * fac, out=B_['g'])         # d(that, times the CMPL factor) / d(mmd^2)
            if rff:
                K.mmd_rff_bwd(G[:n0], th[:n0], B_['diff'], B_['g'], 2.0 * c / n0)
                K.mmd_rff_bwd(G[n0:], th[n0:], B_['diff'], B_['g'], -2.0 * c / n1)
                K.gemm(dz, G, B_['W'], True, True, alpha=a)
            else:
                K.mmd_identity_bwd(dz[:n0], B_['diff'], B_['g'], 2.0 / n0)
                K.mmd_identity_bwd(dz[n0:], B_['diff'], B_['g'], -2.0 / n1)
            p.DZMMD.index_add_(0, it['idx'], dz)

    def _loss_scalars(self, after=None, terms_elsewhere=False, bump_counters=False):
        """RECL, KLD, PERT, YL, ELBO, CMPL (src/DrVAE.py:611-624) as device scalars.  ``terms_elsewhere``
        (with ``after``): this launch only parks on the flag and advances the counters; the caller has another
        chain assemble the scalars (they are a leaf of the step: only the host reads them)."""
        cfg, p = self.cfg, self.plan
        L = cfg.L
        nll = p.NLLP if getattr(self, '_nll_fused', False) else (p.NLLC if getattr(self, '_nll_cs', False) else p.NLL)
        # (per-tile / per-chunk partials: a row's sum is its term)
        rl = nll.shape[1] if nll.dim() == 2 else 1
        if p.universal:      # normalisers and group masks are per-row weights written by dv_batch_masks
            terms = [(nll[:p.o3], p.w_recl[:p.o3], 1.0, 0, rl)]
            if cfg.has_pert:
                terms.append((nll[p.o3:], p.w_pert[:L * p.Np], 1.0, 2, rl))
                terms.append((p.KLZ2, p.c_klz2, 1.0, 1))
        else:
            terms = [(nll[:p.o3], None, 1.0 / (L * p.n_tot), 0)]
            if cfg.has_pert and p.Np:
                terms.append((nll[p.o3:], None, 1.0 / (L * max(1., p.n_pairs)), 2))
                terms.append((p.KLZ2, p.c_klz2, 1.0, 1))           # beta_pert*rate/(L N) lives on the device
        if cfg.kind == 'pvae':
            terms.append((p.KLP, p.c_klp, 1.0, 1) if p.universal else (p.KLP, None, 1.0 / p.n_tot, 1))
        if cfg.has_y:
            terms.append((p.KLDrow, None, 1.0 / (L * p.n_tot), 1))
            terms.append((p.YLrow, p.w_yl, 1.0, 3) if p.universal else
                         (p.YLrow, None, 1.0 / (L * max(1., p.n_lab)), 3))
        if p.DZMMD is not None:
            terms.append((p.MMDval, None, 1.0 / p.n_tot, 4))
        bump = ()
        if after is not None or bump_counters:     # this launch also advances the step / Philox counters
            bump = [(self.step_dev, 1)] + ([(self.rng_ctr, self._rng_pending)] if getattr(self, '_rng_pending', 0) else [])
            self._rng_pending = 0
            self._ctr_bumped = True
        # (train steps also add their scalars to ``loss_sum``: a ``fit`` epoch reads the sums once, at its end)
        K.loss_assemble(self.arena.loss, [] if terms_elsewhere else terms, p.w_elbo, p.w_cmpl, after=after, bump=bump,
                        halt=self.sync_err, accum=self.loss_sum if self.fuse_bwd else None)

    # --------------------------------------------------------------------- backward
    def backward(self):
        """The hand-written backward pass, scheduled like ``forward``: the side chain's share (``_side_backward``; in the
        dual-graph schedule followed by ``_side_graph_tail``) next to the main chain -- decoder, perturbation function,
        samples, encoder -- which joins the side chain's d/dz contributions where it first needs them."""
        cfg, p = self.cfg, self.plan
        B, Np, L, Z1, X = p.B, p.Np, cfg.L, cfg.dim_z1, cfg.dim_x
        Q = p.c_enc.out[-1]
        Qmu, Qlv = Q[:, :Z1], Q[:, Z1:]
        DQ = p.DQ
        Z1blk, DZ1 = p.ZDEC[:L * B], p.DZDEC[:L * B]
        # ---- side chain: y-marginalisation, fprop, classifier -> DZ1B (its share of d/dz1), DZ2F
        mode = self._mode()

        # (dual-graph schedule) the classifier's weight gradient is a leaf -- only the optimiser reads it --
        # and the side chain is the one the join waits for: it runs AFTER the side chain has published its
        # data gradients, and the optimiser launch gates that slice of the arena on a flag of its own
        # Under data parallelism every gradient (and the loss tail) must be final before the exchange: with the plain
        # two-graph split ``replay`` makes the launching stream wait for the side stream before the all-reduce, so the
        # leaf work may still move behind the join (only the optimiser half cannot: it follows the exchange); the
        # overlapped / captured exchanges keep everything in front of the join
        split_kind = getattr(self, '_split_kind', False)
        late = mode == 5 and self._late_ok() and split_kind in (False, True)
        leaf = []
        # ... and HALF of the optimiser sweep moves there too: the decoder heads (the tail of the arena, half of
        # all parameters) are final and no longer read once the heads' backward products are through -- the
        # launch after them publishes that -- so the side chain updates them next to the main chain's tail
        g0 = self.arena.grad.storage_offset()
        side_ok, hs = self._side_adam_layout()
        side_adam = late and not split_kind and side_ok
        # the gradient exchange captured INTO the step's graph (data parallelism, ``split_kind == 'captured'``): every gradient
        # and the loss scalars are final in front of the collective, so no leaf work moves behind the join -- but the side chain,
        # idle behind it, draws the NEXT step's noise (the main chain's graph then no longer starts with the draw) and the
        # sweep's first workgroup orders the next step behind that.  Measured (one-rank RCCL, same box): cfg 2 0.2042 -> 0.2007 ms,
        # cfg 4 0.1799 -> 0.1769; the heads' half of the sweep behind the collective on the side chain's 64 CUs as well
        # (a flag published on entry of the main chain's sweep): 0.2124 / 0.1871 -- half the arena through a quarter of the
        # chip's bandwidth takes longer than the whole sweep on the rest (profiles/r06_experiments.md)
        cap_fork = self._cap_fork(mode, split_kind, side_ok)
        # the loss scalars (a leaf: only the host / the exchange reads them) are assembled by the side chain behind
        # the join, once the main chain has published that its reconstruction rows are final
        side_loss = side_adam or (late and split_kind is True and len(self.L_decx) > 1 and not self.wbranch.on)

        # ---- main chain: reconstruction terms, d/d(mu, pre-softplus) straight from the per-row
        # coefficients, then back through the decoder (the three big GEMMs)
        PX = p.c_decx.out[-1]
        if not self.fuse_bwd and cfg.type_rec != 'diag_gaussian':
            K.rec_nll_rows(p.NLL, p.XIN, PX, kind=cfg.type_rec, shift=REC_ACT[cfg.type_rec][1], xidx=p.tgt, coef=p.c_nll,
                           dpre=p.DPX)
        elif not self.fuse_bwd:
            K.nll_rows_bwd(p.DPX[:, :X], p.DPX[:, X:], p.c_nll, p.XIN, PX[:, :X], PX[:, X:], mode=GAUSS_SIGMA,
                           xidx=p.tgt, sd_act='softplus', sd_shift=1e-3)
        if mode == 5 and self._rec == 'side':
            self._side_backward(Qmu, Qlv, Z1blk, mode, late, leaf)
            self._side_graph_tail(late, leaf, side_loss, side_adam, hs, cap_fork=cap_fork)
            return
        if mode < 2:
            self.branch.fork()
        elif mode == 3:
            self.branch._forked = True       # one fork/join per step: the side chain simply continues
        p.c_decx.backward(p.DPX, p.dec_in, [[(p.DZDEC, 1.0, 0.0)]] + [None] * (len(p.dec_in) - 1),
                          wbranch=self.wbranch if self.wbranch.on else None,
                          publish_after_last=(self.flags[4:5], self.step_dev, 1) if side_loss else None,
                          db_last_done=bool(self.fuse_bwd and getattr(self, '_nll_cs', False)))
        if p.DZMMD is not None:
            # model-level MMD penalty (use_s extension): its gradient w.r.t. the z1 / z2 samples was computed in
            # forward() through the block-level MMD kernels (see ``_mmd_penalty``)
            p.DZDEC[:p.o3].add_(p.DZMMD)
        if self._after_decoder_bwd is not None:
            self._after_decoder_bwd()        # decoder_x gradients are final: graph split point of the overlapped exchange
        if mode != 5:
            with self.branch:
                self._side_backward(Qmu, Qlv, Z1blk, mode, late, leaf)
            self.branch.join()
        # (only where the join does not wait: every workgroup of the consumer polls the flag, and a long wait -- VFAE:
        # its side chain is the longer one, 30 us/step -- slows the very chain it waits for: 0.184 -> 0.208 ms)
        # ... and only where the parked grid is a small fraction of what the chip holds resident (256 CUs x 8
        # workgroups): a consumer grid that filled the chip would leave the side chain nowhere to run
        fold_join = (mode == 5 and side_loss and self.fold_join and cfg.has_pert
                     and (B * Z1 + 255) // 256 <= 256)
        park = bump = None
        if fold_join:
            # no launch of its own for the join: the first consumer of the side chain's gradients (below) parks on
            # the flag itself, and the counters ride on the sample-backward launch
            park = (self.flags[1:2], self.step_dev, self.sync_err[0:2])
            bump = [(self.step_dev, 1)] + ([(self.rng_ctr, self._rng_pending)] if getattr(self, '_rng_pending', 0) else [])
            self._rng_pending = 0
            self._ctr_bumped = True
        elif mode == 5:    # the launch that assembles the loss scalars also parks on the side chain's flag
            self._loss_scalars(after=(self.flags[1:2], self.step_dev, self.sync_err[0:2], 1, K.WAIT_SPINS),
                               terms_elsewhere=side_loss)
        elif mode == 3:         # (the step / Philox counters ride on this launch: two launches less in front of the optimiser)
            self._loss_scalars(bump_counters=True)
        if mode == 5 and late and not split_kind:      # (the step counter is advanced before the optimiser launch: counter + 0 by then)
            if cfg.has_y:
                lc = self.L_clf[0]
                lo = min(lc.dW.storage_offset(), lc.db.storage_offset()) - g0
                hi = max(lc.dW.storage_offset() + span(lc.dW), lc.db.storage_offset() + lc.db.numel()) - g0
            else:       # (no classifier, no leaf gradient in flight: the first workgroup's elements stand in for the slice --
                lo, hi = 0, 4       # the gate is what orders the NEXT step behind the side chain's tail, see ``_tail_gated``)
            self._adam_gate = (self.flags[6:7] if self._tail_gated() else self.flags[3:4], self.step_dev, 0,
                               self.sync_err[6:8], lo, hi)
            self._adam_n = hs if side_adam else None
        if cap_fork:
            # (the sweep's first workgroup also parks on the side chain's "tail through" flag: what orders the NEXT step --
            # its first launch reads the noise the side chain has drawn -- behind it)
            self._adam_gate = (self.flags[3:4], self.step_dev, 0, self.sync_err[6:8], 0, 4)
        if cfg.has_pert:
            P2 = p.c_z2F.out[-1]
            # everything that hangs on the z2Fz1 samples, one launch: scatter-back of the decoded
            # copies, reparam backward, KL(q(z2|x2)||p(z2|z1)) with free bits (src/DrVAE.py:466,482-487)
            # wrt both arguments, the residual path, and the side chain's share of d/dz1
            K.z2f_post_bwd(p.DP2, DZ1, DQ[B:] if Np else None, p.DZ2F if cfg.has_y else None,     # (no classifier: no gradient into the z2Fz1 samples but the decoder's)
                           p.DZDEC[p.o3:] if Np else None, p.pair_slot,
                           p.E2F, P2, Q[B:] if Np else None, p.c_klz2, p.KLZ2raw, cfg.kl_min,
                           p.DZ1B if cfg.has_y else None, L, B, Np, park=park,
                           prior=(p.c_klp[B:], p.KLPraw[B:]) if (cfg.kind == 'pvae' and Np) else None)
            # perturbation function: mu = z1 + z1 W^T + b, logvar head
            p.c_z2F.backward(p.DP2, [Z1blk], [[(DZ1, 1.0, 1.0)]])
        # (VFAE: the side chain's share of d/dz1 is a second source of the sample backward below, no summing launch of
        # its own.  The join stays the one-workgroup launch above: VFAE's side chain is the longer one, and 59 parked
        # workgroups polling for it slow the very chain they wait for -- cfg 4 0.164 -> 0.170 ms)
        add1 = p.DZ1B if (cfg.has_y and not cfg.has_pert) else None
        # ---- back through the samples into q(z1|x1): L z1-samples (+ L z2-samples for pairs) per row,
        # plus the row-aligned KL(q(z1|x)||p(z1|z3,y)) gradients of the row's fprop rows
        fp = cfg.has_y and p.Mf
        K.reparam_bwd_seg(DQ[:B, :Z1], DQ[:B, Z1:], p.DZDEC, p.E12, Qlv, p.zseg_ptr, p.zseg_rows,
                          extra=p.DQFP if fp else None, ex_ptr=p.q_ptr if fp else None,
                          ex_rows=p.q_rows if fp else None, bump=bump, dz_add=add1,
                          # PVAE's prior term KL(q || N(0,I)) (src/PVAE.py): its gradient rides on the two launches that
                          # write q's gradient rows (here: rows [0, B); ``z2f_post_bwd``: the pairs' q(z2|x2) rows) instead
                          # of a dv_kl_rows_bwd launch behind them (cfg 1: one launch less on the critical chain)
                          prior=(p.c_klp, p.KLPraw, cfg.kl_min, Qmu) if cfg.kind == 'pvae' else None)
        p.c_enc.backward(DQ, p.enc_in, None,
                         publish_first=(self.flags[5:6], self.step_dev, 0) if ((late or cap_fork) and self.noise_ahead) else None)

    def _side_backward(self, Qmu, Qlv, Z1blk, mode, late, leaf):
        """the side chain's share of the backward pass: y-marginalisation, fprop blocks, classifier -> ``DZ1B`` (its share
        of d/dz1) and ``DZ2F``.  ``late``: the classifier's weight gradient is deferred (appended to ``leaf``: a leaf of the step
        -- only the optimiser reads it -- that runs behind the side chain's publish)"""
        cfg, p = self.cfg, self.plan
        B, Np, L, Z1 = p.B, p.Np, cfg.L, cfg.dim_z1

        def wgrad_clf(*args):
            ws = getattr(p, 'SNWS', None)       # (row-split workspace: plans with >= 1024 classifier rows)
            if late:
                # (its entry also tells the main chain that the side chain's data gradients are final: see below)
                leaf.append(lambda pub=None: K.smalln_bwd_weight(*args, publish=pub, ws=ws))
            else:
                K.smalln_bwd_weight(*args, ws=ws)

        if self.fuse_bwd and mode == 1:
            self._loss_scalars()         # leaf work, off the critical path
        if cfg.has_y and cfg.cont:
            Y, Z3 = cfg.dim_y, cfg.dim_z3
            PZ1, Q3, QYm = p.c_dz1.out[-1], p.c_top.out[-1], p.c_clf.out[-1]
            K.ycont_bwd(None, p.CFP, QYm, p.ylab, p.has_y_i32, Y_LOGVAR_CONT, p.c_yl, p.c_kld, p.DFPIN[:, Z1:],
                        p.DZ3IN[:, Z3:], B, sqerr=cfg.kind == 'vfae')     # cfp[r] = c_kld[r]: one fprop row per row
            K.kl_rows_bwd(p.DQFP[:, :Z1], p.DQFP[:, Z1:], p.DPZ1[:, :Z1], p.DPZ1[:, Z1:], p.CFP, p.KL1raw,
                          Qmu, Qlv, PZ1[:, :Z1], PZ1[:, Z1:], qidx=p.fp_q, free_bits=True, kl_min=cfg.kl_min)
            p.c_dz1.backward(p.DPZ1, [p.Z3IN], [[(p.DZ3IN, 1.0, 0.0)]])
            K.kl_rows_bwd(p.DQ3[:, :Z3], p.DQ3[:, Z3:], None, None, p.CFP, p.KL3raw, Q3[:, :Z3], Q3[:, Z3:],
                          prior=(0.0, 0.0), free_bits=True, kl_min=cfg.kl_min, dz=p.DZ3IN[:, :Z3], eps=p.E3)
            p.c_top.backward(p.DQ3, [p.FPIN], [[(p.DFPIN, 1.0, 0.0)]])
            K.rows_segment_sum(p.DZ1B, p.DFPIN, seg_ptr=p.fp_ptr, beta=0.0, width=Z1)
            # the y columns of both fprop inputs carry d/d(y sample); labeled rows: the log-likelihood
            K.ycont_bwd(p.DLOG, None, QYm, p.ylab, p.has_y_i32, Y_LOGVAR_CONT, p.c_yl, p.c_kld, p.DFPIN[:, Z1:],
                        p.DZ3IN[:, Z3:], B, sqerr=cfg.kind == 'vfae')
            if cfg.kind == 'drvae' and cfg.clf_z1z2:
                p.c_clf.backward(p.DLOG, [Z1blk, p.D], [[(p.DZ1B, 1.0, 1.0)], [(p.DZ2F, 1.0, 0.0), (p.DZ1B, -1.0, 1.0)]])
            elif cfg.kind == 'drvae':
                p.c_clf.backward(p.DLOG, [p.Z2F], [[(p.DZ2F, 1.0, 0.0)]])
            else:
                p.c_clf.backward(p.DLOG, [Z1blk], [[(p.DZ1B, 1.0, 1.0)]])
        elif cfg.has_y:
            Y = cfg.dim_y
            if not self.fuse_bwd:
                K.ymarg_bwd(p.CFP, p.DQY, p.QY, p.label_r, p.fp_ptr, p.KLFP, p.log_prior, p.c_kld, p.c_yl)
            if p.Mf:
                Z3 = cfg.dim_z3
                PZ1, Q3 = p.c_dz1.out[-1], p.c_top.out[-1]
                # KL(q(z1|x) || p(z1|z3,y)): gradient to p (decoder_z1 heads) and, row-aligned, to q
                if not self._fprop_tail():       # (else: left the classifier-head launch of the forward pass)
                    K.kl_rows_bwd(p.DQFP[:, :Z1], p.DQFP[:, Z1:], p.DPZ1[:, :Z1], p.DPZ1[:, Z1:], p.CFP, p.KL1raw,
                                  Qmu, Qlv, PZ1[:, :Z1], PZ1[:, Z1:], qidx=p.fp_q, free_bits=True,
                                  kl_min=cfg.kl_min)
                # KL(q(z3|z1,y) || N(0,I)) + the sample path: in the epilogue of the data-gradient product that produces
                # d/dz3 (``DV_EPI_KLQ``: the side chain in front of the join is a chain of dependent launches -- one less),
                # else a row pass behind it
                klq = dict(out=p.DQ3, q=Q3, eps=p.E3, coef=p.CFP, raw=p.KL3raw, kl_min=cfg.kl_min, Z=Z3) \
                    if (self.fuse_bwd and T.get('klq_epi')) else None
                if not p.c_dz1.backward(p.DPZ1, [p.Z3IN], [[(p.DZ3IN, 1.0, 0.0)]], klq=klq):
                    K.kl_rows_bwd(p.DQ3[:, :Z3], p.DQ3[:, Z3:], None, None, p.CFP, p.KL3raw, Q3[:, :Z3], Q3[:, Z3:],
                                  prior=(0.0, 0.0), free_bits=True, kl_min=cfg.kl_min, dz=p.DZ3IN[:, :Z3], eps=p.E3)
                p.c_top.backward(p.DQ3, [p.FPIN], [[(p.DFPIN, 1.0, 0.0)]])
                # z1 feeds one (labeled) or Y (unlabeled) fprop rows: their d/dz1 is summed per z1 row -- inside
                # the classifier's data-gradient launch where that launch writes DZ1B anyway, else on its own
                seg_in_clf = self.clf_small and not (cfg.kind == 'drvae' and not cfg.clf_z1z2)
                if not seg_in_clf:
                    K.rows_segment_sum(p.DZ1B, p.DFPIN, seg_ptr=p.fp_ptr, beta=0.0, width=Z1)
            # classifier
            b1 = 1.0 if p.Mf else 0.0
            seg = (p.DFPIN, p.fp_ptr) if p.Mf else None
            two = cfg.kind == 'drvae' and cfg.clf_z1z2
            if self.clf_small:
                lc = self.L_clf[0]
                if two:      # input [z1, z2F - z1]: d/dz1 gets W1 - W2, d/dz2F gets W2
                    K.smalln_bwd_data([(p.DZ1B, 0, 1.0, b1, Z1, -1.0), (p.DZ2F, Z1, 1.0, 0.0)], p.DQY, p.QY, lc.W,
                                      seg=seg)
                    wgrad_clf(lc.dW, lc.db, p.DQY, p.QY, Z1blk, p.D)
                elif cfg.kind == 'drvae':
                    K.smalln_bwd_data([(p.DZ2F, 0, 1.0, 0.0)], p.DQY, p.QY, lc.W)
                    wgrad_clf(lc.dW, lc.db, p.DQY, p.QY, p.Z2F)
                    if not p.Mf:
                        p.DZ1B.zero_()
                else:
                    K.smalln_bwd_data([(p.DZ1B, 0, 1.0, b1)], p.DQY, p.QY, lc.W, seg=seg)
                    wgrad_clf(lc.dW, lc.db, p.DQY, p.QY, Z1blk)
            else:
                K.softmax_clamp_bwd(p.DLOG, p.DQY, p.QY, sigmoid1=cfg.clf_1sig)
                if two:
                    p.c_clf.backward(p.DLOG, [Z1blk, p.D],
                                     [[(p.DZ1B, 1.0, b1)], [(p.DZ2F, 1.0, 0.0), (p.DZ1B, -1.0, 1.0)]])
                elif cfg.kind == 'drvae':
                    p.c_clf.backward(p.DLOG, [p.Z2F], [[(p.DZ2F, 1.0, 0.0)]])
                    if not p.Mf:
                        p.DZ1B.zero_()
                else:
                    p.c_clf.backward(p.DLOG, [Z1blk], [[(p.DZ1B, 1.0, b1)]])

    def _cap_fork(self, mode, split_kind, side_ok):
        """data-parallel step with the exchange captured into its graph: does the side chain, idle behind the join, draw the
        NEXT step's noise (``noise_ahead``, as in the single-GPU step)?"""
        return bool(mode == 5 and split_kind == 'captured' and side_ok and self._late_ok() and self.cfg.has_y
                    and T.get('dp_fork'))

    def _side_graph_tail(self, late, leaf, side_loss, side_adam, hs, cap_fork=False):
        """(dual-graph schedule) what the side chain's graph runs behind its backward pass: publish its data gradients, the
        deferred leaf launches, the decoder heads' half of the optimiser sweep, the loss scalars, the NEXT step's noise, its
        own counters"""
        cfg, p = self.cfg, self.plan
        if cap_fork:
            K.flag_publish(self.flags[1:2], self.side_ctr)         # DZ1B / DZ2F / every side gradient is final (the join)
            if self.noise_ahead:      # the next step's draws: this step's readers are through once the encoder backward has started
                K.flag_wait(self.flags[5:6], self.side_ctr, self.sync_err[10:12])
                K.fill_normal_rows(p.noise, p.noise_desc, self.seed, self.rng_ctr)
            K.counters_add2(self.side_ctr, 1, self.side_t, 1, publish=(self.flags[3:4], self.side_ctr, 1))
            return
        # DZ1B / DZ2F / side gradients are final: published on entry of the first leaf launch behind them (the
        # classifier's weight gradient) where there is one, else by a launch of its own
        tail = (self.fold_tail & 1) and late and len(leaf) > 0
        if tail:
            leaf[0](pub=(self.flags[1:2], self.side_ctr, 1))
            for fn in leaf[1:]:
                fn()
        else:
            K.flag_publish(self.flags[1:2], self.side_ctr)
            for fn in leaf:
                fn()
        if late:
            if side_loss:
                a = self.arena
                # (entry of this launch = the deferred leaf launches, i.e. the classifier's weight gradient, are through:
                # flag 6, what the main chain's optimiser launch gates its classifier slice on)
                K.flag_wait(self.flags[4:5], self.side_ctr, self.sync_err[8:10],
                            publish=(self.flags[6:7], self.side_ctr, 1) if self._tail_gated() else None)
                if side_adam:
                    K.adam_l2(a.param[hs:a.n_live], a.grad[hs:a.n_live], a.exp_avg[hs:a.n_live],
                              a.exp_avg_sq[hs:a.n_live], self.side_t, lr=cfg.learning_rate,
                              weight_decay=cfg.weight_decay, halt=self.sync_err)
                self._loss_scalars()   # a leaf too; the wait above also covers the main chain's NLL rows
            if self.noise_ahead:
                # the next step's N(0,1) draws: every reader of this step's is through once the encoder
                # backward has started (the main chain publishes that), and the Philox counter has advanced;
                # the draw launch parks on that flag itself (``fold_tail``) or behind a wait launch
                w5 = (self.flags[5:6], self.side_ctr, self.sync_err[10:12])
                if self.fold_tail & 2:
                    K.fill_normal_rows(p.noise, p.noise_desc, self.seed, self.rng_ctr, park=w5)
                else:
                    K.flag_wait(*w5)
                    K.fill_normal_rows(p.noise, p.noise_desc, self.seed, self.rng_ctr)
            # ... and now the side chain's late work is final: published by the counter launch on entry
            if self.fold_tail & 4:
                K.counters_add2(self.side_ctr, 1, self.side_t, 1, publish=(self.flags[3:4], self.side_ctr, 1))
                return
            K.flag_publish(self.flags[3:4], self.side_ctr)
        K.counters_add2(self.side_ctr, 1, self.side_t, 1)

    # -------------------------------------------------------------------- optimiser
    def optimizer_step(self, gscale=1.0):
        """torch.optim.Adam with coupled L2 on EVERY parameter (src/DGMMixin.py:36)."""
        cfg, a = self.cfg, self.arena
        if getattr(self, '_ctr_bumped', False):
            self._ctr_bumped = False          # the loss-scalar launch of this step already advanced them
        elif getattr(self, '_rng_pending', 0):
            K.counters_add2(self.step_dev, 1, self.rng_ctr, self._rng_pending)
            self._rng_pending = 0
        else:
            K.counter_add(self.step_dev, 1)
        if self._rec == 'both' and self.sched == 5 and self._dual_capable():
            # eager step: the side chain's counters follow, and so does the "side chain's tail is through" flag that the
            # NEXT captured step's first launch waits for (published on entry: counter + 1 = the advanced value)
            K.counters_add2(self.side_ctr, 1, self.side_t, 1, publish=(self.flags[3:4], self.side_ctr, 1))
        step = K.adamax_l2 if cfg.optim_alg == 'adamax' else K.adam_l2    # exp_avg_sq doubles as Adamax's exp_inf
        n = a.n_live                      # parameters without gradients sit behind it (untouched, like torch)
        if self._adam_n is not None:      # dual-graph step: the side chain sweeps the rest (the decoder heads)
            n, self._adam_n = self._adam_n, None
        kw = {}
        if self._adam_gate is not None:   # dual-graph step: the classifier's dW may still be in flight on the side chain
            kw['gate'], self._adam_gate = self._adam_gate, None
        step(a.param[:n], a.grad[:n], a.exp_avg[:n], a.exp_avg_sq[:n], self.step_dev, lr=cfg.learning_rate,
             weight_decay=cfg.weight_decay, gscale=gscale, halt=self.sync_err, **kw)

    def train_step(self, noise=None, allreduce=None):
        """forward + backward (+ gradient all-reduce) + Adam + iteration count: the body of
        ``run_on_batch(train_mode=True)`` (src/DGMMixin.py:91-126)."""
        self.training = True
        self.join_side()
        if noise is not None:
            self.set_noise(noise)
        else:
            self.draw_noise(bump=False)
        self.fuse_bwd = True
        try:
            self.forward()
            self.backward()
            if allreduce is not None:
                allreduce(self.arena.xchg)
            self.optimizer_step()
        finally:
            self.fuse_bwd = False
        self.iters += 1

    def losses(self):
        """OrderedDict of python floats (one device->host copy; the only sync of a step)."""
        self.join_side()         # (the loss scalars are assembled by the side chain's tail: see join_side)
        v = self.arena.loss.detach().cpu().tolist()
        if self._side_graph is not None:
            self.check_sync()
        keys = ['RECL', 'KLD', 'PERT', 'YL', 'MMD', 'ELBO', 'CMPL']
        if self.cfg.kind == 'pvae':
            keys.remove('YL')
        if self.cfg.kind == 'vfae':
            keys.remove('PERT')
        return OrderedDict((k, v[LOSS_IDX[k]]) for k in keys)
