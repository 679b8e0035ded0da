"""Index lists, coefficient vectors and buffers of the fused train step for ONE batch structure (which rows are
pairs / labeled).  Built once per structure and cached by ``FusedStep``; labels, inputs and noise are data and
are refreshed in place."""
import math

import numpy as np
import torch

from . import kernels as K
from . import tuning as T
from .arena import N_LOSS
from .chain import _Chain, _pad4

LOSS_IDX = {'RECL': 0, 'KLD': 1, 'PERT': 2, 'YL': 3, 'MMD': 4, 'ELBO': 5, 'CMPL': 6}


class _Plan:
    """Index lists, coefficient vectors and buffers for one batch structure."""

    def __init__(self, eng, rows, has_x2, has_y, counts, key, universal=False):
        cfg, dev = eng.cfg, eng.dev
        self.key, self.rows = key, rows
        # universal: the structure passed in is "every row a pair, every row unlabeled" (all class slots
        # materialised); which rows really are pairs / labeled is per-batch DATA, turned into coefficient and weight
        # vectors on the device by dv_batch_masks -- one plan and one captured graph for ANY batch composition
        self.universal = bool(universal)
        L, Y, X, Z1, Z3 = cfg.L, cfg.dim_y, cfg.dim_x, cfg.dim_z1, cfg.dim_z3
        B = self.B = len(rows)
        self.pair_host = np.nonzero(has_x2)[0]
        Np = self.Np = len(self.pair_host)
        n_lab = int(has_y.sum())
        # (universal plans) normalisers handed over as data -- this rank's rows are a slice of a global batch
        self.global_counts = bool(universal and counts is not None)
        if counts is None:
            counts = (B, Np, n_lab)
        self.n_tot, self.n_pairs, self.n_lab = [float(c) for c in counts]
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32, device=dev)
        zf = lambda *s: torch.zeros(*s, device=dev)

        def mat(rws, cols):      # row stride padded to 16 B so that rows allow vector access
            return torch.zeros(rws, _pad4(cols), device=dev)[:, :cols]

        self.pair_idx = i32(self.pair_host)
        Me = B + Np
        self.o2, self.o3 = L * B, L * B + L * Np            # ZDEC block offsets (z2 | z2Fz1-of-pairs)
        Md = L * B + 2 * L * Np
        tgt = np.concatenate([np.tile(np.arange(B), L), np.tile(B + np.arange(Np), L), np.tile(B + np.arange(Np), L)])
        self.tgt = i32(tgt)
        self.pidx = i32((np.arange(L)[:, None] * B + self.pair_host[None, :]).reshape(-1))
        self.qz2_idx = i32(B + np.arange(Np))
        # stacked input source [x1 ; x2] and the row list that builds XIN in one gather
        self.XSRC = torch.zeros(2 * B, X, device=dev)
        self.xin_idx = i32(np.concatenate([np.arange(B), B + self.pair_host]))
        # q row of every sample row of ZDEC[:o3] (z1 samples, then z2 samples drawn from q(z1|x1) of the pairs)
        self.z_src_idx = i32(np.concatenate([np.tile(np.arange(B), L), np.tile(self.pair_host, L)]))
        slot = np.full(B, -1, np.int64)
        slot[self.pair_host] = np.arange(Np)
        self.pair_slot = i32(slot)
        # ZDEC row that receives the z2Fz1 sample of (l, i) (pairs only)
        self.pert_out_idx = i32(np.where(slot[None, :] >= 0, (L * B + L * Np) + np.arange(L)[:, None] * Np + slot[None, :],
                                         -1).reshape(-1))
        # CSR: q row i -> its sample rows in ZDEC[:o3]
        zrows = [[l * B + i for l in range(L)] + ([L * B + l * Np + slot[i] for l in range(L)] if slot[i] >= 0 else [])
                 for i in range(B)]
        self.zseg_ptr = i32(np.concatenate([[0], np.cumsum([len(r) for r in zrows])]))
        self.zseg_rows = i32(np.concatenate(zrows) if B else np.zeros(0))
        self.z2_ptr = i32(np.arange(Np + 1) * L)
        self.z2_rows = i32((np.arange(L)[None, :] * Np + np.arange(Np)[:, None]).reshape(-1))
        # ---- noise arena: one flat buffer, one Philox launch per step
        sizes = [Me * X, L * B * Z1, L * Np * Z1, L * B * Z1 if cfg.has_pert else 0]
        # fprop rows
        self.Mf = 0
        if cfg.has_y:
            # fprop rows per data row (per sample l): true class | every class; the regression head always
            # conditions on ONE y (the target, or a sample of q(y|.))
            nf_row = np.ones(B, np.int64) if cfg.cont else np.where(has_y, 1, Y)
            fp_ptr = np.concatenate([[0], np.cumsum(np.tile(nf_row, L))])
            self.Mf = int(fp_ptr[-1])
            fl, fi, fslot, fcls = [], [], [], []
            for l in range(L):
                for i in range(B):
                    if has_y[i] or cfg.cont:
                        fl.append(l); fi.append(i); fslot.append(0); fcls.append(0)
                    else:
                        for j in range(Y):
                            fl.append(l); fi.append(i); fslot.append(j); fcls.append(j)
            self.fp_l_host, self.fp_i_host = np.asarray(fl, np.int64), np.asarray(fi, np.int64)
            self.fp_slot_host = np.asarray(fslot, np.int64)
            self.fp_ptr = i32(fp_ptr)
            # the same CSR over ALL sample rows of ZDEC[:o3] (the z2 samples of the pairs feed no fprop row): fan-out of
            # the encoder heads' sample epilogue straight into the z1 columns of the fprop input
            self.fp_ptr_ext = i32(np.concatenate([fp_ptr, np.full(L * Np, fp_ptr[-1])]))
            self.fp_src = i32(self.fp_l_host * B + self.fp_i_host)
            self.fp_cls = i32(np.asarray(fcls, np.int64))
            self.fp_q = i32(self.fp_i_host)
            order = np.argsort(self.fp_i_host, kind='stable')
            self.q_rows = i32(order)
            self.q_ptr = i32(np.concatenate([[0], np.cumsum(np.bincount(self.fp_i_host, minlength=B))]))
            self.label_r = i32(np.zeros(L * B, np.int64))
            self._has_y_host = has_y.copy()
            self._fp_lab_host = has_y[self.fp_i_host]
            self.has_y_dev = torch.as_tensor(has_y, device=dev)
            self.fp_i_dev = torch.as_tensor(self.fp_i_host, device=dev)
            self.fp_lab_dev = torch.as_tensor(self._fp_lab_host, device=dev)
            self.fp_slot_dev = i32(self.fp_slot_host)
            self.has_y_i32, self.fp_lab_i32 = i32(has_y), i32(self._fp_lab_host)
            sizes.append(self.Mf * Z3)
            if cfg.cont:
                sizes.append(L * B * Y)                      # eps of the y samples (unlabeled rows use them)
        self.noise = zf(int(sum(sizes)))
        views, o = [], 0
        for s in sizes:
            views.append(self.noise[o:o + s])
            o += s
        self.EX = views[0].view(Me, X)
        self.E1 = views[1].view(L * B, Z1)
        self.E2 = views[2].view(L * Np, Z1)
        self.E12 = self.noise[sizes[0]:sizes[0] + sizes[1] + sizes[2]].view(L * B + L * Np, Z1)
        self.E2F = views[3].view(L * B, Z1) if cfg.has_pert else None
        self.E3 = views[4].view(self.Mf, Z3) if cfg.has_y else None
        self.EY = views[5].view(L * B, Y) if (cfg.has_y and cfg.cont) else None
        # row descriptors of the on-device draws (dv_fill_normal_rows): {offset, width, draw id, GLOBAL row}.
        # A draw is identified by what it is in the reference's order of draws (x1 / x2 input
        # noise, then per sample l: z1, z2, z2Fz1 eps, then z3 eps per (l, class)) and by the row's position in
        # the GLOBAL minibatch (``eng.row0`` + position in this rank's shard), never by where it sits in a
        # buffer: the values do not depend on grouping, stacking or the number of ranks
        grow = int(eng.row0) + np.asarray(rows, np.int64)
        desc, off = [], 0

        def seg(n_rows, width, draw, g):
            nonlocal off
            if n_rows:
                r = np.arange(n_rows, dtype=np.int64)
                desc.append(np.stack([off + r * width, np.full(n_rows, width), np.broadcast_to(draw, (n_rows,)),
                                      np.broadcast_to(g, (n_rows,))], 1))
            off += n_rows * width
        lB, lN = np.repeat(np.arange(L), B), np.repeat(np.arange(L), Np)
        seg(B, X, 0, grow)
        seg(Np, X, 1, grow[self.pair_host])
        seg(L * B, Z1, 2 + lB, np.tile(grow, L))
        seg(L * Np, Z1, 2 + L + lN, np.tile(grow[self.pair_host], L))
        if cfg.has_pert:
            seg(L * B, Z1, 2 + 2 * L + lB, np.tile(grow, L))
        if cfg.has_y:
            seg(self.Mf, Z3, 2 + 3 * L + self.fp_l_host * Y + self.fp_slot_host, grow[self.fp_i_host])
            if cfg.cont:
                seg(L * B, Y, 2 + 3 * L + L * Y + lB, np.tile(grow, L))
        assert off == self.noise.numel() and off < 2 ** 31
        self.noise_desc = i32(np.concatenate(desc) if desc else np.zeros((0, 4)))
        # ---- activations / gradients
        self.XIN = mat(Me, X)
        self.ZDEC, self.DZDEC = mat(Md, Z1), mat(Md, Z1)
        self.c_enc = _Chain(eng.L_enc, Me, dev)
        self.c_decx = _Chain(eng.L_decx, Md, dev)
        self.enc_in, self.dec_in = [self.XIN], [self.ZDEC]
        self.DZMMD = None
        if cfg.use_s:
            # one_hot(s) columns of the encoder / decoder inputs: a second input source of their first layers
            # (no concatenated copy: the GEMM reads [x | onehot] from two operands); the class of every stacked
            # row is data, refreshed per batch by ``set_s_host``
            S = cfg.dim_s
            assert len(cfg.h_en_z1) >= 1 and len(cfg.h_de_x) >= 1, 'use_s: hidden layers in encoder_z1 and decoder_x'
            self.SOHe, self.SOHd = mat(Me, S), mat(Md, S)
            self.s_enc, self.s_dec = i32(np.zeros(Me)), i32(np.zeros(Md))
            self.enc_in, self.dec_in = [self.XIN, self.SOHe], [self.ZDEC, self.SOHd]
            self._hx_host = np.asarray(has_x2).astype(bool)
            if cfg.use_MMD:
                self.DZMMD = zf(self.o3, Z1)
                self.MMDval = zf(1)
        self.DQ = zf(Me, 2 * Z1)
        self.DPX = mat(Md, 2 * X if cfg.type_rec == 'diag_gaussian' else X)
        self.NLL = zf(Md)
        # per-tile partial sums of the reconstruction rows when they come out of the decoder-heads launch itself
        # (dv_gemm_heads, DV_HEADS_NLL): row r's log-likelihood = NLLP[r].sum()
        self.NLLP = zf(Md, K.heads_tiles(X))
        # chip-filling heads (wide configuration): the NLL row pass behind the plain product also emits the heads' bias
        # gradient -- per-chunk row partials and per-row-block column sums (``kernels.nll_rows_raw_cs``)
        self.NLLC = self.NLLWS = None
        if cfg.type_rec == 'diag_gaussian' and X % 4 == 0 and Md > 0 and (T.get('nll_cs') == 2 or not eng._heads_small(self.DPX)):
            chunks, rbs = K.nll_raw_cs_shape(Md, X)
            self.NLLC, self.NLLWS = zf(Md, chunks), zf(rbs, 2 * X)
        if cfg.has_pert:
            self.c_z2F = _Chain(eng.L_z2F, L * B, dev, resid_cols=Z1)
            self.Z2F, self.D, self.DZ2F = mat(L * B, Z1), mat(L * B, Z1), mat(L * B, Z1)
            self.DP2 = zf(L * B, 2 * Z1)
            self.KLZ2, self.KLZ2raw = zf(L * Np), zf(L * Np)
            self.TQ, self.TP = zf(L * Np, 2 * Z1), zf(L * Np, 2 * Z1)
        if cfg.kind == 'pvae':
            self.KLP, self.KLPraw = zf(Me), zf(Me)
        if cfg.has_y:
            R, Mf = L * B, self.Mf
            self.c_clf = _Chain(eng.L_clf, R, dev)
            self.QY, self.DQY, self.DLOG = zf(R, Y), zf(R, Y), zf(R, 1 if cfg.clf_1sig else Y)
            # many classifier rows (wide configuration: 4096): the single-Linear head's weight gradient splits its rows over
            # workgroups through this workspace (``kernels.smalln_bwd_weight(ws=...)``)
            self.SNWS = zf(K.smalln_ws_numel(Y, 2 * Z1)) if (R >= 1024 and Y <= 8) else None
            self.ylab = zf(B, Y)                            # regression targets (type_y='cont')
            # log p(y): the uniform prior as a scalar, a class prior given as data as a device vector
            self.log_prior = math.log(1.0 / Y) if cfg.prior_y is None else \
                torch.log(torch.tensor(cfg.prior_y, dtype=torch.float64)).float().to(dev)
            self.DZ1B = mat(R, Z1)
            self.YLrow, self.KLDrow = zf(R), zf(R)
            self.c_top = _Chain(eng.L_top, Mf, dev)
            self.c_dz1 = _Chain(eng.L_dz1, Mf, dev)
            self.FPIN, self.DFPIN = mat(Mf, Z1 + Y), mat(Mf, Z1 + Y)
            self.Z3IN, self.DZ3IN = mat(Mf, Z3 + Y), mat(Mf, Z3 + Y)
            self.DQ3, self.DPZ1, self.DQFP = zf(Mf, 2 * Z3), zf(Mf, 2 * Z1), zf(Mf, 2 * Z1)
            self.KL3, self.KL3raw, self.KL1, self.KL1raw = zf(Mf), zf(Mf), zf(Mf), zf(Mf)
            self.KLFP, self.CFP = zf(max(Mf, 1)), zf(max(Mf, 1))
        # ---- per-row loss coefficients dCMPL/d(row term) (src/DrVAE.py:611-624)
        self.beta = None
        self.c_nll = zf(Md)
        self.c_nll[:self.o3] = -1.0 / (L * self.n_tot)
        if cfg.has_y:
            self.c_kld = torch.full((L * B,), 1.0 / (L * self.n_tot), device=dev)
            self.c_yl = torch.full((L * B,), -cfg.yloss_rate / (L * max(1., self.n_lab)), device=dev)
        if cfg.kind == 'pvae':
            self.c_klp = torch.full((Me,), 1.0 / self.n_tot, device=dev)
        if cfg.has_pert:
            self.c_klz2 = zf(L * Np)
        self.w_elbo = zf(3)
        self.w_cmpl = zf(N_LOSS)
        if self.universal:
            assert (Np <= B if cfg.has_pert else Np == 0) and not cfg.cont     # (Np < B: pairs-first feeds, bucketed)
            self.hx_dev = i32(np.zeros(B)) if cfg.has_pert else None      # flags / labels of an explicit batch
            self.hy_dev = i32(np.zeros(B)) if cfg.has_y else None
            self.y_dev = i32(np.zeros(B)) if cfg.has_y else None
            self.w_recl, self.w_pert, self.w_yl = zf(2 * L * B), zf(L * B), zf(L * B)
            self.beta_dev = zf(1)
            self.gcounts_dev = i32(np.zeros(2)) if self.global_counts else None    # (N_pairs, N_labeled) of an explicit batch
            if not cfg.has_y:
                self.c_yl = None
            # rows the feed guarantees to be labeled (one fprop row, the structure's own label path): a static flag
            self.one_slot = i32(has_y.astype(np.int32)) if (cfg.has_y and has_y.any()) else None
        self._cfg = cfg
        self.x1 = self.x2 = None
        self.feed = None        # graph-resident input feed (drvae_amd.data.DeviceBatcher.begin_epoch)
        self.feed_active = False   # ... and whether it is the source of the NEXT train step: explicit data
        #                            (FusedStep.set_batch, DeviceBatcher.feed) switches it off, begin_epoch on

    @property
    def live_feed(self):
        """the installed epoch feed if it is the current input source, else None (inputs come from XSRC)"""
        return self.feed if self.feed_active else None

    def set_s_host(self, sv):
        """nuisance classes of this batch's rows (``use_s`` extension): one-hot columns of the stacked encoder
        and decoder rows, and the row lists of the model-level MMD penalty (per data group and sample)"""
        cfg = self._cfg
        L, B, Np = cfg.L, self.B, self.Np
        sv = np.asarray(sv).astype(np.int64).reshape(-1)
        assert sv.shape[0] == B and sv.min() >= 0 and sv.max() < cfg.dim_s
        sp = sv[self.pair_host]
        enc = np.concatenate([sv, sp])
        dec = np.concatenate([np.tile(sv, L), np.tile(sp, L), np.tile(sp, L)])
        self.s_enc.copy_(torch.as_tensor(enc, dtype=torch.int32))
        self.s_dec.copy_(torch.as_tensor(dec, dtype=torch.int32))
        K.rows_gather(self.SOHe, None, None, onehot_cls=self.s_enc, n_classes=cfg.dim_s, width=0)
        K.rows_gather(self.SOHd, None, None, onehot_cls=self.s_dec, n_classes=cfg.dim_s, width=0)
        if cfg.use_MMD:
            # groups in the reference's order (src/DrVAE.py:585-608; src/PVAE.py:441-453; src/VFAE.py:421-433): the
            # penalty is evaluated per group and Monte-Carlo sample, on z1 and (pairs) z2 (src/DrVAE.py:537-540)
            hx = self._hx_host
            hy = self._has_y_host.astype(bool) if cfg.has_y else np.zeros(B, bool)
            if cfg.kind == 'drvae':
                masks = [hy & ~hx, ~hy & ~hx, hy & hx, ~hy & hx]
            elif cfg.kind == 'pvae':
                masks = [~hx, hx]
            else:
                masks = [hy, ~hy]
            slot = np.full(B, -1, np.int64)
            slot[self.pair_host] = np.arange(Np)
            dev = self.ZDEC.device
            self.mmd_calls = []
            for m in masks:
                idx = np.nonzero(m)[0]
                if len(idx) == 0:
                    continue
                sind = [torch.as_tensor((sv[idx] == k).astype(np.int64), device=dev) for k in range(cfg.dim_s)]
                # (rows in / rows out of every category as index lists: host knowledge, so that the step itself
                # needs no torch.nonzero -- it would synchronise, which a hipGraph capture cannot)
                pairs = [(torch.as_tensor(np.nonzero(sv[idx] == k)[0], device=dev),
                          torch.as_tensor(np.nonzero(sv[idx] != k)[0], device=dev)) for k in range(cfg.dim_s)]
                for l in range(L):
                    self.mmd_calls.append((torch.as_tensor(l * B + idx, device=dev), sind, pairs))
                    if hx[idx[0]]:
                        self.mmd_calls.append((torch.as_tensor(self.o2 + l * Np + slot[idx], device=dev), sind, pairs))
            self.mmd_sig = sv.tobytes()          # a captured step is valid for THIS composition of nuisance classes
            # the same penalty as explicit launch lists (``FusedStep._mmd_penalty``: no autograd inside the step): per call
            # and category pair the rows of both sides as ONE index list into the stacked sample rows [side 0 | side 1],
            # and the term's weight 1 / L (two categories: the first pair only; else the mean over the categories,
            # src/DGMMixin.py:42-66).  None when a side is empty (the reference then compares with one random row: the
            # step falls back to the block-level operators)
            items, ncat = [], cfg.dim_s
            for rows, _, pairs in self.mmd_calls:
                for k in range(1 if ncat == 2 else ncat):
                    i0, i1 = pairs[k]
                    if i0.numel() == 0 or i1.numel() == 0:
                        items = None
                        break
                    g = torch.cat([rows.index_select(0, i0), rows.index_select(0, i1)])
                    items.append(dict(idx=g.long(), idx32=g.to(torch.int32), n0=int(i0.numel()), n1=int(i1.numel()),
                                      w=(1.0 if ncat == 2 else 1.0 / ncat) / L))
                if items is None:
                    break
            self.mmd_items = items
            self.__dict__.pop('_mmd_bufs', None)

    def set_s_device(self, s_dev):
        """nuisance classes of this batch's rows from a DEVICE tensor (``DeviceBatcher.feed``: rows drawn on the device):
        the one-hot columns of the stacked encoder / decoder rows are rebuilt device to device, no host round trip.
        Not with the model-level MMD penalty, whose row lists are host knowledge (``set_s_host``)."""
        cfg = self._cfg
        assert not cfg.use_MMD, 'use_MMD: the penalty needs the nuisance classes on the host (set_s_host)'
        L = cfg.L
        sv = s_dev.reshape(-1).to(torch.int32)
        sp = sv.index_select(0, self.pair_idx.long()) if self.Np else sv[:0]
        self.s_enc.copy_(torch.cat([sv, sp]))
        self.s_dec.copy_(torch.cat([sv.repeat(L), sp.repeat(L), sp.repeat(L)]))
        K.rows_gather(self.SOHe, None, None, onehot_cls=self.s_enc, n_classes=cfg.dim_s, width=0)
        K.rows_gather(self.SOHd, None, None, onehot_cls=self.s_dec, n_classes=cfg.dim_s, width=0)

    def set_labels_host(self, yv):
        """class labels of this batch's rows (host ints; only the labeled rows' entries matter)"""
        cfg = self._cfg
        yv = np.asarray(yv).astype(np.int64).reshape(-1)
        lab = np.where(self._has_y_host, yv, 0)
        self.label_r.copy_(torch.as_tensor(np.tile(lab, cfg.L), dtype=torch.int32))
        if self.Mf:
            cls = np.where(self._fp_lab_host, yv[self.fp_i_host], self.fp_slot_host)
            self.fp_cls.copy_(torch.as_tensor(cls, dtype=torch.int32))
        self._refresh_onehot()

    def set_labels_device(self, y_dev):
        """same from a device tensor of B labels (device-resident input pipeline: no host sync)"""
        cfg = self._cfg
        y32 = y_dev.reshape(-1).to(torch.int32)
        self.label_r.copy_(torch.where(self.has_y_dev, y32, torch.zeros_like(y32)).repeat(cfg.L))
        if self.Mf:
            self.fp_cls.copy_(torch.where(self.fp_lab_dev, y32[self.fp_i_dev], self.fp_slot_dev))
        self._refresh_onehot()

    def _refresh_onehot(self):
        # one-hot class columns of the decoder_z1 input [z3 | onehot(y)] (src/DrVAE.py:355)
        if self._cfg.has_y and self.Mf:
            Z3, Y = self._cfg.dim_z3, self._cfg.dim_y
            K.rows_gather(self.Z3IN[:, Z3:], None, None, onehot_cls=self.fp_cls, n_classes=Y, width=0)
            if not self._cfg.cont:      # ... and of the encoder_z3 input [z1 | onehot(y)] (src/DrVAE.py:341)
                K.rows_gather(self.FPIN[:, self._cfg.dim_z1:], None, None, onehot_cls=self.fp_cls, n_classes=Y, width=0)

    def set_beta(self, beta):
        """(re)write the coefficients that depend on the perturbation annealing coefficient
        (0.01 on the very first iteration, 1.0 afterwards with the driver settings)."""
        if self.beta == beta:
            return
        cfg, L = self._cfg, self._cfg.L
        self.beta = beta
        if self.universal:       # the per-row coefficients are rewritten every step by dv_batch_masks: it reads beta here
            self.beta_dev.fill_(beta)
            self.w_elbo.copy_(torch.tensor([1.0, -1.0, beta * cfg.pertloss_rate if cfg.has_pert else 0.0]))
        elif cfg.has_pert:
            self.c_nll[self.o3:] = -beta * cfg.pertloss_rate / (L * max(1., self.n_pairs))
            self.c_klz2.fill_(beta * cfg.kl_qz2pz2_rate / (L * self.n_tot))
            self.w_elbo.copy_(torch.tensor([1.0, -1.0, beta * cfg.pertloss_rate]))
        else:
            self.w_elbo.copy_(torch.tensor([1.0, -1.0, 0.0]))
        w = [0.0] * N_LOSS
        w[LOSS_IDX['ELBO']] = -1.0
        if cfg.has_y:
            w[LOSS_IDX['YL']] = -cfg.yloss_rate
        if cfg.use_s and cfg.use_MMD:
            w[LOSS_IDX['MMD']] = -cfg.mmd_rate         # src/DrVAE.py:623-624
        self.w_cmpl.copy_(torch.tensor(w))
