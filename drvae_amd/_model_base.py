"""Shared wiring of the three ELBO models.  The reference spells the same skeleton out
three times (src/DrVAE.py, src/PVAE.py, src/VFAE.py); here one base class owns block
construction, inference ``forward`` and the mapping to the fused train step, and the
three public classes only carry their constructor signatures."""
import warnings
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import blocks as blk
from . import engine as E
from .DGMMixin import DeepGenerativeModelMixin
from .fit import FitMixin


class ELBOModel(FitMixin, DeepGenerativeModelMixin, nn.Module):
    kind = None             # 'drvae' | 'pvae' | 'vfae'

    def _init_common(self, args):
        """store every ctor argument as an attribute (the reference does this with inspect,
        src/DrVAE.py:71-74) and set the hyper-parameters it hard-codes (src/DrVAE.py:79-97)."""
        device = args.pop('device', None)
        weight_norm = args.pop('weight_norm', False)
        for k, v in args.items():
            setattr(self, k, v)
        self.wn = bool(weight_norm)   # reference: hard-coded False (src/DrVAE.py:79); exposed here
        self.bn = False
        self.prior_mu, self.prior_sg = 0., 1.
        if self.weight_decay is None:
            self.weight_decay = 0.
        self.kl_min = 2.
        self.anneal_learning_rate = False
        self.anneal_kl, self.anneal_kl_itermax = False, 100
        self.anneal_yloss, self.anneal_yloss_itermax = False, 1
        self.finished_training_iters = 0
        self.add_noise = False        # set by the caller / fit(add_noise=...) (src/DrVAE.py:769)
        self._check_supported()
        self.nprng = np.random.RandomState(self.random_seed)
        torch.manual_seed(self.random_seed)
        self._build_blocks()
        self._create_optimizer()
        if device is None:
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.to(device)

    def _check_supported(self):
        if self.type_rec not in ('diag_gaussian', 'binary', 'poisson'):
            raise ValueError("type_rec must be 'diag_gaussian', 'binary' or 'poisson'")      # src/DrVAE.py:124-131
        # 'binary' / 'poisson' point at blk.BernoulliDecoder / blk.PoissonDecoder, which the reference's blocks.py
        # does not define (its constructor raises AttributeError there): built here as labelled EXTENSIONS
        bad = []
        # use_s=True is an EXTENSION here: the reference crashes on it (torch.cat([z, s_raw]) of a float and an
        # integer tensor, src/DrVAE.py:438), so it is built from the evident intent -- one_hot(s) appended to the
        # inputs of encoder_z1 and decoder_x, and (use_MMD) the model-level MMD penalty -- and checked against the
        # oracle's restatement of that intent only (no reference output exists)
        if getattr(self, 'use_s', False) and (len(self.dim_h_en_z1) < 1 or len(self.dim_h_de_x) < 1):
            bad.append('use_s=True without hidden layers in encoder_z1 / decoder_x')
        if getattr(self, 'type_y', 'discrete') not in ('discrete', 'cont'):
            raise ValueError('Invalid type_y')
        if getattr(self, 'type_y', 'discrete') == 'cont' and self.kind == 'vfae' and getattr(self, 'semi_supervised', False):
            # the reference's own semi-supervised regression branch crashes (src/VFAE.py:386 passes the
            # 1-tuple returned by sample() on to torch.cat): nothing to be identical to.  Supervised-only
            # (labeled rows, squared-error y-loss, src/VFAE.py:351) runs and is supported
            bad.append("semi-supervised VFAE with type_y='cont'")
        if getattr(self, 'type_y', 'discrete') == 'cont' and not isinstance(getattr(self, 'prior_y', 'uniform'), str):
            bad.append("type_y='cont' with a data prior")
        if getattr(self, 'clf_1sig', False) and self.dim_y != 2:
            raise ValueError('Invalid combination of clf_1sig and dim_y')      # src/DrVAE.py:161
        # input_x_dropout (--x-dropout) is accepted and has NO effect, exactly as in the reference: its MLP
        # computes the dropped inputs and then concatenates the ORIGINAL ones (src/blocks.py:158-161)
        if self.dropout_rate > 0:
            bad.append('dropout_rate > 0 (hard-coded 0. by every reference driver)')
        pr = getattr(self, 'prior_y', 'uniform')
        if pr is not None and not isinstance(pr, str):       # a class prior given as data (src/DrVAE.py:83-85)
            assert isinstance(pr, np.ndarray) and len(pr) == self.dim_y
        if getattr(self, 'use_c', False) or getattr(self, 'use_m', False):
            bad.append('use_c/use_m')
        if bad:
            raise NotImplementedError('not covered by the fused MI355X step: ' + '; '.join(bad))

    # ------------------------------------------------------------------- blocks
    def _build_blocks(self):
        """same sub-module names and construction order as the reference, so that state_dict
        keys (and the parameter-init RNG stream) line up: src/DrVAE.py:112-183,
        src/PVAE.py:106-153, src/VFAE.py:103-165."""
        hp = dict(nonlin=self.nonlinearity, weight_norm=self.wn, batch_norm=self.bn, dropout_rate=self.dropout_rate)
        pri = dict(prior_mu=self.prior_mu, prior_sg=self.prior_sg)
        Z1 = self.dim_z1
        s_in = [self.dim_s] if getattr(self, 'use_s', False) else []       # src/DrVAE.py:134-135,179-180
        self.encoder_z1 = blk.DiagGaussianModule([self.dim_x] + s_in, self.dim_h_en_z1, Z1,
                                                 input_dropout_rates=[self.input_x_dropout] + [0.] * len(s_in), **pri, **hp)
        if self.kind in ('drvae', 'pvae'):
            self.dim_z2 = Z1
            self.decoder_z2Fz1 = blk.DiagGaussianModuleLinear([Z1], [], Z1, bias_only=False, **pri, **hp)
        if self.kind in ('drvae', 'vfae'):
            clf_in = [Z1, Z1] if (self.kind == 'drvae' and self.clf_z1z2) else [Z1]
            if getattr(self, 'type_y', 'discrete') == 'cont':      # regression head (src/DrVAE.py:166-169)
                self.encoder_y = blk.DiagGaussianModule(clf_in, self.dim_h_clf, self.dim_y, fixed_variance=0.05 ** 2,
                                                        constrain_means=True, **hp)
            else:
                self.encoder_y = blk.CategoricalDecoder(clf_in, self.dim_h_clf,
                                                        1 if getattr(self, 'clf_1sig', False) else self.dim_y, **hp)
            top_h = self.dim_h_en_z3 if self.kind == 'drvae' else self.dim_h_en_z2
            top_z = self.dim_z3 if self.kind == 'drvae' else self.dim_z2
            top = blk.DiagGaussianModule([Z1, self.dim_y], top_h, top_z, **pri, **hp)
            setattr(self, 'encoder_z3' if self.kind == 'drvae' else 'encoder_z2', top)
            self.decoder_z1 = blk.DiagGaussianModule([top_z, self.dim_y], self.dim_h_de_z1, Z1, **hp)
        dec = {'diag_gaussian': blk.DiagGaussianSigmaModule, 'binary': blk.BernoulliDecoder,
               'poisson': blk.PoissonDecoder}[self.type_rec]                 # src/DrVAE.py:124-129
        self.decoder_x = dec([Z1] + s_in, self.dim_h_de_x, self.dim_x, **hp)

    def _step_config(self):
        top = 'dim_z3' if self.kind == 'drvae' else 'dim_z2'
        return E.StepConfig(
            kind=self.kind, dim_x=self.dim_x, dim_y=getattr(self, 'dim_y', 2), dim_z1=self.dim_z1,
            dim_z3=getattr(self, top, self.dim_z1), h_en_z1=list(self.dim_h_en_z1),
            h_de_z1=list(getattr(self, 'dim_h_de_z1', [])),
            h_en_z3=list(getattr(self, 'dim_h_en_z3' if self.kind == 'drvae' else 'dim_h_en_z2', [])),
            h_de_x=list(self.dim_h_de_x), h_clf=list(getattr(self, 'dim_h_clf', [])), nonlin=self.nonlinearity,
            weight_norm=self.wn, L=self.L, learning_rate=self.learning_rate, weight_decay=self.weight_decay,
            add_noise_var=self.add_noise_var, yloss_rate=getattr(self, 'yloss_rate', 1.),
            kl_qz2pz2_rate=getattr(self, 'kl_qz2pz2_rate', 1.), pertloss_rate=getattr(self, 'pertloss_rate', 0.),
            anneal_perturb_rate_itermax=getattr(self, 'anneal_perturb_rate_itermax', 0),
            anneal_perturb_rate_offset=getattr(self, 'anneal_perturb_rate_offset', 0),
            clf_z1z2=getattr(self, 'clf_z1z2', True), semi_supervised=getattr(self, 'semi_supervised', True),
            kl_min=self.kl_min, optim_alg=self.optim_alg, clf_1sig=bool(getattr(self, 'clf_1sig', False)),
            type_y=getattr(self, 'type_y', 'discrete'), type_rec=self.type_rec,
            use_s=bool(getattr(self, 'use_s', False)), dim_s=int(getattr(self, 'dim_s', 2)),
            use_MMD=bool(getattr(self, 'use_s', False) and getattr(self, 'use_MMD', False)),
            mmd_rate=float(getattr(self, 'mmd_rate', 1.)), kernel_MMD=getattr(self, 'kernel_MMD', 'rbf_fourier'),
            prior_y=None if (getattr(self, 'prior_y', None) is None or isinstance(getattr(self, 'prior_y', None), str))
            else tuple(float(v) for v in self.prior_y))

    # ---------------------------------------------------------------- inference
    @torch.no_grad()
    def forward(self, x1, s=[]):
        """Inference with the posterior MEANS (no sampling): src/DrVAE.py:253-311,
        src/PVAE.py:203-246, src/VFAE.py:178-215."""
        self.eval()
        x1 = x1.to(next(self.parameters()).device, torch.float32)
        cond = self._s_inputs(s, x1)
        qz1 = self.encoder_z1([x1] + cond)
        z1 = qz1[0]
        res = OrderedDict(z1=z1, qz1=qz1)
        if self.kind in ('drvae', 'pvae'):
            pz2 = self.decoder_z2Fz1([z1])
            z2 = pz2[0]
            res.update(z2=z2, pz2=pz2)
        if self.kind in ('drvae', 'vfae'):
            if self.kind == 'drvae':
                clf_in = [z1, z2 - z1] if self.clf_z1z2 else [z2]
            else:
                clf_in = [z1]
            qy = self.encoder_y(clf_in)
            res.update(**self._pred_proba(qy))
        px1 = self.decoder_x([z1] + cond)
        res.update(px1=px1, x1_rec=px1[0])
        if self.kind in ('drvae', 'pvae'):
            px2 = self.decoder_x([z2] + cond)
            res.update(px2=px2, x2_pert=px2[0])
        return res

    def _s_inputs(self, s, like):
        """[one_hot(s)] when the model conditions on the nuisance variable (src/DrVAE.py:265-268), else []"""
        if not getattr(self, 'use_s', False):
            return []
        return [blk.one_hot(torch.as_tensor(s).to(like.device), self.dim_s)]

    @torch.no_grad()
    def forward_w_pert_identity(self, x1, x2, s=[]):
        """Inference assuming the perturbation function is the identity
        (src/DrVAE.py:185-251, src/PVAE.py:155-201)."""
        if self.kind == 'vfae':
            raise AttributeError('VFAE has no perturbation path')
        self.eval()
        dev = next(self.parameters()).device
        x1, x2 = x1.to(dev, torch.float32), x2.to(dev, torch.float32)
        cond = self._s_inputs(s, x1)
        qz1 = self.encoder_z1([x1] + cond)
        z1 = qz1[0]
        res = OrderedDict(z1=z1, qz1=qz1)
        if self.kind == 'drvae':
            qy = self.encoder_y([z1, z1 - z1] if self.clf_z1z2 else [z1])
            res.update(**self._pred_proba(qy))
        px2 = self.decoder_x([z1] + cond)
        qz2 = self.encoder_z1([x2] + cond)
        px2_rec = self.decoder_x([qz2[0]] + cond)
        res.update(px2=px2, x2_pert=px2[0], z2=qz2[0], qz2=qz2, px2_rec=px2_rec, x2_rec=px2_rec[0])
        return res

    def _pred_proba(self, qy):
        """(pred, proba) of src/DrVAE.py:212-220: class + probabilities, or mean + log-variance"""
        if getattr(self, 'type_y', 'discrete') == 'cont':
            return dict(pred=qy[0], proba=qy[1])
        return dict(pred=self.encoder_y.most_probable(*qy), proba=qy[0])

    def predict(self, **kwargs):
        res = self.forward(**kwargs)
        return res['pred'].squeeze().cpu().numpy(), res['proba'].cpu().numpy()

    def reconstruct(self, **kwargs):
        res = self.forward(**kwargs)
        out = [res['x1_rec'].cpu().numpy(), [t.cpu().numpy() for t in res['px1']]]
        if 'px2' in res:
            out += [res['x2_pert'].cpu().numpy(), [t.cpu().numpy() for t in res['px2']]]
        return tuple(out)

    def transform(self, **kwargs):
        res = self.forward(**kwargs)
        if 'z2' in res:
            return res['z1'].cpu().numpy(), res['z2'].cpu().numpy()
        return res['z1'].cpu().numpy()

    # -------------------------------------------------------------- empty groups
    def _warn_empty_groups(self, has_x2, has_y):
        """the reference warns about empty data groups in the minibatch (src/DrVAE.py:588-608)"""
        hx = np.asarray(has_x2.cpu() if torch.is_tensor(has_x2) else has_x2).astype(bool).reshape(-1)
        hy = np.asarray(has_y.cpu() if torch.is_tensor(has_y) else has_y).astype(bool).reshape(-1)
        if self.kind == 'drvae':
            names = {'Labeled Singleton': hy & ~hx, 'Unlabeled Singleton': ~hy & ~hx,
                     'Labeled Paired perturbation': hy & hx, 'Unlabeled Paired perturbation': ~hy & hx}
        elif self.kind == 'pvae':
            names = {'Singleton': ~hx, 'Paired perturbation': hx}
        else:
            names = {'labeled': hy, 'unlabeled': ~hy}
        for n, m in names.items():
            if not m.any():
                warnings.warn('No %s data in the minibatch' % n)
