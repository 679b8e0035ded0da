"""Drop-in for the reference's ``blocks`` module (``import blocks as blk``,
src/DrVAE.py:21, src/PVAE.py:21, src/VFAE.py:21).

Same public names, constructor signatures, ``forward(list_of_tensors)`` convention,
return types (tuples / 1-tuples / 1-element lists) and ``state_dict`` keys as
src/blocks.py, so the reference's model classes consume these blocks unchanged when
this module is installed as ``sys.modules['blocks']``.  All
arithmetic -- Linear(+WeightNorm)+activation GEMMs, reparameterisation, diagonal-
Gaussian KL, Gaussian log-likelihood over genes, the categorical head -- runs in the
hand-written HIP kernels behind ``drvae_amd.ops``; tensors must live on the GPU.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import layers as lyr
from . import ops
from ._lib import GAUSS_LOGVAR, GAUSS_SIGMA

# name -> module, as src/blocks.py:21-24 (the modules are parameter-free; MLP fuses the
# activation into the producing GEMM's epilogue instead of calling them)
nonlinearities = {}
for _name, _factory in (('tanh', nn.Tanh), ('sigmoid', nn.Sigmoid), ('softmax', lambda: nn.Softmax(dim=-1)),
                        ('softplus', nn.Softplus), ('softsign', nn.Softsign), ('relu', nn.ReLU),
                        ('leaky_relu', lambda: nn.LeakyReLU(0.1)), ('elu', nn.ELU), ('selu', nn.SELU)):
    nonlinearities[_name] = _factory()


def _fusable(module):
    """name of the fused-epilogue activation equivalent to an activation module, or None"""
    if isinstance(module, nn.ELU):
        return 'elu' if module.alpha == 1.0 else None
    if isinstance(module, nn.Softplus):
        return 'softplus' if (module.beta in (1, 1.0) and module.threshold in (20, 20.0)) else None
    if isinstance(module, nn.LeakyReLU):
        return 'leaky_relu' if abs(module.negative_slope - 0.1) < 1e-12 else None
    for cls, name in ((nn.Sigmoid, 'sigmoid'), (nn.Tanh, 'tanh'), (nn.ReLU, 'relu'), (nn.SELU, 'selu'),
                      (nn.Softsign, 'softsign')):
        if isinstance(module, cls):
            return name
    return None


def _apply_linear(layer, inputs, act='identity', shift=0.0):
    g = getattr(layer, 'g', None)
    if isinstance(layer, lyr.WeightNormLinear) and layer._pending_data_init:
        layer._data_init(torch.cat(list(inputs), 1))
    return ops.linear_act(inputs, layer.weight, layer.bias, g, act, shift)


def _run_sequential(seq, inputs, shift=0.0):
    """Evaluate an nn.Sequential of [Dropout|BatchNorm|Linear|activation] modules, fusing each
    Linear with the activation module that follows it into one GEMM launch."""
    mods = list(seq._modules.values())      # not children(): it de-duplicates the shared activation module
    x, i, shifted = list(inputs), 0, (shift == 0.0)
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            act = _fusable(mods[i + 1]) if i + 1 < len(mods) else None
            last = (i + (2 if act else 1)) >= len(mods)
            x = [_apply_linear(m, x, act or 'identity', shift if last else 0.0)]
            shifted = shifted or last
            i += 2 if act else 1
        else:
            t = x[0] if len(x) == 1 else torch.cat(x, 1)
            x = [m(t)]
            i += 1
    out = x[0] if len(x) == 1 else torch.cat(x, 1)
    return out if shifted else out + shift


# ------------------------------------------------------------------------ MMD kernels
def rbf(x1, x2, gamma=1.):
    """src/blocks.py:29-32.  The reference implementation raises on torch >= 0.4
    (``squeeze_(2)`` of a 2-D tensor); the intended Gram matrix is returned here."""
    d2 = ((x1[None, :, :] - x2[:, None, :]) ** 2).sum(2)
    return torch.exp(-d2 * gamma).t()


def mmd_criterion(z, sind, kernel='rbf_fourier', pairs=None):
    """Minus the MMD (``mmd_objective`` with ``kernel``) between the latent rows of every category of the nuisance
    variable and the rows outside it, averaged over the categories; with two categories only the first pair
    (the body of src/DGMMixin.py:42-66).  ``sind``: one 0/1 indicator vector per category.  A side without rows is
    replaced by one random N(0,1) row, like the reference.  ``pairs``: the (rows in, rows out) index lists of the
    categories when the caller has them already (``torch.nonzero`` synchronises with the host: not inside a hipGraph
    capture)."""
    mmd = 0.
    for k, ind in enumerate(sind):
        if pairs is not None:
            ind0, ind1 = pairs[k]
        else:
            flat = ind.reshape(-1)
            ind0 = torch.nonzero(flat != 0).reshape(-1)
            ind1 = torch.nonzero(flat == 0).reshape(-1)
        z0 = z.index_select(0, ind0) if ind0.numel() else torch.empty(1, z.size(1), device=z.device).normal_()
        z1 = z.index_select(0, ind1) if ind1.numel() else torch.empty(1, z.size(1), device=z.device).normal_()
        mmd = mmd - mmd_objective(z0, z1, kernel=kernel)
        if len(sind) == 2:
            return mmd
    return mmd / len(sind)


def poly(x1, x2, degree=2, gamma=1., bias=1.):
    """src/blocks.py:34-35 (the x1 x2^T product runs on the MFMA GEMM)."""
    gram = ops.linear_act([x1], x2.contiguous(), None)
    return torch.pow(gamma * gram + bias, degree)


def identity(x1, x2):
    """src/blocks.py:37-38: one HIP launch each way (``ops.MMDIdentity``).  No ATen fallback: host tensors are refused
    by the launcher like everywhere else in the package."""
    return ops.MMDIdentity.apply(x1, x2)


def mmd_fourier(x1, x2, bandwidth=2., dim_r=500):
    """Random-Fourier-feature MMD (src/blocks.py:40-55): draws W~N(0,1) (Z,dim_r) then
    b~U(0,1) (dim_r), in that order, from the framework generator like the reference."""
    z = x1.size(1)
    rnd_a = torch.empty(z, dim_r, device=x1.device).normal_()
    rnd_b = torch.empty(dim_r, device=x1.device).uniform_()
    a, c = math.sqrt(2. / bandwidth) / math.sqrt(z), math.sqrt(2. / dim_r)
    return ops.MMDRff.apply(x1, x2, rnd_a, rnd_b, a, c)


kernels = {'rbf': rbf, 'poly': poly, 'identity': identity, 'rbf_fourier': mmd_fourier}


def mmd_objective(x1, x2, kernel='rbf', bandwidths=1. / (2 * (np.array([1., 2., 5., 8., 10]) ** 2))):
    """MMD score between two row sets (src/blocks.py:59-76)."""
    fn = kernels[kernel]
    if kernel == 'identity':
        return torch.sqrt(fn(x1, x2))
    if kernel == 'rbf_fourier':
        return torch.sqrt(fn(x1, x2, bandwidth=2.))
    # the bandwidth mixture, its means and its derivative on HIP row kernels around the three Gram products on the MFMA
    # GEMM (``ops.MMDMix``).  No ATen fallback (host tensors are refused by the launchers); the row kernels hold at most
    # 8 bandwidths (the reference's default: 5)
    if len(bandwidths) > 8:
        raise NotImplementedError('mmd_objective(kernel=%r): at most 8 bandwidths (got %d)' % (kernel, len(bandwidths)))
    if x1.size(1) != x2.size(1):
        raise ValueError('mmd_objective: the two row sets differ in width (%d, %d)' % (x1.size(1), x2.size(1)))
    gam = [math.sqrt(x1.size(1)) * float(bw) for bw in bandwidths]
    return torch.sqrt(ops.MMDMix.apply(x1, x2, kernel, gam))


def one_hot(y, max_dim):
    """(n,) or (n,1) integer labels -> (n,max_dim) float one-hot; None for None/empty
    (src/blocks.py:78-92; pure: does not reshape its argument in place)."""
    if y is None or len(y) == 0:
        return None
    idx = y.detach().reshape(-1, 1).long()
    out = torch.zeros(idx.size(0), max_dim, device=idx.device)
    out.scatter_(1, idx, 1.0)
    return out


def _keep_mask_like(t, keep):
    """Bernoulli(keep) 0/1 mask, drawn on the tensor's device from torch's generator (tests inject the reference's)"""
    return torch.empty_like(t, memory_format=torch.contiguous_format).bernoulli_(keep)


class BatchNorm1d(nn.BatchNorm1d):
    """``nn.BatchNorm1d`` of the reference's MLP (src/blocks.py:137-149: ``bn_input`` / ``bn{i}``, affine, momentum 0.1,
    eps 1e-5) with forward and backward on the HIP kernels; state_dict keys and running-statistics semantics are the
    parent's (incl. ``num_batches_tracked``)."""

    def forward(self, x):
        if x.dim() != 2 or self.momentum is None or not self.affine or not self.track_running_stats:
            raise NotImplementedError('BatchNorm1d: (rows, features) input, affine, tracked statistics, fixed momentum')
        if self.training:
            if x.shape[0] < 2:      # (torch: "Expected more than 1 value per channel when training")
                raise ValueError('Expected more than 1 value per channel when training, got input size %s' % (tuple(x.shape),))
            self.num_batches_tracked.add_(1)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                              self.momentum, self.eps)


class Dropout(nn.Dropout):
    """``nn.Dropout`` between hidden layers of the reference's MLP (src/blocks.py:140-141): keep mask drawn on the
    device, applied (forward and backward) by ``dv_mask_scale``; identity in eval mode."""

    def forward(self, x):
        if not self.training or self.p == 0.:
            return x
        keep = 1.0 - self.p
        if keep <= 0.:
            return x * 0.
        return ops.dropout(x, _keep_mask_like(x, keep), keep)


# -------------------------------------------------------------------------------- MLP
class MLP(nn.Module):
    """Deterministic MLP over the concatenation of its inputs, returning the last hidden
    layer (src/blocks.py:95-164).  Sub-modules of ``self.model`` are named ``bn_input``,
    ``dropout{i}``, ``linear{i}``, ``activ{i}``, ``bn{i}`` exactly as in the reference."""

    def __init__(self, input_dims, hidden_dims, nonlin='softplus', weight_norm=False, batch_norm=False,
                 dropout_rate=0., input_dropout_rates=None):
        super().__init__()
        self.input_dims, self.hidden_dims, self.nonlin = input_dims, hidden_dims, nonlin
        self.weight_norm, self.batch_norm, self.dropout_rate = weight_norm, batch_norm, dropout_rate
        if input_dropout_rates is None:
            input_dropout_rates = [max(0., dropout_rate - 0.3)] * len(input_dims)
        if len(input_dropout_rates) != len(input_dims):
            raise ValueError('MLP: input_dropout_rates is not the same length as input_dims %s %s'
                             % (input_dropout_rates, input_dims))
        self.input_dropout_rates = input_dropout_rates
        # kept for state/API parity; the reference computes input dropout and then discards
        # it (src/blocks.py:159-161), so it is never applied
        self.input_dropouts = nn.ModuleList([nn.Dropout(p=r) for r in input_dropout_rates])
        make = lyr.WeightNormLinear if weight_norm else nn.Linear
        mods = OrderedDict()
        width = int(np.asarray(input_dims).sum())
        if batch_norm:
            mods['bn_input'] = BatchNorm1d(width, affine=True)
        for i, h in enumerate(hidden_dims, start=1):
            if i > 1 and dropout_rate > 0.:
                mods['dropout%d' % i] = Dropout(p=dropout_rate)
            mods['linear%d' % i] = make(width, h)
            mods['activ%d' % i] = nonlinearities[nonlin]
            if batch_norm:
                mods['bn%d' % i] = BatchNorm1d(h, affine=True)
            width = h
        self.model = nn.Sequential(mods)

    def forward(self, inputs):
        assert len(inputs) == len(self.input_dims)
        return _run_sequential(self.model, inputs)

    def features(self, inputs):
        """like forward, but an empty trunk hands the un-concatenated inputs to the next
        Linear (which reads both sources directly) instead of materialising torch.cat"""
        assert len(inputs) == len(self.input_dims)
        if len(self.model) == 0:
            return list(inputs)
        return [_run_sequential(self.model, inputs)]


# ----------------------------------------------------------------------------- mixins
def _noise_like(t):
    return torch.empty_like(t, memory_format=torch.contiguous_format).normal_()


class _DiagGaussianOps:
    """Shared implementation of the two Gaussian mixins of src/blocks.py:166-240.  The second
    distribution parameter is a log-variance (``_MODE = GAUSS_LOGVAR``, prior attribute ``prior_lv``) or
    a standard deviation (``GAUSS_SIGMA``, ``prior_sg``); every method is one HIP launch
    (plus one for its backward)."""
    _MODE = None
    _PRIOR2 = None

    def _prior(self):
        return float(self.prior_mu), float(getattr(self, self._PRIOR2))

    def sample(self, mu, second):
        """reparameterised draw; a 1-tuple, because callers concatenate tuples (src/DrVAE.py:350)"""
        return (ops.reparam(mu, second, _noise_like(mu), self._MODE),)

    def kldivergence_perx(self, mu_q, second_q, mu_p, second_p):
        """KL(q || p) per row"""
        return ops.kl_rows(mu_q, second_q, mu_p, second_p, self._MODE)

    def kldivergence(self, mu_q, second_q, mu_p, second_p):
        return self.kldivergence_perx(mu_q, second_q, mu_p, second_p).sum()

    def kldivergence_from_prior_perx(self, mu, second):
        """KL(q || scalar prior) per row"""
        pm, p2 = self._prior()
        return ops.kl_rows_prior(mu, second, pm, p2, self._MODE)

    def kldivergence_from_prior(self, mu, second):
        return self.kldivergence_from_prior_perx(mu, second).sum()

    def logp_perx(self, sample, mu, second):
        """Gaussian log-density of ``sample`` summed over features, per row"""
        return ops.nll_rows(sample, mu, second, self._MODE)

    def logp(self, sample, mu, second):
        return self.logp_perx(sample, mu, second).sum()

    def logp_prior_perx(self, sample):
        pm, p2 = self._prior()
        return ops.nll_rows(sample, torch.full_like(sample, pm), torch.full_like(sample, p2), self._MODE)

    def logp_prior(self, sample):
        return self.logp_prior_perx(sample).sum()


class GaussianLogVarMixin(_DiagGaussianOps):
    """Diagonal Gaussian parametrised by (mu, log sigma^2): src/blocks.py:166-202."""
    _MODE, _PRIOR2 = GAUSS_LOGVAR, 'prior_lv'


class GaussianSigmaMixin(_DiagGaussianOps):
    """Diagonal Gaussian parametrised by (mu, sigma): src/blocks.py:204-240."""
    _MODE, _PRIOR2 = GAUSS_SIGMA, 'prior_sg'


def _head(suffix, make, n_in, n_out, dropout_rate, activation=None):
    mods = OrderedDict()
    if dropout_rate > 0.:
        mods['dropout_' + suffix] = Dropout(p=dropout_rate)
    mods['linear_' + suffix] = make(n_in, n_out)
    if activation is not None:
        mods['activ_' + suffix] = nonlinearities[activation]
    return nn.Sequential(mods)


def _trunk_width(input_dims, hidden_dims):
    return hidden_dims[-1] if len(hidden_dims) > 0 else int(np.asarray(input_dims).sum())


# ---------------------------------------------------------------------------- modules
class DiagGaussianModule(GaussianLogVarMixin, nn.Module):
    """inputs -> (mu, logvar): MLP trunk ``nnet`` + heads ``encoder_mu.linear_mu`` and
    ``encoder_lv.linear_lv``; ``logvar = lv(h) - 2`` (src/blocks.py:243-301)."""

    def __init__(self, input_dims, hidden_dims, output_dim, nonlin='softplus', weight_norm=False, batch_norm=False,
                 dropout_rate=0., input_dropout_rates=None, prior_mu=0., prior_sg=1., constrain_means=False,
                 fixed_variance=None):
        super().__init__()
        self.nnet = MLP(input_dims=input_dims, hidden_dims=hidden_dims, nonlin=nonlin, weight_norm=weight_norm,
                        batch_norm=batch_norm, dropout_rate=dropout_rate, input_dropout_rates=input_dropout_rates)
        self.output_dim, self.constrain_means = output_dim, constrain_means
        self.fixed_variance = None if fixed_variance is None else (torch.zeros(1) + fixed_variance).log()
        make = lyr.WeightNormLinear if weight_norm else nn.Linear
        width = _trunk_width(input_dims, hidden_dims)
        self.encoder_mu = _head('mu', make, width, output_dim, dropout_rate,
                                'sigmoid' if constrain_means else None)
        self.encoder_lv = _head('lv', make, width, output_dim, dropout_rate)
        # plain tensors, not buffers -- as in the reference (src/blocks.py:288-289)
        self.prior_mu = torch.zeros(1) + prior_mu
        self.prior_lv = (torch.zeros(1) + prior_sg ** 2).log()

    def forward(self, inputs):
        h = self.nnet.features(inputs)
        mu = _run_sequential(self.encoder_mu, h)
        if self.fixed_variance is not None:
            logvar = self.fixed_variance.to(mu.device).expand_as(mu)
        else:
            logvar = _run_sequential(self.encoder_lv, h, shift=-2.0)
        return mu, logvar


class DiagGaussianModuleLinear(GaussianLogVarMixin, nn.Module):
    """The perturbation function p(z2|z1): ``mu = x + x W_mu^T + b`` (or ``x + b`` when
    ``bias_only``), ``logvar = Linear(x) - 2`` (src/blocks.py:304-361)."""

    def __init__(self, input_dims, hidden_dims, latent_dim, nonlin='softplus', weight_norm=False, batch_norm=False,
                 dropout_rate=0., input_dropout_rates=None, prior_mu=0., prior_sg=1., constrain_means=False,
                 bias_only=False):
        super().__init__()
        self.input_dims, self.hidden_dims, self.nonlin = input_dims, hidden_dims, nonlin
        self.weight_norm, self.batch_norm, self.dropout_rate = weight_norm, batch_norm, dropout_rate
        self.input_dropout_rates = input_dropout_rates
        self.constrain_means, self.bias_only = constrain_means, bias_only
        assert len(input_dims) == 1 and input_dims[0] == latent_dim     # must not change dimensionality
        if constrain_means:
            raise NameError("name 'modules_mu' is not defined")         # reference behaviour, src/blocks.py:335
        # the logvar head is a plain Linear whatever weight_norm says (src/blocks.py:332); it is
        # created first so that the parameter-init RNG stream matches the reference's
        head_lv = _head('lv', nn.Linear, latent_dim, latent_dim, dropout_rate)
        self.W_mu = nn.Parameter(torch.empty(latent_dim, latent_dim).uniform_(-0.0001, 0.0001))
        self.bias_mu = nn.Parameter(torch.empty(latent_dim).uniform_(-0.0001, 0.0001))
        self.encoder_lv = head_lv
        self.prior_mu = torch.zeros(1) + prior_mu
        self.prior_lv = (torch.zeros(1) + prior_sg ** 2).log()

    def forward(self, inputs):
        assert len(inputs) == len(self.input_dims)
        x = inputs[0] if len(inputs) == 1 else torch.cat(inputs, 1)
        if self.bias_only:
            mu = x + self.bias_mu.expand_as(x)
        else:
            mu = x + ops.linear_act([x], self.W_mu, self.bias_mu)
        logvar = _run_sequential(self.encoder_lv, [x], shift=-2.0)
        return mu, logvar


class DiagGaussianSigmaModule(GaussianSigmaMixin, nn.Module):
    """inputs -> (mu, std) with ``std = softplus(sg(h)) + 1e-3`` -- the data decoder
    p(x|z) (src/blocks.py:364-416)."""

    def __init__(self, input_dims, hidden_dims, latent_dim, nonlin='softplus', weight_norm=False, batch_norm=False,
                 dropout_rate=0., input_dropout_rates=None, prior_mu=0., prior_sg=1., constrain_means=False):
        super().__init__()
        self.nnet = MLP(input_dims=input_dims, hidden_dims=hidden_dims, nonlin=nonlin, weight_norm=weight_norm,
                        batch_norm=batch_norm, dropout_rate=dropout_rate, input_dropout_rates=input_dropout_rates)
        self.latent_dim, self.constrain_means = latent_dim, constrain_means
        make = lyr.WeightNormLinear if weight_norm else nn.Linear
        width = _trunk_width(input_dims, hidden_dims)
        self.encoder_mu = _head('mu', make, width, latent_dim, dropout_rate,
                                'sigmoid' if constrain_means else None)
        self.encoder_sg = _head('sg', make, width, latent_dim, dropout_rate, 'softplus')
        self.prior_mu = torch.zeros(1) + prior_mu
        self.prior_sg = torch.zeros(1) + prior_sg

    def forward(self, inputs):
        h = self.nnet.features(inputs)
        mu = _run_sequential(self.encoder_mu, h)
        std = _run_sequential(self.encoder_sg, h, shift=1e-3)
        return mu, std


class _SingleHeadDecoder(nn.Module):
    """MLP trunk ``nnet`` + one Linear head with a fused activation: the anatomy of ``CategoricalDecoder``
    (src/blocks.py:419-455) for the two data decoders src/DrVAE.py:124-129 names and src/blocks.py never defines."""
    head, lin, act, shift = None, None, 'identity', 0.0

    def __init__(self, input_dims, hidden_dims, reconstruction_dim, nonlin='softplus', weight_norm=False,
                 batch_norm=False, dropout_rate=0., input_dropout_rates=None):
        super().__init__()
        self.nnet = MLP(input_dims=input_dims, hidden_dims=hidden_dims, nonlin=nonlin, weight_norm=weight_norm,
                        batch_norm=batch_norm, dropout_rate=dropout_rate, input_dropout_rates=input_dropout_rates)
        self.reconstruction_dim = reconstruction_dim
        make = lyr.WeightNormLinear if weight_norm else nn.Linear
        mods = OrderedDict()
        if dropout_rate > 0.:
            mods['dropout'] = Dropout(p=dropout_rate)
        mods[self.lin] = make(_trunk_width(input_dims, hidden_dims), reconstruction_dim)
        setattr(self, self.head, nn.Sequential(mods))

    def _head_out(self, inputs):
        h = self.nnet.features(inputs)
        seq = getattr(self, self.head)
        if hasattr(seq, 'dropout'):
            h = [seq.dropout(h[0] if len(h) == 1 else torch.cat(h, 1))]
        return _apply_linear(getattr(seq, self.lin), h, act=self.act, shift=self.shift)

    def logp(self, x, v):
        return torch.sum(self.logp_perx(x, v))


class BernoulliDecoder(_SingleHeadDecoder):
    """EXTENSION (``type_rec='binary'``, src/DrVAE.py:124-125; absent from the reference's blocks.py):
    inputs -> [clamp(sigmoid(linear_p(h)), 1e-10, 1-1e-10)]; log p(x|.) = sum x log p + (1-x) log(1-p)."""
    head, lin, act = 'decoder_p', 'linear_p', 'sigmoid'

    def forward(self, inputs):
        return [torch.clamp(self._head_out(inputs), min=1e-10, max=1. - 1e-10)]

    def logp_perx(self, x, ps):
        return (x * torch.log(ps) + (1. - x) * torch.log(1. - ps)).sum(1)

    def sample(self, ps):
        return torch.bernoulli(ps)

    def most_probable(self, ps):
        return (ps > 0.5).to(ps.dtype)


class PoissonDecoder(_SingleHeadDecoder):
    """EXTENSION (``type_rec='poisson'``, src/DrVAE.py:128-129; absent from the reference's blocks.py):
    inputs -> [softplus(linear_r(h)) + 1e-6]; log p(x|.) = sum x log rate - rate - lgamma(x+1)."""
    head, lin, act, shift = 'decoder_r', 'linear_r', 'softplus', 1e-6

    def forward(self, inputs):
        return [self._head_out(inputs)]

    def logp_perx(self, x, rate):
        return (x * torch.log(rate) - rate - torch.lgamma(x + 1.)).sum(1)

    def sample(self, rate):
        return torch.poisson(rate)


class CategoricalDecoder(nn.Module):
    """inputs -> [clamped class probabilities] (a 1-element list), plus the categorical
    log-likelihood / KL / entropy helpers (src/blocks.py:419-486)."""

    def __init__(self, input_dims, hidden_dims, reconstruction_dim, nonlin='softplus', weight_norm=False,
                 batch_norm=False, dropout_rate=0., input_dropout_rates=None):
        super().__init__()
        self.nnet = MLP(input_dims=input_dims, hidden_dims=hidden_dims, nonlin=nonlin, weight_norm=weight_norm,
                        batch_norm=batch_norm, dropout_rate=dropout_rate, input_dropout_rates=input_dropout_rates)
        self.reconstruction_dim = reconstruction_dim
        make = lyr.WeightNormLinear if weight_norm else nn.Linear
        width = _trunk_width(input_dims, hidden_dims)
        mods = OrderedDict()
        if dropout_rate > 0.:
            mods['dropout_p'] = Dropout(p=dropout_rate)
        mods['linear_p'] = make(width, reconstruction_dim)
        mods['activ_p'] = nn.Softmax(dim=-1) if reconstruction_dim > 1 else nn.Sigmoid()
        self.decoder_p = nn.Sequential(mods)

    def forward(self, inputs):
        h = self.nnet.features(inputs)
        if hasattr(self.decoder_p, 'dropout_p'):
            h = [self.decoder_p.dropout_p(h[0] if len(h) == 1 else torch.cat(h, 1))]
        logits = _apply_linear(self.decoder_p.linear_p, h)
        return [ops.softmax_clamp(logits, sigmoid1=(self.reconstruction_dim == 1))]

    def sample(self, ps):
        return torch.multinomial(ps, 1)

    def logp_perx(self, x, ps):
        return ops.cat_logp_rows(ps, x)

    def logp(self, x, ps):
        return torch.sum(self.logp_perx(x, ps))

    def entropy(self, ps):
        return ops.cat_entropy_rows(ps).sum()

    def kldivergence_perx(self, ps, prior):
        return ops.cat_kl_elem(ps, prior)

    def kldivergence(self, ps, prior):
        return self.kldivergence_perx(ps, prior).sum()

    def most_probable(self, ps):
        return ops.cat_most_probable(ps)
