"""Oracle, block level: functional fp32 CPU restatement of reference ``src/layers.py``
and ``src/blocks.py``.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Everything is a pure function of explicit tensors: parameters come in as a dict
keyed by the reference's ``state_dict`` names, noise comes in as explicit ``eps``
tensors (the reference draws them from the global CPU generator, blocks.py:172,210).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = float(np.log(2 * np.pi))  # blocks.py:196,234 use float(np.log(2*np.pi))


# --------------------------------------------------------------------------- layers
def weightnorm_linear(x, weight, g, bias):
    """layers.py:38-40 -- out = (g / ||W_row||_2) * (x W^T) + b."""
    out = F.linear(x, weight)
    scale = g / torch.norm(weight, 2, 1)
    return scale.expand_as(out) * out + bias.expand_as(out)


def linear(x, p, prefix):
    """nn.Linear or WeightNormLinear depending on whether ``<prefix>.g`` exists
    (blocks.py:118-121 picks the layer class; layers.py:17-23 adds ``g``)."""
    w, b = p[prefix + '.weight'], p[prefix + '.bias']
    if prefix + '.g' in p:
        return weightnorm_linear(x, w, p[prefix + '.g'], b)
    return F.linear(x, w, b)


def made_mask(d_pre, d, m_pre, output_layer, rev_order=False):
    """layers.py:84-133 -- MADE connectivity mask (numpy, integer work)."""
    m_pre = np.asarray(m_pre)
    mask = np.zeros((d_pre, d), dtype=np.float32)
    if not output_layer:
        m = np.arange(1, np.max(m_pre)).astype(int)
        while len(m) < d:
            m = np.hstack((m, m))
        m = m[:d]
        for j in range(d):
            mask[m_pre <= m[j], j] = 1
    else:
        m = np.arange(1, d + 1)
        if rev_order:
            m = m[::-1]
        for j in range(d):
            mask[m_pre < m[j], j] = 1
    return m, mask


# ------------------------------------------------------------------- nonlinearities
def activation(name, x):
    """blocks.py:21-24 name -> function table."""
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'softmax':
        return torch.softmax(x, dim=-1)
    if name == 'softplus':
        return F.softplus(x)
    if name == 'softsign':
        return F.softsign(x)
    if name == 'relu':
        return torch.relu(x)
    if name == 'leaky_relu':
        return F.leaky_relu(x, 0.1)
    if name == 'elu':
        return F.elu(x)
    if name == 'selu':
        return F.selu(x)
    raise KeyError(name)


# ------------------------------------------------------------------------------ MLP
def mlp(inputs, p, prefix, n_hidden, nonlin):
    """blocks.py:153-164 -- concat inputs on dim 1, then linear{i}+activ{i}.
    Input dropout is computed and discarded by the reference (blocks.py:159-161);
    batch norm / hidden dropout are hard-coded off by every model (DrVAE.py:79-80)."""
    h = torch.cat(inputs, 1)
    for i in range(1, n_hidden + 1):
        h = activation(nonlin, linear(h, p, '%s.model.linear%d' % (prefix, i)))
    return h


def batch_norm_rows(x, p, prefix, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d(affine=True) as blocks.py:137-149 instantiates it, written out: training normalises by the batch
    mean and BIASED variance and moves the running statistics by ``momentum`` (running_var with the unbiased
    variance); eval normalises by the running statistics.  Returns (y, new running_mean, new running_var)."""
    w, b = p[prefix + '.weight'], p[prefix + '.bias']
    rm, rv = p[prefix + '.running_mean'], p[prefix + '.running_var']
    if training:
        n = x.shape[0]
        mu = x.mean(0)
        var = ((x - mu) ** 2).mean(0)
        y = (x - mu) / torch.sqrt(var + eps) * w + b
        return y, (1 - momentum) * rm + momentum * mu.detach(), (1 - momentum) * rv + momentum * var.detach() * n / (n - 1)
    return (x - rm) / torch.sqrt(rv + eps) * w + b, rm, rv


def mlp_options(inputs, p, prefix, n_hidden, nonlin, batch_norm=False, dropout_masks=None, training=True):
    """blocks.py:135-164 with the options every model hard-codes off: ``bn_input``, then per layer [dropout{i} (i > 1)]
    linear{i} activ{i} [bn{i}].  ``dropout_masks``: {i: keep mask} of a train-mode pass (rate 0.5 -> scale 2).
    Returns (output, {bn prefix: (running_mean, running_var)})."""
    h = torch.cat(inputs, 1)
    stats = {}

    def bn(name, h):
        q = '%s.model.%s' % (prefix, name)
        y, rm, rv = batch_norm_rows(h, p, q, training)
        stats[q] = (rm, rv)
        return y
    if batch_norm:
        h = bn('bn_input', h)
    for i in range(1, n_hidden + 1):
        if dropout_masks and i in dropout_masks and training:
            m = dropout_masks[i]
            h = h * m / float(m.mean().new_tensor(0.5))
        h = activation(nonlin, linear(h, p, '%s.model.linear%d' % (prefix, i)))
        if batch_norm:
            h = bn('bn%d' % i, h)
    return h, stats


def one_hot(y, max_dim):
    """blocks.py:78-92 (intended semantics: (n,)|(n,1) ints -> (n,max_dim) floats)."""
    if y is None or len(y) == 0:
        return None
    idx = y.reshape(-1, 1).long()
    out = torch.zeros(idx.size(0), max_dim)
    out.scatter_(1, idx, 1.0)
    return out


# ------------------------------------------------------------ Gaussian, logvar form
def sample_logvar(mu, logvar, eps):
    """blocks.py:170-174."""
    return eps * torch.exp(0.5 * logvar) + mu


def kl_logvar_rows(mu_q, lv_q, mu_p, lv_p):
    """blocks.py:180-182."""
    return -0.5 * torch.sum(1 - lv_p + lv_q - ((mu_q - mu_p) ** 2 + lv_q.exp()) / lv_p.exp(), dim=1)


def kl_logvar_prior_rows(mu, lv, prior_mu=0., prior_sg=1.):
    """blocks.py:188-190 with the 1-element priors of blocks.py:288-289."""
    pm = (torch.zeros(1) + prior_mu).expand_as(mu)
    pl = (torch.zeros(1) + prior_sg ** 2).log().expand_as(lv)
    return kl_logvar_rows(mu, lv, pm, pl)


def logp_logvar_rows(sample, mu, logvar):
    """blocks.py:195-196."""
    return -0.5 * torch.sum(LOG_2PI + logvar + ((sample - mu) ** 2) / logvar.exp(), dim=1)


def logp_logvar_prior_rows(sample, prior_mu=0., prior_sg=1.):
    """blocks.py:201-202."""
    pm = (torch.zeros(1) + prior_mu).expand_as(sample)
    pl = (torch.zeros(1) + prior_sg ** 2).log().expand_as(sample)
    return logp_logvar_rows(sample, pm, pl)


# ------------------------------------------------------------- Gaussian, sigma form
def sample_sigma(mu, std, eps):
    """blocks.py:208-211."""
    return eps * std + mu


def kl_sigma_rows(mu_q, std_q, mu_p, std_p):
    """blocks.py:217-220."""
    return -0.5 * torch.sum(1. - torch.log(std_p ** 2) + torch.log(std_q ** 2)
                            - ((mu_q - mu_p) ** 2 + std_q ** 2) / std_p ** 2, dim=1)


def logp_sigma_rows(sample, mu, std):
    """blocks.py:233-234 -- the reconstruction NLL reduction over genes."""
    return -0.5 * torch.sum(LOG_2PI + torch.log(std ** 2) + ((sample - mu) ** 2) / (std ** 2), dim=1)


# --------------------------------------------------------------------- block forward
def diag_gaussian(inputs, p, prefix, n_hidden, nonlin, constrain_means=False, fixed_variance=None):
    """DiagGaussianModule.forward, blocks.py:291-301 -> (mu, logvar)."""
    h = mlp(inputs, p, prefix + '.nnet', n_hidden, nonlin)
    mu = linear(h, p, prefix + '.encoder_mu.linear_mu')
    if constrain_means:
        mu = torch.sigmoid(mu)
    logvar = linear(h, p, prefix + '.encoder_lv.linear_lv') - 2.
    if fixed_variance is not None:
        logvar = (torch.zeros(1) + fixed_variance).log().expand_as(mu)
    return mu, logvar


def diag_gaussian_linear(inputs, p, prefix, bias_only=False):
    """DiagGaussianModuleLinear.forward, blocks.py:349-361 -> (mu, logvar);
    the logvar head is a plain nn.Linear regardless of weight_norm (blocks.py:332)."""
    x = torch.cat(inputs, 1)
    if bias_only:
        mu = x + p[prefix + '.bias_mu'].expand_as(x)
    else:
        mu = x + F.linear(x, p[prefix + '.W_mu']) + p[prefix + '.bias_mu'].expand_as(x)
    logvar = F.linear(x, p[prefix + '.encoder_lv.linear_lv.weight'],
                      p[prefix + '.encoder_lv.linear_lv.bias']) - 2.
    return mu, logvar


def diag_gaussian_sigma(inputs, p, prefix, n_hidden, nonlin, constrain_means=False):
    """DiagGaussianSigmaModule.forward, blocks.py:410-416 -> (mu, std)."""
    h = mlp(inputs, p, prefix + '.nnet', n_hidden, nonlin)
    mu = linear(h, p, prefix + '.encoder_mu.linear_mu')
    if constrain_means:
        mu = torch.sigmoid(mu)
    std = F.softplus(linear(h, p, prefix + '.encoder_sg.linear_sg')) + 1e-3
    return mu, std


# EXTENSION: the data decoders src/DrVAE.py:124-129 names but src/blocks.py never defines.  Same
# anatomy as CategoricalDecoder (MLP trunk ``nnet`` + one Linear head + activation, a 1-element list returned):
def bernoulli(inputs, p, prefix, n_hidden, nonlin):
    """[clamp(sigmoid(linear_p(h)), 1e-10, 1-1e-10)] -- the probability clamp of blocks.py:463"""
    h = mlp(inputs, p, prefix + '.nnet', n_hidden, nonlin)
    return [torch.clamp(torch.sigmoid(linear(h, p, prefix + '.decoder_p.linear_p')), min=1e-10, max=1. - 1e-10)]


def bernoulli_logp_rows(x, ps):
    return (x * torch.log(ps) + (1. - x) * torch.log(1. - ps)).sum(1)


POISSON_RATE_SHIFT = 1e-6


def poisson(inputs, p, prefix, n_hidden, nonlin):
    """[softplus(linear_r(h)) + 1e-6]: strictly positive rates"""
    h = mlp(inputs, p, prefix + '.nnet', n_hidden, nonlin)
    return [F.softplus(linear(h, p, prefix + '.decoder_r.linear_r')) + POISSON_RATE_SHIFT]


def poisson_logp_rows(x, rate):
    return (x * torch.log(rate) - rate - torch.lgamma(x + 1.)).sum(1)


def categorical(inputs, p, prefix, n_hidden, nonlin, reconstruction_dim):
    """CategoricalDecoder.forward, blocks.py:456-463 -> clamped class probabilities."""
    h = mlp(inputs, p, prefix + '.nnet', n_hidden, nonlin)
    a = linear(h, p, prefix + '.decoder_p.linear_p')
    if reconstruction_dim > 1:
        ps = torch.softmax(a, dim=-1)
    else:
        s = torch.sigmoid(a)
        ps = torch.cat((1. - s, s), 1)
    return torch.clamp(ps, min=1e-10, max=1. - 1e-10)


def categorical_logp_rows(x, ps):
    """blocks.py:473-474: -nll_loss(log ps, labels, reduce=False)."""
    return -F.nll_loss(ps.log(), x.reshape(-1).long(), reduction='none')


def categorical_entropy(ps):
    """blocks.py:476-477."""
    return -(ps * torch.log(ps)).sum()


def categorical_kl_elem(ps, prior):
    """blocks.py:479-480 (elementwise; callers sum over dim 1)."""
    return -ps * (torch.log(prior).expand_as(ps) - torch.log(ps))


def categorical_most_probable(ps):
    """blocks.py:485-486."""
    return torch.max(ps, dim=1)[1]


# ------------------------------------------------------------------------------ MMD
def mmd_identity(x1, x2):
    """blocks.py:37-38."""
    return ((x1.mean(0) - x2.mean(0)) ** 2).sum()


def mmd_poly(x1, x2, degree=2, gamma=1., bias=1.):
    """blocks.py:34-35."""
    return torch.pow(gamma * x1.mm(x2.t()) + bias, degree)


def mmd_fourier(x1, x2, rnd_a, rnd_b, bandwidth=2.):
    """blocks.py:40-55 with the random features passed in explicitly
    (rnd_a ~ N(0,1) (Z,dim_r), rnd_b ~ U(0,1) (dim_r); drawn in that order)."""
    z = x1.size(1)
    dim_r = rnd_a.size(1)
    rW = math.sqrt(2. / bandwidth) * rnd_a / math.sqrt(z)
    rb = 2 * math.pi * rnd_b
    c = math.sqrt(2. / dim_r)
    rf0 = c * torch.cos(x1.mm(rW) + rb.expand(x1.size(0), dim_r))
    rf1 = c * torch.cos(x2.mm(rW) + rb.expand(x2.size(0), dim_r))
    return ((rf0.mean(0) - rf1.mean(0)) ** 2).sum()


MMD_BANDWIDTHS = 1. / (2 * (np.array([1., 2., 5., 8., 10]) ** 2))


def mmd_objective(x1, x2, kernel='rbf', bandwidths=MMD_BANDWIDTHS, rnd_a=None, rnd_b=None):
    """blocks.py:59-76.  'rbf' raises on torch>=0.4 in the reference (squeeze_(2) on a
    2-D tensor, blocks.py:32) so it is not restated."""
    if kernel == 'identity':
        return torch.sqrt(mmd_identity(x1, x2))
    if kernel == 'rbf_fourier':
        return torch.sqrt(mmd_fourier(x1, x2, rnd_a, rnd_b, bandwidth=2.))
    if kernel != 'poly':
        raise NotImplementedError(kernel)
    a = b = c = 0
    for bw in bandwidths:
        a = a + mmd_poly(x1, x1, gamma=math.sqrt(x1.size(1)) * bw) / len(bandwidths)
        c = c + mmd_poly(x2, x2, gamma=math.sqrt(x2.size(1)) * bw) / len(bandwidths)
        b = b + mmd_poly(x1, x2, gamma=math.sqrt(x1.size(1)) * bw) / len(bandwidths)
    return torch.sqrt(a.mean() - 2 * b.mean() + c.mean())


def mmd_criterion(z, sind, kernel='rbf_fourier', normals=(), uniforms=()):
    """DGMMixin.py:42-66: minus the MMD between the latent rows of every category of the nuisance variable
    and the rows outside it, averaged over the categories (two categories: the first pair only).  A side
    without rows is replaced by ONE random N(0,1) row.  ``normals`` / ``uniforms``: the draws in the order
    the reference consumes them (per category: [the random row], W ~ N(0,1), b ~ U(0,1)).  The reference
    needs two missing imports supplied to run at all (pinned by tests/golden/make_golden.py)."""
    normals, uniforms = list(normals), list(uniforms)
    total = 0.
    for ind in sind:
        i0 = torch.nonzero(ind.reshape(-1) != 0).reshape(-1)
        i1 = torch.nonzero(ind.reshape(-1) == 0).reshape(-1)
        z0 = z.index_select(0, i0) if i0.numel() else normals.pop(0)
        z1 = z.index_select(0, i1) if i1.numel() else normals.pop(0)
        if kernel == 'rbf_fourier':
            m = mmd_objective(z0, z1, kernel, rnd_a=normals.pop(0), rnd_b=uniforms.pop(0))
        else:
            m = mmd_objective(z0, z1, kernel)
        total = total - m
        if len(sind) == 2:
            return total
    return total / len(sind)


# ------------------------------------------------------------------- train-step bits
def free_bits(kl_rows, kl_min=2.0):
    """DGMMixin.py:68-75: max(KL_row, kl_min) on the per-row KL."""
    return torch.max(kl_rows, torch.tensor([kl_min]).expand_as(kl_rows))


def anneal_coef(iter_num, iter_max=1000, iter_offset=0):
    """DGMMixin.py:77-89."""
    if iter_num - iter_offset > 0:
        return min(1., 0.01 + (iter_num - iter_offset) / (1. * iter_max))
    return 0.01
