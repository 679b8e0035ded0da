"""Deterministic inputs for the golden cases (shared by ``make_golden.py``, which runs
the reference on them, and by the tests, which run the oracle / the HIP path on them).
All randomness is ``np.random.RandomState`` (legacy generator: stream frozen across
NumPy versions), so inputs regenerate identically on the GPU box."""
import os
from collections import OrderedDict

import numpy as np

from oracle import models_ref as M

HERE = os.path.dirname(os.path.abspath(__file__))


def sample_index(n):
    """16 fixed positions in a flattened array of n elements."""
    return (np.arange(16, dtype=np.int64) * 7919 + 13) % n


def load(name):
    with np.load(os.path.join(HERE, name + '.npz')) as z:
        return {k: z[k] for k in z.files}


# ----------------------------------------------------------------------- block cases
def _lin(out, prefix, n_in, n_out, wn):
    out[prefix + '.weight'] = (n_out, n_in)
    out[prefix + '.bias'] = (n_out,)
    if wn:
        out[prefix + '.g'] = (n_out,)


def _fill(shapes, seed, scale=None):
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    fan = 1
    for k, shp in shapes.items():
        if k.endswith('.g'):
            a = rs.uniform(0.5, 1.5, shp)
        elif k.endswith('W_mu') or k.endswith('bias_mu'):
            a = rs.uniform(-0.3, 0.3, shp)
        else:
            if k.endswith('.weight'):
                fan = shp[1]
            a = rs.uniform(-1, 1, shp) * (scale or 1.0) / np.sqrt(fan)
        out[k] = a.astype(np.float32)
    return out


def block_inputs(tag):
    seed = sum(ord(ch) * (i + 1) for i, ch in enumerate(tag))
    rs = np.random.RandomState(seed)
    f = lambda *s: rs.standard_normal(s).astype(np.float32)
    n = 7
    sh = OrderedDict()
    if tag == 'G1':
        _lin(sh, '', 13, 5, True)
        sh = OrderedDict((k[1:], v) for k, v in sh.items())
        return dict(params=_fill(sh, seed), x=f(n, 13), dy=f(n, 5))
    if tag in ('G2d', 'G2e'):      # the MLP's batch_norm / hidden-dropout options (src/blocks.py:135-151)
        prm = OrderedDict()

        def bn(prefix, width):
            prm[prefix + '.weight'] = rs.uniform(0.5, 1.5, width).astype(np.float32)
            prm[prefix + '.bias'] = (0.3 * rs.standard_normal(width)).astype(np.float32)
            prm[prefix + '.running_mean'] = (0.2 * rs.standard_normal(width)).astype(np.float32)
            prm[prefix + '.running_var'] = rs.uniform(0.5, 2.0, width).astype(np.float32)
            prm[prefix + '.num_batches_tracked'] = np.asarray(3, np.int64)

        def lin(prefix, n_in, n_out):
            prm[prefix + '.weight'] = (rs.uniform(-1, 1, (n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
            prm[prefix + '.bias'] = (rs.uniform(-1, 1, n_out) / np.sqrt(n_in)).astype(np.float32)
        if tag == 'G2d':
            bn('model.bn_input', 13)
        lin('model.linear1', 13, 11)
        if tag == 'G2d':
            bn('model.bn1', 11)
        lin('model.linear2', 11, 6)
        if tag == 'G2d':
            bn('model.bn2', 6)
        return dict(params=prm, xa=f(n, 9), xb=f(n, 4), dy=f(n, 6), seed=seed)
    if tag.startswith('G2'):
        wn = tag in ('G2b', 'G2c')
        _lin(sh, 'model.linear1', 13, 11, wn)
        _lin(sh, 'model.linear2', 11, 6, wn)
        return dict(params=_fill(sh, seed), xa=f(n, 9), xb=f(n, 4), dy=f(n, 6))
    if tag.startswith('G3'):
        wn = tag == 'G3b'
        _lin(sh, 'nnet.model.linear1', 13, 11, wn)
        _lin(sh, 'encoder_mu.linear_mu', 11, 5, wn)
        _lin(sh, 'encoder_lv.linear_lv', 11, 5, wn)
        return dict(params=_fill(sh, seed), xa=f(n, 9), xb=f(n, 4), eps=f(n, 5), s=f(n, 5),
                    mu_p=f(n, 5), lv_p=(0.5 * f(n, 5)))
    if tag.startswith('G4'):
        wn = tag == 'G4b'
        _lin(sh, 'nnet.model.linear1', 5, 11, wn)
        _lin(sh, 'encoder_mu.linear_mu', 11, 17, wn)
        _lin(sh, 'encoder_sg.linear_sg', 11, 17, wn)
        return dict(params=_fill(sh, seed, 2.0), z=f(n, 5), x=f(n, 17), eps=f(n, 17), mu_p=f(n, 17),
                    sd_p=np.abs(f(n, 17)) + 0.2)
    if tag.startswith('G5'):
        sh['W_mu'] = (5, 5)
        sh['bias_mu'] = (5,)
        _lin(sh, 'encoder_lv.linear_lv', 5, 5, False)
        return dict(params=_fill(sh, seed), z=f(n, 5))
    if tag in ('G6a', 'G6b'):
        rdim = 3 if tag == 'G6a' else 1
        _lin(sh, 'decoder_p.linear_p', 10, rdim, False)
        ncls = 3 if tag == 'G6a' else 2
        prior = np.full((n, ncls), 1.0 / ncls, np.float32)
        return dict(params=_fill(sh, seed, 3.0), za=f(n, 5), zb=f(n, 5),
                    y=rs.randint(0, ncls, (n, 1)).astype(np.int64), prior=prior)
    if tag == 'G6c':
        _lin(sh, 'nnet.model.linear1', 5, 7, False)
        _lin(sh, 'decoder_p.linear_p', 7, 2, False)
        # huge logits so that softmax saturates and the 1e-10 clamp is active
        return dict(params=_fill(sh, seed, 60.0), za=3 * f(n, 5), y=rs.randint(0, 2, (n, 1)).astype(np.int64),
                    prior=np.tile(np.array([[0.3, 0.7]], np.float32), (n, 1)))
    if tag == 'G7':
        return dict(x1=f(6, 5), x2=0.5 * f(9, 5) + 0.3, rnd_a=f(5, 500),
                    rnd_b=rs.uniform(0, 1, 500).astype(np.float32))
    if tag == 'G8':
        return dict(y=np.array([[0], [3], [1], [1], [2]], np.int64))
    if tag == 'G9':
        return dict(kl=np.array([0.1, 1.999, 2.0, 2.001, 7.5, -1.0], np.float32),
                    anneal_args=[(0, 1, 0), (1, 1, 0), (5, 100, 0), (5, 100, 5), (6, 100, 5), (2000, 1000, 0)])
    if tag in ('G10a', 'G10b'):   # reconstruction metrics: (rows, genes) = (40, 13) and (150, 978)
        m, x = (40, 13) if tag == 'G10a' else (150, 978)
        xx = (f(m, x) * 1.3 + 0.4).astype(np.float32)
        return dict(x=xx, x_rec=(0.8 * xx + 0.5 * f(m, x)).astype(np.float32),
                    std=(np.abs(f(m, x)) * 0.3 + 0.2).astype(np.float32))
    if tag == 'G11':   # dataset wrapping / sampler weights
        def side(n, paired, off):
            d = {'x1': f(n, 6), 's': np.zeros(n), 'y': rs.randint(0, 2, n), 'ycont': rs.rand(n),
                 'has_y': (rs.rand(n) < 0.6).astype(np.int64), 'cid': off + np.arange(n) % 5}
            if paired:
                d['x2'] = f(n, 6)
            return d
        return dict(sing=side(9, False, 0), pair=side(6, True, 3),
                    labels=np.array([3, 3, 7, 7, 7, 7, 1, -1, -1, -1, 3, 1]), ratio=0.25, token=-1)
    raise KeyError(tag)


# ----------------------------------------------------------------------- fit policy / y metrics
def fit_policy_cases():
    """scripted per-epoch validation objectives for the early-stopping / snapshot policy of ``fit``"""
    rs = np.random.RandomState(77)
    up = 1.0 + 0.01 * np.arange(200)
    plateau = np.concatenate([np.linspace(0.5, 1.5, 12), 1.5 - 0.002 * np.arange(188)])
    noisy = 1.0 + 0.3 * np.sin(np.arange(200) / 7.0) + 0.1 * rs.standard_normal(200) + 0.004 * np.arange(200)
    late = np.concatenate([np.full(45, 1.0), 1.0 + 0.05 * np.arange(30), np.full(125, 2.0)])
    withnan = plateau.copy()
    withnan[[3, 20, 21]] = np.nan
    negative = -2.0 + 0.01 * np.arange(200)     # threshold test multiplies by 0.999: sign matters
    out = OrderedDict()
    for kind in ('drvae', 'pvae', 'vfae'):
        for nm, objs in (('up', up), ('plateau', plateau), ('noisy', noisy), ('late', late)):
            for es in (True, False):
                out['%s_%s_%s' % (kind, nm, 'es' if es else 'noes')] = dict(
                    kind=kind, epochs=120, objs=objs, early_stop=es, n_batches=3)
    out['drvae_nan_es'] = dict(kind='drvae', epochs=120, objs=withnan, early_stop=True, n_batches=2)
    out['drvae_neg_es'] = dict(kind='drvae', epochs=150, objs=negative, early_stop=True, n_batches=1)
    out['vfae_short_es'] = dict(kind='vfae', epochs=10, objs=plateau, early_stop=True, n_batches=2)
    out['pvae_short_noes'] = dict(kind='pvae', epochs=10, objs=plateau, early_stop=False, n_batches=2)
    out['drvae_interrupt'] = dict(kind='drvae', epochs=30, objs=up, early_stop=True, n_batches=4, interrupt_at=50)
    out['drvae_interrupt_first'] = dict(kind='drvae', epochs=30, objs=plateau, early_stop=False, n_batches=4,
                                        interrupt_at=3)
    return out


def sampler_cases():
    """imbalanced 'cell line' ids of a training set; batch sizes as the drivers use them"""
    rs = np.random.RandomState(77)
    out = OrderedDict()
    cid = np.concatenate([np.full(n, c) for c, n in enumerate([3, 40, 7, 120, 25, 60, 11, 134])])
    out['cid8'] = dict(cid=rs.permutation(cid).astype(np.int64), batch_size=48, epochs=300)
    out['small'] = dict(cid=np.array([0, 0, 0, 1, 1, 2, 2, 2, 2, 2], np.int64), batch_size=16, epochs=50)     # len < batch
    out['exact'] = dict(cid=(np.arange(96) % 5).astype(np.int64), batch_size=32, epochs=100)                    # len % batch == 0
    return out


def masked_linear_cases():
    """stacks of (in_features, out_features, output_layer, rev_order); layer i > 0 is built on the ``get_m()`` of
    layer i - 1 (the first on ``m_pre=None``)"""
    out = OrderedDict()
    out['made_6'] = [(6, 9, False, False), (9, 4, False, False), (4, 6, True, False)]
    out['made_6_rev'] = [(6, 9, False, True), (9, 6, True, True)]
    out['made_cond'] = [((5, 3), 12, False, False), (12, 7, False, False), (7, 5, True, False)]
    out['made_cond_rev'] = [([4, 2, 1], 8, False, True), (8, 4, True, True)]
    out['made_narrow'] = [(7, 3, False, False), (3, 7, True, False)]
    return out


def y_metric_cases():
    rs = np.random.RandomState(4242)
    out = OrderedDict()
    n = 57
    y = rs.randint(0, 2, n).astype(np.int64)
    p1 = np.clip(0.5 + 0.25 * (2 * y - 1) * rs.rand(n) + 0.2 * rs.standard_normal(n), 0.01, 0.99).astype(np.float32)
    out['Y2'] = dict(ylab=y, proba=np.stack([1 - p1, p1], 1), pred=(p1 > 0.5).astype(np.int64))
    pt = np.round(p1 * 8) / 8                      # heavy ties in the scores
    out['Y2ties'] = dict(ylab=y, proba=np.stack([1 - pt, pt], 1).astype(np.float32), pred=(pt > 0.5).astype(np.int64))
    y3 = rs.randint(0, 3, n).astype(np.int64)
    lg = rs.standard_normal((n, 3)) + 1.5 * np.eye(3)[y3]
    p3 = (np.exp(lg) / np.exp(lg).sum(1, keepdims=True)).astype(np.float32)
    out['Y3'] = dict(ylab=y3, proba=p3, pred=p3.argmax(1).astype(np.int64))
    yc = rs.rand(n).astype(np.float32)
    out['Ycont'] = dict(ylab=yc, pred=(0.6 * yc + 0.2 + 0.15 * rs.standard_normal(n)).astype(np.float32)[:, None],
                        proba=np.full((n, 1), np.log(0.05 ** 2), np.float32), cont=True)
    out['Y2oneclass'] = dict(ylab=np.ones(9, np.int64), proba=out['Y2']['proba'][:9].copy(),
                             pred=out['Y2']['pred'][:9].copy())
    return out


# ----------------------------------------------------------------------- model cases
def tiny_spec(kind, **over):
    kw = dict(kind=kind, dim_x=13, dim_y=2, dim_z1=5, dim_z3=4, h_en_z1=[7], h_de_z1=[6], h_en_z3=[6],
              h_de_x=[8], h_clf=[], L=2, learning_rate=5e-3)
    kw.update(over)
    return M.ModelSpec(**kw)


def _flags(pattern):
    """pattern: string of 'a' (labeled single) 'b' (unlabeled single) 'c' (labeled pair) 'd' (unlabeled pair)"""
    has_y = np.array([ch in 'ac' for ch in pattern], np.int64)
    has_x2 = np.array([ch in 'cd' for ch in pattern], np.int64)
    return has_y, has_x2


MODEL_CASES = OrderedDict([
    # name: (spec factory, rows, group pattern or None, n_steps, full-output?)
    ('tiny_drvae', (lambda: tiny_spec('drvae'), 'acbdaabcdbacab', 3, True)),
    ('tiny_drvae_nolp', (lambda: tiny_spec('drvae', dim_y=3, h_clf=[3], L=3), 'abdbadabbdaa', 2, True)),
    ('tiny_drvae_wn', (lambda: tiny_spec('drvae', weight_norm=True), 'cadbcabdbca', 2, True)),
    ('tiny_drvae_only_up', (lambda: tiny_spec('drvae', L=1), 'ddddd', 2, True)),
    ('tiny_drvae_adamax', (lambda: tiny_spec('drvae', optim_alg='adamax'), 'acbdaabcdbacab', 3, True)),
    ('tiny_drvae_prior', (lambda: tiny_spec('drvae', dim_y=3, prior_y=[0.2, 0.5, 0.3]), 'abdbcdabbdac', 2, True)),
    ('tiny_drvae_1sig', (lambda: tiny_spec('drvae', clf_1sig=True, h_clf=[3]), 'cadbcabdbca', 2, True)),
    ('tiny_vfae_prior_1sig', (lambda: tiny_spec('vfae', clf_1sig=True, prior_y=[0.7, 0.3]), 'abbabaabbb', 2, True)),
    ('tiny_drvae_cont', (lambda: tiny_spec('drvae', type_y='cont', dim_y=1), 'acbdaabcdbacab', 3, True)),
    ('tiny_vfae_cont_sup', (lambda: tiny_spec('vfae', type_y='cont', dim_y=1, semi_supervised=False, h_clf=[3]),
                            'aababaaa', 2, True)),
    ('tiny_pvae', (lambda: tiny_spec('pvae'), 'bdbbdddbdb', 3, True)),
    ('tiny_vfae', (lambda: tiny_spec('vfae', dim_y=3), 'abbabaabbb', 3, True)),
    ('tiny_vfae_sup', (lambda: tiny_spec('vfae', semi_supervised=False, add_noise_var=0.), 'aababaaa', 2, True)),
    ('cfg1_pvae', (lambda: M.ModelSpec(kind='pvae', L=1), 150, 3, False)),
    ('cfg2_drvae', (lambda: M.ModelSpec(kind='drvae', L=2), 150, 3, False)),
    ('cfg4_vfae', (lambda: M.ModelSpec(kind='vfae', L=2, add_noise_var=0.), 150, 3, False)),
    # BASELINE.json configs[4] at its per-GPU size (the MFMA-bound stress configuration; ~25 s per step on the CPU)
    ('cfg5_wide', (lambda: M.ModelSpec(kind='drvae', L=4, dim_x=20000, dim_z1=200, dim_z3=200, h_en_z1=[2048],
                                       h_de_x=[2048]), 1024, 2, False)),
])


# the cases small enough for the CPU stand-in kernels of tests/test_engine_cpu.py (python loops over rows)
SMALL_MODEL_CASES = [n for n in MODEL_CASES if n != 'cfg5_wide']


def model_case(name):
    mk, rows, steps, full = MODEL_CASES[name]
    spec = mk()
    seed = sum(ord(ch) * (i + 1) for i, ch in enumerate(name))
    if isinstance(rows, str):
        batch = M.make_batch(spec, len(rows), seed=seed)
        has_y, has_x2 = _flags(rows)
        if spec.kind == 'pvae':
            has_y[:] = 0
        if spec.kind == 'vfae':
            has_x2[:] = 0
        # rebuild x2 for the overridden pairing (zero-imputed singletons)
        rs = np.random.RandomState(seed + 1)
        x2 = (batch['x1'] + 0.1 * rs.standard_normal(batch['x1'].shape)).astype(np.float32)
        batch['x2'] = x2 * has_x2[:, None].astype(np.float32)
        batch['has_y'], batch['has_x2'] = has_y, has_x2
        n = len(rows)
    else:
        n = rows
        batch = M.make_batch(spec, n, seed=1234)
    if spec.type_y == 'cont':          # regression targets in (0,1) (the head's means are sigmoid-constrained)
        batch['y'] = np.random.RandomState(seed + 2).rand(n, spec.dim_y).astype(np.float32)
    noises = [M.make_noise(spec, n, seed=seed + 100 + i) for i in range(steps)]
    return dict(name=name, spec=spec, batch=batch, noises=noises, param_seed=123, full=full)


def mmd_criterion_cases():
    """inputs of ``DGMMixin._get_mmd_criterion`` (src/DGMMixin.py:42-66): latent rows z, one 0/1 indicator
    vector per category of the nuisance variable s, and the N(0,1) / U(0,1) draws in the order the reference
    consumes them (per category: [a random row when the category or its complement is empty], W, b)."""
    rs = np.random.RandomState(909)
    n, z, dim_r = 12, 5, 500
    out = OrderedDict()
    lab2 = np.array([1, 0, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1])
    lab3 = np.array([0, 1, 2, 0, 1, 0, 2, 1, 0, 2, 1, 0])
    lab3e = np.array([0, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0])       # category 2 has no rows

    def draws(k, empties=()):
        nor, uni = [], []
        for c in range(k):
            if c in empties:
                nor.append(rs.standard_normal((1, z)).astype(np.float32))
            nor.append(rs.standard_normal((z, dim_r)).astype(np.float32))
            uni.append(rs.rand(dim_r).astype(np.float32))
        return nor, uni
    for tag, lab, k, kernel, empties, ndraw in (('two', lab2, 2, 'rbf_fourier', (), 1), ('three', lab3, 3, 'rbf_fourier', (), 3),
                                                ('three_empty', lab3e, 3, 'rbf_fourier', (2,), 3),
                                                ('three_identity', lab3, 3, 'identity', (), 0)):
        nor, uni = draws(ndraw, tuple(e for e in empties)) if ndraw else ([], [])
        if tag == 'two':        # (two categories: the reference returns after the first one)
            sind = [(lab == 1).astype(np.int64), (lab == 0).astype(np.int64)]
        else:
            sind = [(lab == c).astype(np.int64) for c in range(k)]
        out[tag] = dict(z=rs.standard_normal((n, z)).astype(np.float32), sind=sind, kernel=kernel, normals=nor,
                        uniforms=uni)
    return out
