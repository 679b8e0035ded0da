"""CPU: the fused step engine's orchestration (stacked forward, HAND-WRITTEN backward,
loss coefficients, Adam wiring) checked against the golden vectors of the reference and
against the oracle, with the HIP launchers replaced by their plain-PyTorch references
(tests/kernel_ref.py).  The kernels themselves are checked on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import models_ref as M
from tests import kernel_ref
from tests.golden import cases as C


def make_engine(spec, params, device='cpu'):
    from drvae_amd import engine as E
    from drvae_amd.arena import ParamArena
    cfg = E.StepConfig(**{k: getattr(spec, k) for k in E.StepConfig.__dataclass_fields__ if hasattr(spec, k)})
    shapes = E.param_shapes(cfg)
    assert list(shapes.items()) == [(k, tuple(v)) for k, v in M.param_shapes(spec).items()]
    arena = ParamArena(shapes, device, frozen=E.frozen_params(cfg))
    arena.load(params)
    return E.FusedStep(cfg, arena), arena


def set_batch(eng, batch, dev='cpu', counts=None):
    t = lambda k: torch.from_numpy(batch[k].copy()).to(dev)
    return eng.set_batch(t('x1'), t('x2'), batch['y'], batch['has_x2'], batch['has_y'], counts=counts)


def close(a, b, rtol, atol):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', C.SMALL_MODEL_CASES)
def test_engine_matches_reference_golden(name, monkeypatch):
    kernel_ref.install(monkeypatch)
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True))
    set_batch(eng, case['batch'])
    # eval-mode loss (no input noise)
    eng.training = False
    eng.set_noise(case['noises'][0])
    eng.forward()
    for k, v in eng.losses().items():
        close(v, gold['eval/' + k], 2e-5, 2e-6)
    nsteps = len(case['noises'])
    for step, noise in enumerate(case['noises']):
        eng.train_step(noise)
        for k, v in eng.losses().items():
            close(v, gold['step%d/%s' % (step, k)], 2e-5, 2e-6)
        if step == 0:
            # gradients of step 0 are still in the arena
            pass
        if step in (0, nsteps - 1):
            for k in arena.shapes:
                a = arena.p(k).numpy()
                if case['full']:
                    close(a, gold['param%d/%s' % (step, k)], 1e-4, 2e-5)
                else:
                    close(a.astype(np.float64).sum(), gold['paramsum%d/%s' % (step, k)], 1e-4, 2e-3)
                    close(a.reshape(-1)[C.sample_index(a.size)], gold['paramsample%d/%s' % (step, k)], 1e-4, 2e-5)
    assert eng.iters == nsteps


@pytest.mark.parametrize('name', C.SMALL_MODEL_CASES)
def test_engine_gradients_match_reference_golden(name, monkeypatch):
    kernel_ref.install(monkeypatch)
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True))
    set_batch(eng, case['batch'])
    eng.training = True
    eng.set_noise(case['noises'][0])
    eng.forward()
    eng.backward()
    for k in arena.shapes:
        g = arena.g(k).numpy()
        if case['full']:
            ref = gold['grad/' + k]
            close(g, ref, 3e-4, 3e-6 * max(1.0, float(np.abs(ref).max())))
        else:
            close(np.sqrt((g.astype(np.float64) ** 2).sum()), gold['gradnorm/' + k], 1e-4, 1e-7)
            close(g.reshape(-1)[C.sample_index(g.size)], gold['gradsample/' + k], 2e-3, 2e-6)


@pytest.mark.parametrize('name', ['tiny_drvae', 'tiny_pvae'])
def test_raw_decoder_heads_match_reference_golden(name, monkeypatch):
    """the wide configuration's train-step path at a size the CPU mirror handles: the decoder heads as a plain product,
    bias + softplus + shift applied by the NLL row pass (``nll_rows_fwdbwd(bias=...)``) -- same golden losses / updates"""
    from drvae_amd import tuning as T
    monkeypatch.setenv('DRVAE_TUNE', 'fuse_heads=0,raw_heads=2')
    T.reload()
    try:
        kernel_ref.install(monkeypatch)
        calls = []
        real = kernel_ref.nll_rows_fwdbwd
        monkeypatch.setattr('drvae_amd.kernels.nll_rows_fwdbwd', lambda *a, **k: (calls.append(k.get('bias') is not None), real(*a, **k))[1])
        case, gold = C.model_case(name), C.load('model_' + name)
        spec = case['spec']
        eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True))
        set_batch(eng, case['batch'])
        for step, noise in enumerate(case['noises']):
            eng.train_step(noise)
            for k, v in eng.losses().items():
                close(v, gold['step%d/%s' % (step, k)], 2e-5, 2e-6)
        nsteps = len(case['noises'])
        for k in arena.shapes:
            close(arena.p(k).numpy(), gold['param%d/%s' % (nsteps - 1, k)], 1e-4, 2e-5)
        assert calls and all(calls)
    finally:
        monkeypatch.delenv('DRVAE_TUNE')
        T.reload()


UNIVERSAL_CASES = [n for n in C.SMALL_MODEL_CASES if 'cont' not in n and n not in ('tiny_vfae_sup',)]


@pytest.mark.parametrize('name', UNIVERSAL_CASES)
def test_universal_plan_matches_reference_golden(name, monkeypatch):
    """N2: the batch-independent plan (every row materialised as a pair with all class slots; group membership as
    device-side masks from dv_batch_masks) reproduces the reference's losses / gradients / parameters for every
    batch composition of the golden cases -- incl. empty groups -- with ONE plan."""
    kernel_ref.install(monkeypatch)
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True))
    eng.universal = True
    p = set_batch(eng, case['batch'])
    assert p.universal and p.key[0] == 'universal'
    eng.training = False
    eng.set_noise(case['noises'][0])
    eng.forward()
    for k, v in eng.losses().items():
        close(v, gold['eval/' + k], 2e-5, 2e-6)
    # gradients of the first train-mode pass
    eng.training = True
    eng.set_noise(case['noises'][0])
    eng.forward()
    eng.backward()
    for k in arena.shapes:
        g = arena.g(k).numpy()
        if case['full']:
            ref = gold['grad/' + k]
            close(g, ref, 3e-4, 3e-6 * max(1.0, float(np.abs(ref).max())))
        else:
            close(np.sqrt((g.astype(np.float64) ** 2).sum()), gold['gradnorm/' + k], 1e-4, 1e-7)
    nsteps = len(case['noises'])
    for step, noise in enumerate(case['noises']):
        eng.train_step(noise)
        for k, v in eng.losses().items():
            close(v, gold['step%d/%s' % (step, k)], 2e-5, 2e-6)
    for k in arena.shapes:
        a = arena.p(k).numpy()
        if case['full']:
            close(a, gold['param%d/%s' % (nsteps - 1, k)], 1e-4, 2e-5)
        else:
            close(a.astype(np.float64).sum(), gold['paramsum%d/%s' % (nsteps - 1, k)], 1e-4, 2e-3)
    assert len(eng._plans) == 1


def test_universal_plan_one_plan_for_any_composition(monkeypatch):
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec('drvae')
    params = M.init_params(spec, 3, as_numpy=True)
    uni, au = make_engine(spec, params)
    uni.universal = True
    for seed, pattern in enumerate(['aabbccdd', 'aaaaaaab', 'dddddddd', 'cdcdabab', 'bbbbaaaa']):
        batch = M.make_batch(spec, 8, seed=seed)
        fl = {'a': (1, 0), 'b': (0, 0), 'c': (1, 1), 'd': (0, 1)}
        batch['has_y'] = np.array([fl[c][0] for c in pattern], np.int64)
        batch['has_x2'] = np.array([fl[c][1] for c in pattern], np.int64)
        batch['x2'] = batch['x2'] * batch['has_x2'][:, None].astype(np.float32)
        noise = M.make_noise(spec, 8, seed=10 + seed)
        one, a1 = make_engine(spec, params)
        a1.param.copy_(au.param); a1.exp_avg.copy_(au.exp_avg); a1.exp_avg_sq.copy_(au.exp_avg_sq)
        one.step_dev.copy_(uni.step_dev); one.iters = uni.iters
        for e in (uni, one):
            set_batch(e, batch)
            e.train_step(noise)
        for (k, a), b in zip(uni.losses().items(), one.losses().values()):
            close(a, b, 2e-5, 2e-6)
        close(au.grad, a1.grad.numpy(), 2e-4, 1e-6)
        close(au.param, a1.param.numpy(), 1e-5, 1e-6)
    assert len(uni._plans) == 1


@pytest.mark.parametrize('kind', ['drvae', 'pvae'])
def test_universal_plan_with_pair_slots(kind, monkeypatch):
    """the bucketed sampler feed's plans: pairs first, pair slots for the first n rows only (n >= the batch's pairs) --
    same losses / gradients / update as the plan built for the batch's exact composition"""
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec(kind)
    params = M.init_params(spec, 3, as_numpy=True)
    uni, au = make_engine(spec, params)
    uni.universal = True
    for seed, (pattern, slots) in enumerate([('ccddabab', 4), ('cdcaabbb', 4), ('dcbbaabb', 2), ('aabbaabb', 2),
                                             ('dddddddc', 8), ('cdabbbaa', 6)]):
        batch = M.make_batch(spec, 8, seed=seed)
        fl = {'a': (1, 0), 'b': (0, 0), 'c': (1, 1), 'd': (0, 1)}
        batch['has_y'] = np.array([fl[c][0] for c in pattern], np.int64)
        batch['has_x2'] = np.array([fl[c][1] for c in pattern], np.int64)
        batch['x2'] = batch['x2'] * batch['has_x2'][:, None].astype(np.float32)
        noise = M.make_noise(spec, 8, seed=10 + seed)
        one, a1 = make_engine(spec, params)
        a1.param.copy_(au.param); a1.exp_avg.copy_(au.exp_avg); a1.exp_avg_sq.copy_(au.exp_avg_sq)
        one.step_dev.copy_(uni.step_dev); one.iters = uni.iters
        uni.universal_pair_slots = slots
        for e in (uni, one):
            p = set_batch(e, batch)
            e.train_step(noise)
        assert uni.plan.universal and uni.plan.Np == slots and uni.plan.DPX.shape[0] == uni.cfg.L * (8 + 2 * slots)
        for (k, a), b in zip(uni.losses().items(), one.losses().values()):
            close(a, b, 2e-5, 2e-6)
        close(au.grad, a1.grad.numpy(), 2e-4, 1e-6)
        close(au.param, a1.param.numpy(), 1e-5, 1e-6)
    assert len(uni._plans) == 4


@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_universal_plan_with_labeled_range(kind, monkeypatch):
    """the bucketed sampler feed's plans, label dimension: rows [a, b) of the batch are labeled for sure and get one
    fprop row (their class) instead of one per class -- same losses / gradients / update as the plan built for the
    batch's exact composition.  Row order of the feed: unlabeled pairs, labeled pairs, labeled singles, unlabeled
    singles (pairs are a prefix, labeled rows one contiguous run)."""
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec(kind)
    params = M.init_params(spec, 3, as_numpy=True)
    uni, au = make_engine(spec, params)
    uni.universal = True
    fl = {'a': (1, 0), 'b': (0, 0), 'c': (1, 1), 'd': (0, 1)}        # (labeled, pair)
    cases = [('dccaabbb', 4, (1, 5)), ('ddccabbb', 4, (2, 4)), ('dcccaaab', 4, (2, 6)), ('ccccaaaa', 4, (0, 8)),
             ('ddddbbbb', 4, (0, 0)), ('dccaabbb', 8, (2, 4))]
    for seed, (pattern, slots, lab) in enumerate(cases):
        batch = M.make_batch(spec, 8, seed=seed)
        batch['has_y'] = np.array([fl[c][0] for c in pattern], np.int64)
        batch['has_x2'] = np.array([fl[c][1] for c in pattern], np.int64)
        batch['x2'] = batch['x2'] * batch['has_x2'][:, None].astype(np.float32)
        noise = M.make_noise(spec, 8, seed=10 + seed)
        one, a1 = make_engine(spec, params)
        a1.param.copy_(au.param); a1.exp_avg.copy_(au.exp_avg); a1.exp_avg_sq.copy_(au.exp_avg_sq)
        one.step_dev.copy_(uni.step_dev); one.iters = uni.iters
        uni.universal_pair_slots, uni.universal_labeled_range = slots, lab
        for e in (uni, one):
            set_batch(e, batch)
            e.train_step(noise)
        Y = uni.cfg.dim_y
        assert uni.plan.universal and uni.plan.Mf == uni.cfg.L * (8 * Y - (lab[1] - lab[0]) * (Y - 1))
        for (k, a), b in zip(uni.losses().items(), one.losses().values()):
            close(a, b, 2e-5, 2e-6)
        close(au.grad, a1.grad.numpy(), 2e-4, 1e-6)
        close(au.param, a1.param.numpy(), 1e-5, 1e-6)
    with pytest.raises(AssertionError):           # an unlabeled row inside the labeled range
        batch['has_y'][:] = 0
        uni.universal_labeled_range = (1, 5)
        set_batch(uni, batch)


def _use_s_case(kind, use_mmd, seed=0):
    spec = C.tiny_spec(kind, use_s=True, dim_s=2, use_MMD=use_mmd, mmd_rate=0.7, kernel_MMD='identity')
    for sd in range(seed, seed + 50):      # every data group must hold both nuisance classes (no random fill-in row)
        batch = M.make_batch(spec, 24, seed=sd)
        hx, hy, s = batch['has_x2'].astype(bool), batch['has_y'].astype(bool), batch['s'].reshape(-1)
        groups = {'drvae': [hy & ~hx, ~hy & ~hx, hy & hx, ~hy & hx], 'pvae': [~hx, hx], 'vfae': [hy, ~hy]}[kind]
        if all(len(set(s[g])) == 2 for g in groups):
            return spec, batch
    raise AssertionError('no suitable batch')


@pytest.mark.parametrize('use_mmd', [False, True])
@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_use_s_extension_matches_oracle(kind, use_mmd, monkeypatch):
    """N4 (extension, no reference output exists: the reference crashes with use_s=True): one_hot(s) conditioning of
    encoder_z1 / decoder_x and the model-level MMD penalty against the oracle's restatement of the intended maths"""
    kernel_ref.install(monkeypatch)
    spec, batch = _use_s_case(kind, use_mmd)
    params = M.init_params(spec, 9, as_numpy=True)
    eng, arena = make_engine(spec, params)
    assert eng.cfg.use_s and eng.cfg.use_MMD == use_mmd
    tr = M.RefTrainer(spec, M.init_params(spec, 9))
    t = lambda k: torch.from_numpy(batch[k].copy())
    eng.set_batch(t('x1'), t('x2'), batch['y'], batch['has_x2'], batch['has_y'], s=batch['s'])
    # gradients of a train-mode pass
    noise = M.make_noise(spec, 24, seed=4)
    ref, _ = tr.loss(batch, noise, True)
    ref['CMPL'].backward()
    eng.training = True
    eng.set_noise(noise)
    eng.forward()
    eng.backward()
    for k, v in eng.losses().items():
        close(v, float(ref[k]), 2e-5, 2e-6)
    if use_mmd:
        assert abs(eng.losses()['MMD']) > 1e-4
    for k, prm in tr.params.items():
        close(arena.g(k), prm.grad.numpy(), 3e-4, 2e-6)
        prm.grad = None
    for step in range(3):
        nz = M.make_noise(spec, 24, seed=10 + step)
        want, _ = tr.step(batch, nz)
        eng.train_step(nz)
        for k, v in eng.losses().items():
            close(v, float(want[k]), 2e-5, 2e-6)
    for k, prm in tr.params.items():
        close(arena.p(k), prm.detach().numpy(), 1e-4, 2e-5)


@pytest.mark.parametrize('type_rec', ['binary', 'poisson'])
@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_bernoulli_poisson_decoders_match_oracle(kind, type_rec, monkeypatch):
    """N4 (extension: src/DrVAE.py:124-129 names BernoulliDecoder / PoissonDecoder, src/blocks.py defines neither):
    the single-head data decoders in the fused step against the oracle's maths -- losses, gradients, 3 Adam steps"""
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec(kind, type_rec=type_rec, add_noise_var=0.0)
    batch = M.make_batch(spec, 16, seed=3)
    assert set(np.unique(batch['x1'])) <= {0.0, 1.0} if type_rec == 'binary' else batch['x1'].max() > 1
    params = M.init_params(spec, 9, as_numpy=True)
    head = 'decoder_x.decoder_p.linear_p.weight' if type_rec == 'binary' else 'decoder_x.decoder_r.linear_r.weight'
    assert head in params and 'decoder_x.encoder_mu.linear_mu.weight' not in params
    eng, arena = make_engine(spec, params)
    tr = M.RefTrainer(spec, M.init_params(spec, 9))
    set_batch(eng, batch)
    noise = M.make_noise(spec, 16, seed=4)
    ref, _ = tr.loss(batch, noise, True)
    ref['CMPL'].backward()
    eng.training = True
    eng.set_noise(noise)
    eng.forward()
    eng.backward()
    for k, v in eng.losses().items():
        close(v, float(ref[k].detach()), 2e-5, 2e-6)
    for k, prm in tr.params.items():
        close(arena.g(k), prm.grad.numpy(), 3e-4, 2e-6)
        prm.grad = None
    for step in range(3):
        nz = M.make_noise(spec, 16, seed=10 + step)
        want, _ = tr.step(batch, nz)
        eng.train_step(nz)
        for k, v in eng.losses().items():
            close(v, float(want[k].detach()), 2e-5, 2e-6)
    for k, prm in tr.params.items():
        close(arena.p(k), prm.detach().numpy(), 1e-4, 2e-5)


@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_wide_step_path_bias_gradient_in_the_nll_pass(kind, monkeypatch):
    """round 5, the chip-filling step's path at a size the CPU mirror handles (tuning switches force it): the decoder heads
    as a plain product; the NLL row pass finishes them AND emits their bias gradient (per-chunk row partials, per-row-block
    column sums summed by a small colsum: ``nll_rows_raw_cs``; no column-sum pass inside the weight-gradient launch) --
    same losses and parameters as the oracle"""
    from drvae_amd import tuning as T
    monkeypatch.setenv('DRVAE_TUNE', 'fuse_heads=0,raw_heads=2,nll_cs=2')
    T.reload()
    try:
        kernel_ref.install(monkeypatch)
        import drvae_amd.kernels as K
        seen = dict(cs=0, adam=[], colsum_rows=[])
        real_cs, real_adam, real_colsum, real_pair = kernel_ref.nll_rows_raw_cs, kernel_ref.adam_l2, kernel_ref.colsum, kernel_ref.linear_bwd_pair
        monkeypatch.setattr(K, 'nll_rows_raw_cs', lambda *a, **k: (seen.__setitem__('cs', seen['cs'] + 1), real_cs(*a, **k))[1])
        monkeypatch.setattr(K, 'adam_l2', lambda p, *a, **k: (seen['adam'].append(p.numel()), real_adam(p, *a, **k))[1])
        monkeypatch.setattr(K, 'colsum', lambda out, X_, **k: (seen['colsum_rows'].append(X_.shape[0]), real_colsum(out, X_, **k))[1])
        no_db = []
        monkeypatch.setattr(K, 'linear_bwd_pair', lambda dW, db, *a, **k: (no_db.append(db is None), real_pair(dW, db, *a, **k))[1])
        spec = C.tiny_spec(kind, dim_x=2056, h_de_x=[8], dim_z1=6)       # 3 gene chunks (the last one ragged)
        n = 70                                                           # 2 row blocks of the pass (64 + ragged)
        # (a gene count that is no multiple of 4 -- 978: an evaluation pass takes the wave-per-row raw pass instead)
        spec_odd = C.tiny_spec(kind, dim_x=978, h_de_x=[8], dim_z1=6)
        eng_o, _ = make_engine(spec_odd, M.init_params(spec_odd, 4, as_numpy=True))
        b_o, n_o = M.make_batch(spec_odd, 20, seed=3), M.make_noise(spec_odd, 20, seed=28)
        set_batch(eng_o, b_o)
        eng_o.training = False
        eng_o.set_noise(n_o)
        raw_calls = []
        real_fwd = kernel_ref.nll_rows_fwd
        monkeypatch.setattr(K, 'nll_rows_fwd', lambda *a, **k: (raw_calls.append(k.get('bias') is not None), real_fwd(*a, **k))[1])
        eng_o.forward()
        ref_o, _ = M.RefTrainer(spec_odd, M.init_params(spec_odd, 4)).loss(b_o, n_o, training=False)
        for k, v in eng_o.losses().items():
            r = float(ref_o[k].detach()) if torch.is_tensor(ref_o[k]) else float(ref_o[k])
            assert abs(v - r) <= 1e-4 * max(1.0, abs(r)), ('eval 978', k, v, r)
        assert raw_calls == [True] and seen['cs'] == 0
        batch, params = M.make_batch(spec, n, seed=3), M.init_params(spec, 4, as_numpy=True)
        eng, arena = make_engine(spec, params)
        set_batch(eng, batch)
        tr = M.RefTrainer(spec, M.init_params(spec, 4))
        # an evaluation pass first: plain heads product + the forward-only row pass (no gradients, no column sums)
        noise0 = M.make_noise(spec, n, seed=29)
        eng.training = False
        eng.set_noise(noise0)
        eng.forward()
        ref0, _ = tr.loss(batch, noise0, training=False)
        for k, v in eng.losses().items():
            r = float(ref0[k].detach()) if torch.is_tensor(ref0[k]) else float(ref0[k])
            assert abs(v - r) <= 1e-4 * max(1.0, abs(r)), ('eval', k, v, r)
        assert seen['cs'] == 1 and not seen['adam']
        seen['cs'] = 0
        eng.training = True
        for step in range(3):
            noise = M.make_noise(spec, n, seed=30 + step)
            eng.train_step(noise)
            ref, _ = tr.step(batch, noise)
            for k, v in eng.losses().items():
                r = float(ref[k].detach()) if torch.is_tensor(ref[k]) else float(ref[k])
                assert abs(v - r) <= 1e-4 * max(1.0, abs(r)), (step, k, v, r)
        for k, prm in tr.params.items():
            close(arena.p(k), prm.detach().numpy(), 2e-4, 5e-5)
        assert seen['cs'] == 3 and seen['adam'] == [arena.n_live] * 3
        assert all(r <= 8 for r in seen['colsum_rows']) and no_db.count(True) == 3
    finally:
        monkeypatch.delenv('DRVAE_TUNE')
        T.reload()


@pytest.mark.parametrize('optim', ['adam', 'adamax'])
def test_pad_columns_stay_zero_through_training(optim, monkeypatch):
    """round-4 advisor: products over padded K / N are only correct while every pad column is exactly 0.  The CPU mirror
    now runs the padded forms for real (``kernel_ref.gemm``: it reads the operands' pads and writes the outputs' pads), so
    this trains a model whose every inner dimension is odd (inner dimensions >= 16 are row-padded in the arena) and checks
    parameters, gradients, both moments and the plan's activation buffers after every step -- and the oracle's losses"""
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec('drvae', dim_x=21, dim_z1=17, dim_z3=18, h_en_z1=[19], h_de_z1=[22], h_en_z3=[23], h_de_x=[25],
                       optim_alg=optim, weight_decay=0.05)
    n = 14
    batch, params = M.make_batch(spec, n, seed=2), M.init_params(spec, 6, as_numpy=True)
    eng, arena = make_engine(spec, params)
    assert len(arena.pads(arena.param)) >= 6                      # there ARE padded rows in this model
    set_batch(eng, batch)
    tr = M.RefTrainer(spec, M.init_params(spec, 6))
    for step in range(4):
        noise = M.make_noise(spec, n, seed=40 + step)
        eng.train_step(noise)
        ref, _ = tr.step(batch, noise)
        for k, v in eng.losses().items():
            r = float(ref[k].detach()) if torch.is_tensor(ref[k]) else float(ref[k])
            assert abs(v - r) <= 1e-4 * max(1.0, abs(r)), (step, k, v, r)
        bufs = [('param', arena.param), ('grad', arena.grad), ('exp_avg', arena.exp_avg)]
        if optim == 'adam':     # (Adamax: u = max(beta2 u, |g| + eps) makes the pads of its infinity norm eps, as torch's does for
            bufs.append(('exp_avg_sq', arena.exp_avg_sq))       # a zero gradient; nothing multiplies by them: 0 / eps = 0)
        for name, buf in bufs:
            assert not any(bool(p.any()) for p in arena.pads(buf)), (step, name)
        p = eng.plan
        for name in ('XIN', 'ZDEC', 'DZDEC', 'Z2F', 'D', 'DZ2F', 'FPIN', 'Z3IN', 'DZ1B'):
            t = getattr(p, name, None)
            if t is not None and t._base is not None and t._base.shape[1] != t.shape[1]:
                assert not bool(t._base[:, t.shape[1]:].any()), (step, name)
        for ch in (p.c_enc, p.c_decx, p.c_z2F, p.c_top, p.c_dz1):
            for t in ch.out + ch.dpre:
                if t._base is not None and t._base.shape[1] != t.shape[1]:
                    assert not bool(t._base[:, t.shape[1]:].any()), step
    for k, prm in tr.params.items():
        close(arena.p(k), prm.detach().numpy(), 2e-4, 5e-5)


def test_chain_refuses_padded_products_over_column_slices(monkeypatch):
    """a column slice of a wider buffer (Q[:, :Z] with Z % 4 != 0) has neighbours behind its last column, not zeros: the
    chain must not run the padded-K / padded-N forms over it"""
    from drvae_amd.chain import _Chain
    wide = torch.zeros(6, 20)
    assert _Chain._pad_ok(torch.zeros(6, 12)[:, :10])             # whole rows of a row-padded buffer
    assert _Chain._pad_ok(torch.zeros(6, 10))                     # unpadded rows
    assert _Chain._pad_ok(wide[:, :8])                            # a multiple of 4: nothing to pad
    assert not _Chain._pad_ok(wide[:, :10])                       # a slice with live neighbours
    assert not _Chain._pad_ok(torch.zeros(6, 12)[:, 1:11])
