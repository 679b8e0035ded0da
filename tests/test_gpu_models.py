"""-m gpu: the model classes (drop-in counterparts of the reference's DrVAE / PVAE / VFAE)
through their reference-style API -- ctor kwargs, state_dict, run_on_batch, forward --
against the golden vectors, plus hipGraph replay == eager launches."""
import os

import numpy as np
import pytest
import torch

from oracle import models_ref as M
from tests.golden import cases as C

pytestmark = pytest.mark.gpu


def build_model(spec, dev):
    from drvae_amd.DrVAE import DrVAE
    from drvae_amd.PVAE import PVAE
    from drvae_amd.VFAE import VFAE
    common = dict(dim_x=spec.dim_x, dim_s=1, dim_y=spec.dim_y, dim_h_en_z1=list(spec.h_en_z1),
                  dim_h_de_x=list(spec.h_de_x), dim_z1=spec.dim_z1, type_rec='diag_gaussian',
                  nonlinearity=spec.nonlin, learning_rate=spec.learning_rate, L=spec.L,
                  weight_decay=spec.weight_decay, add_noise_var=spec.add_noise_var, use_MMD=False, random_seed=123,
                  weight_norm=spec.weight_norm, optim_alg=spec.optim_alg, device=dev)
    pert = dict(kl_qz2pz2_rate=spec.kl_qz2pz2_rate, pertloss_rate=spec.pertloss_rate,
                anneal_perturb_rate_itermax=spec.anneal_perturb_rate_itermax,
                anneal_perturb_rate_offset=spec.anneal_perturb_rate_offset)
    ycfg = dict(dim_h_de_z1=list(spec.h_de_z1), dim_h_clf=list(spec.h_clf), yloss_rate=spec.yloss_rate,
                clf_1sig=spec.clf_1sig, prior_y='uniform' if spec.prior_y is None else np.asarray(spec.prior_y),
                type_y=spec.type_y)
    if spec.kind == 'drvae':
        return DrVAE(dim_h_en_z3=list(spec.h_en_z3), dim_z3=spec.dim_z3, clf_z1z2=spec.clf_z1z2, **common, **pert,
                     **ycfg)
    if spec.kind == 'pvae':
        return PVAE(**common, **pert)
    return VFAE(dim_h_en_z2=list(spec.h_en_z3), dim_z2=spec.dim_z3, semi_supervised=spec.semi_supervised, **common,
                **ycfg)


def kwargs_for(spec, batch, dev):
    t = lambda k: torch.from_numpy(batch[k].copy()).to(dev)
    if spec.kind == 'drvae':
        return dict(x1=t('x1'), x2=t('x2'), s=t('s'), y=t('y'), has_x2=t('has_x2'), has_y=t('has_y'))
    if spec.kind == 'pvae':
        return dict(x1=t('x1'), x2=t('x2'), s=t('s'), has_x2=t('has_x2'))
    return dict(x1=t('x1'), s=t('s'), y=t('y'), has_y=t('has_y'))


@pytest.mark.parametrize('name', ['tiny_drvae', 'tiny_drvae_nolp', 'tiny_drvae_wn', 'tiny_drvae_adamax', 'tiny_drvae_prior',
                                  'tiny_drvae_1sig', 'tiny_vfae_prior_1sig', 'tiny_drvae_cont', 'tiny_vfae_cont_sup', 'tiny_pvae',
                                  'tiny_vfae',
                                  'tiny_vfae_sup', 'cfg2_drvae'])
def test_run_on_batch_matches_reference(name, dev):
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    model = build_model(spec, dev)
    params = M.init_params(spec, case['param_seed'], as_numpy=True)
    assert list(model.state_dict().keys()) == list(params.keys())          # reference state_dict names/order
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model.add_noise = spec.add_noise_var > 0
    kw = kwargs_for(spec, case['batch'], dev)
    ev = model.run_on_batch(train_mode=False, noise=case['noises'][0], **kw)
    for k, v in ev.items():
        np.testing.assert_allclose(float(v), gold['eval/' + k], rtol=1e-4, atol=1e-5)
    for step, noise in enumerate(case['noises']):
        losses = model.run_on_batch(train_mode=True, noise=noise, **kw)
        assert list(losses.keys()) == [k[len('step0/'):] for k in gold if k.startswith('step0/')]
        for k, v in losses.items():
            np.testing.assert_allclose(float(v), gold['step%d/%s' % (step, k)], rtol=1e-4, atol=1e-5)
    assert model.finished_training_iters == len(case['noises'])
    # the nn.Parameters alias the arena: state_dict reflects the fused Adam updates
    last = len(case['noises']) - 1
    for k, v in model.state_dict().items():
        a = v.cpu().numpy()
        if case['full']:
            np.testing.assert_allclose(a, gold['param%d/%s' % (last, k)], rtol=2e-4, atol=5e-5)
        else:
            np.testing.assert_allclose(a.reshape(-1)[C.sample_index(a.size)], gold['paramsample%d/%s' % (last, k)],
                                       rtol=2e-4, atol=5e-5)


def test_inference_forward_uses_means(dev):
    """forward(): posterior means only, no sampling (src/DrVAE.py:253-311) vs the oracle blocks."""
    from oracle import blocks_ref as B
    spec = C.tiny_spec('drvae')
    model = build_model(spec, dev)
    params = M.init_params(spec, 3, as_numpy=True)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    x1 = torch.from_numpy(M.make_batch(spec, 9, seed=2)['x1'])
    res = model.forward(x1=x1.to(dev))
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    mu1, _ = B.diag_gaussian([x1], p, 'encoder_z1', 1, 'elu')
    mu2, _ = B.diag_gaussian_linear([mu1], p, 'decoder_z2Fz1')
    qy = B.categorical([mu1, mu2 - mu1], p, 'encoder_y', 0, 'elu', 2)
    px1 = B.diag_gaussian_sigma([mu1], p, 'decoder_x', 1, 'elu')
    px2 = B.diag_gaussian_sigma([mu2], p, 'decoder_x', 1, 'elu')
    c = lambda a, b: np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-4, atol=1e-5)
    c(res['z1'], mu1); c(res['z2'], mu2); c(res['proba'], qy); c(res['x1_rec'], px1[0]); c(res['px1'][1], px1[1])
    c(res['x2_pert'], px2[0])
    assert (res['pred'].cpu().numpy() == qy.argmax(1).numpy()).all()
    pred, proba = model.predict(x1=x1)
    assert pred.shape == (9,) and proba.shape == (9, 2)
    assert not model.training


def test_graph_replay_equals_eager(dev):
    from tests.test_engine_cpu import make_engine, set_batch
    spec = M.ModelSpec(kind='drvae', L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    batch = M.make_batch(spec, 150, seed=5)
    eager, a0 = make_engine(spec, params, dev)
    graph, a1 = make_engine(spec, params, dev)
    for e in (eager, graph):
        set_batch(e, batch, dev)
        e.train_step()                       # iteration 0 (beta_pert = 0.01) always eager
    graph.capture()
    for _ in range(4):
        eager.train_step()
        graph.replay()
    torch.cuda.synchronize()
    assert graph.iters == eager.iters == 5
    assert torch.equal(a0.param, a1.param)
    assert eager.losses() == graph.losses()


@pytest.mark.parametrize('kind', ['drvae', 'vfae', 'pvae'])
def test_partitioned_replay_equals_eager(kind, dev):
    """the production launch mode of the train step -- dual graphs ordered by device flags, the split of the
    compute units chosen by ``tune_partition`` (which must restore the training state), replays inside
    ``partition()`` -- is bitwise the eager step, and no device-side wait ever timed out (PVAE, round 5: its side
    chain is the step's tail only)"""
    from tests.test_engine_cpu import make_engine, set_batch
    spec = M.ModelSpec(kind=kind, L=1 if kind == 'pvae' else 2)
    params = M.init_params(spec, 3, as_numpy=True)
    batch = M.make_batch(spec, 150, seed=5)
    eager, a0 = make_engine(spec, params, dev)
    graph, a1 = make_engine(spec, params, dev)
    for e in (eager, graph):
        set_batch(e, batch, dev)
        e.train_step()
    graph.capture()
    assert graph._side_graph is not None, 'the dual-graph schedule was not taken'
    graph.tune_partition(candidates=(32, 64), steps=4)
    with graph.partition():
        for _ in range(6):
            graph.replay()
    for _ in range(6):
        eager.train_step()
    torch.cuda.synchronize()
    graph.check_sync()
    assert graph.iters == eager.iters == 7
    assert torch.equal(a0.param, a1.param) and torch.equal(a0.exp_avg_sq, a1.exp_avg_sq)
    assert eager.losses() == graph.losses()


@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_captured_exchange_forks_behind_the_collective(kind, dev):
    """data parallelism with the gradient exchange captured INTO the step's graph (round 6): the side chain, idle behind the
    join, draws the NEXT step's noise as in the single-GPU step (the main chain's graph no longer starts with the draw; the
    sweep's first workgroup orders the next step behind the side chain's tail).  With a capturable stand-in collective (a
    device copy of the buffer onto itself through a scratch: the schedule does not depend on what the collective computes) the
    replays are bitwise the eager steps with the same collective, and bitwise the two-graph split form; no wait times out."""
    from tests.test_engine_cpu import make_engine, set_batch
    spec = M.ModelSpec(kind=kind, L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    batch = M.make_batch(spec, 150, seed=5)
    engines = [make_engine(spec, params, dev) for _ in range(3)]
    scratch = [torch.empty_like(a.xchg) for _, a in engines]

    def make_ar(i):
        def ar(buf):         # (in place, stream-ordered, capturable: what an all-reduce over one rank is)
            scratch[i].copy_(buf)
            buf.copy_(scratch[i])
        return ar
    for i, (e, _) in enumerate(engines):
        set_batch(e, batch, dev)
        e.train_step(allreduce=make_ar(i))
    (eager, a0), (cap, a1), (split, a2) = engines
    cap.capture(split_for_allreduce='captured', allreduce=make_ar(1))
    assert cap._side_graph is not None and len(cap._graphs) == 1
    assert cap.noise_ahead, 'the captured exchange step does not draw ahead on the side chain'
    split.capture(split_for_allreduce=True)
    for e, ar in ((cap, None), (split, make_ar(2))):
        with e.partition(64):
            for _ in range(8):
                e.replay(ar)
    for _ in range(8):
        eager.train_step(allreduce=make_ar(0))
    torch.cuda.synchronize()
    cap.check_sync()
    split.check_sync()
    assert cap.iters == eager.iters == split.iters == 9
    for a in (a1, a2):
        assert torch.equal(a0.param, a.param) and torch.equal(a0.exp_avg, a.exp_avg) and torch.equal(a0.exp_avg_sq, a.exp_avg_sq)
    assert eager.losses() == cap.losses() == split.losses()
    # ... and the switch that turns the fork off gives the same numbers (the round-5 form: everything in front of the join)
    from drvae_amd import tuning
    os.environ['DRVAE_TUNE'] = 'dp_fork=0'
    tuning.reload()
    try:
        plain, a3 = make_engine(spec, params, dev)
        set_batch(plain, batch, dev)
        plain.train_step(allreduce=make_ar(0))
        plain.capture(split_for_allreduce='captured', allreduce=make_ar(0))
        assert not plain.noise_ahead
        with plain.partition(64):
            for _ in range(8):
                plain.replay(None)
        torch.cuda.synchronize()
        plain.check_sync()
    finally:
        os.environ.pop('DRVAE_TUNE')
        tuning.reload()
    assert torch.equal(a0.param, a3.param) and eager.losses() == plain.losses()


def test_checkpoint_roundtrip_and_errors(dev, tmp_path):
    spec = C.tiny_spec('vfae', dim_y=3)
    m1, m2 = build_model(spec, dev), build_model(spec, dev)
    with torch.no_grad():
        for p in m1.parameters():
            p.add_(0.01)
    f = str(tmp_path / 'ckpt.pth')
    m1.save_to_file(f)
    m2.load_params_from_file(f)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    from drvae_amd.DrVAE import DrVAE
    with pytest.raises(ValueError):
        DrVAE(dim_x=5, dim_s=1, dim_y=2, type_rec='laplace', device=dev)      # src/DrVAE.py:124-131 ('binary' / 'poisson': extensions)
    with pytest.raises(ValueError):
        DrVAE(dim_x=5, dim_s=1, dim_y=2, type_rec='diag_gaussian', optim_alg='sgd', device=dev)


@pytest.mark.parametrize('tag', ['G10a', 'G10b'])
def test_eval_x_reconstruction_matches_reference(tag, dev):
    """N1: RMSE / variance-weighted R^2 / mean per-row Pearson / mean logL (src/DGMMixin.py:128-156)."""
    gold = C.load('blocks')
    c = C.block_inputs(tag)
    spec = C.tiny_spec('pvae', dim_x=c['x'].shape[1])
    model = build_model(spec, dev)
    t = lambda a: torch.from_numpy(a.copy())
    got = model.eval_x_reconstruction(t(c['x']), t(c['x_rec']), t(c['std']))
    for k in ('rmse', 'r2', 'pearr', 'll'):
        np.testing.assert_allclose(got[k], float(gold['%s/%s' % (tag, k)]), rtol=1e-5)
    assert np.isnan(model.eval_x_reconstruction(t(c['x']), t(c['x_rec']))['ll'])
    from tests import kernel_ref as R
    import drvae_amd.kernels as K
    x, r = t(c['x']).to(dev), t(c['x_rec']).to(dev)
    rows, rr = torch.empty(x.shape[0], 6, device=dev), torch.empty(x.shape[0], 6, device=dev)
    cols, rc = (torch.empty(3, x.shape[1], dtype=torch.float64, device=dev) for _ in range(2))
    K.recon_row_stats(rows, x, r); R.recon_row_stats(rr, x, r)
    K.col_moments(cols, x, r); R.col_moments(rc, x, r)
    np.testing.assert_allclose(rows.cpu().numpy(), rr.cpu().numpy(), rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(cols.cpu().numpy(), rc.cpu().numpy(), rtol=1e-12, atol=1e-9)


def test_device_batcher_feeds_one_captured_graph(dev):
    """N2: a device-resident dataset, stratified on-device batches, ONE plan + ONE hipGraph for all
    batches (only rows/labels change, device to device) == eager steps fed the same rows via set_batch."""
    from drvae_amd import data as D
    from tests.test_engine_cpu import make_engine
    spec = M.ModelSpec(kind='drvae', L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    big = M.make_batch(spec, 1200, seed=9)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y')).to(dev)
    w = D.compute_balanced_weights(np.arange(1200) % 7)
    bat = D.DeviceBatcher(ds, w, 150, seed=5)
    fed, a1 = make_engine(spec, params, dev)
    eager, a0 = make_engine(spec, params, dev)
    plan = bat.bind(fed)
    idxs = [bat.next_indices() for _ in range(5)]
    # iteration 0 eager on both (beta_pert = 0.01), then capture once and replay with fresh data
    bat.feed(idxs[0])
    fed.train_step()
    fed.capture()
    for i in idxs[1:]:
        bat.feed(i)
        fed.replay()
    assert fed.plan is plan and len(fed._plans) == 1
    for k, i in enumerate(idxs):
        ic = i.cpu()
        eager.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), bat.has_x2, bat.has_y)
        eager.train_step()
    torch.cuda.synchronize()
    assert torch.equal(a0.param, a1.param)
    assert eager.losses() == fed.losses()


def test_device_batcher_feeds_nuisance_conditioned_models(dev):
    """N4 + N2: a ``use_s`` model on device-drawn batches -- the one-hot(s) columns of the encoder / decoder inputs are
    rebuilt device to device by ``DeviceBatcher.feed`` -- through ONE captured graph == eager steps fed the same rows
    and nuisance classes from the host"""
    from drvae_amd import data as D
    from tests.test_engine_cpu import make_engine
    spec = C.tiny_spec('drvae', use_s=True, dim_s=2)
    params = M.init_params(spec, 3, as_numpy=True)
    big = M.make_batch(spec, 400, seed=9)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y')).to(dev)
    bat = D.DeviceBatcher(ds, D.compute_balanced_weights(np.arange(400) % 7), 24, seed=5)
    fed, a1 = make_engine(spec, params, dev)
    eager, a0 = make_engine(spec, params, dev)
    bat.bind(fed)
    idxs = [bat.next_indices() for _ in range(5)]
    bat.feed(idxs[0])
    fed.train_step()
    fed.capture()
    for i in idxs[1:]:
        bat.feed(i)
        fed.replay()
    for i in idxs:
        eager.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), bat.has_x2, bat.has_y, s=ds.s[i].cpu())
        eager.train_step()
    torch.cuda.synchronize()
    assert len({int(v) for i in idxs for v in ds.s.reshape(-1)[i].tolist()}) == 2      # both classes occur
    assert torch.equal(a0.param, a1.param)
    assert eager.losses() == fed.losses()
    with pytest.raises(NotImplementedError):
        bat.begin_epoch()


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_graph_resident_epoch_feed(kind, dev):
    """N2/N3: the epoch's index table lives on the device and the captured step gathers batch
    (optimiser step - epoch base) itself: an epoch is n graph replays with no other host work, and
    equals eager steps fed the same rows from the host."""
    from drvae_amd import data as D
    from tests.test_engine_cpu import make_engine
    spec = M.ModelSpec(kind=kind, L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    big = M.make_batch(spec, 900, seed=9)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y')).to(dev)
    w = D.compute_balanced_weights(np.arange(900) % 7)
    bat = D.DeviceBatcher(ds, w, 150, seed=5)
    fed, a1 = make_engine(spec, params, dev)
    eager, a0 = make_engine(spec, params, dev)
    bat.bind(fed)
    tables = []
    for epoch in range(2):
        tables.append(bat.begin_epoch(n_batches=3).clone())
        if epoch == 0:
            fed.capture()
        for _ in range(3):
            fed.replay()
    assert len(fed._plans) == 1 and fed.iters == 6
    assert not torch.equal(tables[0], tables[1])
    eager.set_structure(bat.has_x2, bat.has_y)
    for tab in tables:
        for b in range(3):
            i = tab[b].long()
            eager.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), bat.has_x2, bat.has_y)
            eager.train_step()
    torch.cuda.synchronize()
    assert torch.equal(a0.param, a1.param)
    assert eager.losses() == fed.losses()


def test_explicit_batch_after_epoch_feed_uses_its_own_data(dev):
    """after DeviceBatcher epochs (graph-resident feed installed on the plan) a train step given
    explicit data of the SAME batch structure must train on that data, not on the stale epoch table -- and
    the captured feed graph must refuse to replay for it."""
    from drvae_amd import data as D
    from tests.test_engine_cpu import make_engine
    spec = M.ModelSpec(kind='pvae', L=1)          # pair-only PVAE: every batch has the same structure
    params = M.init_params(spec, 3, as_numpy=True)
    big = M.make_batch(spec, 600, seed=9)
    big['has_x2'][:] = 1
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y')).to(dev)
    bat = D.DeviceBatcher(ds, torch.ones(600), 48, seed=5)
    fed, a1 = make_engine(spec, params, dev)
    bat.bind(fed)
    bat.begin_epoch(n_batches=2)
    fed.capture()
    fed.replay(); fed.replay()
    # same structure, explicit rows: must come out exactly like an engine that never saw a feed
    rows = torch.arange(100, 148, device=dev)
    noise = M.make_noise(spec, 48, seed=11)
    plain, a0 = make_engine(spec, params, dev)
    a0.param.copy_(a1.param); a0.exp_avg.copy_(a1.exp_avg); a0.exp_avg_sq.copy_(a1.exp_avg_sq)
    plain.step_dev.copy_(fed.step_dev); plain.iters = fed.iters
    for e in (fed, plain):
        e.set_batch(ds.x1[rows], ds.x2[rows], None, bat.has_x2, bat.has_y)
        e.train_step(noise)
    torch.cuda.synchronize()
    assert fed.plan.feed is not None and fed.plan.live_feed is None
    assert fed.losses() == plain.losses() and torch.equal(a0.param, a1.param)
    with pytest.raises(AssertionError):
        fed.replay()                              # the captured graph reads the epoch table: wrong source now
    bat.begin_epoch(n_batches=2)                  # ... and is valid again once an epoch feed is current
    fed.replay()
    torch.cuda.synchronize()


def test_model_move_keeps_training_state(dev):
    """``.cpu()`` / ``.to(dev)`` swap ``prm.data``; the model must notice, carry the Adam state over
    and keep the parameters aliased to the arena the fused step updates (same results as a model never moved)."""
    case = C.model_case('tiny_drvae')
    spec = case['spec']
    params = M.init_params(spec, case['param_seed'], as_numpy=True)
    kw = kwargs_for(spec, case['batch'], dev)
    out = []
    for move in (False, True):
        model = build_model(spec, dev)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        model.add_noise = spec.add_noise_var > 0
        model.run_on_batch(train_mode=True, noise=case['noises'][1], **kw)
        if move:
            model.cpu()
            assert model._engine is None and next(model.parameters()).device.type == 'cpu'
            model.to(dev)
        model.run_on_batch(train_mode=True, noise=case['noises'][2], **kw)
        model._assert_arena_aliased()
        assert model.finished_training_iters == 2
        out.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        if move:
            with pytest.raises(TypeError):
                model.double()
            sd = {k: v.clone() for k, v in model.state_dict().items()}
            model.load_state_dict(sd, assign=True)
            model._assert_arena_aliased()
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize('name', ['tiny_drvae', 'tiny_drvae_nolp', 'tiny_drvae_only_up', 'tiny_pvae', 'tiny_vfae', 'cfg2_drvae'])
def test_universal_plan_matches_reference_gpu(name, dev):
    """N2: the batch-independent plan on the real kernels (dv_batch_masks, the labeled-slot branch of dv_ymarg_*,
    row-weighted loss terms) against the reference's golden losses, eager and captured"""
    from tests.test_engine_cpu import make_engine, set_batch
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True), dev)
    eng.universal = True
    p = set_batch(eng, case['batch'], dev)
    assert p.universal
    eng.training = False
    eng.set_noise(case['noises'][0])
    eng.forward()
    for k, v in eng.losses().items():
        np.testing.assert_allclose(v, gold['eval/' + k], rtol=1e-4, atol=2e-5)
    for step, noise in enumerate(case['noises']):
        eng.train_step(noise)
        for k, v in eng.losses().items():
            np.testing.assert_allclose(v, gold['step%d/%s' % (step, k)], rtol=1e-4, atol=2e-5)
    nsteps = len(case['noises'])
    for k in arena.shapes:
        a = arena.p(k).cpu().numpy()
        if case['full']:
            np.testing.assert_allclose(a, gold['param%d/%s' % (nsteps - 1, k)], rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize('name', ['tiny_drvae', 'cfg2_drvae', 'tiny_vfae'])
def test_bucketed_universal_plan_matches_reference_gpu(name, dev):
    """N2, the bucketed sampler feed's plans against the reference's golden losses: the golden batch with its rows in the
    feed's order (unlabeled pairs | labeled pairs | labeled singles | unlabeled singles; the loss is a sum over rows and
    the injected noise is addressed by row, so the reference values stand), pair slots and labeled range rounded the way
    ``DeviceBatcher(pair_bucket=16, label_bucket=8)`` rounds them -- cfg 2 at its full size: 150 rows, L=2"""
    from tests.test_engine_cpu import make_engine, set_batch
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    batch = {k: np.asarray(v) for k, v in case['batch'].items()}
    hx, hy = batch['has_x2'].reshape(-1).astype(bool), batch['has_y'].reshape(-1).astype(bool)
    if spec.kind == 'vfae':
        hx = np.zeros_like(hy)
    grp = np.where(hx, hy.astype(int), 3 - hy.astype(int))
    order = np.argsort(grp, kind='stable')
    B, w, v = len(order), 16, 8
    n_p, n_up, n_l = int(hx.sum()), int((hx & ~hy).sum()), int(hy.sum())
    slots = min(B, max(min(w, B), -(-n_p // w) * w)) if spec.kind != 'vfae' else None
    a, b = min(-(-n_up // v) * v, B), (n_up + n_l) // v * v
    lab = (a, b) if b > a else None
    pb = {k: (x[order] if (hasattr(x, 'shape') and x.shape[:1] == (B,)) else x) for k, x in batch.items()}
    perm = lambda nz: {k: (x[order] if k in ('nx1', 'nx2') else x[..., order, :]) for k, x in nz.items()}
    eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True), dev)
    eng.universal, eng.universal_pair_slots, eng.universal_labeled_range = True, slots, lab
    p = set_batch(eng, pb, dev)
    assert p.universal and (slots is None or p.Np == slots) and (lab is None or p.Mf < spec.L * B * spec.dim_y)
    eng.training = False
    eng.set_noise(perm(case['noises'][0]))
    eng.forward()
    for k, val in eng.losses().items():
        np.testing.assert_allclose(val, gold['eval/' + k], rtol=1e-4, atol=2e-5)
    for step, noise in enumerate(case['noises']):
        eng.train_step(perm(noise))
        for k, val in eng.losses().items():
            np.testing.assert_allclose(val, gold['step%d/%s' % (step, k)], rtol=1e-4, atol=2e-5)
    nsteps = len(case['noises'])
    for k in arena.shapes:
        got = arena.p(k).cpu().numpy()
        if case['full']:
            np.testing.assert_allclose(got, gold['param%d/%s' % (nsteps - 1, k)], rtol=2e-4, atol=5e-5)
        else:
            np.testing.assert_allclose(got.astype(np.float64).sum(), gold['paramsum%d/%s' % (nsteps - 1, k)], rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize('kind,label_bucket', [('drvae', None), ('pvae', None), ('drvae', 4), ('vfae', 8)])
def test_sampler_mode_pair_buckets_gpu(kind, label_bucket, dev):
    """N2, mode='sampler' with pair_bucket: batches re-ordered pairs first, each replayed on the captured plan of its
    number-of-pairs bucket (a few graphs, switched from step to step; plans first met in the second epoch are captured
    then) -- bitwise the eager steps of the same plans handed the same rows explicitly."""
    from drvae_amd import data as D
    from tests.test_engine_cpu import make_engine
    spec = M.ModelSpec(kind=kind, L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    big = M.make_batch(spec, 640, seed=9)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y')).to(dev)
    w = D.compute_balanced_weights(np.arange(640) % 7)
    bat = D.DeviceBatcher(ds, w, 64, seed=5, mode='sampler', pair_bucket=4, label_bucket=label_bucket)
    fed, a1 = make_engine(spec, params, dev)
    eager, a0 = make_engine(spec, params, dev)
    eager.universal = True
    bat.bind(fed)
    used, n_cap = set(), []
    for epoch in range(2):
        tab = bat.begin_epoch(n_batches=5).clone()
        bat.prepare_epoch(lambda e: (n_cap.append(e.plan.key), e.capture()))
        for k in range(5):
            bat.select(k)
            used.add(fed.plan.key)
            fed.replay()
            i = tab[k].long()
            hx, hy = ds.has_x2[i].cpu().numpy(), ds.has_y[i].cpu().numpy()
            n = int(hx.sum())
            if spec.kind != 'vfae':
                assert hx[:n].all() and n <= fed.plan.Np < n + 4 or fed.plan.Np == 4
            eager.universal_pair_slots = fed.plan.Np
            eager.universal_labeled_range = bat.batch_specs[k][1:] if label_bucket else None
            if label_bucket:
                a, b = bat.batch_specs[k][1:]
                assert hy[a:b].all() and (b - a > int(hy.sum()) - 2 * label_bucket or (a, b) == (0, 0))
                assert fed.plan.Mf == spec.L * (64 * spec.dim_y - (b - a) * (spec.dim_y - 1))
            eager.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), hx, hy)
            eager.train_step()
    # a re-drawn table WITHOUT new captures: a composition not met before runs on the cheapest captured plan that
    # serves it (pair slots for all its pairs, a labeled range inside its run of labeled rows)
    tab = bat.begin_epoch(n_batches=5).clone()          # (same table size: the captured steps read this table)
    n_before, fell_back = len(n_cap), 0
    for k in range(5):
        bat.select(k)
        fed.replay()
        i = tab[k].long()
        hx, hy = ds.has_x2[i].cpu().numpy(), ds.has_y[i].cpu().numpy()
        key = fed.plan.key
        slots, lab = (key[3] if len(key) > 3 else 64), (key[4:6] if len(key) > 4 else (0, 0))
        fell_back += (slots,) + tuple(lab) != tuple(bat.batch_specs[k])
        assert key in set(n_cap) and (spec.kind == 'vfae' or int(hx.sum()) <= slots) and hy[lab[0]:lab[1]].all()
        eager.universal_pair_slots, eager.universal_labeled_range = slots, lab
        eager.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), hx, hy)
        eager.train_step()
    torch.cuda.synchronize()
    assert len(used) > 1 and len(n_cap) == len(set(n_cap)) == n_before and used <= set(n_cap)
    assert eager.losses() == fed.losses()
    assert torch.equal(a0.param, a1.param)
    fed.check_sync()


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_sampler_mode_epoch_in_one_graph(kind, dev):
    """N2, mode='sampler': an epoch of WeightedRandomSampler-style batches (i.i.d. rows, any group mix per batch) is
    n replays of ONE captured graph -- the composition of each batch reaches the step as device-side masks read
    through the epoch's index table -- and equals, bitwise, eager steps handed the same rows and flags explicitly;
    against per-structure plans the same batches agree under injected noise."""
    import drvae_amd.kernels as K
    from drvae_amd import data as D
    from tests.test_engine_cpu import make_engine
    spec = M.ModelSpec(kind=kind, L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    big = M.make_batch(spec, 640, seed=9)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y')).to(dev)
    w = D.compute_balanced_weights(np.arange(640) % 7)
    bat = D.DeviceBatcher(ds, w, 64, seed=5, mode='sampler')
    fed, a1 = make_engine(spec, params, dev)
    eager, a0 = make_engine(spec, params, dev)
    eager.universal = True
    bat.bind(fed)
    tables = []
    for epoch in range(2):
        tables.append(bat.begin_epoch(n_batches=3).clone())
        if epoch == 0:
            fed.capture()
        for _ in range(3):
            fed.replay()
    assert len(fed._plans) == 1 and fed.iters == 6
    comps = set()
    for tab in tables:
        for b in range(3):
            i = tab[b].long()
            hx, hy = ds.has_x2[i].cpu().numpy(), ds.has_y[i].cpu().numpy()
            comps.add((int(hx.sum()), int(hy.sum())))
            eager.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), hx, hy)
            eager.train_step()
    torch.cuda.synchronize()
    assert len(comps) > 1 and len(eager._plans) == 1
    assert eager.losses() == fed.losses()
    assert torch.equal(a0.param, a1.param)
    # one of these batches on its own per-structure plan, injected noise: same losses and gradients
    i = tables[1][2].long()
    hx, hy = ds.has_x2[i].cpu().numpy(), ds.has_y[i].cpu().numpy()
    noise = M.make_noise(spec, 64, seed=21)
    one, a2 = make_engine(spec, params, dev)
    a2.param.copy_(a0.param)
    one.iters = eager.iters                  # (same perturbation annealing coefficient)
    res = []
    for e in (eager, one):
        e.set_batch(ds.x1[i], ds.x2[i], ds.y[i].cpu(), hx, hy)
        e.training, e.fuse_bwd = True, False
        e.set_noise(noise)
        e.forward()
        e.backward()
        torch.cuda.synchronize()
        res.append((e.losses(), e.arena.grad.clone()))
    assert not one.plan.universal
    for (k, a), b in zip(res[0][0].items(), res[1][0].values()):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5)
    rel = float((res[0][1] - res[1][1]).norm() / res[1][1].norm())
    assert rel < 1e-4, rel


@pytest.mark.parametrize('use_mmd', [False, True])
@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_use_s_extension_matches_oracle_gpu(kind, use_mmd, dev):
    """N4 extension on the real kernels: one_hot(s) as a second GEMM operand of encoder_z1 / decoder_x, the
    model-level MMD penalty through the block-level MMD operators; losses, gradients and 3 Adam steps vs the oracle"""
    from tests.test_engine_cpu import _use_s_case, make_engine
    spec, batch = _use_s_case(kind, use_mmd)
    params = M.init_params(spec, 9, as_numpy=True)
    eng, arena = make_engine(spec, params, dev)
    tr = M.RefTrainer(spec, M.init_params(spec, 9))
    t = lambda k: torch.from_numpy(batch[k].copy()).to(dev)
    eng.set_batch(t('x1'), t('x2'), batch['y'], batch['has_x2'], batch['has_y'], s=batch['s'])
    noise = M.make_noise(spec, 24, seed=4)
    ref, _ = tr.loss(batch, noise, True)
    ref['CMPL'].backward()
    eng.training = True
    eng.set_noise(noise)
    eng.forward()
    eng.backward()
    for k, v in eng.losses().items():
        np.testing.assert_allclose(v, float(ref[k].detach()), rtol=1e-4, atol=2e-5)
    for k, prm in tr.params.items():
        np.testing.assert_allclose(arena.g(k).cpu().numpy(), prm.grad.numpy(), rtol=1e-3, atol=5e-6)
        prm.grad = None
    for step in range(3):
        nz = M.make_noise(spec, 24, seed=10 + step)
        want, _ = tr.step(batch, nz)
        eng.train_step(nz)
        for k, v in eng.losses().items():
            np.testing.assert_allclose(v, float(want[k].detach()), rtol=1e-4, atol=2e-5)
    for k, prm in tr.params.items():
        np.testing.assert_allclose(arena.p(k).cpu().numpy(), prm.detach().numpy(), rtol=2e-4, atol=5e-5)
    # the captured step (conditioning AND the cross-row MMD penalty: its row lists are plan data) == eager launches
    twin, arena2 = make_engine(spec, params, dev)
    twin.set_batch(t('x1'), t('x2'), batch['y'], batch['has_x2'], batch['has_y'], s=batch['s'])
    twin.iters = eng.iters
    for dst, src in ((arena2.param, arena.param), (arena2.exp_avg, arena.exp_avg), (arena2.exp_avg_sq, arena.exp_avg_sq),
                     (twin.step_dev, eng.step_dev), (twin.rng_ctr, eng.rng_ctr), (twin.side_ctr, eng.side_ctr),
                     (twin.side_t, eng.side_t)):
        dst.copy_(src)
    twin.capture()
    for _ in range(3):
        eng.train_step()
        twin.replay()
    torch.cuda.synchronize()
    twin.check_sync()
    assert torch.equal(arena.param, arena2.param)
    assert eng.losses() == twin.losses() and all(np.isfinite(v) for v in eng.losses().values())
    if use_mmd:                             # another composition of nuisance classes: the captured row lists are stale
        twin.plan.set_s_host(1 - np.asarray(batch['s']).reshape(-1)[twin.plan.rows])
        with pytest.raises(AssertionError, match='capture again'):
            twin.replay()


def test_use_s_model_api_with_fourier_mmd(dev):
    """the model class with ``use_s=True, use_MMD=True, kernel_MMD='rbf_fourier'`` (the reference's constructor
    defaults for the penalty): trains through ``run_on_batch``, reports a non-zero MMD term, infers with ``s``"""
    from drvae_amd.DrVAE import DrVAE
    spec = C.tiny_spec('drvae')
    model = DrVAE(dim_x=spec.dim_x, dim_s=3, dim_y=2, dim_h_en_z1=[7], dim_h_de_z1=[6], dim_h_en_z3=[6], dim_h_de_x=[8],
                  dim_h_clf=[], dim_z1=5, dim_z3=4, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-3, L=2,
                  weight_decay=0.05, add_noise_var=0.01, use_s=True, use_MMD=True, kernel_MMD='rbf_fourier', mmd_rate=2.0,
                  pertloss_rate=0.05, random_seed=1, device=dev)
    assert tuple(model.encoder_z1.nnet.model.linear1.weight.shape) == (7, spec.dim_x + 3)
    assert tuple(model.decoder_x.nnet.model.linear1.weight.shape) == (8, 5 + 3)
    batch = M.make_batch(spec, 48, seed=2)
    g = torch.Generator().manual_seed(0)
    s = torch.randint(0, 3, (48, 1), generator=g)
    t = lambda k: torch.from_numpy(batch[k].copy()).to(dev)
    kw = dict(x1=t('x1'), x2=t('x2'), s=s, y=t('y'), has_x2=t('has_x2'), has_y=t('has_y'))
    model.add_noise = True
    first = {k: float(v) for k, v in model.run_on_batch(train_mode=True, **kw).items()}
    for _ in range(30):
        last = {k: float(v) for k, v in model.run_on_batch(train_mode=True, **kw).items()}
    assert all(np.isfinite(v) for v in last.values()) and first['MMD'] < 0 and last['MMD'] < 0
    assert last['CMPL'] < first['CMPL']
    res = model.forward(t('x1'), s=s)
    assert tuple(res['x1_rec'].shape) == (48, spec.dim_x) and bool(torch.isfinite(res['x1_rec']).all())
    ev = model.run_on_batch(train_mode=False, **kw)
    assert np.isfinite(float(ev['MMD']))


@pytest.mark.parametrize('kernel', ['rbf_fourier', 'identity'])
@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_mmd_penalty_as_launch_lists_equals_the_block_level_path(kind, kernel, dev):
    """round 6: the model-level MMD penalty inside the fused step as explicit launch lists (no autograd) against the
    block-level operators it replaced (``DRVAE_TUNE=mmd_explicit=0``): the same random Fourier features (torch's generator,
    same seed, same order of draws), the same kernels -- penalty value, its gradient w.r.t. the samples and three train steps
    agree to rounding; three nuisance classes (the mean over the categories) and two (the first pair only)"""
    from drvae_amd import tuning
    from tests.test_engine_cpu import make_engine
    for dim_s in (2, 3):
        spec = C.tiny_spec(kind, use_s=True, dim_s=dim_s, use_MMD=True, mmd_rate=0.7, kernel_MMD=kernel)
        batch = M.make_batch(spec, 48, seed=3)
        hx, hy = batch['has_x2'].astype(bool).reshape(-1), batch['has_y'].astype(bool).reshape(-1)
        sv = np.zeros(48, np.int64)                        # every data group holds every class: round-robin inside each group
        for grp in ([hy & ~hx, ~hy & ~hx, hy & hx, ~hy & hx] if kind == 'drvae' else [hy, ~hy]):
            sv[np.nonzero(grp)[0]] = np.arange(int(grp.sum())) % dim_s
        params = M.init_params(spec, 9, as_numpy=True)
        t = lambda k: torch.from_numpy(batch[k].copy()).to(dev)
        out = {}
        for mode in (1, 0):
            os.environ['DRVAE_TUNE'] = 'mmd_explicit=%d' % mode
            tuning.reload()
            try:
                eng, arena = make_engine(spec, params, dev)
                eng.set_batch(t('x1'), t('x2'), batch['y'], batch['has_x2'], batch['has_y'], s=sv)
                assert (eng.plan.mmd_items is not None) and len(eng.plan.mmd_items) >= 2
                torch.manual_seed(5)
                eng.training = True
                eng.set_noise(M.make_noise(spec, 48, seed=4))
                eng.forward()
                eng.backward()
                res = dict(mmd=eng.plan.MMDval.clone(), dz=eng.plan.DZMMD.clone(), grad=arena.grad.clone(), loss=eng.losses())
                for step in range(3):
                    eng.train_step(M.make_noise(spec, 48, seed=10 + step))
                res.update(param=arena.param.clone(), last=eng.losses())
                out[mode] = res
            finally:
                os.environ.pop('DRVAE_TUNE')
                tuning.reload()
        a, b = out[1], out[0]
        assert abs(float(b['mmd'])) > 1e-4
        np.testing.assert_allclose(a['mmd'].cpu().numpy(), b['mmd'].cpu().numpy(), rtol=2e-5)
        assert float((a['dz'] - b['dz']).norm() / b['dz'].norm()) < 2e-5
        assert float((a['grad'] - b['grad']).norm() / b['grad'].norm()) < 2e-5
        for k in a['loss']:
            np.testing.assert_allclose(a['loss'][k], b['loss'][k], rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(a['last'][k], b['last'][k], rtol=1e-4, atol=1e-5)
        assert float((a['param'] - b['param']).norm() / b['param'].norm()) < 1e-4


@pytest.mark.parametrize('type_rec', ['binary', 'poisson'])
@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_bernoulli_poisson_decoders_gpu(kind, type_rec, dev):
    """N4 extension on the real kernels (dv_rec_nll_rows + single-head decoder GEMMs): losses, gradients and Adam
    steps vs the oracle, the captured step, and the model class through its reference-style API"""
    import drvae_amd.kernels as K
    from tests import kernel_ref as R
    from tests.test_engine_cpu import make_engine, set_batch
    # kernel vs its reference, odd sizes, gathered x rows
    g = torch.Generator().manual_seed(1)
    M_, X_ = 37, 101
    a = torch.randn(M_, X_, generator=g).to(dev) * 3
    v = torch.sigmoid(a) if type_rec == 'binary' else torch.nn.functional.softplus(a) + 1e-6
    xs = ((torch.rand(20, X_, generator=g) > 0.5).float() if type_rec == 'binary' else
          torch.poisson(torch.rand(20, X_, generator=g) * 4)).to(dev)
    xi = torch.randint(0, 20, (M_,), generator=g).to(torch.int32).to(dev)
    coef = torch.randn(M_, generator=g).to(dev)
    outs = []
    for L_ in (K, R):
        o, d = torch.zeros(M_, device=dev), torch.zeros(M_, X_, device=dev)
        L_.rec_nll_rows(o, xs, v, kind=type_rec, shift=1e-6 if type_rec == 'poisson' else 0.0, xidx=xi, coef=coef, dpre=d)
        outs.append((o, d))
    torch.cuda.synchronize()
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=2e-4, atol=1e-5)
    # fused step vs oracle
    spec = C.tiny_spec(kind, type_rec=type_rec, add_noise_var=0.0)
    batch = M.make_batch(spec, 16, seed=3)
    eng, arena = make_engine(spec, M.init_params(spec, 9, as_numpy=True), dev)
    tr = M.RefTrainer(spec, M.init_params(spec, 9))
    set_batch(eng, batch, dev)
    for step in range(3):
        nz = M.make_noise(spec, 16, seed=10 + step)
        want, _ = tr.step(batch, nz)
        eng.train_step(nz)
        for k, val in eng.losses().items():
            np.testing.assert_allclose(val, float(want[k].detach()), rtol=1e-4, atol=2e-5)
    for k, prm in tr.params.items():
        np.testing.assert_allclose(arena.p(k).cpu().numpy(), prm.detach().numpy(), rtol=2e-4, atol=5e-5)
    eng.capture()
    eng.replay()
    torch.cuda.synchronize()
    assert all(np.isfinite(val) for val in eng.losses().values())
    # model class
    spec2 = C.tiny_spec(kind, type_rec=type_rec)
    model = build_model(spec2, dev) if False else None
    from drvae_amd.DrVAE import DrVAE
    from drvae_amd.VFAE import VFAE
    common = dict(dim_x=spec.dim_x, dim_s=1, dim_y=2, dim_h_en_z1=[7], dim_h_de_z1=[6], dim_h_de_x=[8], dim_h_clf=[],
                  dim_z1=5, type_rec=type_rec, nonlinearity='elu', learning_rate=5e-3, L=2, weight_decay=0.05,
                  use_MMD=False, random_seed=1, device=dev)
    model = DrVAE(dim_h_en_z3=[6], dim_z3=4, pertloss_rate=0.05, **common) if kind == 'drvae' else \
        VFAE(dim_h_en_z2=[6], dim_z2=4, **common)
    assert type(model.decoder_x).__name__ == ('BernoulliDecoder' if type_rec == 'binary' else 'PoissonDecoder')
    t = lambda k: torch.from_numpy(batch[k].copy()).to(dev)
    kw = dict(x1=t('x1'), s=t('s'), y=t('y'), has_y=t('has_y'))
    if kind == 'drvae':
        kw.update(x2=t('x2'), has_x2=t('has_x2'))
    l0 = float(model.run_on_batch(train_mode=True, **kw)['CMPL'])
    for _ in range(40):
        l1 = float(model.run_on_batch(train_mode=True, **kw)['CMPL'])
    assert np.isfinite(l1) and l1 < l0
    res = model.forward(t('x1'))
    assert tuple(res['x1_rec'].shape) == (16, spec.dim_x) and len(res['px1']) == 1
    lp = model.decoder_x.logp_perx(t('x1'), *res['px1'])
    assert tuple(lp.shape) == (16,) and bool(torch.isfinite(lp).all())
