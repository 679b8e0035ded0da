"""N3: the ``fit`` protocol.  CPU: the early-stopping / snapshot controller against decisions
recorded from the reference's unmodified ``fit`` (tests/golden/fit.npz), the prediction metrics
against the reference's scikit-learn values, and an end-to-end ``fit`` on a tiny model with the
HIP launchers replaced by their references.  GPU: ``fit`` fed by the device batcher (graph replay)."""
import os

import numpy as np
import pytest
import torch

from tests import kernel_ref
from tests.golden import cases as C


@pytest.fixture(scope='module')
def G():
    return C.load('fit')


@pytest.mark.parametrize('name', list(C.fit_policy_cases()))
def test_early_stopping_matches_reference_fit(G, name):
    from drvae_amd.fit import EarlyStopping
    c = C.fit_policy_cases()[name]
    ctl = EarlyStopping(c['epochs'], c['early_stop'], patience={'vfae': 40}.get(c['kind'], 50))
    snaps, means, bests, hits, cont, ran = [], [], [], 0, 0, 0
    per_epoch = c['n_batches']
    stop_iter = c.get('interrupt_at')
    for epoch in range(1, c['epochs'] + 1):
        if stop_iter is not None and stop_iter <= epoch * per_epoch:      # KeyboardInterrupt inside this epoch
            if ctl.on_interrupt():
                snaps.append(ran)
            break
        d = ctl.update(epoch, c['objs'][epoch - 1])
        ran += 1
        means.append(d['rolling_mean'])
        bests.append(d['best_before'])
        if d['snapshot']:
            snaps.append(epoch)
        hits += d['patience_hit']
        cont += d['continuing']
        if d['stop']:
            break
    assert snaps == list(G[name + '/snapshots'])
    assert ran == int(G[name + '/epochs_run'])
    assert hits == int(G[name + '/early_stopped']) and cont == int(G[name + '/continuing'])
    fmt = lambda a: np.array([float('{:.4f}'.format(v)) for v in a])          # the reference logs {:.4f}
    np.testing.assert_array_equal(fmt(means), G[name + '/rolling_mean'])
    np.testing.assert_array_equal(fmt(bests), G[name + '/best_before'])


@pytest.mark.parametrize('tag', list(C.y_metric_cases()))
def test_y_metrics_match_reference(G, tag):
    from drvae_amd import metrics as MET
    c = C.y_metric_cases()[tag]
    if c.get('cont'):
        got = MET.eval_y_regression(torch.from_numpy(c['pred']), torch.from_numpy(c['ylab']))
        for k in ('rmse', 'r2', 'pearr'):
            assert got[k] == pytest.approx(float(G['%s/%s' % (tag, k)]), rel=1e-6)
        return
    got = MET.eval_y_prediction(torch.from_numpy(c['pred']), torch.from_numpy(c['proba']),
                                torch.from_numpy(c['ylab']), c['proba'].shape[1])
    for k in ('acc', 'auroc', 'aupr'):
        want = float(G['%s/%s' % (tag, k)])
        if np.isnan(want):
            assert np.isnan(got[k])
        else:
            assert got[k] == pytest.approx(want, rel=1e-6 if k == 'acc' else 1e-12)
        rb = '%s/ref_with_blk/%s' % (tag, k)       # the reference's own macro branch, run with its missing name supplied
        if rb in G:
            assert got[k] == pytest.approx(float(G[rb]), rel=1e-6 if k == 'acc' else 1e-12)


def test_y_metrics_random_vs_sklearn():
    skm = pytest.importorskip('sklearn.metrics')
    from drvae_amd import metrics as MET
    rs = np.random.RandomState(3)
    for n, levels in ((5, None), (400, None), (400, 7), (1000, 3)):
        y = rs.randint(0, 2, n)
        y[:2] = [0, 1]
        sc = rs.rand(n) if levels is None else rs.randint(0, levels, n) / levels
        ty, ts = torch.from_numpy(y), torch.from_numpy(sc)
        assert MET.roc_auc(ty, ts) == pytest.approx(skm.roc_auc_score(y, sc), rel=1e-12)
        assert MET.average_precision(ty, ts) == pytest.approx(skm.average_precision_score(y, sc), rel=1e-12)


def _tiny_model(kind, **kw):
    import drvae_amd
    from drvae_amd.DrVAE import DrVAE
    from drvae_amd.PVAE import PVAE
    from drvae_amd.VFAE import VFAE
    common = dict(dim_x=13, dim_s=1, dim_y=2, dim_h_en_z1=[7], dim_h_de_x=[8], dim_z1=5, type_rec='diag_gaussian',
                  nonlinearity='elu', learning_rate=5e-3, L=2, weight_decay=0.01, add_noise_var=0.01, use_MMD=False,
                  random_seed=5, epochs=3, batch_size=8)
    common.update(kw)
    if kind == 'drvae':
        return DrVAE(dim_h_de_z1=[6], dim_h_en_z3=[6], dim_h_clf=[], dim_z3=4, pertloss_rate=0.05, **common)
    if kind == 'pvae':
        return PVAE(pertloss_rate=0.05, **common)
    return VFAE(dim_h_de_z1=[6], dim_h_en_z2=[6], dim_h_clf=[], dim_z2=4, semi_supervised=True, **common)


def _tiny_dataset(kind, n, seed, device='cpu'):
    from drvae_amd import data as D
    rs = np.random.RandomState(seed)
    y = rs.randint(0, 2, n)
    x1 = (rs.standard_normal((n, 13)) + 0.8 * (2 * y[:, None] - 1) * (np.arange(13) % 3 == 0)).astype(np.float32)
    hx = (np.arange(n) % 3 == 0).astype(np.int64)
    x2 = ((x1 * 0.7 + 0.2) * hx[:, None]).astype(np.float32)
    hy = (np.arange(n) % 4 != 1).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    if kind == 'vfae':
        return D.VFAEDataset(t(x1), t(np.zeros(n, np.int64)), t(y), t(hy))
    return D.DrVAEDataset(t(x1), t(x2), t(np.zeros(n, np.int64)), t(y), t(hx), t(hy))


class _Loader(list):
    dataset = None


def _loader(ds, bs):
    idx = np.arange(len(ds))
    ld = _Loader([tuple(getattr(ds, f)[idx[i:i + bs]] for f in (ds.FIELDS if hasattr(ds, 'FIELDS')
                                                                   else ('x1', 's', 'y', 'has_y')))
                  for i in range(0, len(ds) - bs + 1, bs)])
    ld.dataset = ds
    return ld


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_fit_end_to_end_cpu(kind, monkeypatch, tmp_path):
    kernel_ref.install(monkeypatch)
    model = _tiny_model(kind, device='cpu')
    logs = []
    model.w2log = lambda *a: logs.append(' '.join(str(e) for e in a))
    tr, va = _tiny_dataset(kind, 40, 1), _tiny_dataset(kind, 24, 2)
    fn = str(tmp_path / 'best.pth')
    perf0, _ = model.evaluate_performance_on_dataset(va)
    model.fit(_loader(tr, 8), _loader(va, 8), add_noise=True, verbose=True, early_stop=True, model_filename=fn)
    assert model.finished_training_iters == 3 * 5 and model.add_noise is True
    assert sum(ln.startswith('====> Epoch') for ln in logs) == 3
    assert any(ln.startswith('training epoch: 1 [0/40 (0%)]\tCMPL:') for ln in logs)
    assert any(ln.startswith('Valid rolling mem:') for ln in logs)
    assert os.path.exists(fn)                      # first epoch is always an improvement over -inf
    perf1, txt = model.evaluate_performance_on_dataset(va, return_full_data=True)
    assert perf1['x1_rmse'] < perf0['x1_rmse']     # it trains
    keys = {'losses', 'x1_rmse', 'x1_r2', 'x1_pearr', 'x1_ll', 'model_class', 'z1'}
    if kind != 'pvae':
        keys |= {'y_acc', 'y_auroc', 'y_aupr', 'pred', 'proba'}
    if kind != 'vfae':
        keys |= {'x2_rmse', 'x2_wI_rmse', 'x2_rec_pearr', 'KL_qz2_qz1', 'KL_qz2_pz2Fz1', 'qz1mu_qz2mu_rmse', 'z2'}
    if kind == 'drvae':
        keys |= {'y_wI_acc', 'y_wI_auroc'}
    assert keys <= set(perf1), keys - set(perf1)
    assert txt.startswith('X1: RMSE:' if kind == 'pvae' else 'Y: Accuracy:')
    # the snapshot is a reference-format state_dict that loads back
    sd = torch.load(fn)
    assert list(sd) == list(model.state_dict())
    model.load_params_from_file(fn)


def test_input_dropout_is_a_noop_like_the_reference(monkeypatch):
    """--x-dropout reaches MLP(input_dropout_rates=...), whose forward drops the inputs and then concatenates
    the undropped ones (src/blocks.py:158-161): same losses with and without it"""
    kernel_ref.install(monkeypatch)
    from oracle import models_ref as M
    ds = _tiny_dataset('drvae', 16, 3)
    out = []
    for rate in (0.0, 0.4):
        model = _tiny_model('drvae', device='cpu', input_x_dropout=rate)
        spec = M.ModelSpec(kind='drvae', dim_x=13, dim_z1=5, dim_z3=4, L=2)
        noise = M.make_noise(spec, 16, seed=11)
        model.add_noise = True
        losses = model.run_on_batch(train_mode=True, noise=noise, x1=ds.x1, x2=ds.x2, s=ds.s, y=ds.y, has_x2=ds.has_x2,
                                    has_y=ds.has_y)
        out.append({k: float(v) for k, v in losses.items()})
    assert out[0] == out[1]


def _cont_dataset(n, seed, device='cpu'):
    from drvae_amd import data as D
    ds = _tiny_dataset('drvae', n, seed, device)
    rs = np.random.RandomState(seed + 10)
    ycont = (0.5 + 0.3 * np.tanh(ds.x1.cpu().numpy()[:, 0]) + 0.05 * rs.standard_normal(n)).clip(0.02, 0.98)
    return D.DrVAEDataset(ds.x1, ds.x2, ds.s, torch.from_numpy(ycont.astype(np.float32)).to(device), ds.has_x2, ds.has_y)


def test_fit_regression_head_cpu(monkeypatch, tmp_path):
    """type_y='cont' (src/DrVAE.py:159-169): sigmoid-constrained Gaussian head, RMSE/R2/Pearson report"""
    kernel_ref.install(monkeypatch)
    model = _tiny_model('drvae', device='cpu', type_y='cont', dim_y=1, epochs=4)
    logs = []
    model.w2log = lambda *a: logs.append(' '.join(str(e) for e in a))
    tr, va = _cont_dataset(40, 1), _cont_dataset(24, 2)
    perf0, _ = model.evaluate_performance_on_dataset(va)
    lv0 = model.encoder_y.encoder_lv.linear_lv.weight.detach().clone()
    model.fit(_loader(tr, 8), _loader(va, 8), add_noise=True, early_stop=True, model_filename=str(tmp_path / 'b.pth'))
    perf1, txt = model.evaluate_performance_on_dataset(va, return_full_data=True)
    assert {'y_rmse', 'y_r2', 'y_pearr', 'y_wI_rmse', 'pred', 'proba'} <= set(perf1) and 'y_acc' not in perf1
    assert txt.startswith('Y: RMSE:') and perf1['y_rmse'] < perf0['y_rmse']
    assert np.allclose(perf1['proba'], np.log(0.05 ** 2))             # the fixed log-variance
    # the unused log-variance head is in the state_dict but never updated (torch's Adam skips grad-less params)
    assert torch.equal(lv0, model.encoder_y.encoder_lv.linear_lv.weight.detach())
    with pytest.raises(NotImplementedError):
        _tiny_model('vfae', device='cpu', type_y='cont', dim_y=1)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_fit_device_batcher_gpu(kind, tmp_path, dev):
    from drvae_amd import data as D
    model = _tiny_model(kind, device='cuda', epochs=4)
    tr, va = _tiny_dataset(kind, 64, 1, 'cuda'), _tiny_dataset(kind, 32, 2, 'cuda')
    w = D.compute_balanced_weights(np.arange(64) % 5)
    batcher = D.DeviceBatcher(tr, w, 16, seed=3)
    fn = str(tmp_path / 'best.pth')
    perf0, _ = model.evaluate_performance_on_dataset(va)
    model.fit(batcher, _loader(va, 8), add_noise=True, verbose=False, early_stop=True, model_filename=fn)
    assert model.finished_training_iters == 4 * 4
    perf1, _ = model.evaluate_performance_on_dataset(va)
    assert perf1['x1_rmse'] < perf0['x1_rmse'] and np.isfinite(float(perf1['losses']['ELBO']))
    assert os.path.exists(fn)
    # same protocol through the tuple-loader path gives a working model too
    if kind == 'drvae':          # regression head through the device batcher (targets fed per step)
        mc = _tiny_model('drvae', device='cuda', type_y='cont', dim_y=1, epochs=3)
        trc, vac = _cont_dataset(64, 1, 'cuda'), _cont_dataset(32, 2, 'cuda')
        p0, _ = mc.evaluate_performance_on_dataset(vac)
        mc.fit(D.DeviceBatcher(trc, w, 16, seed=3), _loader(vac, 8), add_noise=True, early_stop=True, model_filename=fn)
        p1, _ = mc.evaluate_performance_on_dataset(vac)
        assert mc.finished_training_iters == 3 * 4 and p1['y_rmse'] < p0['y_rmse']
    m2 = _tiny_model(kind, device='cuda', epochs=2)
    m2.fit(_loader(tr, 16), _loader(va, 8), add_noise=False, early_stop=False, model_filename=fn)
    assert m2.finished_training_iters == 2 * 4
    # the reference's sampler semantics on the device (any group mix per batch): same protocol, one graph
    m3 = _tiny_model(kind, device='cuda', epochs=3)
    p0, _ = m3.evaluate_performance_on_dataset(va)
    m3.fit(D.DeviceBatcher(tr, w, 16, seed=4, mode='sampler'), _loader(va, 8), add_noise=True, early_stop=False,
           model_filename=fn)
    p1, _ = m3.evaluate_performance_on_dataset(va)
    assert m3.finished_training_iters == 3 * 4 and m3.engine().universal
    assert p1['x1_rmse'] < p0['x1_rmse'] and np.isfinite(float(p1['losses']['ELBO']))
    # ... with a captured step per number-of-pairs bucket (batches re-ordered pairs first)
    m4 = _tiny_model(kind, device='cuda', epochs=3)
    m4.fit(D.DeviceBatcher(tr, w, 16, seed=4, mode='sampler', pair_bucket=4, label_bucket=4), _loader(va, 8), add_noise=True,
           early_stop=False, model_filename=fn)
    p4, _ = m4.evaluate_performance_on_dataset(va)
    assert m4.finished_training_iters == 3 * 4 and p4['x1_rmse'] < p0['x1_rmse']
    assert len(m4.engine()._captures) > 1
    if kind != 'vfae':
        # same draws, same model: the epochs differ by the order of rows inside a batch only (summation order, noise rows)
        assert abs(p4['x1_rmse'] - p1['x1_rmse']) < 0.05 * p1['x1_rmse']


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_device_epoch_mean_is_the_mean_of_the_step_objectives(kind, dev):
    """the epoch's 'Avg train loss' of the device-batcher path comes from the running sums the loss-scalar launch keeps
    on the device (``FusedStep.loss_sum``): it must equal the mean of the per-step objectives read one step at a time"""
    from drvae_amd import data as D
    w = D.compute_balanced_weights(np.arange(64) % 5)
    means = []
    for per_step in (False, True):
        model = _tiny_model(kind, device='cuda', epochs=1)
        tr = _tiny_dataset(kind, 64, 1, 'cuda')
        batcher = D.DeviceBatcher(tr, w, 16, seed=3)
        if not per_step:
            means.append([model._epoch_device(batcher, e, False) for e in range(3)])
            continue
        got = []
        orig = model._epoch_device_body

        def body(eng, bat, epoch, verbose):      # the same replays, scalars read after every step
            tot = 0.0
            for b in range(len(bat)):
                eng.replay()
                torch.cuda.synchronize()
                tot += float(model._train_objective(model._loss_tensors(eng)))
            model.finished_training_iters = eng.iters
            return tot / len(bat)
        model._epoch_device_body = body
        means.append([model._epoch_device(batcher, e, False) for e in range(3)])
    a, b = np.array(means[0]), np.array(means[1])
    assert np.all(np.isfinite(a)) and np.all(np.abs(a) > 1e-3)
    np.testing.assert_allclose(a, b, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_fit_tuple_loader_runs_one_captured_graph(kind, tmp_path, dev):
    """fit() fed by a plain DataLoader whose batches differ in their mix of pairs / labels (what the reference's
    WeightedRandomSampler pipeline yields): every batch goes through ONE batch-independent plan and ONE captured
    graph, and trains exactly like eager launches on the same plan."""
    tr, va = _tiny_dataset(kind, 64, 1, 'cuda'), _tiny_dataset(kind, 32, 2, 'cuda')
    out = []
    for mode in (True, 'eager'):
        model = _tiny_model(kind, device='cuda', epochs=3)
        model.universal_plan = mode
        torch.manual_seed(0)
        model.fit(_loader(tr, 16), _loader(va, 8), add_noise=True, verbose=False, early_stop=False,
                  model_filename=str(tmp_path / 'm.pth'))
        eng = model.engine()
        # one plan per batch SIZE (16 training rows; the whole-set evaluations of fit use the 64- / 32-row ones), none per mix
        assert eng.universal and all(k[0] == 'universal' for k in eng._plans) and len(eng._plans) == 3
        assert model.finished_training_iters == 3 * 4
        if mode is True:
            assert eng._graphs and eng._graph_key == ('universal', 16, 0)
        out.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_captured_evaluation_equals_step_by_step(kind, dev):
    """round 4 (N1 off the host): ``evaluate_performance_on_dataset`` of an HBM-resident dataset is one graph replay +
    one copy; every number equals the step-by-step path's (same kernels; the loss scalars draw fresh Philox noise in both,
    so they are compared on a model whose losses do not depend on the draw: means-only terms exactly, the rest loosely),
    it follows the parameters (a train step between two evaluations changes the result), and it is re-captured when the
    dataset's tensors are replaced."""
    from drvae_amd import fit as F
    model = _tiny_model(kind, device='cuda', epochs=2)
    va = _tiny_dataset(kind, 48, 2, 'cuda')
    g = lambda k: getattr(va, k, None)
    ref, txt_ref = model._evaluate(g('x1'), g('x2'), g('s'), g('y'), g('has_x2'), g('has_y'))
    got, txt = model.evaluate_performance_on_dataset(va)
    ev = F._EvalGraph.get(model, va)
    assert ev is not None and ev.graph is not None, 'the captured path was not taken'
    assert txt == txt_ref
    for k, v in ref.items():
        if k in ('losses', 'model_class'):
            continue
        assert (np.isnan(v) and np.isnan(got[k])) or abs(got[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, got[k], v)
    for k, v in ref['losses'].items():           # sampled terms: same distribution, another draw
        assert abs(float(got['losses'][k]) - float(v)) <= 0.2 * max(1.0, abs(float(v))), (k, float(got['losses'][k]), float(v))
    again, _ = model.evaluate_performance_on_dataset(va)
    assert again['x1_rmse'] == got['x1_rmse'] and F._EvalGraph.get(model, va) is ev
    # parameters move -> the replay sees them
    kw = dict(zip(va.FIELDS if hasattr(va, 'FIELDS') else ('x1', 's', 'y', 'has_y'),
                  (getattr(va, f) for f in (va.FIELDS if hasattr(va, 'FIELDS') else ('x1', 's', 'y', 'has_y')))))
    for _ in range(5):
        model.run_on_batch(train_mode=True, **model._batch_kwargs(tuple(kw.values())))
    moved, _ = model.evaluate_performance_on_dataset(va)
    ref2, _ = model._evaluate(g('x1'), g('x2'), g('s'), g('y'), g('has_x2'), g('has_y'))
    assert moved['x1_rmse'] != got['x1_rmse'] and abs(moved['x1_rmse'] - ref2['x1_rmse']) <= 1e-6 * max(1.0, ref2['x1_rmse'])
    # a dataset whose tensors were replaced is captured again
    va.x1 = va.x1.clone()
    assert F._EvalGraph.get(model, va) is not ev


@pytest.mark.gpu
@pytest.mark.parametrize('X', [1100, 978])
def test_captured_evaluation_beyond_the_register_resident_rows(X, dev):
    """round 5: rows of up to 1024 genes take the one-pass reconstruction statistics behind RAW decoder heads
    (dv_recon_rows / dv_col_moments(r_bias)); wider rows the separate passes behind finished heads -- both equal the
    step-by-step evaluation, metric by metric"""
    from drvae_amd import data as D, fit as F, kernels as K
    model = _tiny_model('drvae', dim_x=X, device='cuda', epochs=2)
    rs = np.random.RandomState(3)
    n = 40
    y = rs.randint(0, 2, n)
    x1 = (rs.standard_normal((n, X)) + 0.5 * (2 * y[:, None] - 1) * (np.arange(X) % 3 == 0)).astype(np.float32)
    hx = (np.arange(n) % 3 == 0).astype(np.int64)
    x2 = ((x1 * 0.7 + 0.2) * hx[:, None]).astype(np.float32)
    hy = (np.arange(n) % 4 != 1).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to('cuda')
    va = D.DrVAEDataset(t(x1), t(x2), t(np.zeros(n, np.int64)), t(y), t(hx), t(hy))
    g = lambda k: getattr(va, k, None)
    ref, _ = model._evaluate(g('x1'), g('x2'), g('s'), g('y'), g('has_x2'), g('has_y'))
    calls = []
    real = K.recon_rows
    K.recon_rows = lambda *a, **kw: (calls.append(kw.get('bias') is not None), real(*a, **kw))[1]
    try:
        got, _ = model.evaluate_performance_on_dataset(va)
    finally:
        K.recon_rows = real
    assert F._EvalGraph.get(model, va) is not None
    assert (len(calls) > 0 and all(calls)) == (X <= K.RECON_ROWS_MAX_X)       # (one pass, raw heads) only up to 1024 genes
    for k, v in ref.items():
        if k in ('losses', 'model_class'):
            continue
        assert (np.isnan(v) and np.isnan(got[k])) or abs(got[k] - v) <= 2e-6 * max(1.0, abs(v)), (k, got[k], v)


@pytest.mark.gpu
def test_captured_evaluation_follows_the_engine(dev):
    """round-4 advisor (medium): the captured evaluation points into the engine's arena / plans; ``.cpu().cuda()`` retires
    the engine (DGMMixin._apply), so the cached graph must go with it -- evaluate, move the model away and back, train,
    evaluate: the numbers are those of the step-by-step path on the CURRENT parameters"""
    from drvae_amd import fit as F
    model = _tiny_model('drvae', device='cuda', epochs=2)
    va = _tiny_dataset('drvae', 48, 2, 'cuda')
    g = lambda k: getattr(va, k, None)
    first, _ = model.evaluate_performance_on_dataset(va)
    ev = F._EvalGraph.get(model, va)
    eng0 = model.engine()
    model.cpu()
    model.cuda()
    assert '_eval_graphs' not in model.__dict__ or not model.__dict__['_eval_graphs']
    batch = tuple(getattr(va, f) for f in va.FIELDS)
    for _ in range(5):
        model.run_on_batch(train_mode=True, **model._batch_kwargs(batch))
    assert model.engine() is not eng0
    got, _ = model.evaluate_performance_on_dataset(va)
    assert F._EvalGraph.get(model, va) is not ev
    ref, _ = model._evaluate(g('x1'), g('x2'), g('s'), g('y'), g('has_x2'), g('has_y'))
    assert got['x1_rmse'] != first['x1_rmse']
    for k in ('x1_rmse', 'x1_pearr', 'x2_rmse', 'y_auroc', 'y_aupr', 'y_acc'):
        assert abs(got[k] - ref[k]) <= 1e-6 * max(1.0, abs(ref[k])), (k, got[k], ref[k])
    # a host-resident label array: the captured path declines (no pageable copy under capture), the step-by-step one runs
    va2 = _tiny_dataset('drvae', 48, 2, 'cuda')
    va2.y = va2.y.cpu()
    assert F._EvalGraph.get(model, va2) is None
    perf, _ = model.evaluate_performance_on_dataset(va2)
    assert np.isfinite(perf['x1_rmse'])
