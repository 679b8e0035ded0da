"""N2 (input pipeline): dataset wrapping and sampler weights against the reference's own outputs
(golden G11), and the stratified device batcher's structure / sampling marginals (CPU)."""
import copy

import numpy as np
import pytest
import torch

from drvae_amd import data as D
from tests.golden import cases as C


@pytest.fixture(scope='module')
def G():
    return C.load('blocks')


def test_balanced_weights_match_reference(G):
    c = C.block_inputs('G11')
    np.testing.assert_allclose(D.compute_balanced_weights(c['labels']).numpy(), G['G11/w_plain'], rtol=1e-12)
    np.testing.assert_allclose(D.compute_balanced_weights(c['labels'], c['ratio'], c['token']).numpy(), G['G11/w_ratio'],
                               rtol=1e-12)
    with pytest.raises(AssertionError):
        D.compute_balanced_weights(c['labels'], unlabeled_data_ratio=0.3)


@pytest.mark.parametrize('mode', ['both', 'pair_only', 'sing_only'])
@pytest.mark.parametrize('rm', [False, True])
def test_wrap_in_dataset_matches_reference(G, mode, rm):
    c = C.block_inputs('G11')
    ds, dd = D.wrap_in_DrVAEDataset(copy.deepcopy(c['sing']), copy.deepcopy(c['pair']), concat=mode, remove_unlabeled=rm)
    for fld in D.DrVAEDataset.FIELDS:
        want = G['G11/%s_%d/%s' % (mode, int(rm), fld)]
        got = getattr(ds, fld).numpy()
        assert got.dtype == want.dtype, fld
        np.testing.assert_array_equal(got, want, err_msg=fld)
    assert len(ds) == len(G['G11/%s_%d/x1' % (mode, int(rm))])
    row = ds[1]
    assert len(row) == 6 and torch.equal(row[0], ds.x1[1])
    with pytest.raises(ValueError):
        D.wrap_in_DrVAEDataset(c['sing'], c['pair'], concat='nope')


def test_device_batcher_structure_and_marginals():
    c = C.block_inputs('G11')
    ds, _ = D.wrap_in_DrVAEDataset(copy.deepcopy(c['sing']), copy.deepcopy(c['pair']))
    w = torch.rand(len(ds), generator=torch.Generator().manual_seed(0)).double() + 0.1
    b = D.DeviceBatcher(ds, w, 12, seed=3)
    assert sum(b.group_counts) == 12
    hy, hx = ds.has_y.numpy().astype(bool), ds.has_x2.numpy().astype(bool)
    # expected composition under the weights, rounded
    for cnt, (gy, gx) in zip(b.group_counts, D._GROUPS):
        share = float(w[(hy == bool(gy)) & (hx == bool(gx))].sum() / w.sum()) * 12
        assert abs(cnt - share) < 1.0
    counts = np.zeros(len(ds))
    for _ in range(600):
        idx = b.next_indices().numpy()
        assert len(idx) == 12
        # fixed group order ls, us, lp, up with fixed sizes
        np.testing.assert_array_equal(hy[idx].astype(int), b.has_y)
        np.testing.assert_array_equal(hx[idx].astype(int), b.has_x2)
        np.add.at(counts, idx, 1)
    # within a group, rows are drawn proportionally to their weights
    for (gy, gx) in D._GROUPS:
        m = (hy == bool(gy)) & (hx == bool(gx))
        if m.sum() > 1:
            np.testing.assert_allclose(counts[m] / counts[m].sum(), (w[m] / w[m].sum()).numpy(), atol=0.04)


@pytest.mark.parametrize('tag', list(C.sampler_cases()))
def test_sampler_mode_matches_reference_pipeline(tag):
    """N2, mode='sampler': the reference's WeightedRandomSampler(weights, len) + DataLoader(drop_last=len>=batch)
    (src/run_drvae.py:150-162), recorded in tests/golden/sampler.npz -- same weights, same number of batches per
    epoch, same batch size, and the per-cell-line draw frequencies of both (the reference's recorded histogram and
    the on-device sampler's) agree with the weights' marginals and with each other."""
    G = C.load('sampler')
    c = C.sampler_cases()[tag]
    cid, n = c['cid'], len(c['cid'])
    w = D.compute_balanced_weights(cid)
    np.testing.assert_allclose(w.numpy(), G[tag + '/weights'], rtol=1e-12)
    ds = D.DrVAEDataset(torch.zeros(n, 3), torch.zeros(n, 3), torch.zeros(n, dtype=torch.int64), torch.zeros(n, 1, dtype=torch.int64),
                        torch.from_numpy((np.arange(n) % 2).astype(np.int32)), torch.from_numpy((np.arange(n) % 3 == 0).astype(np.int32)))
    bat = D.DeviceBatcher(ds, w, c['batch_size'], seed=5, mode='sampler')
    assert len(bat) == int(G[tag + '/n_batches'])
    assert [bat.batch_size] == G[tag + '/batch_rows'].tolist()
    ncls = int(cid.max()) + 1
    hist = np.zeros(ncls, np.int64)
    draws = int(G[tag + '/draws'])
    for _ in range(draws // bat.batch_size):
        np.add.at(hist, cid[bat.next_indices().numpy()], 1)
    p = np.array([w.numpy()[cid == k].sum() for k in range(ncls)]) / float(w.sum())
    for h, tot in ((hist, hist.sum()), (G[tag + '/class_hist'], draws)):
        sd = np.sqrt(tot * p * (1 - p))
        assert np.all(np.abs(h - tot * p) < 5 * sd + 1), (h, tot * p)
    # balanced weights: every cell line equally likely (that is their purpose, src/utils.py:292-300)
    np.testing.assert_allclose(p, 1.0 / ncls, rtol=1e-9)


@pytest.mark.parametrize('tag', list(C.sampler_cases()))
def test_cpu_generator_reproduces_the_reference_index_stream(tag):
    """N2, DeviceBatcher(mode='sampler', generator='cpu'): under torch.manual_seed the epoch tables equal, bit for bit,
    the batches the reference's DataLoader + WeightedRandomSampler pipeline yielded (first two epochs recorded in
    tests/golden/sampler.npz by tests/golden/make_golden.py) -- through begin_epoch (graph-resident feed) and through
    next_indices (host-driven feed) alike"""
    G = C.load('sampler')
    c = C.sampler_cases()[tag]
    cid, n = c['cid'], len(c['cid'])
    w = D.compute_balanced_weights(cid)
    ds = D.DrVAEDataset(torch.zeros(n, 3), torch.zeros(n, 3), torch.zeros(n, dtype=torch.int64), torch.zeros(n, 1, dtype=torch.int64),
                        torch.from_numpy((np.arange(n) % 2).astype(np.int32)), torch.from_numpy((np.arange(n) % 3 == 0).astype(np.int32)))
    want = [G['%s/epoch%d_idx' % (tag, e)] for e in range(2)]
    bat = D.DeviceBatcher(ds, w, c['batch_size'], seed=5, mode='sampler', generator='cpu')
    torch.manual_seed(1234)
    for e in range(2):
        got = bat._reference_epoch().numpy()
        np.testing.assert_array_equal(got, want[e])
    torch.manual_seed(1234)
    bat2 = D.DeviceBatcher(ds, w, c['batch_size'], seed=5, mode='sampler', generator='cpu')
    stream = np.stack([bat2.next_indices().numpy() for _ in range(2 * len(bat2))])
    np.testing.assert_array_equal(stream, np.concatenate(want))


def test_sampler_mode_epoch_table(monkeypatch):
    """the epoch's index table of mode='sampler': len(dataset) // batch rows of i.i.d. draws over ALL rows (any group
    mix per batch), one batch-independent plan bound to the engine"""
    from oracle import models_ref as M
    from tests import kernel_ref
    from tests.test_engine_cpu import make_engine
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec('drvae')
    eng, _ = make_engine(spec, M.init_params(spec, 3, as_numpy=True))
    big = M.make_batch(spec, 100, seed=4)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y'))
    bat = D.DeviceBatcher(ds, D.compute_balanced_weights(np.arange(100) % 7), 16, seed=2, mode='sampler')
    p = bat.bind(eng)
    assert p.universal and eng.universal and len(bat) == 6
    tab = bat.begin_epoch()
    assert tuple(tab.shape) == (6, 16) and int(tab.min()) >= 0 and int(tab.max()) < 100
    comp = {tuple(np.bincount(2 * big['has_y'][r] + big['has_x2'][r], minlength=4)) for r in tab.numpy()}
    assert len(comp) > 1            # the composition varies from batch to batch
    # explicit path: feed() hands the batch's own flags to the masks kernel
    idx = bat.feed()
    assert torch.equal(p.hx_dev.long(), ds.has_x2[idx].long()) and torch.equal(p.y_dev.long(), ds.y.reshape(-1)[idx].long())
    eng.train_step()
    assert all(np.isfinite(v) for v in eng.losses().values())


def test_sampler_mode_pair_buckets(monkeypatch):
    """pair_bucket: the same draws as the plain sampler feed, every batch re-ordered pairs first; every batch gets the
    plan whose pair slots are its number of pairs rounded up to the bucket width (and the table feeds all of them)"""
    from oracle import models_ref as M
    from tests import kernel_ref
    from tests.test_engine_cpu import make_engine
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec('drvae')
    big = M.make_batch(spec, 100, seed=4)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y'))
    w = D.compute_balanced_weights(np.arange(100) % 7)
    tabs = []
    for bucket in (None, 4):
        eng, _ = make_engine(spec, M.init_params(spec, 3, as_numpy=True))
        bat = D.DeviceBatcher(ds, w, 16, seed=2, mode='sampler', pair_bucket=bucket)
        bat.bind(eng)
        tabs.append(bat.begin_epoch().numpy().copy())
    plain, sorted_ = tabs
    hx = big['has_x2'].reshape(-1)
    np.testing.assert_array_equal(np.sort(plain, 1), np.sort(sorted_, 1))
    for k, row in enumerate(sorted_):
        n = int(hx[row].sum())
        assert hx[row][:n].all() and not hx[row][n:].any()
        assert bat.batch_slots[k] == min(16, max(4, -(-n // 4) * 4))
        # stable: pairs and singletons each keep the order they were drawn in
        np.testing.assert_array_equal(row[:n], plain[k][hx[plain[k]] == 1])
        np.testing.assert_array_equal(row[n:], plain[k][hx[plain[k]] == 0])
    keys = {('universal', 16, 0) if s == 16 else ('universal', 16, 0, int(s)) for s in set(bat.batch_slots.tolist()) | {16}}
    assert keys <= set(eng._plans) and all(eng._plans[k].live_feed is eng._plans[('universal', 16, 0)].feed for k in keys)
    with pytest.raises(RuntimeError):
        bat.select(0)               # nothing captured yet
    # no pairs in the model: buckets are switched off
    spec_v = C.tiny_spec('vfae')
    eng_v, _ = make_engine(spec_v, M.init_params(spec_v, 3, as_numpy=True))
    bat_v = D.DeviceBatcher(ds, w, 16, seed=2, mode='sampler', pair_bucket=4)
    bat_v.bind(eng_v)
    assert bat_v.pair_bucket is None


def test_sampler_mode_label_buckets(monkeypatch):
    """label_bucket: every batch in the order unlabeled pairs | labeled pairs | labeled singles | unlabeled singles;
    its plan has pair slots for (at least) its pairs and a labeled range inside its run of labeled rows"""
    from oracle import models_ref as M
    from tests import kernel_ref
    from tests.test_engine_cpu import make_engine
    kernel_ref.install(monkeypatch)
    spec = C.tiny_spec('drvae')
    big = M.make_batch(spec, 100, seed=4)
    t = lambda k: torch.from_numpy(big[k].copy())
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y'))
    w = D.compute_balanced_weights(np.arange(100) % 7)
    eng, _ = make_engine(spec, M.init_params(spec, 3, as_numpy=True))
    bat = D.DeviceBatcher(ds, w, 16, seed=2, mode='sampler', pair_bucket=4, label_bucket=2)
    plain = D.DeviceBatcher(ds, w, 16, seed=2, mode='sampler')
    eng2, _ = make_engine(spec, M.init_params(spec, 3, as_numpy=True))
    bat.bind(eng)
    plain.bind(eng2)
    tab, ref = bat.begin_epoch().numpy(), plain.begin_epoch().numpy()
    hx, hy = big['has_x2'].reshape(-1), big['has_y'].reshape(-1)
    np.testing.assert_array_equal(np.sort(tab, 1), np.sort(ref, 1))
    some_range = False
    for row, (P, a, b) in zip(tab, bat.batch_specs):
        px, py = hx[row].astype(bool), hy[row].astype(bool)
        n_p, n_up, n_l = int(px.sum()), int((px & ~py).sum()), int(py.sum())
        grp = np.where(px, py.astype(int), 3 - py.astype(int))
        assert (np.diff(grp) >= 0).all()                      # the four groups in order
        assert n_p <= P <= max(4, n_p + 3) and px[:n_p].all() and not px[n_p:].any()
        assert py[n_up:n_up + n_l].all() and n_l == py.sum()
        assert (a, b) == (0, 0) or (n_up <= a < b <= n_up + n_l and a % 2 == 0 and b % 2 == 0 and b - a > n_l - 4)
        some_range |= b > a
        assert eng._plans[eng.set_structure_universal(16, P, (a, b)).key].live_feed is not None
    assert some_range
    # models without labels: the label buckets are switched off
    spec_p = C.tiny_spec('pvae')
    eng_p, _ = make_engine(spec_p, M.init_params(spec_p, 3, as_numpy=True))
    bat_p = D.DeviceBatcher(ds, w, 16, seed=2, mode='sampler', pair_bucket=4, label_bucket=2)
    bat_p.bind(eng_p)
    assert bat_p.label_bucket is None and bat_p.pair_bucket == 4


@pytest.mark.gpu
def test_epoch_feed_follows_a_dataset_edited_in_place(dev):
    """round-4 advisor: the graph-resident feed reads a row-padded COPY of a 978-gene dataset; an in-place edit of the
    dataset between two epochs (normalisation, augmentation) must reach it at the next ``begin_epoch``"""
    from oracle import models_ref as M
    from tests.test_engine_cpu import make_engine
    spec = C.tiny_spec('drvae', dim_x=13)
    eng, _ = make_engine(spec, M.init_params(spec, 3, as_numpy=True), dev)
    big = M.make_batch(spec, 64, seed=4)
    t = lambda k: torch.from_numpy(big[k].copy()).to(dev)
    ds = D.DrVAEDataset(t('x1'), t('x2'), t('s'), t('y'), t('has_x2'), t('has_y'))
    bat = D.DeviceBatcher(ds, torch.ones(64), 16, seed=2)
    bat.bind(eng)
    bat.begin_epoch()
    fd = eng.plan.feed
    assert fd.x1.data_ptr() != ds.x1.data_ptr() and torch.equal(fd.x1, ds.x1)      # 13 genes: a padded copy
    ds.x1.mul_(2.0).add_(1.0)
    assert not torch.equal(fd.x1, ds.x1)
    bat.begin_epoch()
    assert eng.plan.feed is fd and torch.equal(fd.x1, ds.x1) and torch.equal(fd.x2, ds.x2)
    assert not bool(fd.x1._base[:, 13:].any())
