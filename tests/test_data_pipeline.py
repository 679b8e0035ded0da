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
