"""CPU, world_size 2 over gloo: the data-parallel train step (rows sharded over ranks,
global normalisers, ONE sum all-reduce of the gradient arena + loss tail, identical fused
Adam on every rank) reproduces the single-process step on the concatenated batch.
The HIP launchers are replaced by their PyTorch references (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import models_ref as M
from tests.golden import cases as C


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _install_refs():
    import drvae_amd.kernels as K
    from tests import kernel_ref as R
    for name in R.FUNCTIONS:
        setattr(K, name, getattr(R, name))


def _run_steps(spec, params, batch, noises, counts, allreduce, lo=None, hi=None):
    from tests.test_engine_cpu import make_engine, set_batch
    if lo is not None:
        batch = {k: v[lo:hi] for k, v in batch.items()}
        noises = [M.slice_noise(n, lo, hi) for n in noises]
    eng, arena = make_engine(spec, params)
    set_batch(eng, batch, counts=counts)
    out = []
    for noise in noises:
        eng.train_step(noise, allreduce=allreduce)
        out.append(arena.loss.clone().numpy())
    return np.stack(out), arena.param.clone().numpy(), arena.grad.clone().numpy()


def _worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    _install_refs()
    from drvae_amd import dist as D
    r, w, _ = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    spec, batch, noises, params = _case(kind)
    n = batch['x1'].shape[0]
    lo, hi = D.shard_rows(n, rank, world)
    counts = D.global_counts(batch['has_x2'][lo:hi], batch['has_y'][lo:hi], kind, spec.semi_supervised)
    res = _run_steps(spec, params, batch, noises, counts, D.allreduce_sum, lo, hi)
    q.put((rank, counts) + res)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _case(kind):
    spec = C.tiny_spec(kind, dim_y=3 if kind == 'vfae' else 2)
    n = 12
    batch = M.make_batch(spec, n, seed=11)
    # uneven group mix between the two shards (so per-shard counts differ from global/2)
    pat = 'acbdab' + 'ddcbbb'
    hy = np.array([ch in 'ac' for ch in pat], np.int64)
    hx = np.array([ch in 'cd' for ch in pat], np.int64)
    if kind == 'pvae':
        hy[:] = 0
    if kind == 'vfae':
        hx[:] = 0
    rs = np.random.RandomState(5)
    x2 = (batch['x1'] + 0.1 * rs.standard_normal(batch['x1'].shape)).astype(np.float32)
    batch.update(x2=x2 * hx[:, None].astype(np.float32), has_x2=hx, has_y=hy)
    noises = [M.make_noise(spec, n, seed=20 + i) for i in range(3)]
    return spec, batch, noises, M.init_params(spec, 9, as_numpy=True)


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_two_ranks_equal_one_rank_on_concatenated_batch(kind, monkeypatch):
    from tests import kernel_ref
    kernel_ref.install(monkeypatch)
    spec, batch, noises, params = _case(kind)
    single = _run_steps(spec, params, batch, noises, None, None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n = batch['x1'].shape[0]
    want_counts = (n, int(batch['has_x2'].sum()), int(batch['has_y'].sum()))
    for rank, counts, losses, prm, grad in got:
        assert tuple(counts) == want_counts
        np.testing.assert_allclose(losses, single[0], rtol=1e-5, atol=1e-6)      # summed loss tail == global loss
        np.testing.assert_allclose(grad, single[2], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(prm, single[1], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(got[0][3], got[1][3])                          # replicas stay bit-identical


def test_shard_rows_and_counts_single_process():
    from drvae_amd import dist as D
    assert D.shard_rows(1200, 3, 8) == (450, 600)
    with pytest.raises(AssertionError):
        D.shard_rows(10, 0, 3)
    assert D.global_counts([1, 0, 1, 1], [1, 1, 0, 0]) == (4, 3, 2)
    assert D.global_counts([1, 0, 1, 1], [1, 1, 0, 0], kind='vfae', semi_supervised=False) == (2, 0, 2)
    assert D.init_from_env() == (0, 1, 0)


# ---------------------------------------------------------------- data parallelism ABOVE the engine (round 5)
# ``model.enable_data_parallel()`` + ``run_on_batch`` / ``fit`` / ``DeviceBatcher(mode='sampler')``: a two-rank job
# trains like one process on the concatenated global batches (normalisers src/DrVAE.py:611-616; the
# input pipeline src/run_drvae.py:150-166)

def _model_case(kind):
    from tests.test_fit import _tiny_dataset, _tiny_model
    model = _tiny_model(kind, device='cpu', epochs=2)
    return model, _tiny_dataset(kind, 64, 1), _tiny_dataset(kind, 24, 2)


def _shard_loader(ds, bs, rank, world):
    """batches of ``bs`` rows per rank: global batch k = rows [k*bs*world, (k+1)*bs*world), rank r takes its slice"""
    from tests.test_fit import _Loader
    fields = ds.FIELDS if hasattr(ds, 'FIELDS') else ('x1', 's', 'y', 'has_y')
    gb = bs * world
    ld = _Loader([tuple(getattr(ds, f)[k + rank * bs:k + (rank + 1) * bs] for f in fields)
                  for k in range(0, len(ds) - gb + 1, gb)])
    ld.dataset = ds
    return ld


def _sampler_epochs(model, tr, bs, dp, n_epochs=2):
    """what ``fit._epoch_device`` does per epoch, minus the hipGraph capture (CPU: eager steps through the feed)"""
    from drvae_amd import data as D
    w = D.compute_balanced_weights(np.arange(len(tr)) % 5)
    bat = D.DeviceBatcher(tr, w, bs, seed=3, mode='sampler')
    eng = model.engine()
    losses, tabs = [], []
    for _ in range(n_epochs):
        bat.bind(eng, dp=dp)
        eng.add_noise, eng.iters = True, model.finished_training_iters
        tabs.append(bat.begin_epoch().clone().numpy())
        for _b in range(len(bat)):
            eng.train_step(allreduce=model._allreduce)
            losses.append(eng.arena.loss.clone().numpy())
        model.finished_training_iters = eng.iters
    return np.stack(losses), np.stack(tabs), bat


def _model_worker(rank, world, port, kind, what, tmp, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), DRVAE_DIST_BACKEND='gloo')
        torch.set_num_threads(1)
        _install_refs()
        import torch.distributed as dist
        model, tr, va = _model_case(kind)
        if rank == 1:       # a replica that starts elsewhere: the broadcast of enable_data_parallel aligns it
            with torch.no_grad():
                for prm in model.parameters():
                    prm.add_(0.01)
        assert model.enable_data_parallel() == (rank, world) and model._dp == (rank, world)
        out = {}
        if what == 'fit':
            logs = []
            model.verbose_log = True
            import builtins
            real_print = builtins.print
            builtins.print = lambda *a, **k: logs.append(' '.join(str(e) for e in a))
            try:
                model.fit(_shard_loader(tr, 8, rank, world), _shard_loader(va, 8, 0, 1), add_noise=True, verbose=False,
                          early_stop=True, model_filename=os.path.join(tmp, 'best_%s.pth' % kind))
            finally:
                builtins.print = real_print
            out['n_log'] = len(logs)
            out['valid'] = [ln for ln in logs if ln.startswith('Valid:')]
        else:
            losses, tabs, bat = _sampler_epochs(model, tr, 8, model._dp)
            out.update(losses=losses, tabs=tabs, gtab=bat.global_table.numpy().copy(),
                       gcounts=model.engine().plan.feed.gcounts.numpy().copy())
        eng = model.engine()
        out.update(param=eng.arena.param.clone().numpy(), iters=model.finished_training_iters,
                   m=eng.arena.exp_avg.clone().numpy())
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def _run_two(kind, what, tmp):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, kind, what, str(tmp), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r, o in got:
        assert isinstance(o, dict), o
    return [o for _, o in got]


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_sampler_feed_epochs_two_ranks_equal_one_rank(kind, monkeypatch, tmp_path):
    """DeviceBatcher(mode='sampler') under dp=(rank, 2): both ranks draw the same global table, run their columns, take
    the batch's GLOBAL (N_pairs, N_labeled) from the table -- and train exactly like one process whose batches are the
    global ones (16 rows) on the same table"""
    from tests import kernel_ref
    kernel_ref.install(monkeypatch)
    model, tr, _ = _model_case(kind)
    single_losses, single_tabs, _ = _sampler_epochs(model, tr, 16, None)
    single_param = model.engine().arena.param.clone().numpy()
    r0, r1 = _run_two(kind, 'sampler', tmp_path)
    np.testing.assert_array_equal(r0['gtab'], r1['gtab'])                    # the shared index table
    np.testing.assert_array_equal(np.concatenate([r0['tabs'], r1['tabs']], 2), single_tabs)
    hx = tr.has_x2.numpy() if hasattr(tr, 'has_x2') else np.zeros(len(tr), np.int64)
    hy = tr.has_y.numpy()
    want = np.stack([hx[r0['gtab']].sum(1) * (kind != 'vfae'), hy[r0['gtab']].sum(1) * (kind != 'pvae')], 1)
    np.testing.assert_array_equal(r0['gcounts'], want)
    assert r0['iters'] == r1['iters'] == 2 * (64 // 16)
    for r in (r0, r1):
        np.testing.assert_allclose(r['losses'], single_losses, rtol=1e-4, atol=1e-6)
        err = np.linalg.norm(r['param'] - single_param) / np.linalg.norm(single_param)
        assert err < 1e-4, err
    np.testing.assert_array_equal(r0['param'], r1['param'])                   # replicas stay bit-identical
    np.testing.assert_array_equal(r0['m'], r1['m'])


@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_model_fit_two_ranks_equal_one_rank(kind, monkeypatch, tmp_path):
    """``model.enable_data_parallel()`` + ``fit`` on per-rank tuple loaders: run_on_batch takes the global counts (a
    three-number host all-reduce), the exchange sums [losses | gradients]; both ranks evaluate the whole validation
    set with the same draws, decide the same, and only rank 0 logs and writes the snapshot"""
    from tests import kernel_ref
    kernel_ref.install(monkeypatch)
    model, tr, va = _model_case(kind)
    logs = []
    model.w2log = lambda *a: logs.append(' '.join(str(e) for e in a))
    model.fit(_shard_loader(tr, 16, 0, 1), _shard_loader(va, 8, 0, 1), add_noise=True, early_stop=True,
              model_filename=str(tmp_path / 'single.pth'))
    single = model.engine().arena.param.clone().numpy()
    r0, r1 = _run_two(kind, 'fit', tmp_path)
    assert r0['iters'] == r1['iters'] == model.finished_training_iters == 2 * 4
    for r in (r0, r1):
        err = np.linalg.norm(r['param'] - single) / np.linalg.norm(single)
        assert err < 1e-4, err
    np.testing.assert_array_equal(r0['param'], r1['param'])
    assert r0['n_log'] > 0 and r1['n_log'] == 0                                # rank 0 speaks for the job
    assert os.path.exists(str(tmp_path / ('best_%s.pth' % kind)))
    # the validation lines of the two-rank job == the single-process ones (same parameters, same evaluation draws)
    vs = [ln.split('\t', 1)[1] for ln in logs if ln.startswith('Valid:')]
    vd = [ln.split('\t', 1)[1] for ln in r0['valid']]
    assert len(vs) == len(vd) == 2
    for a, b in zip(vs, vd):
        fa = [float(x) for x in __import__('re').findall(r'-?\d+\.\d+', a)]
        fb = [float(x) for x in __import__('re').findall(r'-?\d+\.\d+', b)]
        np.testing.assert_allclose(fa, fb, rtol=2e-3, atol=2e-3)
