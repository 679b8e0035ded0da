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
