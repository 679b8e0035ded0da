"""-m gpu: the multi-rank train step on real kernels.  Two ranks share the one GPU of the test box and
exchange through gloo (RCCL refuses two ranks on one device), which exercises everything but the
transport: global-count normalisation, the graph split around the exchange, the two-piece overlapped
all-reduce (decoder block early, the rest + loss scalars late) and replica consistency."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK='0')
        import torch.distributed as dist
        from drvae_amd import dist as D
        from oracle import models_ref as M
        from tests.test_engine_cpu import make_engine, set_batch
        D.init_from_env(backend='gloo')
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(0)
        spec = M.ModelSpec(kind='drvae', L=2)
        params = M.init_params(spec, 3, as_numpy=True)
        full = M.make_batch(spec, 96, seed=5)
        lo, hi = D.shard_rows(96, rank, world)
        batch = {k: v[lo:hi] for k, v in full.items()}
        counts = D.global_counts(batch['has_x2'], batch['has_y'])
        assert counts[0] == 96
        res = {}
        for mode in ('overlap', 'split', 'eager'):
            eng, arena = make_engine(spec, params, dev)
            eng.seed = 77 + rank
            set_batch(eng, batch, dev, counts=counts)
            eng.train_step(allreduce=D.allreduce_sum)
            if mode == 'eager':
                for _ in range(3):
                    eng.train_step(allreduce=D.allreduce_sum)
            else:
                eng.capture(split_for_allreduce=mode if mode == 'overlap' else True)
                assert len(eng._graphs) == (3 if mode == 'overlap' else 2)
                ar = D.OverlappedAllReduce() if mode == 'overlap' else D.allreduce_sum
                for _ in range(3):
                    eng.replay(ar)
            torch.cuda.synchronize()
            eng.check_sync()
            res[mode] = (arena.param.clone().cpu(), arena.loss.clone().cpu())
        same = all(torch.equal(res['eager'][i], res[m][i]) for m in ('overlap', 'split') for i in (0, 1))
        # cfg 2 size (150 rows per rank) on on-device Philox noise keyed by GLOBAL row: the two-rank job must train
        # exactly like one process on the concatenated 300 rows (same seed, same draws per global row)
        full = M.make_batch(spec, 300, seed=6)
        lo, hi = D.shard_rows(300, rank, world)
        shard = {k: v[lo:hi] for k, v in full.items()}
        counts = D.global_counts(shard['has_x2'], shard['has_y'])
        eng, arena = make_engine(spec, params, dev)
        eng.seed, eng.row0 = 4242, lo
        set_batch(eng, shard, dev, counts=counts)
        eng.train_step(allreduce=D.allreduce_sum)
        eng.capture(split_for_allreduce=True)
        for _ in range(3):
            eng.replay(D.allreduce_sum)
        torch.cuda.synchronize()
        eng.check_sync()
        one, a1 = make_engine(spec, params, dev)
        one.seed = 4242
        set_batch(one, full, dev)
        one.train_step()
        for _ in range(3):
            one.train_step()
        torch.cuda.synchronize()
        # (norm-wise: Adam turns a gradient that is pure summation-order noise into a full +-lr move of that element)
        perr = float((arena.param - a1.param).norm() / a1.param.norm())
        gerr = float((arena.grad - a1.grad).norm() / a1.grad.norm())
        assert gerr < 1e-4, gerr
        lerr = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(eng.losses().values(), one.losses().values())
                   if b != 0.0)
        same = same and perr < 1e-4 and lerr < 1e-4
        if not (perr < 1e-4 and lerr < 1e-4):
            raise AssertionError('2-rank Philox job != 1-rank job on the concatenated batch: %g %g' % (perr, lerr))
        # replicas agree: compare rank 0's parameters with this rank's
        ref = res['overlap'][0].clone().to(dev)
        dist.broadcast(ref, src=0)
        replicas = torch.equal(ref.cpu(), res['overlap'][0])
        q.put((rank, bool(same), bool(replicas), bool(torch.isfinite(res['overlap'][1]).all())))
        dist.destroy_process_group()
    except Exception as e:       # surface the failure instead of a hung join
        import traceback
        q.put((rank, 'error', traceback.format_exc(), str(e)))


def test_two_rank_overlapped_exchange_matches_eager(dev):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for o in out:
        assert o[1] != 'error', o[2]
        assert o[1] and o[2] and o[3], o


def test_step_path_over_rccl_single_rank(dev):
    """the data-parallel step path over the REAL transport: a one-rank RCCL communicator in a child process
    (process group, split graphs, collective launches and their stream events between the graph replays, CU
    partition) -- single exchange and two overlapped pieces -- trains to exactly the losses of the plain
    single-GPU step"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--steps', '12', '--warmup', '3', '--no-cpu-baseline',
           '--no-roofline', '--no-steady', '--no-extras']

    def run(*args, **env):
        e = dict(os.environ, DRVAE_SIDE_CUS='64', **env)
        e.pop('RANK', None)
        e.pop('WORLD_SIZE', None)
        out = subprocess.run(cmd + list(args), env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
        return json.loads(line)
    plain = run()
    single = run(DRVAE_FORCE_DP='1', MASTER_PORT='29561')
    pieces = run('--dp-exchange', 'overlap', DRVAE_FORCE_DP='1', MASTER_PORT='29562')
    captured = run('--dp-exchange', 'captured', DRVAE_FORCE_DP='1', MASTER_PORT='29563')
    assert plain['finite'] and single['finite'] and pieces['finite'] and captured['finite']
    assert single['losses_last_step'] == plain['losses_last_step'] == pieces['losses_last_step'] == captured['losses_last_step']
    assert all(v == 0 for v in single['chain_wait_ticks'][0::2] + pieces['chain_wait_ticks'][0::2] + captured['chain_wait_ticks'][0::2])
    # round 6: the headline of a data-parallel run stays the two-graph form, the captured form (RCCL's all-reduce inside the step's
    # graph, the side chain drawing the next step's noise behind the join) is probed after it in the same process and rides in the same line
    assert captured['config']['dp_exchange'] == 'captured' and 'exchange_modes' not in captured
    em = single['exchange_modes']
    assert em['single']['ms_per_step'] == single['ms_per_step']
    assert em['captured'].get('finite') is True and em['captured']['noise_drawn_ahead'] is True, em
    assert em['captured']['ms_per_step'] < 1.5 * single['ms_per_step']


def test_bench_self_launch_two_ranks_one_gpu(dev):
    """``python bench.py --gpus 2`` with no launcher: the parent starts two fresh rank processes before touching
    the GPU and relays rank 0's line.  On the one-GPU test box the ranks share the device and exchange through
    gloo (DRVAE_DIST_BACKEND); the weak-scaling job line must report both ranks' rows and finite losses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, DRVAE_DIST_BACKEND='gloo', DRVAE_SIDE_CUS='64')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        e.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '3',
                          '--no-cpu-baseline', '--no-roofline', '--no-steady', '--no-extras'], env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['finite'] and r['config']['global_batch'] == 300
    assert r['config']['dist_backend'] == 'gloo' and r['config']['rccl_ranks'] == 0
    assert all(v == 0 for v in r['chain_wait_ticks'][0::2])


def _bench_env():
    e = dict(os.environ, DRVAE_DIST_BACKEND='gloo', DRVAE_SIDE_CUS='64')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        e.pop(k, None)
    return e


def test_bench_self_launch_eight_ranks_one_gpu(dev):
    """the launch path of the 8-GPU scaling run on the one-GPU box: ``python bench.py --gpus 8`` starts eight fresh
    rank processes (a free port, RANK / LOCAL_RANK / WORLD_SIZE set, rank -> device modulo the devices present, gloo
    since RCCL refuses two ranks per device), relays rank 0's single JSON line and exits 0"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '6', '--warmup', '2',
                          '--no-cpu-baseline', '--no-roofline', '--no-steady', '--no-extras'], env=_bench_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 8 and r['finite'] and r['config']['global_batch'] == 1200 and r['scaling'] == 'weak'
    assert r['config']['parallelism'] == 'dp8' and r['config']['dist_backend'] == 'gloo'
    assert r['value'] == pytest.approx(1200 * 2 * 6 / (r['ms_per_step'] * 6e-3), rel=1e-3)      # whole-job samples/s
    assert r['exchange'] is not None and r['exchange']['bytes'] > 9e6
    assert all(v == 0 for v in r['chain_wait_ticks'][0::2])


def test_bench_self_launch_ends_the_job_when_a_rank_dies(dev):
    """a rank that dies (test hook --fail-rank) must end the whole job with a non-zero exit code well inside the
    time-out: the launcher ends the surviving ranks (they sit in a collective that can never complete) -- plain child
    processes, nothing is ever re-exec'ed"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--steps', '6', '--warmup', '2',
                          '--no-cpu-baseline', '--no-roofline', '--no-steady', '--no-extras', '--fail-rank', '2'],
                         env=_bench_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert time.time() - t0 < 300



# ------------------------------------------------------------- data parallelism above the engine (round 5)
def _fit_case(kind, dev, n=512):
    """a small model + an HBM-resident dataset whose group mix varies from batch to batch"""
    from tests.test_fit import _tiny_dataset, _tiny_model
    model = _tiny_model(kind, device=dev, epochs=2, dim_x=40, dim_h_en_z1=[32], dim_h_de_x=[24], dim_z1=8)
    rs = np.random.RandomState(3)
    from drvae_amd import data as D
    y = rs.randint(0, 2, n)
    x1 = (rs.standard_normal((n, 40)) + 0.8 * (2 * y[:, None] - 1) * (np.arange(40) % 3 == 0)).astype(np.float32)
    hx = (rs.rand(n) < 0.45).astype(np.int64)
    hy = (rs.rand(n) < 0.6).astype(np.int64)
    x2 = ((x1 * 0.7 + 0.2) * hx[:, None]).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    if kind == 'vfae':
        ds = D.VFAEDataset(t(x1), t(np.zeros(n, np.int64)), t(y), t(hy))
    else:
        ds = D.DrVAEDataset(t(x1), t(x2), t(np.zeros(n, np.int64)), t(y), t(hx), t(hy))
    w = D.compute_balanced_weights(np.arange(n) % 7)
    return model, ds, w


def _fit_epochs(model, ds, w, bs, buckets, n_epochs=2):
    """``fit``'s training half: ``_epoch_device`` per epoch (bind with the model's dp state, epoch table, captured
    graphs, replays); returns per-epoch mean objectives and the tables this rank ran"""
    from drvae_amd import data as D
    buckets = dict(buckets)
    mode = buckets.pop('mode', 'sampler')
    bat = D.DeviceBatcher(ds, w, bs, seed=11, mode=mode, **buckets)
    model.add_noise = True
    means, tabs = [], []
    for ep in range(n_epochs):
        means.append(model._epoch_device(bat, ep + 1, False))
        tabs.append(model.engine().plan.feed.table.clone().cpu().numpy())
    return means, tabs, bat


def _fit_worker(rank, world, port, kind, buckets, backend, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK='0', DRVAE_DIST_BACKEND=backend, DRVAE_SIDE_CUS='64')
        if world == 1:
            os.environ['DRVAE_FORCE_DP'] = '1'
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        model, ds, w = _fit_case(kind, dev)
        assert model.enable_data_parallel() == (rank, world)
        assert model._allreduce is not None and model._dp == (rank, world)
        if buckets.get('loader'):
            # tuple loaders: ``fit`` -> ``_epoch_loader`` -> one batch-independent captured step, the three global counts
            # of every batch from a host all-reduce (``run_on_batch`` under data parallelism)
            from tests.test_dp_gloo import _shard_loader
            from tests.test_fit import _tiny_dataset, _tiny_model
            model = _tiny_model(kind, device=dev, epochs=2)
            assert model.enable_data_parallel() == (rank, world)
            model.w2log = lambda *a: None
            tr, va = _tiny_dataset(kind, 64, 1, 'cuda'), _tiny_dataset(kind, 24, 2, 'cuda')
            model.fit(_shard_loader(tr, 16 // world, rank, world), _shard_loader(va, 8, 0, 1), add_noise=True, early_stop=False,
                      model_filename='/tmp/dp_fit_%d.pth' % os.getpid())
            eng = model.engine()
            torch.cuda.synchronize()
            assert eng.universal and eng._graphs and len(eng._graphs) == 2
            q.put((rank, dict(param=eng.arena.param.cpu().numpy(), iters=model.finished_training_iters)))
            dist.barrier()
            dist.destroy_process_group()
            return
        if buckets.get('eval_only'):
            # the whole-set evaluation with its rows sharded over the ranks against the SAME evaluation (same parameters,
            # same Philox counter) of the whole set on this rank alone
            eng = model.engine()
            ctr = eng.rng_ctr.clone()
            out = {}
            for tag, shard in (('sharded', True), ('whole', False), ('again', True)):
                model.shard_evaluation = shard
                eng.rng_ctr.copy_(ctr)
                perf, txt = model.evaluate_performance_on_dataset(ds)
                out[tag] = {k: (float(v) if not isinstance(v, (dict, str)) else v) for k, v in perf.items() if k != 'losses'}
                out[tag].update({'loss_' + k: float(v) for k, v in perf['losses'].items()})
                out[tag]['txt'] = txt
            from drvae_amd.fit import _EvalGraph
            ev = model._eval_graphs[id(ds)]
            out['rows'] = (ev.lo, ev.hi) if ev.dp is not None else None
            q.put((rank, out))
            dist.barrier()
            dist.destroy_process_group()
            return
        means, tabs, bat = _fit_epochs(model, ds, w, 32 // world, buckets)
        eng = model.engine()
        torch.cuda.synchronize()
        eng.check_sync()
        assert len(eng._graphs) == 2                     # split graphs around the exchange
        perf, _ = model.evaluate_performance_on_dataset(ds)
        if getattr(bat, 'global_table', None) is None:
            bat.global_table = torch.zeros(1, 1)
        q.put((rank, dict(means=means, tabs=tabs, gtab=bat.global_table.cpu().numpy(), param=eng.arena.param.cpu().numpy(),
                          iters=model.finished_training_iters, x1_pearr=perf['x1_pearr'],
                          elbo=float(perf['losses']['ELBO']))))
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def _run_fit_ranks(world, kind, buckets, backend='gloo'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29300 + (os.getpid() * 7 + world) % 300
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, kind, buckets, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=420) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=90)
    for r, o in got:
        assert isinstance(o, dict), o
    return [o for _, o in got]


@pytest.mark.parametrize('kind', ['drvae', 'vfae'])
def test_fit_epochs_sampler_feed_two_ranks_equal_one_rank(dev, kind):
    """``model.enable_data_parallel()`` + ``fit``'s device epochs on the exact-sampler feed.  Two ranks
    (one GPU, gloo) with 16 rows each: every rank draws the same global table (device generator, shared seed), runs its
    columns on the batch-independent plan, takes the batch's global (N_pairs, N_labeled) from the table (dv_batch_feed:
    masks.gcounts), replays split graphs around ONE exchange -- and trains like one process on 32-row batches of the same
    table with the Philox draws keyed by global row: losses 1e-4, parameters 1e-4 norm-wise, replicas bit-identical"""
    model, ds, w = _fit_case(kind, dev)
    means, tabs, bat = _fit_epochs(model, ds, w, 32, {})
    torch.cuda.synchronize()
    single = model.engine().arena.param.cpu().numpy()
    perf, _ = model.evaluate_performance_on_dataset(ds)
    r0, r1 = _run_fit_ranks(2, kind, {})
    np.testing.assert_array_equal(r0['gtab'], r1['gtab'])
    for ep in range(2):
        np.testing.assert_array_equal(np.concatenate([r0['tabs'][ep], r1['tabs'][ep]], 1), tabs[ep])
    assert r0['iters'] == r1['iters'] == model.finished_training_iters == 2 * (512 // 32)
    for r in (r0, r1):
        np.testing.assert_allclose(r['means'], means, rtol=1e-4)
        err = np.linalg.norm(r['param'] - single) / np.linalg.norm(single)
        assert err < 1e-4, err
        assert abs(r['x1_pearr'] - perf['x1_pearr']) < 1e-3
    np.testing.assert_array_equal(r0['param'], r1['param'])
    assert r0['x1_pearr'] == r1['x1_pearr'] and r0['elbo'] == r1['elbo']      # same evaluation draws on every rank


@pytest.mark.parametrize('buckets', [dict(pair_bucket=4), dict(pair_bucket=4, label_bucket=4)], ids=['pairs', 'pairs+labels'])
def test_fit_epochs_bucketed_sampler_feed_two_ranks(dev, buckets):
    """the same with bucketed plans (every rank re-orders ITS columns pairs first and switches between its own captured
    plans; the exchange sits between the graphs, so ranks on different plans still meet in it): replicas bit-identical,
    the ranks' columns are a permutation of the shared global table, and (pair buckets) the job trains like one process
    given the ranks' re-ordered columns as its table.  With label buckets a row that is labeled for sure has ONE fprop
    row, whose z3 draw is another Philox draw than the true-class slot's of the every-slot plan: equal in distribution,
    not in value -- there the objective has to agree statistically"""
    kind = 'drvae'
    r0, r1 = _run_fit_ranks(2, kind, buckets)
    np.testing.assert_array_equal(r0['param'], r1['param'])
    np.testing.assert_array_equal(np.sort(np.concatenate([r0['tabs'][1], r1['tabs'][1]], 1), 1), np.sort(r0['gtab'], 1))
    assert np.all(np.isfinite(r0['means'])) and r0['means'] == r1['means']
    from drvae_amd import data as D
    model, ds, w = _fit_case(kind, dev)
    bat = D.DeviceBatcher(ds, w, 32, seed=11, mode='sampler')
    model.add_noise = True
    eng = model.engine()
    means = []
    for ep in range(2):
        # one process, 32-row batches = [rank 0's re-ordered 16 | rank 1's]: same rows at the same global positions
        tab = torch.from_numpy(np.concatenate([r0['tabs'][ep], r1['tabs'][ep]], 1)).to(dev)
        bat.bind(eng)
        eng.add_noise, eng.iters = True, model.finished_training_iters
        bat.begin_epoch(table=tab)
        for _ in range(len(bat)):
            eng.train_step()
        model.finished_training_iters = eng.iters
    torch.cuda.synchronize()
    single = eng.arena.param.cpu().numpy()
    err = np.linalg.norm(r0['param'] - single) / np.linalg.norm(single)
    if 'label_bucket' in buckets:
        assert err < 0.05, err
    else:
        assert err < 1e-4, err


def test_fit_epochs_over_rccl_single_rank(dev):
    """the model-level DP path over the REAL transport: a one-rank RCCL communicator (DRVAE_FORCE_DP=1) in a child
    process -- enable_data_parallel, sampler feed with global counts, split graphs, the collective between them -- trains
    exactly like the plain single-process epochs on the same table"""
    kind = 'drvae'
    model, ds, w = _fit_case(kind, dev)
    means, tabs, bat = _fit_epochs(model, ds, w, 32, {})
    torch.cuda.synchronize()
    single = model.engine().arena.param.cpu().numpy()
    (r0,) = _run_fit_ranks(1, kind, {}, backend='nccl')
    np.testing.assert_array_equal(r0['tabs'][1], tabs[1])
    np.testing.assert_allclose(r0['means'], means, rtol=1e-4)
    err = np.linalg.norm(r0['param'] - single) / np.linalg.norm(single)
    assert err < 1e-4, err


def test_fit_on_tuple_loaders_two_ranks_equal_one_rank(dev):
    """``fit`` fed by per-rank tuple loaders under ``enable_data_parallel`` on the GPU: every batch goes through ONE
    batch-independent captured step (split graphs around the exchange); its normalisers are the global batch's -- N_total
    = the plan's, (N_pairs, N_labeled) handed over per batch (``set_batch(counts=...)`` -> ``dv_batch_masks_desc.gcounts``)
    -- and the job trains like one process on the concatenated batches"""
    from tests.test_dp_gloo import _shard_loader
    from tests.test_fit import _tiny_dataset, _tiny_model
    kind = 'drvae'
    model = _tiny_model(kind, device=dev, epochs=2)
    model.w2log = lambda *a: None
    tr, va = _tiny_dataset(kind, 64, 1, 'cuda'), _tiny_dataset(kind, 24, 2, 'cuda')
    model.fit(_shard_loader(tr, 16, 0, 1), _shard_loader(va, 8, 0, 1), add_noise=True, early_stop=False,
              model_filename='/tmp/dp_fit_single_%d.pth' % os.getpid())
    torch.cuda.synchronize()
    single = model.engine().arena.param.cpu().numpy()
    r0, r1 = _run_fit_ranks(2, kind, dict(loader=True))
    assert r0['iters'] == r1['iters'] == model.finished_training_iters == 2 * 4
    np.testing.assert_array_equal(r0['param'], r1['param'])
    err = np.linalg.norm(r0['param'] - single) / np.linalg.norm(single)
    assert err < 1e-4, err


def test_fit_epochs_stratified_feed_two_ranks(dev):
    """the stratified device feed under data parallelism: every rank's batch has the same composition, the normalisers are
    world x the local counts, the ranks draw ``world x c`` rows per group from the shared generator and take their share --
    replicas bit-identical, and one process given the ranks' batches side by side (explicit 32-row batches, Philox draws
    keyed by global row) trains the same"""
    from tests.test_engine_cpu import make_engine
    kind = 'drvae'
    r0, r1 = _run_fit_ranks(2, kind, dict(mode='stratified'))
    np.testing.assert_array_equal(r0['param'], r1['param'])
    assert np.all(np.isfinite(r0['means'])) and r0['means'] == r1['means']
    assert not np.array_equal(r0['tabs'][1], r1['tabs'][1])                   # different rows on the two ranks
    model, ds, w = _fit_case(kind, dev)
    model.add_noise = True
    eng = model.engine()
    eng.add_noise = True
    t = lambda a: a
    for ep in range(2):
        tab = np.concatenate([r0['tabs'][ep], r1['tabs'][ep]], 1)           # (batches, 32): rank 0's rows | rank 1's
        for b in range(tab.shape[0]):
            idx = torch.from_numpy(tab[b]).long().to(dev)
            eng.iters = model.finished_training_iters
            eng.set_batch(ds.x1[idx], ds.x2[idx], ds.y[idx].cpu().numpy(), ds.has_x2[idx].cpu().numpy(), ds.has_y[idx].cpu().numpy())
            eng.train_step()
            model.finished_training_iters = eng.iters
    torch.cuda.synchronize()
    single = eng.arena.param.cpu().numpy()
    err = np.linalg.norm(r0['param'] - single) / np.linalg.norm(single)
    assert err < 1e-4, err


@pytest.mark.parametrize('kind', ['drvae', 'vfae', 'pvae'])
def test_whole_set_evaluation_is_sharded_by_rows_under_data_parallelism(dev, kind):
    """round 6 (missing 5): under ``enable_data_parallel`` the whole-set evaluation of an HBM-resident dataset runs
    every rank on its n / world rows (loss pass with the whole set's normalisers and Philox draws keyed by the row's position
    in the whole set, inference, per-row statistics, float64 column moments), the partials meet in ONE all-reduce and the
    same finalising launches give every metric of the one-rank evaluation: losses 1e-5 (fp32 row sums in another order),
    reconstruction metrics 1e-6, accuracy / ROC-AUC / AP exactly (integer pair counts); identical on every rank"""
    r0, r1 = _run_fit_ranks(2, kind, {'eval_only': True})
    assert r0['rows'] == (0, 256) and r1['rows'] == (256, 512)
    for r in (r0, r1):
        a, b = r['sharded'], r['whole']
        assert set(a) == set(b)
        for k in a:
            if k in ('model_class', 'txt'):
                assert a[k] == b[k] or k == 'txt'
            elif k.startswith('loss_'):
                assert abs(a[k] - b[k]) <= 2e-5 * max(1.0, abs(b[k])), (k, a[k], b[k])
            elif k.startswith('y_'):
                assert a[k] == b[k], (k, a[k], b[k])
            elif np.isnan(b[k]):
                assert np.isnan(a[k]), k
            else:
                assert abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(b[k])), (k, a[k], b[k])
        assert r['again'] == r['sharded']                    # replays of the two graphs + the exchange: reproducible
    assert r0['sharded'] == r1['sharded']                    # identical on every rank: early stopping decides alike
