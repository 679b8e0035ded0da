"""-m gpu: size-independent properties of the fused step at sizes the CPU oracle cannot reach
in seconds (BASELINE.json configs[4]-class shapes): shard additivity (the data-parallel
contract on one device), bitwise reproducibility, and a finite-difference check of the
hand-written backward."""
import numpy as np
import pytest
import torch

from oracle import models_ref as M
from tests.test_engine_cpu import make_engine, set_batch

pytestmark = pytest.mark.gpu


def _params(spec, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    fan = 1
    for k, shp in M.param_shapes(spec).items():
        if k.endswith('.weight'):
            fan = shp[1]
        scale = 1e-4 if (k.endswith('W_mu') or k.endswith('bias_mu')) else fan ** -0.5
        out[k] = ((torch.rand(*shp, generator=g) * 2 - 1) * scale).numpy()
    return out


def _noise(spec, n, seed):
    g = torch.Generator().manual_seed(seed)
    f = lambda *s: torch.randn(*s, generator=g).numpy()
    return {'nx1': f(n, spec.dim_x), 'nx2': f(n, spec.dim_x), 'ez1': f(spec.L, n, spec.dim_z1),
            'ez2': f(spec.L, n, spec.dim_z1), 'ez2F': f(spec.L, n, spec.dim_z1),
            'ez3': f(spec.L, spec.dim_y, n, spec.dim_z3)}


def _fwd_bwd(spec, params, batch, noise, dev, counts=None, lo=None, hi=None):
    if lo is not None:
        batch = {k: v[lo:hi] for k, v in batch.items()}
        noise = M.slice_noise(noise, lo, hi)
    eng, arena = make_engine(spec, params, dev)
    set_batch(eng, batch, dev, counts=counts)
    eng.set_noise(noise)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    return arena.grad.clone(), eng


def test_wide_config_shard_additivity_and_reproducibility(dev):
    """20000 genes, z=200, enc/dec 2048, L=4 (BASELINE configs[4] shapes; 256 rows here):
    grad(shard 0) + grad(shard 1) with GLOBAL normalisers == grad(full batch), loss tail included."""
    spec = M.ModelSpec(kind='drvae', dim_x=20000, dim_z1=200, dim_z3=200, h_en_z1=[2048], h_de_x=[2048], L=4)
    n = 256
    params = _params(spec, 1)
    batch = M.make_batch(spec, n, seed=3)
    noise = _noise(spec, n, 4)
    full, _ = _fwd_bwd(spec, params, batch, noise, dev)
    full2, _ = _fwd_bwd(spec, params, batch, noise, dev)
    assert torch.equal(full, full2)                                   # no atomics: bitwise reproducible
    counts = (n, int(batch['has_x2'].sum()), int(batch['has_y'].sum()))
    g0, _ = _fwd_bwd(spec, params, batch, noise, dev, counts, 0, 96)          # uneven shards
    g1, _ = _fwd_bwd(spec, params, batch, noise, dev, counts, 96, n)
    s = g0 + g1
    rel = float((s - full).norm() / full.norm())
    assert rel < 2e-5, rel
    np.testing.assert_allclose(s[-8:].cpu().numpy(), full[-8:].cpu().numpy(), rtol=1e-4, atol=1e-4)   # loss scalars
    assert bool(torch.isfinite(full).all())


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_backward_matches_finite_differences_at_baseline_size(kind, dev):
    """<grad CMPL, d> == (CMPL(theta + e d) - CMPL(theta - e d)) / 2e for a random direction d
    (fixed noise): checks the whole hand-written backward at 978 genes / batch 150."""
    spec = M.ModelSpec(kind=kind, L=2)
    params = M.init_params(spec, 11, as_numpy=True)
    batch, noise = M.make_batch(spec, 150, seed=5), M.make_noise(spec, 150, seed=6)
    grad, eng = _fwd_bwd(spec, params, batch, noise, dev)
    arena = eng.arena
    n = arena.n_params
    g = torch.Generator().manual_seed(0)
    d = torch.randn(n, generator=g).to(dev)
    # relative perturbation so that every layer moves comparably
    d = d * arena.param.abs().clamp(min=1e-3)
    theta = arena.param.clone()
    lin = float((grad[:n].double() * d.double()).sum())

    def cmpl(t):
        arena.param.copy_(t)
        eng.set_noise(noise)
        eng.forward()
        return float(arena.loss[6])

    # CMPL ~ 2e3 is only resolved to ~2e-4 in fp32, so the step must move it by >> that
    eps = 1e-2
    fd = (cmpl(theta + eps * d) - cmpl(theta - eps * d)) / (2 * eps)
    arena.param.copy_(theta)
    assert abs(fd - lin) <= 3e-2 * max(1.0, abs(lin)), (fd, lin)


def test_cfg5_full_size_shard_additivity_reproducibility_and_training(dev):
    """BASELINE configs[4] at its stated per-GPU size: 20000 genes, z1=z3=200, enc/dec 2048, 1024 rows, L=4.
    (a) grad(rows 0..511) + grad(rows 512..1023) with GLOBAL normalisers == grad(all 1024 rows), loss scalars
    included (the data-parallel contract); (b) bitwise reproducibility; (c) a directional finite difference of
    the hand-written backward; (d) three captured train steps with on-device Philox noise stay finite and move
    the parameters."""
    spec = M.ModelSpec(kind='drvae', dim_x=20000, dim_z1=200, dim_z3=200, h_en_z1=[2048], h_de_x=[2048], L=4)
    n = 1024
    params = _params(spec, 1)
    batch = M.make_batch(spec, n, seed=3)
    noise = _noise(spec, n, 4)
    full, eng = _fwd_bwd(spec, params, batch, noise, dev)
    assert bool(torch.isfinite(full).all())
    # (c) finite difference along a random relative direction
    arena = eng.arena
    npar = arena.n_params
    g = torch.Generator().manual_seed(0)
    d = torch.randn(npar, generator=g).to(dev) * arena.param.abs().clamp(min=1e-3)
    theta = arena.param.clone()
    lin = float((full[:npar].double() * d.double()).sum())

    def cmpl(t):
        arena.param.copy_(t)
        eng.set_noise(noise)
        eng.forward()
        return float(arena.loss[6])
    eps = 1e-2
    fd = (cmpl(theta + eps * d) - cmpl(theta - eps * d)) / (2 * eps)
    arena.param.copy_(theta)
    assert abs(fd - lin) <= 5e-2 * max(1.0, abs(lin)), (fd, lin)
    del eng, arena
    torch.cuda.empty_cache()
    full2, _ = _fwd_bwd(spec, params, batch, noise, dev)
    assert torch.equal(full, full2)                                   # (b)
    del full2, _
    counts = (n, int(batch['has_x2'].sum()), int(batch['has_y'].sum()))
    g0, e0 = _fwd_bwd(spec, params, batch, noise, dev, counts, 0, 512)
    l0 = e0.arena.loss.clone()
    del e0
    torch.cuda.empty_cache()
    g1, e1 = _fwd_bwd(spec, params, batch, noise, dev, counts, 512, n)
    l1 = e1.arena.loss.clone()
    s = g0 + g1
    rel = float((s - full).norm() / full.norm())
    assert rel < 2e-5, rel                                            # (a)
    # (d) captured steps on Philox noise
    e1.train_step()
    e1.capture()
    p0 = e1.arena.param.clone()
    for _ in range(3):
        e1.replay()
    torch.cuda.synchronize()
    losses = e1.losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    assert float((e1.arena.param - p0).abs().max()) > 0 and bool(torch.isfinite(e1.arena.param).all())


def test_timed_out_chain_wait_fails_loudly(dev, monkeypatch):
    """a device-side wait that times out must never train on stale data silently.  The waits
    are captured with a one-poll bound (the side chain's first wait then gives up before the main chain has
    published): from that step on the loss scalars are NaN, the optimiser leaves the parameters untouched,
    and a plain ``replay()`` loop that never reads the losses raises within two polling periods."""
    import drvae_amd.kernels as K
    import drvae_amd.schedule as S
    spec = M.ModelSpec(kind='drvae', L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    batch = M.make_batch(spec, 150, seed=5)
    eng, arena = make_engine(spec, params, dev)
    set_batch(eng, batch, dev)
    eng.train_step()
    monkeypatch.setattr(K, 'WAIT_SPINS', 1)
    monkeypatch.setattr(S, 'SYNC_POLL', 8)
    eng.capture()
    assert eng._side_graph is not None, 'dual-graph schedule expected on the GPU'
    torch.cuda.synchronize()
    before = arena.param.clone()
    with pytest.raises(RuntimeError, match='chain wait timed out'):
        for _ in range(2 * 8 + 1):
            eng.replay()
    torch.cuda.synchronize()
    assert int(eng.sync_err[0::2].abs().sum()) != 0
    frozen = arena.param.clone()
    eng._sync_event = None
    try:
        for _ in range(4):
            eng.replay()
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    assert torch.equal(arena.param, frozen)                # halted: the sweep no longer touches the parameters
    assert all(np.isnan(v) for v in arena.loss[:7].cpu().tolist())
    with pytest.raises(RuntimeError, match='chain wait timed out'):
        eng.losses()
    del before


def test_losses_and_eager_steps_after_replay_need_no_host_sync(dev, monkeypatch):
    """round-3 advisor: with tail gating nothing on the main stream is ordered behind the side chain's tail (its half of
    the optimiser sweep, the loss scalars) -- ``losses()``, an eager ``train_step`` and an evaluation forward join the
    side stream themselves now.  Two engines run the same captured steps; one reads its losses / parameters right after
    ``replay()`` (no ``torch.cuda.synchronize()``), the other after a full device sync: identical."""
    import drvae_amd.tuning as T
    vals = T._parse()
    vals['tail_gate'] = 2            # the gated tail also with a resident batch (default: graph-resident feeds only)
    monkeypatch.setattr(T, '_VALUES', vals)
    spec = M.ModelSpec(kind='drvae', L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    batch = M.make_batch(spec, 150, seed=5)
    outs = []
    for synced in (False, True):
        eng, arena = make_engine(spec, params, dev)
        eng.seed = 11
        set_batch(eng, batch, dev)
        eng.train_step()
        eng.capture()
        assert eng._side_graph is not None, 'dual-graph schedule expected on the GPU'
        with eng.partition(64):
            got = []
            for _ in range(6):
                eng.replay()
                if synced:
                    torch.cuda.synchronize()
                got.append(tuple(eng.losses().values()))
            eng.replay()
            if synced:
                torch.cuda.synchronize()
            eng.train_step()                       # eager step straight behind a replay
            if synced:
                torch.cuda.synchronize()
            got.append(tuple(eng.losses().values()))
        torch.cuda.synchronize()
        outs.append((got, arena.param.clone()))
        del eng, arena
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


def test_philox_draws_are_keyed_by_global_row(dev):
    """SURVEY 8(e) "RNG under DP": a rank that owns rows [rB, (r+1)B) of the global minibatch draws exactly the
    values a single process draws for those rows -- for every draw of the step (input noise, z1 / z2 / z2Fz1 / z3
    eps), whatever the stacking.  Then: two shards' Philox train-step gradients sum to the full batch's."""
    spec = M.ModelSpec(kind='drvae', L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    n, B = 300, 150
    full_batch = M.make_batch(spec, n, seed=5)
    counts = (n, int(full_batch['has_x2'].sum()), int(full_batch['has_y'].sum()))
    engs = []
    for (lo, hi) in ((0, n), (0, B), (B, n)):
        eng, arena = make_engine(spec, params, dev)
        eng.seed, eng.row0 = 77, lo
        set_batch(eng, {k: v[lo:hi] for k, v in full_batch.items()}, dev, counts=counts)
        eng.draw_noise()
        engs.append((eng, arena, lo, hi))
    torch.cuda.synchronize()
    (ef, af, _, _) = engs[0]
    pf = ef.plan
    L, Y = spec.L, spec.dim_y
    for (e, a, lo, hi) in engs[1:]:
        p = e.plan
        m = hi - lo
        assert torch.equal(p.EX[:m], pf.EX[lo:hi])
        pr_f = {int(r): k for k, r in enumerate(pf.pair_host)}
        for k, r in enumerate(p.pair_host):
            kf = pr_f[int(r) + lo]
            assert torch.equal(p.EX[m + k], pf.EX[n + kf])
            for l in range(L):
                assert torch.equal(p.E2[l * p.Np + k], pf.E2[l * pf.Np + kf])
        for l in range(L):
            assert torch.equal(p.E1[l * m:(l + 1) * m], pf.E1[l * n + lo:l * n + hi])
            assert torch.equal(p.E2F[l * m:(l + 1) * m], pf.E2F[l * n + lo:l * n + hi])
        key_f = {(int(l_), int(i_), int(s_)): f for f, (l_, i_, s_) in
                 enumerate(zip(pf.fp_l_host, pf.fp_i_host, pf.fp_slot_host))}
        for f, (l_, i_, s_) in enumerate(zip(p.fp_l_host, p.fp_i_host, p.fp_slot_host)):
            assert torch.equal(p.E3[f], pf.E3[key_f[(int(l_), int(i_) + lo, int(s_))]])
    # a second draw event differs, and moments are those of N(0,1)
    first = pf.noise.clone()
    ef.draw_noise()
    torch.cuda.synchronize()
    assert float((pf.noise - first).abs().max()) > 0
    assert abs(float(first.mean())) < 1e-2 and abs(float(first.std()) - 1) < 1e-2
    assert abs(float((first ** 4).mean()) - 3) < 1e-1 and bool(torch.isfinite(first).all())
    # Philox train step: shard gradients add up to the full batch's (same draws per global row)
    grads = []
    for (e, a, lo, hi) in engs:
        e.rng_ctr.zero_()
        e.training, e.fuse_bwd = True, False
        e.draw_noise()
        e.forward()
        e.backward()
        torch.cuda.synchronize()
        grads.append(a.xchg.clone())
    s = grads[1] + grads[2]
    rel = float((s - grads[0]).norm() / grads[0].norm())
    assert rel < 2e-5, rel


def test_random_model_shapes_captured_step_vs_oracle(dev):
    """Random model shapes (odd gene counts, latent sizes that are no multiple of a tile or of 4, 2-4 classes, 1-3
    Monte-Carlo samples, random group mixes) -- three train steps with injected noise: the eager fused step against
    the CPU oracle (losses 1e-4 relative, parameters after the optimiser), and the captured dual-graph step (Philox
    noise) bitwise against the eager one.  Covers every fusion of the step at sizes the golden cases do not."""
    from hypothesis import given, settings, strategies as st, HealthCheck, Phase
    from tests.test_engine_cpu import make_engine, set_batch

    @settings(max_examples=6, deadline=None, derandomize=True, database=None, phases=[Phase.generate],
              suppress_health_check=list(HealthCheck))
    @given(kind=st.sampled_from(['drvae', 'vfae', 'pvae']), X=st.integers(37, 300), Z1=st.integers(5, 70),
           Z3=st.integers(5, 70), Y=st.integers(2, 4), L=st.integers(1, 3), B=st.integers(9, 70),
           seed=st.integers(0, 1000))
    def run(kind, X, Z1, Z3, Y, L, B, seed):
        spec = M.ModelSpec(kind=kind, dim_x=X, dim_y=Y, dim_z1=Z1, dim_z3=Z3, h_en_z1=[2 * Z1 + 3], h_de_z1=[Z1 + 7],
                           h_en_z3=[Z3 + 5], h_de_x=[3 * Z1 + 1], L=L)
        rs = np.random.RandomState(seed)
        batch = M.make_batch(spec, B, seed=seed)
        if kind != 'vfae':
            batch['has_x2'] = (rs.rand(B) < 0.5).astype(np.int64)
            batch['x2'] = batch['x2'] * batch['has_x2'][:, None].astype(np.float32)
        if kind != 'pvae':
            batch['has_y'] = (rs.rand(B) < 0.6).astype(np.int64)
        params = M.init_params(spec, seed + 1, as_numpy=True)
        tr = M.RefTrainer(spec, M.init_params(spec, seed + 1))
        eng, arena = make_engine(spec, params, dev)
        set_batch(eng, batch, dev)
        for it in range(3):
            noise = M.make_noise(spec, B, seed=seed + 10 + it)
            ref, _ = tr.step(batch, noise)
            eng.train_step(noise)
            got = eng.losses()
            for k, v in got.items():
                np.testing.assert_allclose(v, float(ref[k].detach()), rtol=2e-4, atol=2e-5, err_msg='%s %s it%d' % (kind, k, it))
        # (norm-wise per tensor: Adam turns a gradient element that is pure summation-order noise into a full +-lr move)
        for k in arena.shapes:
            a, b = arena.p(k).cpu().numpy().ravel(), tr.params[k].detach().numpy().ravel()
            err = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
            assert err < 2e-3, '%s param %s: %g' % (kind, k, err)
        # captured (dual-graph where the model has a side chain) == eager, bitwise, on on-device noise
        e0, a0 = make_engine(spec, params, dev)
        e1, a1 = make_engine(spec, params, dev)
        for e in (e0, e1):
            set_batch(e, batch, dev)
            e.train_step()
        e1.capture()
        for _ in range(3):
            e0.train_step()
            e1.replay()
        torch.cuda.synchronize()
        e1.check_sync()
        assert torch.equal(a0.param, a1.param), kind
        assert e0.losses() == e1.losses()
        # the pad columns of the row-padded weights (inner dimension no multiple of 4) are zero and stay zero: the
        # forward products run over the padded K
        for a in (arena, a0, a1):
            for buf in (a.param, a.grad, a.exp_avg, a.exp_avg_sq):
                for pad in a.pads(buf):
                    assert not pad.any(), kind
        # ... and so are the pad columns of every row-padded activation buffer of the plan and of its layer chains
        # (nothing may write past a row's live columns: the next product reads them as part of K)
        from drvae_amd.chain import _Chain, _whole_rows
        for e in (eng, e0, e1):
            bufs = [v for v in vars(e.plan).values() if torch.is_tensor(v)]
            for c in (v for v in vars(e.plan).values() if isinstance(v, _Chain)):
                bufs += list(c.out) + list(c.dpre)
            n_pad = 0
            for t in bufs:
                if t.dim() == 2 and t.dtype == torch.float32 and _whole_rows(t) and t._base.shape[1] != t.shape[1]:
                    n_pad += 1
                    assert not t._base[:, t.shape[1]:].any(), (kind, tuple(t.shape))
            assert n_pad > 0
    run()
