"""-m gpu: the fused train step (real HIP kernels through the C-ABI) against the golden
vectors of the reference and against the CPU oracle -- ELBO, KL, recon NLL, y-loss within
BASELINE.json's 1e-4 relative fp32 tolerance; gradients and post-Adam parameters too."""
import numpy as np
import pytest
import torch

from oracle import models_ref as M
from tests.golden import cases as C
from tests.test_engine_cpu import make_engine, set_batch

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4          # BASELINE.json north_star: 1e-4 relative, fp32
# Gradients and post-Adam parameters are held to the SAME 1e-4, norm-wise per parameter tensor (|| a - ref || / || ref ||):
# that is the quantity the north_star's tolerance can be asked of.  Element-wise the bound is looser BY CONSTRUCTION and the
# looser figures below are derived, not chosen: one gradient element is an fp32 sum of K products (K = rows x L, up to 20000
# per element in the decoder heads) accumulated in another order than the reference's, so it differs by up to
# ~sqrt(K) 2^-24 sum|terms| -- relative to an element whose terms CANCEL (|sum| << sum|terms|) that exceeds 1e-4 although
# every term is exact to 6e-8: GRAD_RTOL 5e-4 with an absolute term of 2e-5 max|ref|.  Adam then turns a gradient element
# that is pure cancellation noise into a move of up to lr = 5e-4 in either direction: 2e-4 / 5e-5 element-wise on the
# parameters after N steps (|p| ~ 0.03-1).  The norm-wise checks are the tight ones.
NORM_RTOL = 1e-4
GRAD_RTOL = 5e-4


def rel_norm(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30))


def close(a, b, rtol, atol):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', list(C.MODEL_CASES))
def test_train_steps_match_reference_golden(name, dev):
    case, gold = C.model_case(name), C.load('model_' + name)
    spec = case['spec']
    eng, arena = make_engine(spec, M.init_params(spec, case['param_seed'], as_numpy=True), dev)
    set_batch(eng, case['batch'], dev)
    eng.training = False
    eng.set_noise(case['noises'][0])
    eng.forward()
    for k, v in eng.losses().items():
        close(v, gold['eval/' + k], LOSS_RTOL, 1e-5)
    # gradients of the first step
    eng.training = True
    eng.set_noise(case['noises'][0])
    eng.forward()
    eng.backward()
    for k in arena.shapes:
        g = arena.g(k).cpu().numpy()
        if case['full']:
            ref = gold['grad/' + k]
            close(g, ref, GRAD_RTOL, 2e-5 * max(1.0, float(np.abs(ref).max())))
            if float(np.abs(ref).max()) > 1e-6:
                assert rel_norm(g, ref) <= NORM_RTOL, (k, rel_norm(g, ref))
        else:
            close(np.sqrt((g.astype(np.float64) ** 2).sum()), gold['gradnorm/' + k], NORM_RTOL, 1e-7)
            ref = gold['gradsample/' + k]
            close(g.reshape(-1)[C.sample_index(g.size)], ref, GRAD_RTOL, 1e-4 * max(1e-3, float(np.abs(ref).max())))
    nsteps = len(case['noises'])
    for step, noise in enumerate(case['noises']):
        eng.train_step(noise)
        for k, v in eng.losses().items():
            close(v, gold['step%d/%s' % (step, k)], LOSS_RTOL, 1e-5)
        if step in (0, nsteps - 1):
            for k in arena.shapes:
                a = arena.p(k).cpu().numpy()
                if case['full']:
                    close(a, gold['param%d/%s' % (step, k)], 2e-4, 5e-5)
                    assert rel_norm(a, gold['param%d/%s' % (step, k)]) <= NORM_RTOL, (step, k)
                else:
                    close(a.astype(np.float64).sum(), gold['paramsum%d/%s' % (step, k)], 2e-4, 5e-3)
                    close(a.reshape(-1)[C.sample_index(a.size)], gold['paramsample%d/%s' % (step, k)], 2e-4, 5e-5)


@pytest.mark.parametrize('kind', ['drvae', 'pvae', 'vfae'])
def test_full_size_rows_vs_oracle(kind, dev):
    """BASELINE sizes, per-row terms (150 rows each) against the oracle on identical inputs."""
    spec = M.ModelSpec(kind=kind, L=2)
    params = M.init_params(spec, 7, as_numpy=True)
    batch, noise = M.make_batch(spec, 150, seed=5), M.make_noise(spec, 150, seed=6)
    tr = M.RefTrainer(spec, M.init_params(spec, 7))
    ref, rows = tr.loss(batch, noise, training=True)
    eng, arena = make_engine(spec, params, dev)
    p = set_batch(eng, batch, dev)
    eng.training = True
    eng.set_noise(noise)
    eng.forward()
    got = eng.losses()
    for k, v in got.items():
        close(v, float(ref[k]), LOSS_RTOL, 1e-5)
    L, B = spec.L, 150
    nll = p.NLL.cpu().numpy()
    recl = nll[:L * B].reshape(L, B).sum(0) / L
    if p.Np:
        pr = p.pair_host
        r2 = nll[p.o2:p.o3].reshape(L, p.Np).sum(0) / L
        recl[pr] += r2
        pert = np.zeros(B, np.float32)
        pert[pr] = nll[p.o3:].reshape(L, p.Np).sum(0) / L
        close(pert, rows['PERT'].numpy(), 1e-4, 1e-2)
    close(recl, rows['RECL'].numpy(), 1e-4, 1e-2)
    if spec.kind != 'pvae':
        yl = p.YLrow.cpu().numpy().reshape(L, B).sum(0) / L
        close(yl, rows['YL'].numpy(), 1e-4, 1e-5)


def test_philox_step_runs_and_is_reproducible(dev):
    spec = M.ModelSpec(kind='drvae', L=2)
    params = M.init_params(spec, 3, as_numpy=True)
    batch = M.make_batch(spec, 150, seed=5)
    out = []
    for _ in range(2):
        eng, arena = make_engine(spec, params, dev)
        set_batch(eng, batch, dev)
        for _ in range(3):
            eng.train_step()
        out.append((eng.losses(), arena.param.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])            # no atomics anywhere: bitwise reproducible
    assert all(np.isfinite(v) for v in out[0][0].values())


@pytest.mark.parametrize('kind,most', [('drvae', 34), ('vfae', 32), ('pvae', 25)])
def test_launch_count_of_the_captured_step(dev, kind, most, monkeypatch):
    """The captured train step at the benchmark's batch shape: how many launches it is made of (every C-ABI call of
    the capture = one launch; round 1: 47 for DrVAE).  A regression guard for the fusions of the step: samples / NLL
    in the heads' epilogues, paired dW||dX launches, waits / publishes / counters riding on neighbours, the fprop
    block's KL rows inside the classifier-head launch."""
    from drvae_amd import _lib
    spec = M.ModelSpec(kind=kind, L=1 if kind == 'pvae' else 2)
    params = M.init_params(spec, 3, as_numpy=True)
    eng, arena = make_engine(spec, params, dev)
    set_batch(eng, M.make_batch(spec, 150, seed=5), dev)
    eng.train_step()
    counts = {'n': 0, 'on': False}
    real = _lib.check

    def counting(code, what):
        if counts['on'] and not what.startswith('dv_gemm_set_option'):
            counts['n'] += 1
        return real(code, what)
    monkeypatch.setattr(_lib, 'check', counting)
    real_capture_main = eng._capture_main

    def capture_main(*a, **k):          # (the warm-up pass in front of the capture is not part of the step)
        counts['on'] = True
        return real_capture_main(*a, **k)
    monkeypatch.setattr(eng, '_capture_main', capture_main)
    eng.capture()
    counts['on'] = False
    print('launch calls of the captured step:', kind, counts['n'])
    assert 0 < counts['n'] <= most, counts
    for _ in range(3):
        eng.replay()
    torch.cuda.synchronize()
    eng.check_sync()
    assert all(np.isfinite(v) for v in eng.losses().values())
