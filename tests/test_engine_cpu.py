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


@pytest.mark.parametrize('name', list(C.MODEL_CASES))
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


@pytest.mark.parametrize('name', list(C.MODEL_CASES))
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
