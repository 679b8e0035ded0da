"""N1: the models' means-only inference passes (``forward``, ``forward_w_pert_identity``) against outputs of
the reference's own model classes (tests/golden/inference.npz).  Runs on CPU with the HIP launchers replaced
by their PyTorch stand-ins (host logic + wiring) and, marked gpu, on the real kernels."""
import numpy as np
import pytest
import torch

from oracle import models_ref as M
from tests import kernel_ref
from tests.golden import cases as C
from tests.test_gpu_models import build_model

CASES = ('tiny_drvae', 'tiny_drvae_nolp', 'tiny_drvae_wn', 'tiny_drvae_cont', 'tiny_drvae_1sig', 'tiny_pvae', 'tiny_vfae',
         'cfg2_drvae', 'cfg4_vfae')


def _flat(res):
    out = {}
    for k, v in res.items():
        if isinstance(v, (tuple, list)):
            for i, t in enumerate(v):
                out['%s.%d' % (k, i)] = t.detach().cpu().numpy()
        else:
            out[k] = v.detach().cpu().numpy()
    return out


def _check(name, dev):
    G = C.load('inference')
    case = C.model_case(name)
    spec = case['spec']
    model = build_model(spec, dev)
    params = M.init_params(spec, case['param_seed'], as_numpy=True)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    b = case['batch']
    t = lambda k: torch.from_numpy(b[k].copy()).to(dev)
    got = {'fwd': _flat(model.forward(t('x1'), t('s')))}
    if spec.kind != 'vfae':
        got['ident'] = _flat(model.forward_w_pert_identity(t('x1'), t('x2'), t('s')))
    else:
        with pytest.raises(AttributeError):
            model.forward_w_pert_identity(t('x1'), t('x1'), t('s'))
    n = 0
    for tag, r in got.items():
        want_keys = {k.split('/')[2].split('@')[0] for k in G if k.startswith('%s/%s/' % (name, tag))}
        assert want_keys == set(r), (want_keys ^ set(r))        # same result dict as the reference
        for k, v in r.items():
            key = '%s/%s/%s' % (name, tag, k)
            if key in G:
                if k == 'pred' and spec.type_y == 'discrete':
                    assert (v.reshape(-1) == G[key].reshape(-1)).mean() >= 0.99       # argmax ties aside
                else:
                    np.testing.assert_allclose(v, G[key], rtol=2e-4, atol=2e-5, err_msg=key)
            else:
                np.testing.assert_allclose(v.astype(np.float64).sum(), G[key + '@sum'], rtol=1e-4, atol=2e-2)
                np.testing.assert_allclose(np.abs(v.astype(np.float64)).sum(), G[key + '@abs'], rtol=1e-4)
                np.testing.assert_allclose(v.reshape(-1)[C.sample_index(v.size)], G[key + '@smp'], rtol=2e-4,
                                           atol=2e-5, err_msg=key)
            n += 1
    assert n >= 8


@pytest.mark.parametrize('name', CASES)
def test_inference_matches_reference_cpu(name, monkeypatch):
    kernel_ref.install(monkeypatch)
    _check(name, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_inference_matches_reference_gpu(name, dev):
    _check(name, dev)
