#!/usr/bin/env python3
"""Build-container-only check of the module-level drop-in seam: the
UNMODIFIED reference model classes (src/DrVAE.py, src/PVAE.py, src/VFAE.py) run on top of
`drvae_amd.blocks` / `drvae_amd.layers` installed as sys.modules['blocks'/'layers'] and must
reproduce the golden vectors the all-reference run produced.  There is no GPU here, so the HIP
launchers are replaced by their PyTorch references (tests/kernel_ref.py): this exercises the
API surface (names, ctor signatures, list/tuple conventions, state_dict keys, autograd
wiring), not the kernels (those are checked on the GPU).  Run by tests/test_dropin_seam.py."""
import contextlib
import io
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')

import drvae_amd.blocks as our_blocks      # noqa: E402
import drvae_amd.kernels as K              # noqa: E402
import drvae_amd.layers as our_layers      # noqa: E402
from tests import kernel_ref               # noqa: E402

for _n in kernel_ref.FUNCTIONS:
    setattr(K, _n, getattr(kernel_ref, _n))
sys.modules['blocks'] = our_blocks
sys.modules['layers'] = our_layers
sys.modules['h5py'] = types.ModuleType('h5py')
sys.path.insert(0, '/root/reference/src')

from tests.golden import make_golden as G   # noqa: E402  (imports the reference model classes)
from tests.golden import cases as C         # noqa: E402
from oracle import models_ref as M          # noqa: E402

assert G.rDrVAE.blk is our_blocks and our_blocks.lyr is our_layers, 'seam not installed'


def main():
    n_checked = 0
    for name in ['tiny_drvae', 'tiny_drvae_nolp', 'tiny_drvae_wn', 'tiny_drvae_only_up', 'tiny_pvae', 'tiny_vfae',
                 'tiny_vfae_sup']:
        case, gold = C.model_case(name), C.load('model_' + name)
        got = G.run_model_case(case)       # reference classes, our blocks underneath
        for k, v in gold.items():
            if k.startswith('grad/') or k.startswith('param'):
                tol = dict(rtol=5e-4, atol=5e-6 * max(1.0, float(np.abs(v).max())))
            else:
                tol = dict(rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(got[k], v, err_msg='%s %s' % (name, k), **tol)
            n_checked += 1
    print('DROPIN_OK', n_checked)


if __name__ == '__main__':
    main()
