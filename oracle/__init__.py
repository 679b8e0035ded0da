"""CPU oracle for the Dr.VAE ELBO hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a from-the-maths restatement, in plain fp32 PyTorch **CPU** ops, of
the arithmetic that rampasek/DrVAE runs on its ELBO training path (reference files
``src/layers.py``, ``src/blocks.py``, ``src/DrVAE.py``, ``src/PVAE.py``,
``src/VFAE.py``, ``src/DGMMixin.py``; every function cites the ``file:line`` it
follows).  The reference's arithmetic *is* PyTorch CPU ATen (a third-party
dependency, README pins ``pytorch=0.3.1``; not vendored), so the restatement uses
the same ATen primitives executed by the torch in this image (2.10).

Pinning: the reference ships **no tests, golden vectors or fixtures** for this path
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the
reference itself, executed in the build container by ``tests/golden/make_golden.py``
(which imports ``/root/reference/src`` with three non-invasive shims and replays a
recorded noise stream); those outputs are committed as ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks the oracle against every one of them.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``drvae_amd``) never does: it fails loudly when
its HIP library is missing.
"""
from . import blocks_ref, models_ref  # noqa: F401
