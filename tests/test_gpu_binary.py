"""The library the GPU box loads is the one this tree's sources build (the box never compiles: the .so travels with the
snapshot, git-ignored).  ``dv_source_hash()`` is baked in at build time (drvae_amd/build.py); here it is recomputed from the
files next to the binary."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_shipped_binary_was_built_from_this_tree():
    from drvae_amd import _lib, build
    lib = _lib.load()
    assert torch.cuda.is_available()
    got = lib.dv_source_hash().decode()
    assert got == build.source_hash(), 'libdrvae_hip.so is stale: rebuild with `python -m drvae_amd.build`'
    assert build.built_hash(_lib.LIB_PATH) == got
    assert lib.dv_abi_version() == _lib.ABI_VERSION


def test_mmd_functions_refuse_host_tensors():
    """no ATen fallback behind ``blocks.identity`` / ``mmd_objective`` (there is no CPU path)"""
    from drvae_amd import blocks as blk
    a, b = torch.randn(5, 4), torch.randn(6, 4)
    for kernel in ('identity', 'poly', 'rbf', 'rbf_fourier'):
        with pytest.raises(RuntimeError, match='no CPU'):
            blk.mmd_objective(a, b, kernel)
    dev = torch.device('cuda:0')
    with pytest.raises(NotImplementedError):
        blk.mmd_objective(a.to(dev), b.to(dev), 'poly', bandwidths=[0.1] * 9)
