"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/drvae_hip.h declares with the declared arity (no compute
calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, 'include', 'drvae_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'(?:int|const char\*)\s+(dv_\w+)\s*\(([^)]*)\)\s*;', src):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ('', 'void') else len(args.split(','))
    return out


@pytest.fixture(scope='module')
def lib():
    from drvae_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_declares_the_whole_abi():
    decls = _header_decls()
    assert len(decls) >= 27
    from drvae_amd import _lib
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name]) == n, (name, n, len(_lib.SIGNATURES[name]))


def test_library_exports_every_symbol(lib):
    for name in _header_decls():
        assert hasattr(lib, name), name
    from drvae_amd import _lib
    assert lib.dv_abi_version() == _lib.ABI_VERSION == 12
    assert 'dv_arm_park' not in _lib.SIGNATURES and not hasattr(lib, 'dv_arm_park')     # no armed (hidden) state
    assert lib.dv_error_string(0) == b'ok'
    assert lib.dv_error_string(-1) == b'invalid argument'


def test_library_identifies_its_sources(lib):
    """the shipped binary says which sources it was built from: dv_source_hash() == sha256 over csrc/* + the header of THIS
    tree (``build.source_hash``), and the same string is found in the file without loading it (``build.built_hash``)"""
    from drvae_amd import build
    want = build.source_hash()
    assert len(want) == 64
    assert lib.dv_source_hash().decode() == want
    assert build.built_hash() == want
    assert not build.needs_build()


def test_build_is_reproducible(tmp_path):
    """two compilations of one source file of one tree give byte-identical objects (paths mapped away, nothing
    time-dependent baked in; the link adds no build id: ``build.build``) -- checked on the smallest source"""
    import subprocess
    from drvae_amd import build
    objs = []
    for i in range(2):
        (tmp_path / str(i)).mkdir()
        o = str(tmp_path / str(i) / 'optim.o')
        subprocess.check_call([build._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                               '-ffile-prefix-map=%s=.' % build.CSRC, '-cuid=dv-optim', '-DDV_SOURCE_HASH="dv-src-sha256:%s"' % ('0' * 64),
                               '-c', 'optim.hip', '-o', o], cwd=build.CSRC)
        objs.append(open(o, 'rb').read())
    assert objs[0] == objs[1]


def test_gemm_desc_layout_matches_header():
    """ctypes struct field order/names == the C struct in the header."""
    from drvae_amd._lib import GemmDesc
    src = open(os.path.join(ROOT, 'include', 'drvae_hip.h')).read()
    body = re.search(r'typedef struct dv_gemm_desc \{(.*?)\} dv_gemm_desc;', src, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for stmt in body.split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(','):
            names.append(re.sub(r'.*[\s\*]', '', part.strip()))
    assert names == [f[0] for f in GemmDesc._fields_]


@pytest.mark.parametrize('cname,pyname', [('dv_wait', 'Wait'), ('dv_bump', 'Bump'), ('dv_loss_term', 'LossTerm'),
                                          ('dv_publish', 'Publish'), ('dv_heads_epi', 'HeadsEpi'),
                                          ('dv_fprop_kl', 'FpropKl'), ('dv_ymarg', 'Ymarg'), ('dv_seg_add', 'SegAdd'),
                                          ('dv_batch_masks_desc', 'BatchMasks'), ('dv_batch_feed_desc', 'BatchFeed'),
                                          ('dv_kl_rows_desc', 'KlRows'), ('dv_nll_raw_cs_desc', 'NllRawCs'),
                                          ('dv_kl_rows_grad', 'KlRowsGrad'), ('dv_z2f_desc', 'Z2F'),
                                          ('dv_recon_rows_desc', 'ReconRows'), ('dv_adam_hyper', 'AdamHyper'),
                                          ('dv_prior_kl', 'PriorKl')])
def test_small_struct_layouts_match_header(cname, pyname):
    from drvae_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'drvae_hip.h')).read()
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), src, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = [re.sub(r'\[\d+\]', '', re.sub(r'.*[\s\*]', '', part.strip())) for st in body.split(';') if st.strip()
             for part in st.split(',')]
    py = getattr(_lib, pyname)
    assert names == [f[0] for f in py._fields_]
    # ... and the field TYPES: pointer / int32 / int64 / float, declaration by declaration
    import ctypes as C
    kinds = []
    for st in body.split(';'):
        st = st.strip()
        if not st:
            continue
        for j, part in enumerate(st.split(',')):
            decl = part.strip() if j == 0 else re.match(r'(?:const\s+)?\w+', st).group(0) + ' ' + part.strip()
            arr = re.search(r'\[(\d+)\]', decl)
            if '*' in decl:
                k = C.c_void_p
            elif decl.startswith('int64_t'):
                k = C.c_int64
            elif decl.startswith('int32_t'):
                k = C.c_int32
            elif decl.startswith('float'):
                k = C.c_float
            else:
                raise AssertionError(decl)
            kinds.append(k * int(arr.group(1)) if arr else k)
    assert [C.sizeof(k) for k in kinds] == [C.sizeof(f[1]) for f in py._fields_], cname
    for k, f in zip(kinds, py._fields_):
        base_k = getattr(k, '_type_', k) if hasattr(k, '_length_') else k
        base_f = getattr(f[1], '_type_', f[1]) if hasattr(f[1], '_length_') else f[1]
        assert (base_k is C.c_float) == (base_f is C.c_float), (cname, f[0])


def test_argument_validation_without_gpu(lib):
    """Entry points reject bad arguments before touching the device (error code, no abort)."""
    assert lib.dv_gemm(None, None) == -1
    assert lib.dv_counter_add(None, 3, 1, None) == -1
    assert lib.dv_softmax_clamp_fwd(None, 0, 5, 0, 0, None, 0, None) == -1   # Y < 1
    # the descriptor entry points of ABI 11: a missing descriptor, and descriptors whose required fields are missing
    import ctypes as C
    from drvae_amd import _lib
    assert lib.dv_batch_feed(None, None, None, None) == -1
    assert lib.dv_kl_rows_fwd(None, None, None) == -1
    assert lib.dv_batch_masks(None, None, 0, None, None, 4, 2, None) == -1
    assert lib.dv_gauss_nll_rows_raw_cs(None, None) == -1
    d = _lib.BatchFeed(B=4, L=1, n_batches=1, X=8)                # B > 0 but no x1 / table / counters
    assert lib.dv_batch_feed(C.byref(d), None, None, None) == -1
    k = _lib.KlRows(n=3, reps=1, Z=5)                              # rows to do, no operands
    assert lib.dv_kl_rows_fwd(C.byref(k), None, None) == -1
    assert lib.dv_kl_rows_bwd(C.byref(k), None, None) == -1       # no gradient descriptor
    assert lib.dv_kl_rows_bwd(C.byref(k), C.byref(_lib.KlRowsGrad()), None) == -1
    assert lib.dv_z2f_post_bwd(None, None, None) == -1
    assert lib.dv_recon_rows(None, None) == -1
    assert lib.dv_kl_rows_fwd_pair(None, None, None) == -1
    assert lib.dv_kl_rows_fwd_pair(C.byref(k), C.byref(k), None) == -1          # rows to do, no operands
    assert lib.dv_adam_l2(None, None, None, None, 8, None, None, None, 0, None) == -1            # no hyper-parameters
    assert lib.dv_adam_l2_gated(None, None, None, None, 8, C.byref(_lib.AdamHyper(lr=1e-3)), None, None, 0, 8, None, 0, None) == -1
    assert lib.dv_adamax_l2(None, None, None, None, 8, None, None, None, 0, None) == -1
    assert lib.dv_recon_rows(C.byref(_lib.ReconRows(M=4, X=8)), None) == -1          # no operands
    assert lib.dv_recon_rows(C.byref(_lib.ReconRows(M=4, X=4096)), None) == -3       # beyond the resident width
    m = _lib.BatchMasks(Np=9, n_tot=4.0)                           # more pair slots than rows
    assert lib.dv_batch_masks(C.byref(m), None, 0, None, None, 4, 2, None) == -1
    assert lib.dv_loss_assemble_after(None, None, 0, None, None, None, None, None, 0, None, None) == -1
    assert lib.dv_rank_metrics(None, 0, None, None, None, 5, 0, 1, 1, None, None, None) == -1
    assert lib.dv_recon_finalize(None, None, 5, 8, None, 1, None, None, None) == -1
    assert lib.dv_mmd_mix_fwd(None, 0, 3, 3, 0, None, 5, None, 0, None, 0, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from drvae_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU/PyTorch fallback'):
        _lib.load()


def test_kernels_refuse_cpu_tensors():
    import torch
    import drvae_amd.kernels as K
    with pytest.raises(RuntimeError, match='no CPU'):
        K.colsum(torch.zeros(3), torch.zeros(2, 3))


def test_mmd_functions_have_no_aten_fallback():
    """``blocks.identity`` / ``mmd_objective`` (every kernel) refuse host tensors like the rest of the package"""
    import torch
    from drvae_amd import blocks as blk
    a, b = torch.randn(5, 4), torch.randn(6, 4)
    for kernel in ('identity', 'poly', 'rbf', 'rbf_fourier'):
        with pytest.raises(RuntimeError, match='no CPU'):
            blk.mmd_objective(a, b, kernel)
    with pytest.raises(RuntimeError, match='no CPU'):
        blk.identity(a, b)
    with pytest.raises(NotImplementedError):
        blk.mmd_objective(a, b, 'poly', bandwidths=[0.1] * 9)


def _kernel_metadata(tmp_path):
    """(name, vgpr_count, vgpr_spill_count, sgpr_spill_count) of every gfx950 kernel in the built library"""
    import shutil
    import subprocess
    from drvae_amd import _lib
    llvm = '/opt/rocm/lib/llvm/bin'
    if not (os.path.exists(os.path.join(llvm, 'llvm-objdump')) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip('needs the ROCm llvm tools and the built library')
    so = shutil.copy(_lib.LIB_PATH, str(tmp_path))
    subprocess.run([os.path.join(llvm, 'llvm-objdump'), '--offloading', so], cwd=str(tmp_path), capture_output=True, check=True)
    out = []
    for f in sorted(os.listdir(str(tmp_path))):
        if 'gfx950' not in f:
            continue
        notes = subprocess.run([os.path.join(llvm, 'llvm-readelf'), '--notes', os.path.join(str(tmp_path), f)],
                               capture_output=True, text=True, check=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r'\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\S+)', line)
            if m:
                cur[m.group(1)] = m.group(2)
                if len(cur) == 4:
                    out.append((cur['name'], int(cur['vgpr_count']), int(cur['vgpr_spill_count']), int(cur['sgpr_spill_count'])))
                    cur = {}
    return out


def test_no_kernel_of_the_product_library_spills_vector_registers(tmp_path):
    """every kernel the dispatcher can select keeps its vector registers out of scratch memory (gfx950 code-object
    metadata of the built library), and the library carries no lab kernels (they live behind -DDV_LAB)"""
    meta = _kernel_metadata(tmp_path)
    assert len(meta) > 40, 'no gfx950 kernels found in the library'
    spilled = [(n, s) for (n, v, s, g) in meta if s != 0]
    assert not spilled, spilled
    assert not [n for (n, v, s, g) in meta if 'gemm_dma_kernel' in n or 'gemm_wp_kernel' in n]
    assert max(v for (n, v, s, g) in meta) <= 256
