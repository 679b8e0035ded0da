"""In-tree build of libdrvae_hip.so (hipcc, gfx950 only).  ``python -m drvae_amd.build``."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdrvae_hip.so')
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'drvae_hip.h')
SOURCES = ['gemm.hip', 'rows.hip', 'optim.hip']
SOURCE_EXT = ('.hip', '.inc', '.h')


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def source_files():
    """every file the library is compiled from: csrc/*.{hip,inc,h} in name order, then the C-ABI header"""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(SOURCE_EXT)] + [HEADER]


def source_hash():
    """hex sha256 over (file name, length, bytes) of ``source_files()`` -- what ``dv_source_hash()`` of a library built
    from this tree returns (``tests/test_cabi.py`` on the CPU, ``tests/test_gpu_kernels.py`` on the GPU box: the shipped
    binary is the one these sources build)"""
    h = hashlib.sha256()
    for path in source_files():
        with open(path, 'rb') as fh:
            data = fh.read()
        h.update(('%s:%d\n' % (os.path.basename(path), len(data))).encode())
        h.update(data)
    return h.hexdigest()


def built_hash(path=LIB):
    """``dv_source_hash()`` of an existing library without loading it into this process's HIP runtime state (the string
    sits in the host part of the shared object)"""
    import re
    with open(path, 'rb') as fh:
        m = re.search(rb'dv-src-sha256:([0-9a-f]{64})', fh.read())
    return m.group(1).decode() if m else None


def needs_build():
    if not os.path.exists(LIB):
        return True
    return built_hash() != source_hash()


LAB_LIB = os.path.join(os.path.dirname(HERE), 'build_lab', 'libdrvae_lab.so')


def build(force=False, verbose=True, lab=False):
    """Compile every HIP source for gfx950 into one shared object next to the package.  ``lab``: the tuning build
    (-DDV_LAB: the GEMM tilings and probe kernels the dispatcher never selects) into build_lab/libdrvae_lab.so --
    load it with DRVAE_HIP_LIB; the product library does not carry them.

    The build is a function of the sources alone: the sources' sha256 is baked in (``dv_source_hash``; a rebuild happens
    exactly when it differs from the tree's), paths are mapped away (``-ffile-prefix-map``), the compilation-unit id is fixed
    (``-cuid``: hipcc's default hashes the output path into the device symbols' order), no build id, no timestamps,
    objects compiled from relative paths inside csrc/ -- two builds of one tree are byte-identical
    (``tests/test_cabi.py::test_build_is_reproducible``)."""
    if lab:
        os.makedirs(os.path.dirname(LAB_LIB), exist_ok=True)
    elif not force and not needs_build():
        return LIB
    sha = source_hash()
    objs = []
    procs = []
    odir = os.path.dirname(LAB_LIB) if lab else CSRC
    for s in SOURCES:
        o = os.path.join(odir, s.replace('.hip', '.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffile-prefix-map=%s=.' % CSRC,
               '-cuid=dv-' + s[:-4]] + (['-DDV_SOURCE_HASH="dv-src-sha256:%s"' % sha] if s == 'optim.hip' else []) + \
              (['-DDV_LAB'] if lab else []) + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    out = LAB_LIB if lab else LIB
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--build-id=none', '-o', out] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    if '--hash' in sys.argv:
        print(source_hash(), built_hash() if os.path.exists(LIB) else None)
    else:
        print(build(force='--force' in sys.argv, lab='--lab' in sys.argv))
