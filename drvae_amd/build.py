"""In-tree build of libdrvae_hip.so (hipcc, gfx950 only).  ``python -m drvae_amd.build``."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdrvae_hip.so')
SOURCES = ['gemm.hip', 'rows.hip', 'optim.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in ('dv_common.h', 'gemm_common.inc', 'gemm_pipe.inc')] + [
        os.path.join(os.path.dirname(HERE), 'include', 'drvae_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


LAB_LIB = os.path.join(os.path.dirname(HERE), 'build_lab', 'libdrvae_lab.so')


def build(force=False, verbose=True, lab=False):
    """Compile every HIP source for gfx950 into one shared object next to the package.  ``lab``: the tuning build
    (-DDV_LAB: the GEMM tilings and probe kernels the dispatcher never selects) into build_lab/libdrvae_lab.so --
    load it with DRVAE_HIP_LIB; the product library does not carry them."""
    if lab:
        os.makedirs(os.path.dirname(LAB_LIB), exist_ok=True)
    elif not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(os.path.dirname(LAB_LIB) if lab else CSRC, s.replace('.hip', '.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-DDV_LAB'] if lab else []) + \
              ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    out = LAB_LIB if lab else LIB
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, lab='--lab' in sys.argv))
