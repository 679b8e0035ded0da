"""-m gpu: every HIP kernel (through the C-ABI) against its plain-PyTorch fp32 reference."""
import numpy as np
import pytest
import torch

from tests import kernel_ref

pytestmark = pytest.mark.gpu


class _OnHost:
    """``tests/kernel_ref.py`` with every call evaluated on HOST copies of its tensor arguments: the reference arithmetic
    is plain PyTorch fp32 on the CPU (no GPU library in it).  Views keep their aliasing -- every distinct device storage
    is mirrored once and each argument re-created as the same strided view of the mirror -- and whatever the reference
    wrote is copied back into the device tensors it was handed."""

    def __getattr__(self, name):
        fn = getattr(kernel_ref, name)
        if not callable(fn):
            return fn

        def call(*args, **kw):
            mirrors = {}

            def host(t):
                if not (torch.is_tensor(t) and t.is_cuda):
                    return t
                st = t.untyped_storage()
                key = st.data_ptr()
                if key not in mirrors:
                    flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st)
                    mirrors[key] = (flat, flat.cpu())
                cpu_flat = mirrors[key][1]
                base = torch.empty(0, dtype=t.dtype).set_(cpu_flat.untyped_storage())
                return torch.as_strided(base, t.size(), t.stride(), t.storage_offset())

            def walk(a):
                if isinstance(a, (list, tuple)):
                    return type(a)(walk(v) for v in a)
                if isinstance(a, dict):
                    return {k: walk(v) for k, v in a.items()}
                return host(a)
            torch.cuda.synchronize()
            out = fn(*walk(list(args)), **walk(kw))
            for dev_flat, cpu_flat in mirrors.values():
                dev_flat.copy_(cpu_flat)
            return out
        return call


R = _OnHost()


@pytest.fixture(scope='module')
def K(dev):
    import drvae_amd.kernels as K
    from drvae_amd import _lib
    _lib.load()
    return K


def rnd(dev, *shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def strided(dev, rows, cols, pad, seed=0):
    """(rows, cols) view into a wider buffer: exercises leading dimensions != cols"""
    buf = rnd(dev, rows, cols + pad, seed=seed)
    return buf[:, :cols]


def close(a, b, rtol=2e-5, atol=2e-5):
    torch.cuda.synchronize()
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol)


def gemm_tol(K_):
    # fp32 fma chain in a different summation order than the reference: ~sqrt(K)*eps relative to |a||b|
    return dict(rtol=2e-4, atol=2e-5 * max(1.0, K_ ** 0.5))


PRODUCT_TILINGS = (0, 1, 2, 3, 17, 40, 46)      # dv_gemm_has_tiling of the product library


def tilings(*ts):
    """parametrisation over GEMM tilings: the product library's run by default, the lab ones (tuning build only:
    python -m drvae_amd.build --lab) carry the ``lab`` marker and are deselected unless asked for (tests/conftest.py)"""
    return [t if t in PRODUCT_TILINGS else pytest.param(t, marks=pytest.mark.lab) for t in ts]


def force_tiling(t):
    """force a GEMM tiling; a product tiling the library does not have is a failure, not a skip"""
    import drvae_amd.kernels as K
    if K.gemm_force_tiling(t) != 0:
        assert t not in PRODUCT_TILINGS, 'product tiling %d missing from the library' % t
        pytest.skip('tiling %d: lab build only' % t)


SHAPES = [(7, 5, 13), (64, 64, 32), (65, 33, 31), (225, 800, 978), (150, 200, 102), (300, 2, 200), (1, 1, 1),
          (130, 257, 100), (600, 100, 6),
          # 16-B aligned rows, K % 4 == 0: eligible for the hand-pipelined LDS-DMA tilings (40: 128x256x16 ring of 3)
          (260, 516, 200), (128, 256, 48), (1000, 300, 64)]


@pytest.mark.parametrize('tiling', tilings(0, 1, 2, 3, 5, 9, 11, 12, 13, 16, 17, 40, 46))
@pytest.mark.parametrize('M,N,Kd', SHAPES)
def test_gemm_forward_epilogue(K, dev, tiling, M, N, Kd):
    from drvae_amd import _lib
    force_tiling(tiling)
    try:
        x, W, b = rnd(dev, M, Kd, seed=1), rnd(dev, N, Kd, seed=2, scale=Kd ** -0.5), rnd(dev, N, seed=3)
        sc = rnd(dev, N, seed=4).abs() + 0.5
        for kw in (dict(), dict(bias=b, act0='elu', act1='elu'),
                   dict(bias=b, scale=sc, split=N // 2, act0='identity', act1='softplus', shift1=1e-3),
                   dict(bias=b, split=N // 2, act0='identity', act1='identity', shift1=-2.0)):
            out, ref = torch.full((M, N), 7.0, device=dev), torch.zeros(M, N, device=dev)
            K.linear_fwd(out, x, W, **kw)
            R.linear_fwd(ref, x, W, **kw)
            close(out, ref, **gemm_tol(Kd))
    finally:
        K.gemm_force_tiling(0)


@pytest.mark.parametrize('tiling', tilings(0, 1, 2, 3, 5, 16, 17, 40, 46))
@pytest.mark.parametrize('M,N,Kd', SHAPES)
def test_gemm_backward_products(K, dev, tiling, M, N, Kd):
    from drvae_amd import _lib
    force_tiling(tiling)
    try:
        x, W = rnd(dev, M, Kd, seed=1), rnd(dev, N, Kd, seed=2, scale=Kd ** -0.5)
        dpre, yprev = rnd(dev, M, N, seed=5), rnd(dev, M, Kd, seed=6)
        ks = rnd(dev, N, seed=7).abs() + 0.5
        # dx = (dpre*ks) W * elu'(yprev), accumulate on top of existing values
        dx, ref = rnd(dev, M, Kd, seed=8), None
        ref = dx.clone()
        K.linear_bwd_data(dx, dpre, W, kscale=ks, alpha=-0.5, beta=1.0, yref=yprev, act='elu')
        R.linear_bwd_data(ref, dpre, W, kscale=ks, alpha=-0.5, beta=1.0, yref=yprev, act='elu')
        close(dx, ref, **gemm_tol(N))
        dx2, ref2 = torch.empty(M, Kd, device=dev), torch.empty(M, Kd, device=dev)
        K.linear_bwd_data(dx2, dpre, W)
        R.linear_bwd_data(ref2, dpre, W)
        close(dx2, ref2, **gemm_tol(N))
        # dW = dpre^T x (+ fused bias gradient)
        dW, db = rnd(dev, N, Kd, seed=9), rnd(dev, N, seed=10)
        rW, rb = dW.clone(), db.clone()
        K.linear_bwd_weight(dW, dpre, x, beta=1.0, dbias=db)
        R.linear_bwd_weight(rW, dpre, x, beta=1.0, dbias=rb)
        close(dW, rW, **gemm_tol(M))
        close(db, rb, **gemm_tol(M))
        dW0, db0 = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev)
        rW0, rb0 = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev)
        K.linear_bwd_weight(dW0, dpre, x, dbias=db0)
        R.linear_bwd_weight(rW0, dpre, x, dbias=rb0)
        close(dW0, rW0, **gemm_tol(M))
        close(db0, rb0, **gemm_tol(M))
    finally:
        K.gemm_force_tiling(0)


def test_gemm_random_shapes_all_products(K, dev):
    """Random shapes around the tile edges (M, N, K not multiples of 32 / 64 / 4, leading dimensions padded or not),
    every product of a layer -- x W^T with a dual-head epilogue, dy W with the activation backward, dy^T x with the
    fused bias gradient, the paired dW+dX launch -- on the dispatcher's own choice and on the seven-per-CU tiling,
    against the plain-PyTorch reference"""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from drvae_amd import _lib
    lib = _lib.load()

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(M=st.integers(1, 700), N=st.integers(1, 700), Kd=st.integers(1, 400), pad=st.sampled_from([0, 1, 2, 4]),
           tiling=st.sampled_from([0, 17]))
    def run(M, N, Kd, pad, tiling):
        K.gemm_force_tiling(tiling)
        try:
            x, W = strided(dev, M, Kd, pad, seed=1), strided(dev, N, Kd, pad, seed=2)
            b, dpre, yprev = rnd(dev, N, seed=3), strided(dev, M, N, pad, seed=5), rnd(dev, M, Kd, seed=6)
            out, ref = torch.full((M, N), 7.0, device=dev), torch.zeros(M, N, device=dev)
            kw = dict(bias=b, split=N // 2, act0='identity', act1='softplus', shift1=1e-3) if N > 1 else dict(bias=b)
            K.linear_fwd(out, x, W, overread=False, **kw)
            R.linear_fwd(ref, x, W, **kw)
            close(out, ref, **gemm_tol(Kd))
            dx, rx = torch.empty(M, Kd, device=dev), torch.empty(M, Kd, device=dev)
            K.linear_bwd_data(dx, dpre, W, yref=yprev, act='elu')
            R.linear_bwd_data(rx, dpre, W, yref=yprev, act='elu')
            close(dx, rx, **gemm_tol(N))
            dW, db, rW, rb = (torch.empty(N, Kd, device=dev), torch.empty(N, device=dev), torch.empty(N, Kd, device=dev),
                              torch.empty(N, device=dev))
            K.linear_bwd_weight(dW, dpre, x, dbias=db)
            R.linear_bwd_weight(rW, dpre, x, dbias=rb)
            close(dW, rW, **gemm_tol(M))
            close(db, rb, **gemm_tol(M))
            if tiling == 0:
                dW2, db2, dx2 = torch.empty_like(dW), torch.empty_like(db), torch.empty_like(dx)
                K.linear_bwd_pair(dW2, db2, dx2, dpre, x, W, yref=yprev, act='elu')
                close(dW2, rW, **gemm_tol(M))
                close(db2, rb, **gemm_tol(M))
                close(dx2, rx, **gemm_tol(N))
        finally:
            K.gemm_force_tiling(0)
    run()

    # chip-filling layers (>= 512 tiles of 32x32 in dy^T x): the paired launch of the seven-per-CU tiling
    @settings(max_examples=8, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(M=st.integers(33, 900), N=st.integers(1200, 2100), Kd=st.integers(450, 700), pad=st.sampled_from([0, 2]))
    def run_big(M, N, Kd, pad):
        x, W = strided(dev, M, Kd, pad, seed=1), strided(dev, N, Kd, pad, seed=2)
        dpre, yprev = strided(dev, M, N, pad, seed=5), rnd(dev, M, Kd, seed=6)
        outs = []
        for L in (K, R):
            dW, db, dx = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev), torch.full((M, Kd), 3.0, device=dev)
            L.linear_bwd_pair(dW, db, dx, dpre, x, W, alpha=0.5, beta_x=1.0, yref=yprev, act='elu')
            outs.append((dW, db, dx))
        close(outs[0][0], outs[1][0], **gemm_tol(M))
        close(outs[0][1], outs[1][1], **gemm_tol(M))
        close(outs[0][2], outs[1][2], **gemm_tol(N))
    run_big()


def test_gemm_mfma_layout_asymmetric(K, dev):
    """A = I with an asymmetric B: catches a transposed C/D register map (guide section 3)."""
    n = 96
    eye = torch.eye(n, device=dev)
    Bm = (torch.arange(n * n, device=dev, dtype=torch.float32).reshape(n, n) % 251) * 0.01
    out = torch.empty(n, n, device=dev)
    K.gemm(out, eye, Bm, True, False)
    close(out, Bm, rtol=0, atol=0)
    K.gemm(out, eye, Bm, True, True)
    close(out, Bm.t(), rtol=0, atol=0)
    K.gemm(out, Bm, eye, False, False)
    close(out, Bm.t(), rtol=0, atol=0)


def test_gemm_split_sources_residual_strided(K, dev):
    M, K1, K2, N = 37, 10, 6, 20          # K1 % 4 != 0 -> scalar load path for the concat
    for K1_ in (K1, 12):
        x1, x2 = strided(dev, M, K1_, 3, seed=1), strided(dev, M, K2, 1, seed=2)
        W, b = rnd(dev, N, K1_ + K2, seed=3), rnd(dev, N, seed=4)
        res = strided(dev, M, N // 2, 5, seed=5)
        buf = torch.zeros(M, N + 4, device=dev)
        out, ref = buf[:, :N], torch.zeros(M, N, device=dev)
        kw = dict(x2=x2, split=N // 2, act0='identity', act1='identity', shift1=-2.0, resid=res, resid_cols=N // 2)
        K.linear_fwd(out, x1, W, b, **kw)
        R.linear_fwd(ref, x1, W, b, **kw)
        close(out, ref, rtol=1e-4, atol=1e-5)
        assert float(buf[:, N:].abs().max()) == 0.0          # padding columns untouched
        # column-sliced W views in the backward products
        dpre = rnd(dev, M, N, seed=6)
        dx2, rx2 = torch.empty(M, K2, device=dev), torch.empty(M, K2, device=dev)
        K.linear_bwd_data(dx2, dpre, W[:, K1_:])
        R.linear_bwd_data(rx2, dpre, W[:, K1_:])
        close(dx2, rx2, rtol=1e-4, atol=1e-5)
        dW = torch.zeros(N, K1_ + K2, device=dev)
        rW = torch.zeros(N, K1_ + K2, device=dev)
        K.linear_bwd_weight(dW[:, K1_:], dpre, x2)
        R.linear_bwd_weight(rW[:, K1_:], dpre, x2)
        close(dW, rW, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('M,N,Kd', [(224, 800, 978), (450, 200, 102), (37, 50, 18), (600, 1956, 601), (8192, 800, 978)])
def test_gemm_over_padded_k_and_n(K, dev, M, N, Kd):
    """``kpad`` / ``npad``: operands whose rows are zero-padded to 16 B (the plan's activation buffers, the arena's
    weights) multiply over the padded K / N -- the LDS-DMA kernels take them -- with the result of the unpadded product;
    pad columns of the outputs receive zeros only."""
    pad4 = lambda n: (n + 3) // 4 * 4
    def mat(r, c, seed, scale=1.0):
        t = torch.zeros(r, pad4(c), device=dev)
        t[:, :c] = rnd(dev, r, c, seed=seed, scale=scale)
        return t[:, :c]
    x, W, b = mat(M, Kd, 1), mat(N, Kd, 2, scale=Kd ** -0.5), rnd(dev, N, seed=3)       # (heads of O(1): the sample is mu + eps exp(lv / 2))
    out, ref = mat(M, N, 4), torch.empty(M, N, device=dev)
    tol = gemm_tol(Kd)
    K.linear_fwd(out, x, W, b, act0='elu', act1='elu', overread=True, kpad=True)
    R.linear_fwd(ref, x, W, b, act0='elu', act1='elu')
    close(out, ref, **tol)
    K.gemm(out, x, W, True, True, overread=True, kpad=True)
    R.gemm(ref, x, W, True, True)
    close(out, ref, **tol)
    if N % 2 == 0:        # the dual-head launch with samples in its epilogue
        eps, z, rz = rnd(dev, M, N // 2, seed=5), mat(M, N // 2, 6), torch.empty(M, N // 2, device=dev)
        kw = dict(split=N // 2, act0='identity', act1='identity', sample=dict(eps=eps, out=z, n_src=M))
        K.linear_heads(out, x, W, b, overread=True, kpad=True, **kw)
        kw['sample'] = dict(eps=eps, out=rz, n_src=M)
        R.linear_heads(ref, x, W, b, **kw)
        close(out, ref, **tol)
        close(z, rz, **tol)
    # backward: dW = dpre^T x and dx = dpre W over the padded N (N here = Kd, the layer's input width)
    dpre = mat(M, N, 7)
    dW, rW, db, rdb = mat(N, Kd, 8), torch.empty(N, Kd, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    dx, rx = mat(M, Kd, 9), torch.empty(M, Kd, device=dev)
    K.linear_bwd_pair(dW, db, dx, dpre, x, W, overread=True, npad=True, npad_x=True)
    R.linear_bwd_pair(rW, rdb, rx, dpre, x, W)
    close(dW, rW, **gemm_tol(M))
    close(db, rdb, **gemm_tol(M))
    close(dx, rx, **gemm_tol(N))
    K.linear_bwd_weight(dW, dpre, x, dbias=db, overread=True, npad=True)
    close(dW, rW, **gemm_tol(M))
    K.linear_bwd_data(dx, dpre, W, overread=True, npad=True)
    close(dx, rx, **gemm_tol(N))
    for t in (out, dW, dx, z if N % 2 == 0 else out):
        assert not t._base[:, t.shape[1]:].any()


@pytest.mark.parametrize('act', ['elu', 'softplus', 'sigmoid', 'tanh', 'relu', 'leaky_relu', 'selu', 'softsign'])
def test_activations(K, dev, act):
    M, N, Kd = 33, 47, 8
    x, W, b = rnd(dev, M, Kd, seed=1, scale=3.0), rnd(dev, N, Kd, seed=2), rnd(dev, N, seed=3)
    y, ry = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    K.linear_fwd(y, x, W, b, act0=act, act1=act)
    R.linear_fwd(ry, x, W, b, act0=act, act1=act)
    close(y, ry, rtol=1e-5, atol=1e-5)
    # derivative from the output agrees with autograd through the torch activation
    pre = (x.cpu() @ W.cpu().t() + b.cpu()).requires_grad_(True)
    kernel_ref.act_fwd(act, pre).backward(torch.ones_like(pre))
    dY = torch.ones(M, N, device=dev)
    K.act_bwd_(dY, y, act0=act, act1=act)
    close(dY, pre.grad, rtol=2e-4, atol=2e-5)


def test_colsum_wn(K, dev):
    X = strided(dev, 301, 77, 3, seed=1)
    out, ref = rnd(dev, 77, seed=2), None
    ref = out.clone()
    K.colsum(out, X, beta=1.0)
    R.colsum(ref, X, beta=1.0)
    close(out, ref, rtol=1e-5, atol=1e-4)
    N, Kd = 45, 978
    W, g = rnd(dev, N, Kd, seed=3, scale=0.05), rnd(dev, N, seed=4).abs() + 0.5
    sc, nm, rsc, rnm = (torch.empty(N, device=dev) for _ in range(4))
    K.wn_scale(sc, nm, W, g)
    R.wn_scale(rsc, rnm, W, g)
    close(sc, rsc, rtol=1e-5)
    close(nm, rnm, rtol=1e-5)
    dWraw = rnd(dev, N, Kd, seed=5)
    dW, dg = rnd(dev, N, Kd, seed=6), rnd(dev, N, seed=7)
    rW, rg = dW.clone(), dg.clone()
    K.wn_bwd(dW, dg, dWraw, W, g, nm, beta=1.0)
    R.wn_bwd(rW, rg, dWraw, W, g, rnm, beta=1.0)
    close(dW, rW, rtol=1e-4, atol=1e-4)
    close(dg, rg, rtol=1e-4, atol=1e-4)
    # against autograd of the closed form (src/layers.py:38-40)
    Wt, gt = W.clone().requires_grad_(True), g.clone().requires_grad_(True)
    x, dy = rnd(dev, 9, Kd, seed=8), rnd(dev, 9, N, seed=9)
    y = (gt / torch.norm(Wt, 2, 1)) * (x @ Wt.t())
    (y * dy).sum().backward()
    dW2, dg2 = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev)
    K.wn_bwd(dW2, dg2, dy.t() @ x, W, g, nm)
    close(dW2, Wt.grad, rtol=2e-4, atol=2e-5)
    close(dg2, gt.grad, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('mode', [0, 1])
def test_reparam(K, dev, mode):
    n, reps, Z, nq = 11, 3, 100, 17
    Q = rnd(dev, nq, 2 * Z, seed=1, scale=0.5)
    if mode == 1:
        Q[:, Z:] = Q[:, Z:].abs() + 0.1
    mu, sd = Q[:, :Z], Q[:, Z:]
    idx = torch.randperm(nq)[:n].to(torch.int32).to(dev)
    eps, sub = rnd(dev, n * reps, Z, seed=2), rnd(dev, n * reps, Z, seed=3)
    for src in (idx, None):
        muv, sdv = (mu, sd) if src is not None else (mu[:n], sd[:n])
        out, out2 = torch.empty(n * reps, Z, device=dev), torch.empty(n * reps, Z, device=dev)
        ro, ro2 = torch.empty_like(out), torch.empty_like(out2)
        K.reparam_fwd(out, muv, sdv, eps, mode=mode, src_idx=src, reps=reps, sub=sub, out2=out2)
        R.reparam_fwd(ro, muv, sdv, eps, mode=mode, src_idx=src, reps=reps, sub=sub, out2=ro2)
        close(out, ro, rtol=1e-6, atol=1e-6)
        close(out2, ro2, rtol=1e-6, atol=1e-6)
        dz = rnd(dev, n * reps, Z, seed=4)
        dQ = rnd(dev, nq, 2 * Z, seed=5)
        rQ = dQ.clone()
        K.reparam_bwd(dQ[:, :Z], dQ[:, Z:], dz, eps, sdv, mode=mode, src_idx=src, reps=reps, beta=1.0)
        R.reparam_bwd(rQ[:, :Z], rQ[:, Z:], dz, eps, sdv, mode=mode, src_idx=src, reps=reps, beta=1.0)
        close(dQ, rQ, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('with_p', [True, False])
def test_kl_rows(K, dev, mode, with_p):
    n, reps, Z, nq = 9, 2, 100, 12
    Rr = n * reps
    Q, P = rnd(dev, nq, 2 * Z, seed=1, scale=0.7), rnd(dev, Rr + 3, 2 * Z, seed=2, scale=0.7)
    if mode == 1:
        Q[:, Z:] = Q[:, Z:].abs() + 0.2
        P[:, Z:] = P[:, Z:].abs() + 0.2
    mq, sq, mp, sp = Q[:, :Z], Q[:, Z:], P[:, :Z], P[:, Z:]
    qidx = torch.randperm(nq)[:n].to(torch.int32).to(dev)
    pidx = torch.randperm(Rr + 3)[:Rr].to(torch.int32).to(dev)
    kw = dict(mode=mode, qidx=qidx, reps=reps, free_bits=True, kl_min=float(Z) * 0.3,
              prior=(0.1, 0.2 if mode == 0 else 1.3))
    pk = dict(mu_p=mp, sd_p=sp, pidx=pidx) if with_p else {}
    out, raw, ro, rr = (torch.empty(Rr, device=dev) for _ in range(4))
    K.kl_rows_fwd(out, raw, mq, sq, **pk, **kw)
    R.kl_rows_fwd(ro, rr, mq, sq, **pk, **kw)
    close(raw, rr, rtol=2e-5, atol=1e-4)
    close(out, ro, rtol=2e-5, atol=1e-4)
    coef = rnd(dev, Rr, seed=3)
    dq, dp = rnd(dev, Rr, 2 * Z, seed=4), rnd(dev, Rr, 2 * Z, seed=5)
    rq, rp = dq.clone(), dp.clone()
    dpa = (dp[:, :Z], dp[:, Z:]) if with_p else (None, None)
    rpa = (rp[:, :Z], rp[:, Z:]) if with_p else (None, None)
    K.kl_rows_bwd(dq[:, :Z], dq[:, Z:], dpa[0], dpa[1], coef, raw, mq, sq, **pk, beta=1.0, **kw)
    R.kl_rows_bwd(rq[:, :Z], rq[:, Z:], rpa[0], rpa[1], coef, rr, mq, sq, **pk, beta=1.0, **kw)
    close(dq, rq, rtol=2e-5, atol=2e-5)
    close(dp, rp, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('X,pad', [(978, 0), (978, 2), (980, 0), (13, 0), (20000, 0)])
def test_nll_rows(K, dev, mode, X, pad):
    M, nx = 23, 9
    x = strided(dev, nx, X, pad, seed=1)
    P = rnd(dev, M, 2 * (X + pad), seed=2)
    mu, sd = P[:, :X], P[:, X + pad:2 * X + pad]
    if mode == 1:
        sd.copy_(torch.nn.functional.softplus(sd) + 1e-3)
    xidx = torch.randint(0, nx, (M,)).to(torch.int32).to(dev)
    out, ro = torch.empty(M, device=dev), torch.empty(M, device=dev)
    K.nll_rows_fwd(out, x, mu, sd, mode=mode, xidx=xidx)
    R.nll_rows_fwd(ro, x, mu, sd, mode=mode, xidx=xidx)
    close(out, ro, rtol=2e-5, atol=1e-3)
    coef = rnd(dev, M, seed=3)
    D, RD = rnd(dev, M, 2 * X, seed=4), None
    RD = D.clone()
    dx, rdx = torch.zeros(M, X, device=dev), torch.zeros(M, X, device=dev)
    kw = dict(mode=mode, xidx=xidx, beta=1.0)
    if mode == 1:
        kw.update(sd_act='softplus', sd_shift=1e-3)
    K.nll_rows_bwd(D[:, :X], D[:, X:], coef, x, mu, sd, dx=dx, **kw)
    R.nll_rows_bwd(RD[:, :X], RD[:, X:], coef, x, mu, sd, dx=rdx, **kw)
    close(D, RD, rtol=2e-4, atol=2e-4)
    close(dx, rdx, rtol=2e-4, atol=2e-4)
    # the one-pass forward+backward agrees with the two separate launches (odd X / odd offsets
    # take the scalar path, even ones the float2 path)
    kw.pop('beta')
    o2 = torch.empty(M, device=dev)
    D2, RD2 = torch.full((M, 2 * X), 7.0, device=dev), torch.empty(M, 2 * X, device=dev)
    K.nll_rows_fwdbwd(o2, D2[:, :X], D2[:, X:], coef, x, mu, sd, **kw)
    R.nll_rows_bwd(RD2[:, :X], RD2[:, X:], coef, x, mu, sd, **kw)
    close(o2, ro, rtol=2e-5, atol=1e-3)
    close(D2, RD2, rtol=2e-4, atol=2e-4)
    if mode == 1:
        # raw heads, forward only (evaluation passes): the same numbers as finishing the heads first
        rawf = rnd(dev, M, 2 * X + pad, seed=7)
        bf = rnd(dev, 2 * X, seed=8)
        o5, o6 = torch.empty(M, device=dev), torch.empty(M, device=dev)
        K.nll_rows_fwd(o5, x, rawf[:, :X], rawf[:, X:2 * X], mode=1, xidx=xidx, bias=(bf[:X], bf[X:]), sd_shift=1e-3)
        R.nll_rows_fwd(o6, x, rawf[:, :X], rawf[:, X:2 * X], mode=1, xidx=xidx, bias=(bf[:X], bf[X:]), sd_shift=1e-3)
        close(o5, o6, rtol=3e-5, atol=2e-3)
        # raw heads: mu / sd hold x W^T, the pass adds the bias and applies softplus + shift itself
        raw = rnd(dev, M, 2 * (X + pad), seed=5)
        rmu, rsd = raw[:, :X], raw[:, X + pad:2 * X + pad]
        b = rnd(dev, 2 * X, seed=6)
        o3, D3, o4, D4 = torch.empty(M, device=dev), torch.full((M, 2 * X), 7.0, device=dev), torch.empty(M, device=dev), \
            torch.empty(M, 2 * X, device=dev)
        K.nll_rows_fwdbwd(o3, D3[:, :X], D3[:, X:], coef, x, rmu, rsd, bias=(b[:X], b[X:]), **kw)
        R.nll_rows_fwdbwd(o4, D4[:, :X], D4[:, X:], coef, x, rmu, rsd, bias=(b[:X], b[X:]), **kw)
        close(o3, o4, rtol=2e-5, atol=1e-3)
        close(D3, D4, rtol=3e-4, atol=3e-4)


@pytest.mark.parametrize('Y,sig', [(2, False), (3, False), (7, False), (2, True)])
def test_categorical(K, dev, Y, sig):
    M = 301
    logits = rnd(dev, M, 1 if sig else Y, seed=1, scale=4.0)
    logits[0] = 60.0 if sig else torch.tensor([60.0] + [-60.0] * (Y - 1), device=dev)   # clamp active
    p, rp = torch.empty(M, Y, device=dev), torch.empty(M, Y, device=dev)
    K.softmax_clamp_fwd(p, logits, sig)
    R.softmax_clamp_fwd(rp, logits, sig)
    close(p, rp, rtol=3e-5, atol=1e-12)          # (device expf against the host's: ~1e-5 relative on probabilities of 1e-4)
    assert float(p.min()) >= 1e-10 * 0.999
    g = rnd(dev, M, Y, seed=2)
    dl, rdl = rnd(dev, M, logits.shape[1], seed=3), None
    rdl = dl.clone()
    K.softmax_clamp_bwd(dl, g, p, sig, beta=1.0)
    R.softmax_clamp_bwd(rdl, g, rp, sig, beta=1.0)
    close(dl, rdl, rtol=1e-4, atol=1e-6)
    labels = torch.randint(0, Y, (M,)).to(torch.int32).to(dev)
    prior = torch.softmax(rnd(dev, M, Y, seed=4), -1)
    logp, kl, ent = torch.empty(M, device=dev), torch.empty(M, Y, device=dev), torch.empty(M, device=dev)
    best = torch.empty(M, dtype=torch.int32, device=dev)
    rlogp, rkl, rent, rbest = torch.empty_like(logp), torch.empty_like(kl), torch.empty_like(ent), best.clone()
    K.cat_terms_fwd(p, labels=labels, prior=prior, logp=logp, kl=kl, ent=ent, best=best)
    R.cat_terms_fwd(rp, labels=labels, prior=prior, logp=rlogp, kl=rkl, ent=rent, best=rbest)
    close(logp, rlogp, rtol=1e-5, atol=1e-5)
    close(kl, rkl, rtol=1e-5, atol=1e-6)
    close(ent, rent, rtol=1e-5, atol=1e-6)
    assert bool((best == rbest).all())
    c1, gk, c2 = rnd(dev, M, seed=5), rnd(dev, M, Y, seed=6), rnd(dev, M, seed=7)
    dpb, rdp = rnd(dev, M, Y, seed=8), None
    rdp = dpb.clone()
    K.cat_terms_bwd(dpb, p, labels=labels, prior=prior, c_logp=c1, g_kl=gk, c_ent=c2, beta=1.0)
    R.cat_terms_bwd(rdp, rp, labels=labels, prior=prior, c_logp=c1, g_kl=gk, c_ent=c2, beta=1.0)
    close(dpb, rdp, rtol=1e-4, atol=1e-4)


def test_ymarg(K, dev):
    Rr, Y = 40, 3
    lab = torch.rand(Rr) < 0.5
    nf = torch.where(lab, torch.ones(Rr, dtype=torch.int64), torch.full((Rr,), Y))
    fp_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), nf.cumsum(0)]).to(torch.int32).to(dev)
    F_ = int(fp_ptr[-1])
    qy = torch.softmax(rnd(dev, Rr, Y, seed=1), -1)
    label = torch.randint(0, Y, (Rr,)).to(torch.int32).to(dev)
    klfp = rnd(dev, F_, seed=2).abs() + 2
    lp = float(np.log(1.0 / Y))
    yl, kld, ryl, rkld = (torch.empty(Rr, device=dev) for _ in range(4))
    K.ymarg_fwd(yl, kld, qy, label, fp_ptr, klfp, lp)
    R.ymarg_fwd(ryl, rkld, qy, label, fp_ptr, klfp, lp)
    close(yl, ryl, rtol=1e-5, atol=1e-6)
    close(kld, rkld, rtol=1e-5, atol=1e-5)
    ck, cy = rnd(dev, Rr, seed=3), rnd(dev, Rr, seed=4)
    cfp, dqy = torch.empty(F_, device=dev), torch.empty(Rr, Y, device=dev)
    rcfp, rdqy = torch.empty(F_, device=dev), torch.empty(Rr, Y, device=dev)
    K.ymarg_bwd(cfp, dqy, qy, label, fp_ptr, klfp, lp, ck, cy)
    R.ymarg_bwd(rcfp, rdqy, qy, label, fp_ptr, klfp, lp, ck, cy)
    close(cfp, rcfp, rtol=1e-5, atol=1e-6)
    close(dqy, rdqy, rtol=1e-5, atol=1e-5)
    # one-launch forward+backward, also with a class prior given as data (log-prior vector)
    for prior in (lp, torch.log(torch.tensor([0.2, 0.5, 0.3])).to(dev)):
        o = [torch.empty(Rr, device=dev), torch.empty(Rr, device=dev), torch.empty(F_, device=dev),
             torch.empty(Rr, Y, device=dev)]
        ro = [torch.empty_like(t) for t in o]
        K.ymarg_fwdbwd(*o, qy, label, fp_ptr, klfp, prior, ck, cy)
        R.ymarg_fwd(ro[0], ro[1], qy, label, fp_ptr, klfp, prior)
        R.ymarg_bwd(ro[2], ro[3], qy, label, fp_ptr, klfp, prior, ck, cy)
        for a, b in zip(o, ro):
            close(a, b, rtol=1e-5, atol=1e-5)


def test_ycont(K, dev):
    Rr, B, Y = 24, 12, 2
    mu = torch.sigmoid(rnd(dev, Rr, Y + 2, seed=1))[:, :Y]
    ylab, eps = torch.rand(B, Y).to(dev), rnd(dev, Rr, Y, seed=2)
    has_y = (torch.arange(B) % 3 != 0).to(torch.int32).to(dev)
    lv = float(np.log(0.05 ** 2))
    for sq in (False, True):
        _ycont_case(K, dev, mu, ylab, eps, has_y, lv, Rr, B, Y, sq)


def _ycont_case(K, dev, mu, ylab, eps, has_y, lv, Rr, B, Y, sq):
    outs = []
    for mod in (K, R):
        yl = torch.empty(Rr, device=dev)
        f1, f2 = torch.zeros(Rr, 7 + Y, device=dev), torch.zeros(Rr, 5 + Y, device=dev)
        mod.ycont_fwd(yl, f1[:, 7:], f2[:, 5:], mu, ylab, has_y, eps, lv, B, sqerr=sq)
        c_yl, c_kld = rnd(dev, Rr, seed=3), rnd(dev, Rr, seed=4)
        d1, d2 = rnd(dev, Rr, 7 + Y, seed=5), rnd(dev, Rr, 5 + Y, seed=6)
        dl, cfp = torch.empty(Rr, Y, device=dev), torch.empty(Rr, device=dev)
        mod.ycont_bwd(None, cfp, mu, ylab, has_y, lv, c_yl, c_kld, d1[:, 7:], d2[:, 5:], B, sqerr=sq)
        mod.ycont_bwd(dl, None, mu, ylab, has_y, lv, c_yl, c_kld, d1[:, 7:], d2[:, 5:], B, sqerr=sq)
        outs.append((yl, f1, f2, dl, cfp))
    for a, b in zip(*outs):
        close(a, b, rtol=2e-5, atol=1e-4)


def test_row_movement(K, dev):
    n, W, Y, ns = 19, 13, 3, 30
    src, noise = strided(dev, ns, W, 2, seed=1), rnd(dev, n, W, seed=2)
    idx = torch.randint(0, ns, (n,)).to(torch.int32).to(dev)
    cls = torch.randint(0, Y, (n,)).to(torch.int32).to(dev)
    out, ro = torch.zeros(n, W + Y + 1, device=dev), torch.zeros(n, W + Y + 1, device=dev)
    K.rows_gather(out, src, idx, noise=noise, sigma=0.01, onehot_cls=cls, n_classes=Y)
    R.rows_gather(ro, src, idx, noise=noise, sigma=0.01, onehot_cls=cls, n_classes=Y)
    close(out, ro, rtol=0, atol=1e-6)
    out2, ro2 = torch.zeros(n, W, device=dev), torch.zeros(n, W, device=dev)
    K.rows_gather(out2, src[:n])
    R.rows_gather(ro2, src[:n])
    close(out2, ro2, rtol=0, atol=0)
    # segment sum with CSR, weights and destination indices
    sizes = torch.randint(0, 4, (7,))
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), sizes.cumsum(0)]).to(torch.int32).to(dev)
    T = int(ptr[-1])
    rows = torch.randint(0, ns, (T,)).to(torch.int32).to(dev)
    w = rnd(dev, T, seed=3)
    didx = torch.randperm(11)[:7].to(torch.int32).to(dev)
    dst, rdst = rnd(dev, 11, W, seed=4), None
    rdst = dst.clone()
    K.rows_segment_sum(dst, src, seg_ptr=ptr, seg_rows=rows, w=w, dst_idx=didx, beta=1.0)
    R.rows_segment_sum(rdst, src, seg_ptr=ptr, seg_rows=rows, w=w, dst_idx=didx, beta=1.0)
    close(dst, rdst, rtol=1e-5, atol=1e-5)
    dst2, rdst2 = rnd(dev, 11, W, seed=5), None
    rdst2 = dst2.clone()
    K.rows_segment_sum(dst2, src, seg_rows=rows[:5] if T >= 5 else None, dst_idx=didx[:5], n=5, beta=1.0)
    R.rows_segment_sum(rdst2, src, seg_rows=rows[:5] if T >= 5 else None, dst_idx=didx[:5], n=5, beta=1.0)
    close(dst2, rdst2, rtol=1e-5, atol=1e-5)
    x, wv = rnd(dev, 1000, seed=6), rnd(dev, 1000, seed=7)
    o, r_ = torch.ones(1, device=dev), torch.ones(1)
    K.weighted_sum(o, x, wv, scale=0.5, beta=1.0)
    close(o, (1 + 0.5 * (x * wv).sum()).reshape(1), rtol=1e-4, atol=1e-4)
    y, ry = rnd(dev, 1000, seed=8), None
    ry = y.clone()
    K.axpby(y, x, a=-2.0, b=0.5)
    close(y, -2 * x + 0.5 * ry, rtol=1e-6, atol=1e-6)


def test_adam_matches_torch_optim(K, dev):
    n = 100003
    p0, g = rnd(dev, n, seed=1), rnd(dev, n, seed=2)
    p = p0.clone()
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    ref = p0.clone().cpu().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=5e-4, weight_decay=0.05)
    for it in range(3):
        gi = g * (it + 1)
        K.counter_add(step, 1)
        K.adam_l2(p, gi, m, v, step, lr=5e-4, weight_decay=0.05)
        ref.grad = gi.cpu().clone()
        opt.step()
    assert int(step.item()) == 3
    close(p, ref.detach(), rtol=1e-6, atol=1e-7)


def _other_queue_stream(K, dev):
    """a stream on a different HARDWARE queue than the current one (HIP multiplexes its streams onto a few
    queues; a parked kernel blocks everything behind it in its queue): probe candidates"""
    for _ in range(16):
        cand = torch.cuda.Stream()
        probe = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        K.flag_wait(probe[0:1], probe[1:2], probe[2:4], add=1, max_spins=20000)
        with torch.cuda.stream(cand):
            K.flag_publish(probe[0:1], probe[1:2], 1)
        torch.cuda.synchronize()
        if int(probe[2]) == 0:
            return cand
    pytest.skip('no second hardware queue')


@pytest.mark.parametrize('n,lo,hi', [(100003, 5000, 5402), (100003, 99990, 100003), (4096, 0, 4096)])
def test_adam_gated_sweep(K, dev, n, lo, hi):
    """dv_adam_l2_gated: same result as the plain sweep; the gated slice is produced by ANOTHER stream that
    publishes the flag afterwards (the optimiser launch is enqueued first and has to park)"""
    p0, g_final = rnd(dev, n, seed=1), rnd(dev, n, seed=2)
    step = torch.ones(1, dtype=torch.int32, device=dev)
    want, m0, v0 = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    K.adam_l2(want, g_final, m0, v0, step, lr=5e-4, weight_decay=0.05)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    g = g_final.clone()
    g[lo:hi] = 777.0                                   # not final yet
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    side = _other_queue_stream(K, dev)
    torch.cuda.synchronize()
    K.adam_l2(p, g, m, v, step, lr=5e-4, weight_decay=0.05, gate=(flag, step, 0, err, lo, hi))
    with torch.cuda.stream(side):
        torch.cuda._sleep(2000000)                     # ~1 ms: the sweep is parked on the gated slice by now
        g[lo:hi] = g_final[lo:hi]
        K.flag_publish(flag, step, 0)
    torch.cuda.synchronize()
    assert int(err[0]) == 0
    close(p, want, rtol=0, atol=0)


def test_publish_and_park_on_side_tail_launches(K, dev):
    """dv_publish on dv_smalln_linear_bwd_weight / dv_counters_add2 (published on entry, before the counters move) and
    dv_wait on dv_fill_normal_rows (the draw launch parks, and reads the Philox counter behind the wait)"""
    flag = torch.zeros(2, dtype=torch.int32, device=dev)
    ctr = torch.tensor([6], dtype=torch.int32, device=dev)
    t = torch.tensor([7], dtype=torch.int32, device=dev)
    K.counters_add2(ctr, 1, t, 1, publish=(flag[0:1], ctr, 1))
    torch.cuda.synchronize()
    assert flag[0].item() == 7 and ctr.item() == 7 and t.item() == 8       # published 6 + 1, THEN advanced
    K.counters_add2(ctr, 1, t, 1)
    torch.cuda.synchronize()
    assert flag[0].item() == 7 and ctr.item() == 8
    # classifier weight gradient with a publish: same numbers as without
    M, N, K1 = 37, 2, 50
    dp, pr = rnd(dev, M, N, seed=1), torch.softmax(rnd(dev, M, N, seed=2), 1)
    a1 = rnd(dev, M, K1, seed=3)
    dW, db, dW2, db2 = (torch.zeros(N, K1, device=dev), torch.zeros(N, device=dev), torch.zeros(N, K1, device=dev),
                        torch.zeros(N, device=dev))
    K.smalln_bwd_weight(dW, db, dp, pr, a1)
    K.smalln_bwd_weight(dW2, db2, dp, pr, a1, publish=(flag[1:2], ctr, 3))
    torch.cuda.synchronize()
    assert flag[1].item() == 11
    close(dW2, dW, rtol=0, atol=0)
    close(db2, db, rtol=0, atol=0)
    # parked draws: the Philox counter is advanced by the publisher, on another queue, before it publishes
    desc = torch.tensor([[0, 40, 0, 0], [40, 40, 0, 1], [80, 33, 1, 0]], dtype=torch.int32, device=dev)
    rng = torch.tensor([5, 0], dtype=torch.int32, device=dev)
    want, got = torch.zeros(113, device=dev), torch.zeros(113, device=dev)
    rng6 = torch.tensor([6, 0], dtype=torch.int32, device=dev)
    K.fill_normal_rows(want, desc, 99, rng6)
    f2 = torch.zeros(1, dtype=torch.int32, device=dev)
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    side = _other_queue_stream(K, dev)
    torch.cuda.synchronize()
    K.fill_normal_rows(got, desc, 99, rng, park=(f2, ctr, err, 1))
    with torch.cuda.stream(side):
        torch.cuda._sleep(2000000)
        K.counter_add(rng, 1)
        K.flag_publish(f2, ctr, 1)
    torch.cuda.synchronize()
    assert int(err[0]) == 0 and int(err[1]) > 0
    close(got, want, rtol=0, atol=0)
    # a wait nobody answers times out, sets the sticky word and still returns
    err2 = torch.zeros(2, dtype=torch.int32, device=dev)
    K.fill_normal_rows(got, desc, 99, rng, park=(f2, ctr, err2, 5, 50))
    torch.cuda.synchronize()
    assert int(err2[0]) == 1


def test_park_and_bump_arguments(K, dev):
    """dv_wait / dv_bump arguments: a rows_segment_sum given ``park`` parks on a flag published later from another
    queue, a reparam_bwd_seg given ``bump`` advances the counters; nothing is remembered between calls"""
    n, W = 64, 100
    src, want = rnd(dev, n, W, seed=1), torch.zeros(n, W, device=dev)
    dst = torch.zeros(n, W, device=dev)
    K.rows_segment_sum(want, src, beta=0.0, width=W, n=n)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ctr = torch.tensor([6], dtype=torch.int32, device=dev)
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    stale = torch.full((n, W), 777.0, device=dev)
    live = stale.clone()
    side = _other_queue_stream(K, dev)
    torch.cuda.synchronize()
    K.rows_segment_sum(dst, live, beta=0.0, width=W, n=n, park=(flag, ctr, err, 1))   # parks: ``live`` is not final yet
    with torch.cuda.stream(side):
        torch.cuda._sleep(2000000)
        live.copy_(src)
        K.flag_publish(flag, ctr, 1)
    torch.cuda.synchronize()
    assert int(err[0]) == 0 and int(err[1]) > 0
    close(dst, want, rtol=0, atol=0)
    K.rows_segment_sum(dst, stale, beta=0.0, width=W, n=n)       # no park argument: runs straight away
    torch.cuda.synchronize()
    assert float(dst[0, 0]) == 777.0
    # bump on reparam_bwd_seg
    nq, Z, L = 5, 8, 2
    dz, eps = rnd(dev, nq * L, Z, seed=2), rnd(dev, nq * L, Z, seed=3)
    lv = rnd(dev, nq, Z, seed=4)
    ptr = torch.arange(0, nq * L + 1, L, dtype=torch.int32, device=dev)
    rows_ = torch.arange(nq * L, dtype=torch.int32, device=dev)
    dmu, dlv, dmu2, dlv2 = (torch.zeros(nq, Z, device=dev) for _ in range(4))
    step = torch.tensor([3], dtype=torch.int32, device=dev)
    rng = torch.tensor([-2, 0], dtype=torch.int32, device=dev)
    K.reparam_bwd_seg(dmu2, dlv2, dz, eps, lv, ptr, rows_)
    K.reparam_bwd_seg(dmu, dlv, dz, eps, lv, ptr, rows_, bump=[(step, 1), (rng, 5)])
    K.reparam_bwd_seg(dmu, dlv, dz, eps, lv, ptr, rows_)         # second launch: no bump
    torch.cuda.synchronize()
    assert step.tolist() == [4] and rng.tolist() == [3, 1]
    close(dmu, dmu2, rtol=0, atol=0)
    close(dlv, dlv2, rtol=0, atol=0)


def test_adamax_matches_torch_optim(K, dev):
    n = 100003
    p0, g = rnd(dev, n, seed=1), rnd(dev, n, seed=2)
    p = p0.clone()
    m, u = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    ref = p0.clone().cpu().requires_grad_(True)
    opt = torch.optim.Adamax([ref], lr=5e-4, weight_decay=0.05)
    for it in range(3):
        gi = g * (0.5 ** it)                          # shrinking gradients: the running max matters
        K.counter_add(step, 1)
        K.adamax_l2(p, gi, m, u, step, lr=5e-4, weight_decay=0.05)
        ref.grad = gi.cpu().clone()
        opt.step()
    close(p, ref.detach(), rtol=1e-6, atol=1e-7)
    close(u, opt.state[ref]['exp_inf'], rtol=1e-6, atol=1e-9)


def test_fill_normal_statistics_and_counter(K, dev):
    n = 1 << 20
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    ctr = torch.zeros(2, dtype=torch.int32, device=dev)
    K.fill_normal(a, 1234, ctr)
    K.fill_normal(b, 1234, ctr)
    close(a, b, rtol=0, atol=0)                       # same (seed, counter) -> same stream
    K.counter_add(ctr, n // 4)
    K.fill_normal(b, 1234, ctr)
    assert float((a - b).abs().max()) > 0             # advanced counter -> fresh stream
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 3).mean())) < 2e-2 and abs(float((a ** 4).mean()) - 3) < 5e-2
    assert bool(torch.isfinite(a).all())
    c2 = torch.tensor([-1, 0], dtype=torch.int32, device=dev)     # 0x00000000ffffffff + 1 carries
    K.counter_add(c2, 1)
    assert c2.tolist() == [0, 1]


def test_loss_assemble(K, dev):
    xs = [rnd(dev, n, seed=i) for i, n in enumerate((596, 148, 300, 300, 7))]
    w1 = rnd(dev, 148, seed=9)
    terms = [(xs[0][:448], None, 0.25, 0), (xs[0][448:], None, 0.5, 2), (xs[1], w1, 1.0, 1), (xs[2], None, 0.1, 1),
             (xs[3], None, -0.3, 3), (xs[4], None, 2.0, 1)]
    w_elbo = torch.tensor([1.0, -1.0, 0.05], device=dev)
    w_cmpl = torch.tensor([0, 0, 0, -1.0, 0, -1.0, 0, 0], device=dev)
    loss, ref = torch.full((8,), 9.0, device=dev), torch.zeros(8, device=dev)
    K.loss_assemble(loss, terms, w_elbo, w_cmpl)
    R.loss_assemble(ref, terms, w_elbo, w_cmpl)
    close(loss, ref, rtol=1e-5, atol=1e-4)
    # the parked variant: the flag is already published; the launch advances the counters at its end
    # (the step counter doubles as the wait's reference: it is read before it is advanced)
    flag = torch.tensor([5], dtype=torch.int32, device=dev)
    step = torch.tensor([4], dtype=torch.int32, device=dev)
    rng = torch.tensor([-3, 7], dtype=torch.int32, device=dev)          # 64-bit (lo, hi): the add carries
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    loss.fill_(9.0)
    K.loss_assemble(loss, terms, w_elbo, w_cmpl, after=(flag, step, err, 1, 1000), bump=[(step, 1), (rng, 5)])
    close(loss, ref, rtol=1e-5, atol=1e-4)
    assert step.tolist() == [5] and rng.tolist() == [2, 8] and int(err[0]) == 0
    K.loss_assemble(loss, terms, w_elbo, w_cmpl, after=(flag, step, err, 1, 1000), bump=[(step, 1)])
    assert int(err[0]) == 1 and step.tolist() == [6]                    # flag 5 < 5 + 1: bounded wait, reported
    # running sums (``accum``): every launch that assembles terms adds its scalars; the parked launch WITHOUT terms
    # (the dual-graph step's join, whose scalars another chain assembles) must not add anything
    acc, racc = torch.full((8,), 0.5, device=dev), torch.full((8,), 0.5, device=dev)
    err.zero_()
    for _ in range(3):
        K.loss_assemble(loss, terms, w_elbo, w_cmpl, accum=acc)
        R.loss_assemble(ref, terms, w_elbo, w_cmpl, accum=racc)
    close(acc, racc, rtol=1e-5, atol=1e-4)
    close(acc, 0.5 + 3 * ref, rtol=1e-5, atol=1e-4)
    flag.fill_(100)
    K.loss_assemble(loss, [], w_elbo, w_cmpl, after=(flag, step, err, 1, 1000), bump=[(step, 1)], accum=acc)
    K.loss_assemble(loss, terms, w_elbo, w_cmpl, after=(flag, step, err, 1, 1000), accum=acc)
    close(acc, 0.5 + 4 * ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('Y,two', [(2, True), (3, False), (8, True), (1, False)])
def test_smalln_linear_head(K, dev, Y, two):
    M, K1, K2 = 301, 100, 100 if two else 0
    a1 = strided(dev, M, K1, 4, seed=1)
    a2 = strided(dev, M, K2, 0, seed=2) if two else None
    W, b = rnd(dev, Y, K1 + K2, seed=3, scale=0.3), rnd(dev, Y, seed=4)
    p, lg, rp, rlg = (torch.empty(M, Y, device=dev) for _ in range(4))
    K.smalln_fwd(p, lg, a1, W, b, a2)
    R.smalln_fwd(rp, rlg, a1, W, b, a2)
    close(lg, rlg, rtol=1e-4, atol=1e-4)
    close(p, rp, rtol=1e-4, atol=1e-6)
    g = rnd(dev, M, Y, seed=5)
    for probs in (p, None):
        d1, d2 = rnd(dev, M, K1, seed=6), rnd(dev, M, max(K2, 1), seed=7)
        r1, r2 = d1.clone(), d2.clone()
        dsts = [(d1, 0, 1.0, 1.0, K1, -1.0)] if two else [(d1, 0, 0.5, 0.0)]
        rds = [(r1, 0, 1.0, 1.0, K1, -1.0)] if two else [(r1, 0, 0.5, 0.0)]
        if two:
            dsts.append((d2, K1, 1.0, 0.0))
            rds.append((r2, K1, 1.0, 0.0))
        K.smalln_bwd_data(dsts, g, probs, W)
        R.smalln_bwd_data(rds, g, rp if probs is not None else None, W)
        close(d1, r1, rtol=1e-4, atol=1e-5)
        close(d2, r2, rtol=1e-4, atol=1e-5)
        # destination 0 started from a segment sum (1 or Y source rows per row): bitwise the two-launch sequence
        cnt = [1 if i % 3 else Y for i in range(M)]
        ptr = torch.tensor(np.concatenate([[0], np.cumsum(cnt)]), dtype=torch.int32, device=dev)
        src = strided(dev, int(ptr[-1]), K1, 2, seed=10)
        K.smalln_bwd_data(dsts, g, probs, W, seg=(src, ptr))
        K.rows_segment_sum(r1, src, seg_ptr=ptr, beta=0.0, width=K1)
        two_step = [(r1,) + tuple(dsts[0][1:3]) + (1.0,) + tuple(dsts[0][4:])] + [(r2,) + tuple(d[1:]) for d in dsts[1:]]
        K.smalln_bwd_data(two_step, g, probs, W)
        close(d1, r1, rtol=0, atol=0)
        close(d2, r2, rtol=0, atol=0)
        dW, db = rnd(dev, Y, K1 + K2, seed=8), rnd(dev, Y, seed=9)
        rW, rb = dW.clone(), db.clone()
        K.smalln_bwd_weight(dW, db, g, probs, a1, a2, beta=1.0)
        R.smalln_bwd_weight(rW, rb, g, rp if probs is not None else None, a1, a2, beta=1.0)
        close(dW, rW, rtol=1e-4, atol=1e-4)
        close(db, rb, rtol=1e-4, atol=1e-4)


def test_kl_rows_second_term(K, dev):
    """dv_kl_rows_fwd with a second, row-aligned term against the scalar prior (own free bits, own raw output) ==
    two launches chained through ``add``"""
    n, Z, Z2 = 45, 100, 37
    Q, P, Q2 = rnd(dev, 20, 2 * Z, seed=1, scale=0.5), rnd(dev, n, 2 * Z, seed=2, scale=0.5), rnd(dev, n, 2 * Z2, seed=3)
    qidx = torch.tensor([i % 20 for i in range(n)], dtype=torch.int32, device=dev)
    for kl_min in (0.0, 40.0):
        one, raw1, raw2 = (torch.empty(n, device=dev) for _ in range(3))
        K.kl_rows_fwd(one, raw1, Q[:, :Z], Q[:, Z:], P[:, :Z], P[:, Z:], qidx=qidx, free_bits=True, kl_min=kl_min,
                      prior=(0.0, 0.0), second=(Q2[:, :Z2], Q2[:, Z2:], raw2))
        t3, r3, two, r1 = (torch.empty(n, device=dev) for _ in range(4))
        K.kl_rows_fwd(t3, r3, Q2[:, :Z2], Q2[:, Z2:], prior=(0.0, 0.0), free_bits=True, kl_min=kl_min)
        K.kl_rows_fwd(two, r1, Q[:, :Z], Q[:, Z:], P[:, :Z], P[:, Z:], qidx=qidx, free_bits=True, kl_min=kl_min, add=t3)
        close(raw1, r1, rtol=1e-6, atol=1e-5)     # (the two-term loop may contract its fmas differently: 1 ulp)
        close(raw2, r3, rtol=1e-6, atol=1e-5)
        close(one, two, rtol=1e-6, atol=1e-5)
        ref, rr1, rr2 = (torch.empty(n, device=dev) for _ in range(3))
        R.kl_rows_fwd(ref, rr1, Q[:, :Z], Q[:, Z:], P[:, :Z], P[:, Z:], qidx=qidx, free_bits=True, kl_min=kl_min,
                      prior=(0.0, 0.0), second=(Q2[:, :Z2], Q2[:, Z2:], rr2))
        close(one, ref, rtol=1e-5, atol=1e-4)
        close(raw2, rr2, rtol=1e-5, atol=1e-4)


def test_fused_extensions_reparam_kl(K, dev):
    n, Z = 37, 100
    Q = rnd(dev, n, 2 * Z, seed=1, scale=0.5)
    mu, lv = Q[:, :Z], Q[:, Z:]
    eps = rnd(dev, n, Z, seed=2)
    # reparam with a scattered second copy
    idx3 = torch.full((n,), -1, dtype=torch.int32)
    idx3[::3] = torch.randperm(20)[:len(idx3[::3])].to(torch.int32)
    idx3 = idx3.to(dev)
    out, o3 = torch.empty(n, Z, device=dev), torch.zeros(20, Z + 4, device=dev)
    ro, r3 = torch.empty(n, Z, device=dev), torch.zeros(20, Z + 4, device=dev)
    K.reparam_fwd(out, mu, lv, eps, out3=o3[:, :Z], out3_idx=idx3)
    R.reparam_fwd(ro, mu, lv, eps, out3=r3[:, :Z], out3_idx=idx3)
    close(out, ro, rtol=1e-6, atol=1e-6)
    close(o3, r3, rtol=1e-6, atol=1e-6)
    # KL vs prior with free bits, fused sample and additive term
    add = rnd(dev, n, seed=3)
    kl, raw, z = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.zeros(n, Z + 2, device=dev)
    rkl, rraw, rz = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.zeros(n, Z + 2, device=dev)
    kw = dict(prior=(0.0, 0.0), free_bits=True, kl_min=30.0, add=add, eps=eps)
    K.kl_rows_fwd(kl, raw, mu, lv, zout=z[:, :Z], **kw)
    R.kl_rows_fwd(rkl, rraw, mu, lv, zout=rz[:, :Z], **kw)
    close(kl, rkl, rtol=2e-5, atol=1e-4)
    close(z, rz, rtol=1e-6, atol=1e-6)
    # backward with the fused sample term
    coef, dz = rnd(dev, n, seed=4), rnd(dev, n, Z, seed=5)
    dq, rq = torch.empty(n, 2 * Z, device=dev), torch.empty(n, 2 * Z, device=dev)
    kwb = dict(prior=(0.0, 0.0), free_bits=True, kl_min=30.0, dz=dz, eps=eps)
    K.kl_rows_bwd(dq[:, :Z], dq[:, Z:], None, None, coef, raw, mu, lv, **kwb)
    R.kl_rows_bwd(rq[:, :Z], rq[:, Z:], None, None, coef, rraw, mu, lv, **kwb)
    close(dq, rq, rtol=2e-5, atol=2e-5)


def _csr(sizes, total, dev, seed):
    g = torch.Generator().manual_seed(seed)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)])
    rows = torch.randint(0, total, (int(ptr[-1]),), generator=g)
    return ptr.to(torch.int32).to(dev), rows.to(torch.int32).to(dev)


def test_reparam_bwd_seg(K, dev):
    nq, Z, R_, F_ = 23, 100, 90, 40
    sd = rnd(dev, nq + 5, 2 * Z, seed=1, scale=0.5)[:, Z:]
    dz, eps, extra = rnd(dev, R_, Z, seed=2), rnd(dev, R_, Z, seed=3), rnd(dev, F_, 2 * Z, seed=4)
    sp, sr = _csr([int(v) for v in torch.randint(1, 5, (nq,), generator=torch.Generator().manual_seed(5))], R_, dev, 6)
    ep, er = _csr([int(v) for v in torch.randint(0, 3, (nq,), generator=torch.Generator().manual_seed(7))], F_, dev, 8)
    for ex in (True, False):
        dq, rq = rnd(dev, nq, 2 * Z, seed=9), None
        rq = dq.clone()
        kw = dict(extra=extra, ex_ptr=ep, ex_rows=er) if ex else {}
        K.reparam_bwd_seg(dq[:, :Z], dq[:, Z:], dz, eps, sd, sp, sr, beta=1.0, **kw)
        R.reparam_bwd_seg(rq[:, :Z], rq[:, Z:], dz, eps, sd, sp, sr, beta=1.0, **kw)
        close(dq, rq, rtol=1e-5, atol=1e-5)


def test_prior_kl_gradient_rides_on_the_gradient_row_launches(K, dev):
    """dv_prior_kl (round 5): the gradient of coef * max(KL(q || N(0,I)), kl_min) added by dv_reparam_bwd_seg (rows [0, B)) and
    dv_z2f_post_bwd (the pairs' q2 rows) == a dv_kl_rows_bwd(beta = 1) launch behind them"""
    L, B, Z = 2, 150, 100
    pairs = torch.arange(B)[torch.arange(B) % 2 == 0]
    Np = len(pairs)
    Me = B + Np
    Q = rnd(dev, Me, 2 * Z, seed=1, scale=0.5)
    coef = rnd(dev, Me, seed=2)
    raw = torch.empty(Me, device=dev)
    K.kl_rows_fwd(torch.empty(Me, device=dev), raw, Q[:, :Z], Q[:, Z:], prior=(0.0, 0.0), free_bits=True, kl_min=0.0)
    kl_min = float(raw.median())
    # rows [0, B): the CSR sample backward
    seg_ptr = (torch.arange(B + 1) * L).to(torch.int32).to(dev)
    seg_rows = (torch.arange(L)[None, :] * B + torch.arange(B)[:, None]).reshape(-1).to(torch.int32).to(dev)
    dz, eps = rnd(dev, L * B, Z, seed=3), rnd(dev, L * B, Z, seed=4)
    ref, got = torch.zeros(Me, 2 * Z, device=dev), torch.zeros(Me, 2 * Z, device=dev)
    K.reparam_bwd_seg(ref[:B, :Z], ref[:B, Z:], dz, eps, Q[:B, Z:], seg_ptr, seg_rows)
    K.reparam_bwd_seg(got[:B, :Z], got[:B, Z:], dz, eps, Q[:B, Z:], seg_ptr, seg_rows, prior=(coef, raw, kl_min, Q[:B, :Z]))
    # the pairs' rows: the z2Fz1 backward
    slot = torch.full((B,), -1, dtype=torch.int32)
    slot[pairs] = torch.arange(Np, dtype=torch.int32)
    slot = slot.to(dev)
    p2, pert = rnd(dev, L * B, 2 * Z, seed=5, scale=0.5), rnd(dev, L * Np, Z, seed=6)
    c2, r2 = rnd(dev, L * Np, seed=7), rnd(dev, L * Np, seed=8).abs() * 40
    for out, pr in ((ref, None), (got, (coef[B:], raw[B:]))):
        K.z2f_post_bwd(torch.zeros(L * B, 2 * Z, device=dev), rnd(dev, L * B, Z, seed=9), out[B:], None, pert, slot,
                       rnd(dev, L * B, Z, seed=10), p2, Q[B:], c2, r2, kl_min, None, L, B, Np, prior=pr)
    K.kl_rows_bwd(ref[:, :Z], ref[:, Z:], None, None, coef, raw, Q[:, :Z], Q[:, Z:], prior=(0.0, 0.0), free_bits=True,
                  kl_min=kl_min, beta=1.0)
    close(got, ref.cpu(), rtol=1e-6, atol=1e-6)      # (to the rounding of a contracted multiply-add)
    # the stand-ins of the CPU suite
    rgot = torch.zeros(Me, 2 * Z, device=dev)
    R.reparam_bwd_seg(rgot[:B, :Z], rgot[:B, Z:], dz, eps, Q[:B, Z:], seg_ptr, seg_rows, prior=(coef, raw, kl_min, Q[:B, :Z]))
    close(rgot[:B], ref[:B].cpu(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('M,Z,Y,H', [(450, 100, 2, 200), (596, 37, 3, 64), (33, 128, 0, 200)])
def test_gemm_epilogue_klq_is_the_kl_row_backward(K, dev, M, Z, Y, H):
    """DV_EPI_KLQ (round 5): the data gradient of a layer whose input is [a sample of q rows | class columns] leaves the
    launch as d/d(mu | logvar) of those rows incl. their prior term == the plain data gradient + dv_kl_rows_bwd(dz, eps)
    behind it; the weight gradient of the paired launch is untouched"""
    Kin = Z + Y
    x, W = rnd(dev, M, Kin, seed=1), rnd(dev, H, Kin, seed=2, scale=Kin ** -0.5)
    dpre = rnd(dev, M, H, seed=3)
    Q, eps = rnd(dev, M, 2 * Z, seed=4, scale=0.5), rnd(dev, M, Z, seed=5)
    coef = rnd(dev, M, seed=6)
    raw = torch.empty(M, device=dev)
    K.kl_rows_fwd(torch.empty(M, device=dev), raw, Q[:, :Z], Q[:, Z:], prior=(0.0, 0.0), free_bits=True, kl_min=0.0)
    kl_min = float(raw.median())
    # reference: two launches + the row pass
    dW0, db0, dx0 = torch.empty(H, Kin, device=dev), torch.empty(H, device=dev), torch.empty(M, Kin, device=dev)
    K.linear_bwd_pair(dW0, db0, dx0, dpre, x, W, overread=False)
    ref = torch.empty(M, 2 * Z, device=dev)
    K.kl_rows_bwd(ref[:, :Z], ref[:, Z:], None, None, coef, raw, Q[:, :Z], Q[:, Z:], prior=(0.0, 0.0), free_bits=True,
                  kl_min=kl_min, dz=dx0[:, :Z], eps=eps)
    dW1, db1, got = torch.empty(H, Kin, device=dev), torch.empty(H, device=dev), torch.full((M, 2 * Z), 7.0, device=dev)
    klq = dict(out=got, q=Q, eps=eps, coef=coef, raw=raw, kl_min=kl_min, Z=Z)
    K.linear_bwd_pair(dW1, db1, None, dpre, x, W, klq=klq)
    close(got, ref.cpu(), rtol=2e-5, atol=2e-5)
    assert torch.equal(dW1, dW0) and torch.equal(db1, db0)
    rgot = torch.empty(M, 2 * Z, device=dev)
    R.linear_bwd_pair(torch.empty(H, Kin, device=dev), torch.empty(H, device=dev), None, dpre, x, W, klq=dict(klq, out=rgot))
    close(rgot, ref.cpu(), **gemm_tol(H))


def test_kl_rows_fwd_pair_is_two_launches(K, dev):
    """dv_kl_rows_fwd_pair (round 5): two independent sets of KL rows (a prior term over every row, a q || p term over
    gathered rows with L repetitions) in one launch == the two launches, bitwise"""
    n, Np, L, Z = 225, 75, 2, 100
    Q = rnd(dev, n, 2 * Z, seed=1, scale=0.5)
    P2 = rnd(dev, L * 150, 2 * Z, seed=2, scale=0.5)
    qidx = (150 + torch.arange(Np)).to(torch.int32).to(dev)
    pidx = (torch.arange(L)[:, None] * 150 + torch.arange(0, 150, 2)[None, :]).reshape(-1).to(torch.int32).to(dev)
    a = ((None, None, Q[:, :Z], Q[:, Z:]), dict(prior=(0.0, 0.0), free_bits=True, kl_min=30.0))
    b = ((None, None, Q[:, :Z], Q[:, Z:], P2[:, :Z], P2[:, Z:]), dict(qidx=qidx, pidx=pidx, reps=L, free_bits=True, kl_min=30.0))
    outs = []
    for paired in (False, True):
        o1, r1 = torch.full((n,), 7.0, device=dev), torch.full((n,), 7.0, device=dev)
        o2, r2 = torch.full((L * Np,), 7.0, device=dev), torch.full((L * Np,), 7.0, device=dev)
        fa, fb = ((o1, r1) + a[0][2:], a[1]), ((o2, r2) + b[0][2:], b[1])
        if paired:
            K.kl_rows_fwd_pair(fa, fb)
        else:
            K.kl_rows_fwd(*fa[0], **fa[1])
            K.kl_rows_fwd(*fb[0], **fb[1])
        outs.append((o1, r1, o2, r2))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize('with_pairs,with_b', [(True, True), (True, False), (False, True)])
def test_z2f_post_bwd(K, dev, with_pairs, with_b):
    L, B, Z = 2, 19, 100
    pairs = torch.arange(B)[torch.arange(B) % 3 == 1] if with_pairs else torch.zeros(0, dtype=torch.int64)
    Np = len(pairs)
    slot = torch.full((B,), -1, dtype=torch.int32)
    slot[pairs] = torch.arange(Np, dtype=torch.int32)
    slot = slot.to(dev)
    dz2f, eps = rnd(dev, L * B, Z, seed=1), rnd(dev, L * B, Z, seed=2)
    p2, q2 = rnd(dev, L * B, 2 * Z, seed=3, scale=0.5), rnd(dev, max(Np, 1), 2 * Z, seed=4, scale=0.5)
    pert = rnd(dev, max(L * Np, 1), Z, seed=5)
    coef, raw = rnd(dev, max(L * Np, 1), seed=6), rnd(dev, max(L * Np, 1), seed=7).abs() * 40
    dz1b = rnd(dev, L * B, Z, seed=8) if with_b else None
    outs = []
    for F_ in (K, R):
        dp2, dz1, dq2 = torch.zeros(L * B, 2 * Z, device=dev), rnd(dev, L * B, Z, seed=9), torch.zeros(max(Np, 1), 2 * Z,
                                                                                                 device=dev)
        F_.z2f_post_bwd(dp2, dz1, dq2[:Np] if Np else None, dz2f, pert[:L * Np] if Np else None, slot, eps, p2,
                        q2[:Np] if Np else None, coef, raw, 30.0, dz1b, L, B, Np)
        outs.append((dp2, dz1, dq2))
    for a, b in zip(*outs):
        close(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('M,N,Kd', [(450, 200, 102), (596, 1956, 600), (37, 9, 5), (5000, 300, 64)])
def test_gemm_pair_matches_two_launches(K, dev, M, N, Kd):
    """dW = dpre^T x and dx = (dpre W)*act'(yref) as one paired launch == the two single launches."""
    x, W = strided(dev, M, Kd, 2, seed=1), rnd(dev, N, Kd, seed=2, scale=Kd ** -0.5)
    dpre, yref = rnd(dev, M, N, seed=3), rnd(dev, M, Kd, seed=4)
    for beta_x in (0.0, 1.0):
        dW, db, dx = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev), rnd(dev, M, Kd, seed=5)
        rW, rb, rx = torch.empty(N, Kd, device=dev), torch.empty(N, device=dev), dx.clone()
        K.linear_bwd_pair(dW, db, dx, dpre, x, W, alpha=0.5, beta_x=beta_x, yref=yref, act='elu', overread=True)
        R.linear_bwd_pair(rW, rb, rx, dpre, x, W, alpha=0.5, beta_x=beta_x, yref=yref, act='elu')
        close(dW, rW, **gemm_tol(M))
        close(db, rb, **gemm_tol(M))
        close(dx, rx, **gemm_tol(N))


@pytest.mark.parametrize('X,pad,Np,Mf', [(978, 2, 37, 0), (13, 0, 0, 40), (980, 0, 5, 333)])
def test_batch_feed(K, dev, X, pad, Np, Mf):
    N, B, L, Y, nb = 211, 50, 2, 3, 4
    x1, x2 = strided(dev, N, X, pad, seed=1), strided(dev, N, X, pad, seed=2)
    y32 = torch.randint(0, Y, (N,), dtype=torch.int32).to(dev)
    table = torch.randint(0, N, (nb, B), dtype=torch.int32).to(dev)
    pair_rows = torch.randperm(B)[:Np].sort().values.to(torch.int32).to(dev) if Np else None
    noise = rnd(dev, B + Np, X, seed=3)
    has_y = (torch.arange(B) % 3 != 0).to(torch.int32).to(dev)
    fp_i = torch.randint(0, B, (Mf,), dtype=torch.int32).to(dev)
    fp_lab = has_y[fp_i.long()].contiguous()
    fp_slot = torch.randint(0, Y, (Mf,), dtype=torch.int32).to(dev)
    for step, base in ((7, 5), (3, 3), (9, 2), (1, 4)):      # batch 2, 0, clamped to 3, clamped to 0
        ctr = torch.tensor([step], dtype=torch.int32, device=dev)
        bs = torch.tensor([base], dtype=torch.int32, device=dev)
        outs = []
        for mod in (K, R):
            xin = strided(dev, B + Np, X, 4 - X % 4 if X % 4 else 0, seed=9).clone()
            label_r = torch.full((L * B,), -1, dtype=torch.int32, device=dev)
            fp_cls = torch.full((Mf,), -1, dtype=torch.int32, device=dev)
            onehot = torch.full((Mf, Y + 1), 7.0, device=dev)[:, :Y]
            mod.batch_feed(xin, x1, x2, y32, table, nb, ctr, bs, pair_rows=pair_rows, noise=noise, sigma=0.01,
                           has_y=has_y, L=L, label_r=label_r, fp_i=fp_i if Mf else None,
                           fp_lab=fp_lab if Mf else None, fp_slot=fp_slot if Mf else None,
                           fp_cls=fp_cls if Mf else None, onehot=onehot if Mf else None, n_classes=Y)
            outs.append((xin, label_r, fp_cls, onehot))
        close(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-6)
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert torch.equal(a, b)
    # regression targets (type_y='cont'): ylab[i,:] = yf[row of slot i,:]
    yf = torch.rand(N, 2).to(dev)
    yl, ryl = torch.empty(B, 2, device=dev), torch.empty(B, 2, device=dev)
    xin, rin = torch.empty(B + Np, X, device=dev), torch.empty(B + Np, X, device=dev)
    K.batch_feed(xin, x1, x2, None, table, nb, ctr, bs, pair_rows=pair_rows, yf=yf, ylab=yl)
    R.batch_feed(rin, x1, x2, None, table, nb, ctr, bs, pair_rows=pair_rows, yf=yf, ylab=ryl)
    assert torch.equal(yl, ryl) and torch.equal(xin, rin)
    # no noise / no labels (PVAE-style call)
    xin, rin = torch.empty(B + Np, X, device=dev), torch.empty(B + Np, X, device=dev)
    K.batch_feed(xin, x1, x2, None, table, nb, ctr, bs, pair_rows=pair_rows)
    R.batch_feed(rin, x1, x2, None, table, nb, ctr, bs, pair_rows=pair_rows)
    assert torch.equal(xin, rin)


@pytest.mark.parametrize('n1,n2,Z,Rr', [(6, 9, 5, 500), (75, 75, 100, 500), (300, 33, 128, 130)])
def test_mmd_rff(K, dev, n1, n2, Z, Rr):
    from drvae_amd import ops
    x1, x2 = strided(dev, n1, Z, 3, seed=1), rnd(dev, n2, Z, seed=2) * 0.5 + 0.3
    W, b = rnd(dev, Z, Rr, seed=3), torch.rand(Rr).to(dev)
    a, c = float(np.sqrt(2.0 / 2.0) / np.sqrt(Z)), float(np.sqrt(2.0 / Rr))
    # row kernels against their stand-ins
    th1, th2 = a * (x1 @ W) + 2 * np.pi * b, a * (x2 @ W) + 2 * np.pi * b
    out = []
    for mod in (K, R):
        diff, m2 = torch.empty(Rr, device=dev), torch.empty(1, device=dev)
        mod.mmd_rff_fwd(diff, m2, th1, th2, c)
        G = torch.empty(n1, Rr, device=dev)
        mod.mmd_rff_bwd(G, th1, diff, torch.tensor([0.7], device=dev), 2.0 * c / n1)
        out.append((diff, m2, G))
    for u, v in zip(*out):
        close(u, v, rtol=2e-4, atol=2e-6)
    # the whole op (GEMM projections + row kernels, forward and backward) against autograd on the formula
    xa, xb = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    m_hip = ops.MMDRff.apply(xa, xb, W, b, a, c)
    (0.7 * m_hip).backward()
    ga, gb = xa.grad.clone(), xb.grad.clone()
    xa.grad = xb.grad = None
    rf = lambda x: c * torch.cos(a * (x @ W) + 2 * np.pi * b[None, :])
    m = ((rf(xa).mean(0) - rf(xb).mean(0)) ** 2).sum()
    (0.7 * m).backward()
    close(m_hip.detach().reshape(1), m.detach().reshape(1), rtol=5e-4, atol=1e-7)
    close(ga, xa.grad, rtol=5e-3, atol=5e-6)
    close(gb, xb.grad, rtol=5e-3, atol=5e-6)


@pytest.mark.parametrize('M,S,Kd', [(7, 5, 13), (224, 100, 800), (300, 100, 100), (65, 33, 70), (596, 978, 600)])
def test_linear_heads_sample(K, dev, M, S, Kd):
    """dv_gemm_heads / DV_HEADS_SAMPLE: (mu | logvar) + the reparameterised samples of every source row through a
    CSR fan-out (several draws per row, some rows none) == heads GEMM followed by reparam_fwd; and the
    identity fan-out with residual, ``out2 = z - sub`` and the scattered copy ``out3``"""
    x, W, b = rnd(dev, M, Kd, seed=1), rnd(dev, 2 * S, Kd, seed=2, scale=Kd ** -0.5), rnd(dev, 2 * S, seed=3)
    sc = rnd(dev, 2 * S, seed=4).abs() + 0.5
    n_src = max(1, (2 * M) // 3)
    cnt = [(i % 3) + (1 if i % 5 == 0 else 0) for i in range(n_src)]          # 0..3 draws per source row
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(cnt)]), dtype=torch.int32, device=dev)
    R_ = int(ptr[-1])
    rows_ = torch.randperm(R_, generator=torch.Generator().manual_seed(5)).to(torch.int32).to(dev)
    eps = strided(dev, R_, S, 3, seed=6)
    for kw in (dict(shift1=-2.0), dict(scale=sc, shift1=-2.0)):
        outs = []
        for L in (K, R):
            q, z = torch.full((M, 2 * S), 7.0, device=dev), torch.full((R_, S), 9.0, device=dev)
            L.linear_heads(q, x, W, b, split=S, sample=dict(eps=eps, out=z, n_src=n_src, seg_ptr=ptr, seg_rows=rows_), **kw)
            outs.append((q, z))
        close(outs[0][0], outs[1][0], **gemm_tol(Kd))
        close(outs[0][1], outs[1][1], rtol=5e-4, atol=5e-5 * max(1.0, Kd ** 0.5))
    if Kd == S:      # DiagGaussianModuleLinear shape: mu = x + x W^T + b (src/blocks.py:357)
        sub = rnd(dev, M, S, seed=7)
        idx3 = torch.tensor([(i // 2 if i % 2 == 0 else -1) for i in range(M)], dtype=torch.int32, device=dev)
        outs = []
        cnt4 = [(i % 3) for i in range(M)]                       # 0..2 extra copies of sample row i (out4: CSR fan-out)
        ptr4 = torch.tensor(np.concatenate([[0], np.cumsum(cnt4)]), dtype=torch.int32, device=dev)
        for L in (K, R):
            q, z, d, o3 = (torch.full(s_, 7.0, device=dev) for s_ in ((M, 2 * S), (M, S), (M, S), ((M + 1) // 2, S)))
            o4 = torch.full((int(ptr4[-1]), S + 3), 7.0, device=dev)
            L.linear_heads(q, x, W, b, split=S, shift1=-2.0, resid=x, resid_cols=S,
                           sample=dict(eps=eps[:M] if R_ >= M else rnd(dev, M, S, seed=8), out=z, n_src=M, sub=sub, out2=d,
                                       out3=o3, out3_idx=idx3, out4=o4[:, :S], out4_ptr=ptr4))
            outs.append((q, z, d, o3, o4))
        for a, b_ in zip(*outs):
            close(a, b_, rtol=5e-4, atol=5e-5 * max(1.0, Kd ** 0.5))


@pytest.mark.parametrize('M,S,Kd', [(7, 5, 13), (65, 33, 70), (596, 978, 600), (150, 64, 128)])
def test_linear_heads_nll(K, dev, M, S, Kd):
    """dv_gemm_heads / DV_HEADS_NLL: gradients w.r.t. (mu | pre-softplus) and per-tile partial row sums of the Gaussian
    log-likelihood == heads GEMM + nll_rows_fwdbwd, x rows addressed through an index list"""
    x, W, b = rnd(dev, M, Kd, seed=1), rnd(dev, 2 * S, Kd, seed=2, scale=Kd ** -0.5), rnd(dev, 2 * S, seed=3)
    nx = max(1, M // 2)
    xt = strided(dev, nx, S, 2, seed=4)
    xidx = torch.tensor([i % nx for i in range(M)], dtype=torch.int32, device=dev)
    coef = rnd(dev, M, seed=5)
    nt = K.heads_tiles(S)
    got, ref = torch.full((M, 2 * S), 7.0, device=dev), torch.zeros(M, 2 * S, device=dev)
    pg, pr = torch.full((M, nt), 3.0, device=dev), torch.zeros(M, nt, device=dev)
    kw = dict(split=S, act0='identity', act1='softplus', shift1=1e-3)
    K.linear_heads(got, x, W, b, nll=dict(x=xt, xidx=xidx, coef=coef, part=pg), **kw)
    R.linear_heads(ref, x, W, b, nll=dict(x=xt, xidx=xidx, coef=coef, part=pr), **kw)
    tol = gemm_tol(Kd)
    # the gradients divide by std^2..std^3 (std >= 1e-3 + softplus): compare relative to their scale
    scale_ = float(ref.abs().max())
    close(got / scale_, ref / scale_, rtol=2e-3, atol=2e-5 * max(1.0, Kd ** 0.5))
    close(pg.sum(1), pr.sum(1), rtol=2e-4, atol=1e-3 * S ** 0.5)
    assert bool(torch.isfinite(got).all()) and bool(torch.isfinite(pg).all())


def test_batch_masks_and_labeled_slots(K, dev):
    """dv_batch_masks against its reference (explicit flags and through an epoch table), the labeled-slot branch of
    dv_ymarg_* (label <= -2), and row-weighted loss terms"""
    B, L, Y = 37, 3, 3
    g = torch.Generator().manual_seed(0)
    hx = (torch.rand(200, generator=g) < 0.4).to(torch.int32).to(dev)
    hy = (torch.rand(200, generator=g) < 0.6).to(torch.int32).to(dev)
    y = torch.randint(0, Y, (200,), generator=g).to(torch.int32).to(dev)
    table = torch.randint(0, 200, (5, B), generator=g).to(torch.int32).to(dev)
    ctr, base = torch.tensor([9], dtype=torch.int32, device=dev), torch.tensor([7], dtype=torch.int32, device=dev)
    beta = torch.tensor([0.01], device=dev)
    outs = []
    for Lb in (K, R):
        for tab in (None, table):
            bufs = dict(c_nll=torch.full((3 * L * B,), 9.0, device=dev), c_klz2=torch.full((L * B,), 9.0, device=dev),
                        c_yl=torch.full((L * B,), 9.0, device=dev), w_recl=torch.full((2 * L * B,), 9.0, device=dev),
                        w_pert=torch.full((L * B,), 9.0, device=dev), w_yl=torch.full((L * B,), 9.0, device=dev),
                        label=torch.full((L * B,), 9, dtype=torch.int32, device=dev), c_klp=torch.full((2 * B,), 9.0, device=dev))
            Lb.batch_masks(B, L, n_tot=float(B), kl_rate=0.7, pert_rate=0.05, yl_rate=1.3, beta=beta, hx=hx, hy=hy, y=y,
                           table=tab, n_batches=5 if tab is not None else 0, ctr=ctr if tab is not None else None,
                           base=base if tab is not None else None, **bufs)
            outs.append(bufs)
    for a, b in zip(outs[:2], outs[2:]):
        for k in a:
            close(a[k].float(), b[k].float(), rtol=1e-6, atol=1e-9)
    assert int((outs[1]['label'] <= -2).sum()) == int(hy[table[2].long()].sum()) * L
    # pair slots for the first Np rows only (pairs-first feeds), standalone and riding on the feed's launch
    Np, X = 16, 12
    x1, x2 = rnd(dev, 200, X, seed=11), rnd(dev, 200, X, seed=12)
    pair_rows = torch.arange(Np, dtype=torch.int32, device=dev)
    got = []
    for Lb in (K, R, 'feed'):
        bufs = dict(c_nll=torch.full((L * (B + 2 * Np),), 9.0, device=dev), c_klz2=torch.full((L * Np,), 9.0, device=dev),
                    c_yl=torch.full((L * B,), 9.0, device=dev), w_recl=torch.full((2 * L * B,), 9.0, device=dev),
                    w_pert=torch.full((L * B,), 9.0, device=dev), w_yl=torch.full((L * B,), 9.0, device=dev),
                    label=torch.full((L * B,), 9, dtype=torch.int32, device=dev), c_klp=torch.full((2 * B,), 9.0, device=dev))
        kw = dict(n_tot=float(B), kl_rate=0.7, pert_rate=0.05, yl_rate=1.3, beta=beta, hx=hx, hy=hy, y=y, Np=Np, **bufs)
        if Lb == 'feed':
            xin, rin = torch.zeros(B + Np, X, device=dev), torch.zeros(B + Np, X, device=dev)
            K.batch_feed(xin, x1, x2, y, table, 5, ctr, base, pair_rows=pair_rows, L=L, masks=kw)
            R.batch_feed(rin, x1, x2, y, table, 5, ctr, base, pair_rows=pair_rows, L=L)
            assert torch.equal(xin, rin)
        else:
            Lb.batch_masks(B, L, table=table, n_batches=5, ctr=ctr, base=base, **kw)
        got.append(bufs)
    for k in got[0]:
        close(got[0][k].float(), got[1][k].float(), rtol=1e-6, atol=1e-9)
        assert torch.equal(got[0][k], got[2][k]), k
    px = hx[table[2].long()][:Np].bool()
    assert torch.equal(got[0]['c_klz2'].reshape(L, Np) != 0, px.expand(L, Np))
    assert (got[0]['w_recl'][L * B + L * Np:] == 9.0).all() and (got[0]['c_klp'][B + Np:] == 9.0).all()
    # ymarg with materialised class slots
    Rr = L * B
    qy = torch.softmax(rnd(dev, Rr, Y, seed=1), 1)
    fp_ptr = torch.arange(0, Rr * Y + 1, Y, dtype=torch.int32, device=dev)
    klfp, c_kld, c_yl = rnd(dev, Rr * Y, seed=2).abs(), rnd(dev, Rr, seed=3), outs[0]['c_yl']
    label = outs[0]['label']
    res = []
    for Lb in (K, R):
        yl, kld, cfp, dqy = (torch.zeros(Rr, device=dev), torch.zeros(Rr, device=dev), torch.zeros(Rr * Y, device=dev),
                             torch.zeros(Rr, Y, device=dev))
        Lb.ymarg_fwdbwd(yl, kld, cfp, dqy, qy, label, fp_ptr, klfp, float(np.log(1.0 / Y)), c_kld, c_yl)
        yl2, kld2, cfp2, dqy2 = torch.zeros_like(yl), torch.zeros_like(kld), torch.zeros_like(cfp), torch.zeros_like(dqy)
        Lb.ymarg_fwd(yl2, kld2, qy, label, fp_ptr, klfp, float(np.log(1.0 / Y)))
        Lb.ymarg_bwd(cfp2, dqy2, qy, label, fp_ptr, klfp, float(np.log(1.0 / Y)), c_kld, c_yl)
        torch.cuda.synchronize()
        assert torch.equal(yl, yl2) and torch.equal(kld, kld2) and torch.equal(cfp, cfp2) and torch.equal(dqy, dqy2)
        res.append((yl, kld, cfp, dqy))
    for a, b in zip(*res):
        close(a, b, rtol=1e-5, atol=1e-6)
    # loss terms with one weight per row of a (rows, row_len) array
    x2d, wrow = rnd(dev, 50, 7, seed=4), rnd(dev, 50, seed=5)
    w_elbo, w_cmpl = torch.tensor([1.0, -1.0, 0.0], device=dev), torch.zeros(8, device=dev)
    la, lb = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    K.loss_assemble(la, [(x2d, wrow, 0.5, 0, 7)], w_elbo, w_cmpl)
    R.loss_assemble(lb, [(x2d, wrow, 0.5, 0, 7)], w_elbo, w_cmpl)
    close(la, lb, rtol=1e-5, atol=1e-5)


def test_batch_masks_global_counts(K, dev):
    """ABI 11, data parallelism: with ``gcounts`` the normalisers N_pairs / N_labeled are the GLOBAL batch's, read from
    the table's per-batch counts (explicit batch: one pair) instead of being counted over this rank's rows -- standalone,
    and riding on the feed's launch"""
    B, L, Y, nb = 24, 2, 2, 4
    g = torch.Generator().manual_seed(3)
    hx = (torch.rand(90, generator=g) < 0.4).to(torch.int32).to(dev)
    hy = (torch.rand(90, generator=g) < 0.6).to(torch.int32).to(dev)
    y = torch.randint(0, Y, (90,), generator=g).to(torch.int32).to(dev)
    gtab = torch.randint(0, 90, (nb, 3 * B), generator=g).to(torch.int32).to(dev)       # three ranks' columns
    table = gtab[:, B:2 * B].contiguous()                                                  # rank 1's
    gc = torch.stack([hx[gtab.long()].sum(1), hy[gtab.long()].sum(1)], 1).to(torch.int32).contiguous()
    ctr, base = torch.tensor([12], dtype=torch.int32, device=dev), torch.tensor([10], dtype=torch.int32, device=dev)
    beta = torch.tensor([1.0], device=dev)
    x1, x2 = rnd(dev, 90, 8, seed=1), rnd(dev, 90, 8, seed=2)
    pair_rows = torch.arange(B, dtype=torch.int32, device=dev)
    got = []
    for how in ('kernel', 'ref', 'feed', 'explicit', 'explicit_ref'):
        bufs = dict(c_nll=torch.full((3 * L * B,), 9.0, device=dev), c_klz2=torch.full((L * B,), 9.0, device=dev),
                    c_yl=torch.full((L * B,), 9.0, device=dev), w_recl=torch.full((2 * L * B,), 9.0, device=dev),
                    w_pert=torch.full((L * B,), 9.0, device=dev), w_yl=torch.full((L * B,), 9.0, device=dev),
                    label=torch.full((L * B,), 9, dtype=torch.int32, device=dev))
        kw = dict(n_tot=float(3 * B), kl_rate=0.7, pert_rate=0.05, yl_rate=1.3, beta=beta, **bufs)
        if how == 'feed':
            xin = torch.zeros(2 * B, 8, device=dev)
            K.batch_feed(xin, x1, x2, y, table, nb, ctr, base, pair_rows=pair_rows, L=L,
                         masks=dict(hx=hx, hy=hy, y=y, gcounts=gc, **kw))
        elif how.startswith('explicit'):
            tb = table[2].long()
            (K if how == 'explicit' else R).batch_masks(B, L, hx=hx[tb].contiguous(), hy=hy[tb].contiguous(),
                                                        y=y[tb].contiguous(), gcounts=gc[2:3].contiguous(), **kw)
        else:
            (K if how == 'kernel' else R).batch_masks(B, L, table=table, n_batches=nb, ctr=ctr, base=base, hx=hx, hy=hy, y=y,
                                                      gcounts=gc, **kw)
        got.append(bufs)
    for k in got[0]:
        for other in got[1:]:
            close(got[0][k].float(), other[k].float(), rtol=1e-6, atol=1e-9)
    n_p, n_l = float(gc[2, 0]), float(gc[2, 1])
    assert n_p > float(hx[table[2].long()].sum()) and n_l > float(hy[table[2].long()].sum())      # really the global ones
    nz = got[0]['w_pert'][got[0]['w_pert'] != 0]
    assert nz.numel() and torch.allclose(nz, torch.full_like(nz, 1.0 / (L * n_p)))
    assert torch.allclose(got[0]['w_yl'], torch.full_like(got[0]['w_yl'], 1.0 / (L * n_l)))
    assert torch.allclose(got[0]['w_recl'][:L * B], torch.full((L * B,), 1.0 / (L * 3 * B), device=dev))


@pytest.mark.parametrize('M,X,pad', [(130, 2052, 0), (64, 1024, 4), (7, 4, 0), (257, 4100, 8)])
def test_nll_rows_raw_pass_with_the_bias_gradient_folded_in(K, dev, M, X, pad):
    """dv_gauss_nll_rows_raw_cs (round 5): the raw-heads NLL forward + backward pass whose workgroups also keep the column
    sums of the gradients they write -- row partials per gene chunk, column sums per row block -- against the host
    reference and against the unfused pass (same gradients, row sums and bias gradient to rounding)"""
    g = torch.Generator().manual_seed(M + X)
    n_src = max(M // 3, 1)
    x = strided(dev, n_src, X, pad, seed=1)
    xidx = torch.randint(0, n_src, (M,), generator=g).to(torch.int32).to(dev)
    raw = rnd(dev, M, 2 * X + pad, seed=2, scale=0.5)      # (softplus(raw + bias) stays away from 0: a well-conditioned 1 / sd^2)
    mu, sd = raw[:, :X], raw[:, X:2 * X]
    bias = rnd(dev, 2 * X, seed=3, scale=0.3)
    coef = rnd(dev, M, seed=4, scale=0.05)
    chunks, rbs = K.nll_raw_cs_shape(M, X)
    assert (chunks, rbs) == R.nll_raw_cs_shape(M, X)
    outs = []
    for Lb in (K, R):
        dpre = torch.full((M, 2 * X + pad), 7.0, device=dev)
        part, ws = torch.full((M, chunks), 7.0, device=dev), torch.full((rbs, 2 * X), 7.0, device=dev)
        Lb.nll_rows_raw_cs(part, dpre[:, :X], dpre[:, X:2 * X], ws, coef, x, mu, sd, (bias[:X], bias[X:]), xidx=xidx, sd_shift=1e-3)
        outs.append((part, dpre, ws))
    (pk, dk, wk), (pr, dr, wr) = outs
    close(dk[:, :2 * X], dr[:, :2 * X], rtol=3e-4, atol=3e-5)
    close(pk, pr, rtol=2e-4, atol=2e-3)
    close(wk, wr, rtol=3e-4, atol=3e-4)
    if pad:
        assert bool((dk[:, 2 * X:] == 7.0).all())
    # against the unfused pass of the same library
    full, d2 = torch.zeros(M, device=dev), torch.zeros(M, 2 * X + pad, device=dev)
    K.nll_rows_fwdbwd(full, d2[:, :X], d2[:, X:2 * X], coef, x, mu, sd, mode=1, xidx=xidx, sd_act='softplus', sd_shift=1e-3,
                      bias=(bias[:X], bias[X:]))
    close(d2[:, :2 * X], dk[:, :2 * X], rtol=2e-6, atol=1e-7)      # (same element function; the compiler contracts it per kernel)
    close(pk.sum(1), full, rtol=2e-5, atol=2e-3)
    db = torch.zeros(2 * X, device=dev)
    K.colsum(db, wk)
    close(db, d2[:, :2 * X].double().sum(0).float(), rtol=2e-4, atol=2e-4)
    # reproducible bit for bit
    part2, ws2, d3 = torch.zeros_like(pk), torch.zeros_like(wk), torch.zeros_like(dk)
    K.nll_rows_raw_cs(part2, d3[:, :X], d3[:, X:2 * X], ws2, coef, x, mu, sd, (bias[:X], bias[X:]), xidx=xidx, sd_shift=1e-3)
    assert torch.equal(part2, pk) and torch.equal(ws2, wk)


@pytest.mark.parametrize('a_kc,b_kc', [(True, True), (True, False), (False, False)])
def test_gemm_ragged_last_tile_column_runs_as_two_launches(K, dev, a_kc, b_kc):
    """round 5: a chip-filling plain product whose N leaves a narrow last column of 128x256 tiles that costs every CU one
    more tile (here 32 x 33 tiles = 4.1 per CU -> 5; 32 x 32 = 4) runs the full tiles and the narrow rest as two launches
    (dv_gemm; ``dv_gemm_tune.opt[6] = -1``: one launch) -- same product either way, against the host reference"""
    M, N, Kd = 4096, 32 * 256 + 64, 64
    A = rnd(dev, *((M, Kd) if a_kc else (Kd, M)), seed=1)
    B = rnd(dev, *((N, Kd) if b_kc else (Kd, N)), seed=2)
    ref = ((A if a_kc else A.t()).cpu() @ (B.t() if b_kc else B).cpu())
    outs = []
    for opt in (0, -1):
        K.gemm_set_option(6, opt)
        try:
            Cm = torch.full((M, N + 4), 7.0, device=dev)[:, :N]
            K.gemm(Cm, A, B, a_kc, b_kc, overread=False)
            close(Cm, ref, **gemm_tol(Kd))
            outs.append(Cm)
        finally:
            K.gemm_set_option(6, 0)
    assert torch.equal(outs[0][:, :32 * 256], outs[1][:, :32 * 256])       # the full tiles: the same kernel, same tiles
    # with accumulate / scaling (beta C + alpha A B): the split carries them to both parts
    C0 = rnd(dev, M, N, seed=3)
    C1 = C0.clone()
    K.gemm(C1, A, B, a_kc, b_kc, alpha=0.5, beta=2.0)
    close(C1, 2.0 * C0.cpu() + 0.5 * ref, **gemm_tol(Kd))


@pytest.mark.parametrize('n,Y,ties', [(3000, 2, True), (517, 2, False), (900, 3, True), (1, 2, False), (40, 2, True)])
def test_rank_metrics_without_a_sort(K, dev, n, Y, ties):
    """dv_rank_metrics (round 5): ROC-AUC / average precision / accuracy of the labeled rows from pair counts, against
    scikit-learn (the reference's own choice, src/DGMMixin.py:163-180) and the sort-based formulation -- tie groups, one
    class only, no rows"""
    from sklearn.metrics import average_precision_score, roc_auc_score
    from drvae_amd import metrics as MET
    g = torch.Generator().manual_seed(n + Y)
    N_all = n + 50
    proba = torch.softmax(torch.randn(N_all, Y, generator=g) * 2.0, 1)
    if ties:
        proba = (proba * 20).round() / 20 + 1e-3          # many tied scores
    y = torch.randint(0, Y, (N_all,), generator=g)
    pred = proba.argmax(1)
    sel = torch.randperm(N_all, generator=g)[:n].sort().values
    pd, yd, prd, sd = proba.to(dev).contiguous(), y.to(torch.int32).to(dev), pred.to(torch.int32).to(dev), sel.to(torch.int32).to(dev)
    n_cls = 1 if Y == 2 else Y
    counts = torch.zeros(n_cls, n, 4, dtype=torch.int32, device=dev)
    out = torch.zeros(2 * n_cls + 1, dtype=torch.float64, device=dev)
    for rep in range(2):          # (second call: the counts came back zeroed)
        K.rank_metrics(out, counts, pd, yd, pred32=prd, sel=sd, c0=1 if Y == 2 else 0, n_cls=n_cls, binary=Y == 2)
        torch.cuda.synchronize()
        assert int(counts.abs().sum()) == 0
        o = out.cpu().numpy()
        ys, ps = y[sel].numpy(), proba[sel].numpy()
        for c in range(n_cls):
            cls = 1 if Y == 2 else c
            pos = (ys > 0) if Y == 2 else (ys == cls)
            if pos.all() or not pos.any():
                assert np.isnan(o[2 * c]) and (o[2 * c + 1] == 0.0 if not pos.any() else True)
                continue
            assert abs(o[2 * c] - roc_auc_score(pos, ps[:, cls])) < 1e-12
            assert abs(o[2 * c + 1] - average_precision_score(pos, ps[:, cls])) < 1e-12
            assert o[2 * c] == MET.roc_auc(torch.from_numpy(pos), torch.from_numpy(ps[:, cls]))      # integer arithmetic: exact
        assert abs(o[2 * n_cls] - float((pred[sel] == y[sel]).float().mean())) < 1e-7
    # degenerate: only positives / only negatives / no rows
    for yy in (torch.ones(N_all, dtype=torch.int32), torch.zeros(N_all, dtype=torch.int32)):
        K.rank_metrics(out, counts, pd, yy.to(dev), pred32=prd, sel=sd, c0=1, n_cls=1, binary=True)
        o = out.cpu().numpy()
        assert np.isnan(o[0]) and (o[1] == 0.0 if int(yy[0]) == 0 else o[1] > 0.99)
    K.rank_metrics(out, counts, pd, yd, pred32=prd, sel=sd[:0], c0=1, n_cls=1, binary=True)
    o = out.cpu().numpy()
    assert np.isnan(o[0]) and o[1] == 0.0 and np.isnan(o[2])


@pytest.mark.parametrize('M,X,with_sel', [(300, 978, False), (300, 978, True), (70, 13, True)])
def test_recon_finalize(K, dev, M, X, with_sel):
    """dv_recon_finalize + dv_col_moments(sel): the float64 combination of the reconstruction partials over a row
    subset == the host formulas of eval_x_reconstruction (src/DGMMixin.py:128-156) on the gathered rows"""
    x, r, sd = rnd(dev, M, X, seed=1), rnd(dev, M, X, seed=2), rnd(dev, M, X, seed=3).abs() + 0.1
    sel = torch.arange(0, M, 3, dtype=torch.int32, device=dev) if with_sel else None
    n = sel.numel() if with_sel else M
    rows, ll = torch.empty(M, 6, device=dev), torch.empty(M, device=dev)
    part = torch.empty(K.col_moment_blocks(n), 3, X, dtype=torch.float64, device=dev)
    out = torch.zeros(4, dtype=torch.float64, device=dev)
    K.recon_row_stats(rows, x, r)
    K.col_moments(None, x, r, sel=sel, part=part)
    K.nll_rows_fwd(ll, x, r, sd, mode=1)
    K.recon_finalize(out, rows, part, X, sel=sel, n=n, ll=ll)
    xs, rs, ss = (t[sel.long()] if with_sel else t for t in (x, r, sd))
    xs, rs, ss = xs.double().cpu(), rs.double().cpu(), ss.double().cpu()
    rmse = float(torch.sqrt(((xs - rs) ** 2).mean()))
    r2 = float(1.0 - ((xs - rs) ** 2).sum() / ((xs - xs.mean(0)) ** 2).sum())
    xc, rc = xs - xs.mean(1, keepdim=True), rs - rs.mean(1, keepdim=True)
    pear = float(((xc * rc).sum(1) / torch.sqrt((xc ** 2).sum(1) * (rc ** 2).sum(1))).mean())
    llm = float((-0.5 * (np.log(2 * np.pi) + 2 * torch.log(ss) + ((xs - rs) / ss) ** 2)).sum(1).mean())
    o = out.cpu().numpy()
    np.testing.assert_allclose(o, [rmse, r2, pear, llm], rtol=2e-6, atol=1e-6)
    # the partial-free form (out given) still adds the blocks up itself
    cols = torch.empty(3, X, dtype=torch.float64, device=dev)
    K.col_moments(cols, x, r, sel=sel)
    close(cols, part.sum(0), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize('M,X', [(300, 978), (2048, 978), (70, 13), (129, 1024), (5, 64), (40000, 100)])
def test_recon_rows_is_the_two_row_passes(K, dev, M, X):
    """dv_recon_rows (round 5): row statistics and log-likelihood rows of a reconstruction in ONE pass ==
    dv_recon_row_stats (bitwise where the rows take the dword path -- odd widths: same summation order; rows of even width
    are read with 8-B loads, a lane then owns column pairs) + dv_gauss_nll_rows_fwd; with raw heads (bias, softplus + shift
    finished on the way) next to dv_col_moments(r_bias)"""
    import torch.nn.functional as F_

    def same(a, b):
        if X % 2:
            assert torch.equal(a, b)
        else:
            close(a, b.cpu(), rtol=1e-5, atol=2e-4)
    x, r, sd = rnd(dev, M, X, seed=1), rnd(dev, M, X, seed=2), rnd(dev, M, X, seed=3).abs() + 0.1
    rows0, ll0 = torch.empty(M, 6, device=dev), torch.empty(M, device=dev)
    K.recon_row_stats(rows0, x, r)
    K.nll_rows_fwd(ll0, x, r, sd, mode=1)
    rows1, ll1 = torch.full((M, 6), 7.0, device=dev), torch.full((M,), 7.0, device=dev)
    K.recon_rows(rows1, ll1, x, r, sd)
    same(rows1, rows0)
    close(ll1, ll0.cpu(), rtol=2e-6, atol=1e-5 * max(1.0, X / 100))       # (16-B aligned rows take nll_rows_fwd's vector path: another order)
    # raw heads
    bm, bs = rnd(dev, X, seed=4), rnd(dev, X, seed=5, scale=0.3)
    raw_s = rnd(dev, M, X, seed=6, scale=0.5)        # (sigma >= ~0.2: a tiny sigma makes the row sum ill-conditioned)
    mu_f = r + bm
    sd_f = F_.softplus(raw_s + bs) + 1e-3
    K.recon_row_stats(rows0, x, mu_f)
    K.nll_rows_fwd(ll0, x, mu_f, sd_f, mode=1)
    K.recon_rows(rows1, ll1, x, r, raw_s, bias=(bm, bs), sd_shift=1e-3)
    same(rows1, rows0)
    close(ll1, ll0.cpu(), rtol=5e-5, atol=2e-4 * max(1.0, X / 100))     # (hardware transcendentals in the raw term)
    K.recon_rows(rows1, None, x, r, raw_s, bias=(bm, bs), sd_shift=1e-3)      # (no log-likelihood rows wanted)
    same(rows1, rows0)
    # the column moments of a raw product: its bias added on the way, over a row subset
    sel = torch.arange(0, M, 3, dtype=torch.int32, device=dev)
    c0, c1 = torch.empty(3, X, dtype=torch.float64, device=dev), torch.empty(3, X, dtype=torch.float64, device=dev)
    K.col_moments(c0, x, mu_f, sel=sel)
    K.col_moments(c1, x, r, sel=sel, r_bias=bm)
    assert torch.equal(c0, c1)
    # rows beyond the register-resident width are refused, not mangled
    if M == 5:
        with pytest.raises(RuntimeError):
            K.recon_rows(torch.empty(4, 6, device=dev), None, rnd(dev, 4, 1025, seed=1), rnd(dev, 4, 1025, seed=2),
                         rnd(dev, 4, 1025, seed=3).abs() + 0.1)


@pytest.mark.parametrize('M,K1,K2,N', [(4096, 200, 200, 2), (1500, 37, 0, 3), (900, 50, 50, 2)])
def test_smalln_weight_gradient_row_split(K, dev, M, K1, K2, N):
    """round 5: with a workspace the single-Linear classifier head's weight gradient splits >= 1024 rows over workgroups
    (two-stage, fixed order): same result as the one-workgroup-per-column-block form and as the host reference;
    reproducible bit for bit; below 1024 rows the workspace is ignored"""
    probs = torch.softmax(rnd(dev, M, N, seed=1), 1)
    g = rnd(dev, M, N, seed=2, scale=0.1)
    a1 = strided(dev, M, K1, 4, seed=3)
    a2 = strided(dev, M, K2, 0, seed=4) if K2 else None
    KT = K1 + K2
    ws = torch.full((K.smalln_ws_numel(N, KT),), 7.0, device=dev)
    outs = []
    for w in (None, ws, ws):
        dW, db = torch.full((N, KT), 0.5, device=dev), torch.full((N,), 0.5, device=dev)
        K.smalln_bwd_weight(dW, db, g, probs, a1, a2, beta=1.0, ws=w)
        outs.append((dW, db))
    rW, rb = torch.full((N, KT), 0.5, device=dev), torch.full((N,), 0.5, device=dev)
    R.smalln_bwd_weight(rW, rb, g, probs, a1, a2, beta=1.0)
    for dW, db in outs:
        close(dW, rW, rtol=2e-4, atol=2e-5 * M ** 0.5)
        close(db, rb, rtol=2e-4, atol=2e-5 * M ** 0.5)
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    if M < 1024:
        assert torch.equal(outs[0][0], outs[1][0]) and bool((ws == 7.0).all())


def test_gemm_ragged_split_carries_the_fused_epilogues(K, dev):
    """the [full tiles | narrow rest] split of a chip-filling product (dv_gemm) with the per-column operands of the fused
    epilogues moving with their columns: bias, WeightNorm scale, the two heads' split on either side of the cut, residual
    columns, accumulate, and the activation backward (yref)"""
    M, N, Kd = 4096, 32 * 256 + 164, 48
    x, W, b = rnd(dev, M, Kd, seed=1), rnd(dev, N, Kd, seed=2, scale=Kd ** -0.5), rnd(dev, N, seed=3)
    sc = rnd(dev, N, seed=4).abs() + 0.5
    res = rnd(dev, M, N, seed=5)
    for kw in (dict(bias=b, act0='elu', act1='elu'),
               dict(bias=b, scale=sc, split=978, act0='identity', act1='softplus', shift1=1e-3),          # split inside the full tiles
               dict(bias=b, split=32 * 256 + 100, act0='identity', act1='softplus', shift1=1e-3),      # ... inside the narrow rest
               dict(bias=b, split=N // 2, act0='identity', act1='identity', shift1=-2.0, resid=res, resid_cols=32 * 256 + 7)):
        out, ref = torch.full((M, N), 7.0, device=dev), torch.zeros(M, N, device=dev)
        K.linear_fwd(out, x, W, **kw)
        R.linear_fwd(ref, x, W, **kw)
        close(out, ref, **gemm_tol(Kd))
        K.gemm_set_option(6, -1)
        try:
            one = torch.zeros(M, N, device=dev)
            K.linear_fwd(one, x, W, **kw)
        finally:
            K.gemm_set_option(6, 0)
        assert torch.equal(one[:, :32 * 256], out[:, :32 * 256])
    # dx = (dpre W) * act'(yref), accumulated: the BWD epilogue across the cut
    dpre, Wt, y = rnd(dev, M, Kd, seed=6), rnd(dev, Kd, N, seed=7), rnd(dev, M, N, seed=8)
    dx, rx = res.clone(), res.clone()
    K.linear_bwd_data(dx, dpre, Wt, yref=y, act='elu', beta=1.0)
    R.linear_bwd_data(rx, dpre, Wt, yref=y, act='elu', beta=1.0)
    close(dx, rx, **gemm_tol(Kd))
