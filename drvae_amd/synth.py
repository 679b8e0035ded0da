"""Synthetic gene-expression minibatches for benchmarks/smoke: no
dataset ships with the reference (.MISSING_LARGE_BLOBS) and there is no network."""
import numpy as np


def make_batch(kind, n_rows, dim_x=978, dim_y=2, seed=1234, row0=0):
    """x1~N(0,1); x2 = x1 + 0.1 N(0,1) for paired rows and exactly 0 for singletons (the
    zero-imputation of wrap_in_DrVAEDataset, src/DrVAE.py:924); y~Bernoulli(.5)/uniform
    classes; groups by GLOBAL row index ``i mod 4`` -> ls,us,lp,up (drvae), ``i mod 2`` ->
    singleton/pair (pvae) or labeled/unlabeled (vfae).  ``row0`` offsets the global index
    (data-parallel shards)."""
    rs = np.random.RandomState(seed + 7919 * (row0 // max(n_rows, 1)))
    x1 = rs.standard_normal((n_rows, dim_x)).astype(np.float32)
    x2 = (x1 + 0.1 * rs.standard_normal((n_rows, dim_x))).astype(np.float32)
    y = rs.randint(0, dim_y, (n_rows, 1)).astype(np.int64)
    i = row0 + np.arange(n_rows)
    if kind == 'drvae':
        has_y, has_x2 = (i % 2 == 0), ((i // 2) % 2 == 1)
    elif kind == 'pvae':
        has_y, has_x2 = np.zeros(n_rows, bool), (i % 2 == 1)
    else:
        has_y, has_x2 = (i % 2 == 0), np.zeros(n_rows, bool)
    x2 = x2 * has_x2[:, None].astype(np.float32)
    return {'x1': x1, 'x2': x2, 's': np.zeros((n_rows, 1), np.int64), 'y': y,
            'has_x2': has_x2.astype(np.int64), 'has_y': has_y.astype(np.int64)}


def gemm_flops_per_step(cfg, n_rows, frac_pair, frac_lab):
    """GEMM FLOPs of one train step (fwd + bwd) by the model of SURVEY.md 8(d):
    backward = 2*forward - (dX of the data matrix)."""
    X, Y, Z, Z3, L = cfg.dim_x, cfg.dim_y, cfg.dim_z1, cfg.dim_z3, cfg.L

    def mlp(n_in, hidden, n_out_heads):
        f, w = 0, n_in
        for h in hidden:
            f += w * h
            w = h
        return 2 * (f + w * n_out_heads)

    f_enc = mlp(X, cfg.h_en_z1, 2 * Z)
    f_dec = mlp(Z, cfg.h_de_x, 2 * X)
    f_z2 = 2 * (2 * Z * Z) if cfg.has_pert else 0
    f_clf = 0
    f_fp = 0
    if cfg.has_y:
        n_in = 2 * Z if (cfg.kind == 'drvae' and cfg.clf_z1z2) else Z
        f_clf = mlp(n_in, cfg.h_clf, Y)
        f_fp = mlp(Z + Y, cfg.h_en_z3, 2 * Z3) + mlp(Z3 + Y, cfg.h_de_z1, 2 * Z)
    p = frac_pair if cfg.has_pert else 0.0
    lab = frac_lab
    fwd = f_enc * (1 + p) + L * (f_dec * (1 + 2 * p) + f_z2 + f_clf + f_fp * (lab + Y * (1 - lab)))
    first = cfg.h_en_z1[0] if cfg.h_en_z1 else 2 * Z
    bwd = 2 * fwd - 2 * X * first * (1 + p)
    return n_rows * (fwd + bwd)
