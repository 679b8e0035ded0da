#!/usr/bin/env python3
"""Every launcher call of one whole-set evaluation (N1), re-issued in isolation and timed: which launches the
evaluation of an 8192-row set is made of (GPU box only).  python tools/eval_shapes.py [rows]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from drvae_amd import fit as F
from tests.kernel_ref import FUNCTIONS
from tools.eval_bench import dataset
from tools.gemm_bench import time_call
from drvae_amd.DrVAE import DrVAE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=1,
              batch_size=150).to('cuda')
model.w2log = lambda *a: None
ds = dataset(n, 1)
model.evaluate_performance_on_dataset(ds)
ev = F._EvalGraph.get(model, ds)
rec = []
LEAF = [f for f in FUNCTIONS if f not in ('linear_fwd', 'linear_bwd_data', 'linear_bwd_weight')]
real = {f: getattr(K, f) for f in LEAF}


def mk(name):
    def f(*a, **kw):
        shp = tuple(tuple(t.shape) for t in a[:3] if torch.is_tensor(t))
        rec.append((name, shp, lambda: real[name](*a, **kw)))
        return real[name](*a, **kw)
    return f


for f in LEAF:
    setattr(K, f, mk(f))
t_seq = time_call(lambda: None, repeats=1)
rec.clear()
ev._sequence()
torch.cuda.synchronize()
for f in LEAF:
    setattr(K, f, real[f])
tot = 0.0
agg = collections.OrderedDict()
for name, shp, fn in rec:
    us = time_call(fn, repeats=5)
    tot += us
    k = (name, shp)
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += us
print('launcher calls of one evaluation of %d rows: %d, sum of isolated times %.1f us (torch ops not included)' % (n, len(rec), tot))
for (name, shp), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    fl = ''
    if name == 'gemm' and len(shp) == 3:
        M_, N_ = shp[0]
        Kd = shp[1][1] if shp[1][0] == M_ else shp[1][0]
        fl = '  %.1f TF/s' % (2.0 * M_ * N_ * Kd * c / us / 1e6)
    print('  %-18s x%d %-46s %9.1f us%s' % (name, c, shp, us, fl))
