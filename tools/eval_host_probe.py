#!/usr/bin/env python3
"""Where a whole-set evaluation's wall time goes between the host and the device (GPU box only): the captured graph alone
(replays back to back, one sync), one replay + sync, and ``evaluate_performance_on_dataset`` (replay + the copy of the
result vector + the host-side dictionary)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd import fit as F
from tools.eval_bench import dataset
from drvae_amd.DrVAE import DrVAE

model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=1,
              batch_size=150).to('cuda')
model.w2log = lambda *a: None
for n in (8192, 2048):
    ds = dataset(n, 1)
    model.evaluate_performance_on_dataset(ds)
    ev = F._EvalGraph.get(model, ds)
    R = 50

    def timed(fn, sync_each):
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(R):
            fn()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return 1e6 * (time.time() - t0) / R
    a = timed(ev.graph.replay, False)
    b = timed(ev.graph.replay, True)
    c = timed(lambda: model.evaluate_performance_on_dataset(ds), False)
    print('%5d rows: graph back to back %.0f us | replay + sync %.0f us | evaluate_performance_on_dataset %.0f us' % (n, a, b, c))
