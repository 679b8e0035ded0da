#!/usr/bin/env python3
"""Where the training part of a ``fit`` epoch goes (cfg-2 model, 8192-row set, 54 steps): the epoch's table draw, the
replays, the closing sync -- wall clock, ms (GPU box only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE
from drvae_amd import data as DD
from tools.eval_bench import dataset

model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=1,
              batch_size=150).to('cuda')
model.w2log = lambda *a: None
model.add_noise = True
tr = dataset(8192, 1, 'cuda')
bat = DD.DeviceBatcher(tr, torch.ones(8192), 150, seed=1)
print('batcher mode:', bat.mode, 'bucketed:', bat.bucketed)
for ep in range(3):
    model._epoch_device(bat, ep, False)
eng = model.engine()
sync = torch.cuda.synchronize
def T(f, n=5):
    best = 1e9
    for _ in range(n):
        sync(); t0 = time.perf_counter(); f(); sync(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
print('whole _epoch_device          %.3f ms' % T(lambda: model._epoch_device(bat, 9, False)))
print('begin_epoch alone            %.3f ms' % T(lambda: bat.begin_epoch()))
def body():
    with eng.partition():
        model._epoch_device_body(eng, bat, 9, False)
bat.begin_epoch()
print('_epoch_device_body (54 steps) %.3f ms' % T(body))
def replays():
    with eng.partition():
        for b in range(len(bat)):
            bat.select(b)
            eng.replay()
bat.begin_epoch()
print('54 x (select + replay)       %.3f ms' % T(replays))
