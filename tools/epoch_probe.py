#!/usr/bin/env python3
"""Where one steady-state ``fit`` epoch at cfg-2 size spends its time outside the steps (round 5): bind / begin_epoch / the 54
replays / the closing sync, each bracketed by a device synchronisation."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE
from drvae_amd import data as DD
from tools.eval_bench import dataset

dev = torch.device('cuda:0')
model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=1,
              batch_size=150).to(dev)
model.w2log = lambda *a: None
model.add_noise = True
tr = dataset(8192, 1, dev)
bat = DD.DeviceBatcher(tr, torch.ones(8192), 150, seed=1)
for ep in range(3):
    model._epoch_device(bat, ep, False)
torch.cuda.synchronize()
eng = model.engine()
sync = torch.cuda.synchronize
T = {}
for rep in range(5):
    sync(); t0 = time.perf_counter()
    bat.bind(eng); sync(); t1 = time.perf_counter()
    bat.begin_epoch(); sync(); t2 = time.perf_counter()
    with eng.partition():
        sync(); t3 = time.perf_counter()
        eng.loss_sum.zero_()
        for b in range(len(bat)):
            eng.replay()
        sync(); t4 = time.perf_counter()
    sync(); t5 = time.perf_counter()
    v = float(eng.loss_sum[0]); eng.check_sync(); t6 = time.perf_counter()
    for k, v_ in (('bind', t1 - t0), ('begin_epoch', t2 - t1), ('partition enter', t3 - t2), ('54 replays', t4 - t3),
                  ('partition exit', t5 - t4), ('read + check_sync', t6 - t5)):
        T.setdefault(k, []).append(v_ * 1e3)
for k, v in T.items():
    print('%-20s %.3f ms (min of 5: %.3f)' % (k, sorted(v)[len(v) // 2], min(v)))
print('per step in the loop: %.4f ms' % (min(T['54 replays']) / len(bat)))
t0 = time.perf_counter()
for ep in range(5):
    model._epoch_device(bat, ep, False)
sync()
print('whole _epoch_device: %.3f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
# ... and with the whole-set evaluations of fit between the epochs (epochs numbered as fit numbers them: the index table of the
# next epoch is drawn ahead behind the running steps)
va = dataset(2048, 2, dev)
model.epochs = 100
for ep in range(3):
    model._epoch_device(bat, ep + 1, False); model.evaluate_performance_on_dataset(tr); model.evaluate_performance_on_dataset(va)
sync()
for label, ev in (('epochs back to back (draw-ahead)', False), ('epochs with the two evaluations in between', True)):
    tt = []
    for ep in range(6):
        sync(); t0 = time.perf_counter()
        model._epoch_device(bat, ep + 4, False)
        sync(); t1 = time.perf_counter()
        if ev:
            model.evaluate_performance_on_dataset(tr); model.evaluate_performance_on_dataset(va)
        tt.append((t1 - t0) * 1e3)
    print('%-46s _epoch_device %.3f ms (min %.3f)' % (label, sorted(tt)[len(tt) // 2], min(tt)))
