#!/usr/bin/env python3
"""host-side profile of steady-state ``fit`` epochs at cfg-2 size (cProfile, cumulative): what fit() spends between the
epoch's replays and the evaluations"""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE
from drvae_amd import data as DD
from tools.eval_bench import dataset
dev = torch.device('cuda:0')
model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=4,
              batch_size=150).to(dev)
model.w2log = lambda *a: None
tr, va = dataset(8192, 1, dev), dataset(2048, 2, dev)
bat = DD.DeviceBatcher(tr, torch.ones(8192), 150, seed=1)
class VL: dataset = va
model.fit(bat, VL(), add_noise=True, verbose=False, early_stop=False, model_filename='/tmp/best.pth')
model.epochs = 10
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.time()
pr.enable()
model.fit(bat, VL(), add_noise=True, verbose=False, early_stop=False, model_filename='/tmp/best.pth')
pr.disable()
torch.cuda.synchronize()
print('10 epochs: %.1f ms per epoch' % ((time.time() - t0) * 100))
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
