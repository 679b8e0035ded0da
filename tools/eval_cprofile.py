#!/usr/bin/env python3
"""Host-side profile (cProfile) of one whole-set evaluation of ``fit`` (GPU box only)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE  # noqa: E402
from drvae_amd import data as D, synth  # noqa: E402

dev = 'cuda'
N = int(os.environ.get('N', 8192))
b = synth.make_batch('drvae', N, 978, 2, seed=1)
t = lambda k: torch.from_numpy(b[k]).to(dev)
ds = D.DrVAEDataset(t('x1'), t('x2'), torch.zeros(N, dtype=torch.int64, device=dev), t('y'), t('has_x2'), t('has_y'))
model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=1,
              batch_size=150).to(dev)
model.add_noise = False
model.w2log = lambda *a: None
for _ in range(3):
    model.evaluate_performance_on_dataset(ds)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    model.evaluate_performance_on_dataset(ds)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
