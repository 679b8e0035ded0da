#!/usr/bin/env python3
"""Wall-clock of the fit protocol at realistic scale (cfg-2 model, 8192-row training set)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE
from drvae_amd import data as D, synth

dev = 'cuda'
N, NV = int(os.environ.get('N', 8192)), 2048
def ds(n, seed):
    b = synth.make_batch('drvae', n, 978, 2, seed=seed)
    t = lambda k: torch.from_numpy(b[k]).to(dev)
    return D.DrVAEDataset(t('x1'), t('x2'), torch.zeros(n, dtype=torch.int64, device=dev), t('y'), t('has_x2'), t('has_y'))
tr, va = ds(N, 1), ds(NV, 2)
model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
              dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
              weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=6,
              batch_size=150)
logs = []
model.w2log = lambda *a: logs.append(' '.join(str(x) for x in a))
bat = D.DeviceBatcher(tr, torch.ones(N), 150, seed=1)
class VL: dataset = va
t0 = time.time()
model.fit(bat, VL(), add_noise=True, verbose=False, early_stop=True, model_filename='/tmp/best.pth')
torch.cuda.synchronize()
dt = time.time() - t0
for ln in logs:
    if ln.startswith(('Train:', 'Valid:')):
        print(ln[:150])
print('epochs %d, %d steps/epoch, total %.2f s -> %.1f ms/epoch' % (model.epochs, len(bat), dt, 1e3 * dt / model.epochs))
# steady state (the first epoch also pays for the capture and the CU-split tuning, which creates the masked streams)
eng = model.engine()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    model._epoch_device(bat, 1, False); torch.cuda.synchronize(); t1 = time.time()
    model.evaluate_performance_on_dataset(tr); torch.cuda.synchronize(); t2 = time.time()
    model.evaluate_performance_on_dataset(va); torch.cuda.synchronize(); t3 = time.time()
print('steady-state epoch: train part %.1f ms (%d steps, %.3f ms/step) | eval train set %.1f ms | eval valid set %.1f ms '
      '| total %.1f ms' % (1e3 * (t1 - t0), len(bat), 1e3 * (t1 - t0) / len(bat), 1e3 * (t2 - t1), 1e3 * (t3 - t2),
                           1e3 * (t3 - t0)))
