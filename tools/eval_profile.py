#!/usr/bin/env python3
"""Where the whole-set evaluation of ``fit`` (N1: losses + means-only inference + metrics over the full
training set, once per epoch) spends its time (GPU box only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE
from drvae_amd import data as D, synth

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


def timed(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print('%-34s %7.2f ms' % (name, 1e3 * (time.time() - t0) / reps))
    return out


kw = dict(x1=ds.x1, x2=ds.x2, s=ds.s, y=ds.y, has_x2=ds.has_x2, has_y=ds.has_y)
timed('evaluate_performance_on_dataset', lambda: model.evaluate_performance_on_dataset(ds))
timed('  run_on_batch(eval losses)', lambda: model.run_on_batch(train_mode=False, **kw))
res = timed('  forward (means)', lambda: model.forward(ds.x1, ds.s))
yidx = torch.nonzero(ds.has_y.reshape(-1)).reshape(-1)
timed('  eval_y_prediction', lambda: model.eval_y_prediction(res['pred'][yidx], res['proba'][yidx], ds.y.reshape(-1)[yidx]))
timed('  eval_x_reconstruction x1', lambda: model.eval_x_reconstruction(ds.x1, *res['px1']))
x2idx = torch.nonzero(ds.has_x2.reshape(-1)).reshape(-1)
timed('  eval_x_reconstruction x2 (gathered)', lambda: model.eval_x_reconstruction(ds.x2[x2idx], res['px2'][0][x2idx], res['px2'][1][x2idx]))
