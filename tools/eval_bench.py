#!/usr/bin/env python3
"""Whole-set evaluation of ``fit`` (N1) at cfg-2 size: ``evaluate_performance_on_dataset`` of an HBM-resident 8192-row
training set and a 2048-row validation set (one hipGraph replay + one copy each), ms per evaluation and the FLOP model
of its forward work.  ``python tools/eval_bench.py [reps]`` (GPU box only; under rocprofv3 --kernel-trace --stats the
kernel table says where the time goes)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd.DrVAE import DrVAE
from drvae_amd import data as D, synth


def dataset(n, seed, dev='cuda'):
    dev = str(dev)
    b = synth.make_batch('drvae', n, 978, 2, seed=seed)
    t = lambda k: torch.from_numpy(b[k]).to(dev)
    return D.DrVAEDataset(t('x1'), t('x2'), torch.zeros(n, dtype=torch.int64, device=dev), t('y'), t('has_x2'), t('has_y'))


def eval_gflop(ds, L=2, X=978, H1=800, Z=100, HD=600, HF=200, Y=2, executed=False):
    """forward FLOPs of one evaluation: the eval-mode loss pass (src/DrVAE.py:367-543 on every row, L samples) + the
    means-only inference (src/DrVAE.py:253-311)"""
    B = len(ds)
    Np = int(ds.has_x2.sum())
    Nl = int(ds.has_y.sum())
    enc = 2.0 * (X * H1 + H1 * 2 * Z)
    dec = 2.0 * (Z * HD + HD * 2 * X)
    z2f = 2.0 * (Z * 2 * Z)
    fp = 2.0 * 2 * ((Z + Y) * HF + HF * 2 * Z)
    loss = (B + Np) * enc + (L * B + 2 * L * Np) * dec + L * B * z2f + L * (Nl + Y * (B - Nl)) * fp
    # (``executed``: the inference pass re-uses q(z1|x1) of the loss pass -- its encoder product is not run a second time)
    infer = B * ((0.0 if executed else enc) + z2f + 2 * dec + 2.0 * 2 * Z * Y)
    return (loss + infer) / 1e9


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for kv in filter(None, os.environ.get('GEMM_OPTS', '').split(',')):      # tuning A/Bs: dv_gemm_tune.opt keys, e.g. GEMM_OPTS=5=-1
        import drvae_amd.kernels as K
        K.gemm_set_option(*[int(v) for v in kv.split('=')])
    model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
                  dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
                  weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=1,
                  batch_size=150).to('cuda')
    model.w2log = lambda *a: None
    out = {}
    for name, n, seed in (('train 8192', 8192, 1), ('valid 2048', 2048, 2)):
        ds = dataset(n, seed)
        model.evaluate_performance_on_dataset(ds)           # capture
        model.evaluate_performance_on_dataset(ds)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            perf, txt = model.evaluate_performance_on_dataset(ds)
        torch.cuda.synchronize()
        ms = 1e3 * (time.time() - t0) / reps
        gf = eval_gflop(ds)
        out[name] = ms
        print('%s: %.3f ms per evaluation, %.1f GFLOP of forward work -> %.1f TFLOP/s (%.3f of the fp32-MFMA peak)   [%s]'
              % (name, ms, gf, gf / ms, gf / ms / 157.3, txt[:60]), flush=True)
    print('both: %.3f ms' % sum(out.values()))


if __name__ == '__main__':
    main()
