import sys, os, torch, numpy as np
sys.path.insert(0, '/root/repo')
import drvae_amd.kernels as K
from drvae_amd import _lib
import tests.kernel_ref as R
lib = _lib.load(); K.gemm_set_option(4, int(sys.argv[1]))
dev = torch.device('cuda:0')
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(*s, generator=g) * scale).to(dev)
for (M, S, Kd) in [(7, 5, 13), (65, 33, 70), (150, 64, 128), (596, 978, 600)]:
    x, W, b = rnd(M, Kd, seed=1), rnd(2 * S, Kd, seed=2, scale=Kd ** -0.5), rnd(2 * S, seed=3)
    xt, coef = rnd(M, S, seed=4), rnd(M, seed=5)
    nt = K.heads_tiles(S)
    got, ref = torch.full((M, 2 * S), 7.0, device=dev), torch.zeros(M, 2 * S, device=dev)
    pg, pr = torch.full((M, nt), 3.0, device=dev), torch.zeros(M, nt, device=dev)
    kw = dict(split=S, act0='identity', act1='softplus', shift1=1e-3)
    K.linear_heads(got, x, W, b, nll=dict(x=xt, coef=coef, part=pg), **kw)
    torch.cuda.synchronize()
    R.linear_heads(ref, x, W, b, nll=dict(x=xt, coef=coef, part=pr), **kw)
    sc = float(ref.abs().max())
    print(M, S, Kd, 'max rel err', float(((got - ref).abs().max()) / sc), float((pg.sum(1) - pr.sum(1)).abs().max()), flush=True)
