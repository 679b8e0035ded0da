#!/usr/bin/env python3
"""The block-level MMD add-on of cfg 4: mmd_objective(z[:n/2], z[n/2:], 'rbf_fourier'), Z=100, dim_r=500,
forward + backward through the HIP-backed blocks, timed from a hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drvae_amd import blocks as blk
dev = 'cuda'
for n in (150, 300, 1024):
    z = torch.randn(n, 100, device=dev, requires_grad=True)
    def step():
        z.grad = None
        m = blk.mmd_objective(z[:n // 2], z[n // 2:], 'rbf_fourier')
        m.backward()
        return m
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        step()
    e1.record(); torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) * 1e3 / 50
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print('n=%d: %.1f us eager (host-launch bound) | %.1f us replayed from a hipGraph, per forward+backward incl. the two RNG draws' % (n, eager, e0.elapsed_time(e1) * 1e3 / 50))
