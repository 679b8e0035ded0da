#!/usr/bin/env python3
"""Round 6: what a hipGraph launch costs the HOST by number of kernel nodes, and whether the device starts on the first
nodes while the host is still enqueueing the rest (K train steps per graph: is the pipeline fill of a drained device
K times longer?).  Kernel = a ~6 us element-wise launch; a chain of N of them per graph."""
import os, sys, time, torch
x = torch.zeros(1 << 21, device='cuda')
sync = torch.cuda.synchronize
for N in (1, 18, 36, 72, 144):
    g = torch.cuda.CUDAGraph()
    x.add_(1.0); sync()
    with torch.cuda.graph(g):
        for _ in range(N):
            x.add_(1.0)
    g.replay(); sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host, total, dev = [], [], []
    for _ in range(20):
        sync()
        t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    # device time of the chain: the second of two back-to-back replays, between events
    for _ in range(5):
        g.replay(); e0.record(); g.replay(); e1.record(); sync()
        dev.append(e0.elapsed_time(e1) * 1e-3)
    h, t, d = sorted(host)[len(host) // 2], sorted(total)[len(total) // 2], min(dev)
    print('N = %3d nodes: host time of the launch call %6.1f us (%.2f us per node) | drained device -> all done %7.1f us | the chain '
          'itself %7.1f us | start-up (fill) %6.1f us' % (N, 1e6 * h, 1e6 * h / N, 1e6 * t, 1e6 * d, 1e6 * (t - d)))
