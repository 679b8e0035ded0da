#!/usr/bin/env python3
"""Round 6: what torch.cuda.synchronize() costs on an IDLE device by how many HIP streams / hardware queues the process has made
(the CU-split tuner creates four pairs of CU-masked streams): is the fixed cost of a short timed region the device sync?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sync = torch.cuda.synchronize


def cost(n=200):
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        sync()
    return 1e6 * (time.perf_counter() - t0) / n


def after_kernel(n=50):
    x = torch.zeros(1024, device='cuda')
    tot = 0.0
    for _ in range(n):
        sync()
        t0 = time.perf_counter()
        x.add_(1.0)
        sync()
        tot += time.perf_counter() - t0
    return 1e6 * tot / n


torch.zeros(1, device='cuda')
print('fresh process: idle sync %.1f us, launch + sync %.1f us' % (cost(), after_kernel()))
ss = [torch.cuda.Stream() for _ in range(8)]
for s in ss:
    with torch.cuda.stream(s):
        torch.zeros(1, device='cuda').add_(1)
print('+ 8 plain streams used once: idle sync %.1f us, launch + sync %.1f us' % (cost(), after_kernel()))
dev = torch.device('cuda', 0)
cfg, eng, arena, batch, desc = bench.build('cfg2', dev, 0, 1)
eng.train_step()
eng.capture()
print('+ engine captured (dual graphs, side stream): idle sync %.1f us, launch + sync %.1f us' % (cost(), after_kernel()))
eng.tune_partition()
print('+ tune_partition (4 pairs of CU-masked streams): idle sync %.1f us, launch + sync %.1f us' % (cost(), after_kernel()))
with eng.partition():
    for _ in range(30):
        eng.replay()
    sync()
    for k in (1, 5, 20, 100):
        sync(); t0 = time.perf_counter()
        for _ in range(k):
            eng.replay()
        t1 = time.perf_counter()
        sync(); t2 = time.perf_counter()
        print('  %3d replays: enqueue %.1f us, total %.1f us = %.1f us per step; the closing sync returned %.1f us after the last enqueue'
              % (k, 1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e6 * (t2 - t0) / k, 1e6 * (t2 - t1)))
    # where the host time of the FIRST replay behind a sync goes: the two graph launches by themselves
    g_main, g_side, side = eng._graphs[0], eng._side_graph, eng.flag_side
    for rep in range(3):
        sync()
        t0 = time.perf_counter(); g_main.replay(); t1 = time.perf_counter()
        with torch.cuda.stream(side):
            t2 = time.perf_counter(); g_side.replay(); t3 = time.perf_counter()
        t4 = time.perf_counter(); g_main.replay(); t5 = time.perf_counter()
        with torch.cuda.stream(side):
            t6 = time.perf_counter(); g_side.replay(); t7 = time.perf_counter()
        sync()
        print('  behind a sync: main graph launch %.1f us, side graph launch %.1f us | the next pair: %.1f / %.1f us'
              % (1e6 * (t1 - t0), 1e6 * (t3 - t2), 1e6 * (t5 - t4), 1e6 * (t7 - t6)))
    eng.iters += 6
    # the same inside the partition (CU-masked streams), and eng.replay() itself with a timer around each part
    import drvae_amd.kernels as K
    with eng.partition():
        side = eng.flag_side
        for rep in range(3):
            sync()
            t0 = time.perf_counter(); g_main.replay(); t1 = time.perf_counter()
            with torch.cuda.stream(side):
                t2 = time.perf_counter(); g_side.replay(); t3 = time.perf_counter()
            sync()
            print('  in partition, behind a sync: main graph launch %.1f us, side graph launch %.1f us' % (1e6 * (t1 - t0), 1e6 * (t3 - t2)))
        eng.iters += 3
        for rep in range(3):
            sync()
            t0 = time.perf_counter(); eng.plan.set_beta(eng.beta_pert()); t1 = time.perf_counter()
            eng.replay(); t2 = time.perf_counter()
            eng.replay(); t3 = time.perf_counter()
            sync()
            print('  in partition, behind a sync: set_beta %.1f us, eng.replay() %.1f us, the next eng.replay() %.1f us' % (1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2)))
