#!/usr/bin/env python3
"""Per-launch timing of one train step WITHOUT a profiler: every launcher call of a step is
recorded as a closure, then each is re-issued 10x from its own hipGraph and timed with HIP
events (so dispatch gaps of graph replay are included, host launch latency is not).  Also
times the main-stream chain, the side-stream chain and the whole step as single-stream graphs."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import drvae_amd.kernels as K
from tests.kernel_ref import FUNCTIONS

def timed_graph(fns, reps=1):
    g = torch.cuda.CUDAGraph()
    for f in fns: f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns: f()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    dev = torch.device('cuda:0')
    for kv in filter(None, os.environ.get('STEP_PROFILE_GEMM_OPTS', '').split(',')):       # dv_gemm_set_option keys
        k, v = kv.split('=')
        from drvae_amd import _lib
        K.gemm_set_option(int(k), int(v))
    cfg, eng, arena, batch, desc = bench.build(wl, dev, 0, 1)
    if len(sys.argv) > 2 and sys.argv[2].startswith('universal'):
        # the batch-independent plan of the sampler feed on the same batch (universal:N = N pair slots, pairs first)
        import numpy as np
        hx = batch['has_x2'].astype(bool)
        order = np.argsort(~hx, kind='stable')
        t = lambda k: torch.from_numpy(batch[k][order]).to(dev)
        eng.universal = True
        eng.universal_pair_slots = int(sys.argv[2].split(':')[1]) if ':' in sys.argv[2] else None
        eng.set_batch(t('x1'), t('x2'), batch['y'][order], hx[order], batch['has_y'][order])
    eng.train_step()
    rec = []
    main_stream = torch.cuda.current_stream().cuda_stream
    LEAF = [n for n in FUNCTIONS if n not in ('linear_fwd', 'linear_bwd_data', 'linear_bwd_weight')]   # wrappers call gemm
    real = {n: getattr(K, n) for n in LEAF}
    def mk(name):
        def f(*a, **kw):
            side = torch.cuda.current_stream().cuda_stream != main_stream
            rec.append((name, side, (lambda: real[name](*a, **kw))))
            return real[name](*a, **kw)
        return f
    for n in LEAF: setattr(K, n, mk(n))
    eng._launch_sequence()
    torch.cuda.synchronize()
    for n in LEAF: setattr(K, n, real[n])
    keep = arena.param.clone(), arena.exp_avg.clone(), arena.exp_avg_sq.clone()
    per = []
    for i, (name, side, fn) in enumerate(rec):
        per.append(timed_graph([fn], reps=10))
    tot = sum(per)
    print('launches: %d (main %d, side %d); sum of per-launch times %.1f us' % (len(rec), sum(not s for _, s, _ in rec), sum(s for _, s, _ in rec), tot))
    agg = collections.OrderedDict()
    for (name, side, _), t in zip(rec, per):
        k = (name, 'side' if side else 'main')
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += t
    for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('  %-18s %-5s x%-3d %8.1f us  (%.1f each)' % (name, where, n, t, t / n))
    print('single-stream graph, main-stream launches only : %.1f us' % timed_graph([f for _, s, f in rec if not s]))
    print('single-stream graph, side-stream launches only : %.1f us' % timed_graph([f for _, s, f in rec if s]))
    print('single-stream graph, all launches in order     : %.1f us' % timed_graph([f for _, _, f in rec]))
    if os.environ.get('STEP_PROFILE_DROP'):
        # upper bound of an offload: the main chain WITHOUT the named launches (as if another chain ran them), on the
        # whole chip and on what a reserve of 32 / 64 CUs would leave it
        drop = set(os.environ['STEP_PROFILE_DROP'].split(','))
        kept = [f for (nm, sd, f) in rec if not sd and nm not in drop]
        gone = [f for (nm, sd, f) in rec if not sd and nm in drop]
        print('main chain without %s: %.1f us (%d launches); the dropped ones alone: %.1f us' % (
            sorted(drop), timed_graph(kept), len(kept), timed_graph(gone)))
        for n in (32, 64):
            pair = eng._part_streams(n)
            if pair is None:
                continue
            with torch.cuda.stream(pair[0]):
                a = timed_graph(kept)
            with torch.cuda.stream(pair[1]):
                b = timed_graph(gone)
            print('  ... on %d CUs: %.1f us; the dropped ones on the %d-CU reserve: %.1f us' % (256 - n, a, n, b))
    if os.environ.get('STEP_PROFILE_MASKS'):
        # the side chain's launches under a CU mask (the reserve of partition()): per launch and as one graph
        for n in [int(x) for x in os.environ['STEP_PROFILE_MASKS'].split(',')]:
            pair = eng._part_streams(n)
            if pair is None:
                continue
            with torch.cuda.stream(pair[1]):
                side = [(nm, f) for (nm, sd, f) in rec if sd]
                ts = [timed_graph([f], reps=10) for _, f in side]
                print('side chain on %d CUs: one graph %.1f us; per launch: %s' % (
                    n, timed_graph([f for _, f in side]), ' '.join('%s=%.1f' % (nm[:14], t) for (nm, _), t in zip(side, ts))))
            with torch.cuda.stream(pair[0]):
                mainl = [(nm, f) for (nm, sd, f) in rec if not sd]
                ts = [timed_graph([f], reps=10) for _, f in mainl]
                print('main chain on %d CUs: one graph %.1f us; per launch: %s' % (
                    256 - n, timed_graph([f for _, f in mainl]), ' '.join('%s=%.1f' % (nm[:14], t) for (nm, _), t in zip(mainl, ts))))
    print('\nsequence:')
    for (name, side, _), t in zip(rec, per):
        print('  %s %-18s %7.1f' % ('S' if side else 'M', name, t))


if __name__ == '__main__':
    main()
