#!/usr/bin/env python3
"""Benchmark of the Dr.VAE ELBO train step on MI355X (contract: see the task brief).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg1|cfg4|wide(=cfg5)]

One "step" = one full train step (Philox noise + stacked forward + hand-written backward
+ gradient all-reduce + fused Adam) over one synthetic minibatch per GPU, inputs resident
in HBM.  Prints ONE JSON line on rank 0 with the whole-job samples/s (rows x L), the
roofline of the dominant kernel (the fp32 MFMA GEMM family, timed with HIP events on its
launch stream) and a bounded CPU baseline (the oracle on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kind, rows/GPU, L, overrides, description)
    'cfg2': ('drvae', 150, 2, {}, 'DrVAE bortezomib shape: 978 genes, z1=z3=100, enc-z1 800, dec-x 600, '
                                  'enc-z3/dec-z1 200, batch 150/GPU, L=2'),
    'cfg1': ('pvae', 150, 1, {}, 'PVAE: 978 genes, z1=100, enc 800, dec 600, batch 150/GPU, L=1'),
    'cfg4': ('vfae', 150, 2, {'add_noise_var': 0.0}, 'VFAE/SSVAE: 978 genes, z1=z2=100, enc 800, dec 600, '
                                                       'batch 150/GPU, L=2'),
    'wide': ('drvae', 1024, 4, {'dim_x': 20000, 'dim_z1': 200, 'dim_z3': 200, 'h_en_z1': [2048],
                                'h_de_x': [2048]},
             'DrVAE wide synthetic: 20000 genes, z1=z3=200, enc 2048, dec 2048 (assumed), batch 1024/GPU, L=4'),
}
WORKLOADS['cfg5'] = WORKLOADS['wide']         # BASELINE.json's name for it
# BASELINE.json configs[3] "exercises y-classifier + MMD loss path": the same VFAE step with the nuisance variable s as a
# model input and the model-level MMD penalty between the nuisance classes' latent samples (src/DGMMixin.py:42-66,
# src/blocks.py:40-76; an EXTENSION -- the reference's own glue raises as shipped) inside the captured step
WORKLOADS['cfg4_mmd'] = ('vfae', 150, 2, {'add_noise_var': 0.0, 'use_s': True, 'dim_s': 2, 'use_MMD': True,
                                          'kernel_MMD': 'rbf_fourier'},
                         'VFAE/SSVAE with use_s + use_MMD (rbf_fourier penalty per data group and sample in the captured '
                         'step): 978 genes, z1=z2=100, enc 800, dec 600, batch 150/GPU, L=2')
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (dense, exact fp32)


def profiled_roofline(workload):
    """What the committed profiler output says about this workload: profiles/rNN_<workload>_roofline.json, written by
    tools/roofline_from_profile.py from the rocprofv3 --kernel-trace --stats summary (GEMM time per step IN the
    running step) and the --pmc passes (memory-side bytes).  rocprofv3 cannot run inside this process, so these are
    the figures of the profiled run of the same command; nothing is baked into this file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s_roofline.json' % workload)))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(files[-1], ROOT)


def build(workload, device, rank, world, seed=123):
    from drvae_amd import engine as E
    from drvae_amd import synth
    from drvae_amd.arena import ParamArena
    kind, rows, L, over, desc = WORKLOADS[workload]
    cfg = E.StepConfig(kind=kind, L=L, **over)
    shapes = E.param_shapes(cfg)
    arena = ParamArena(shapes, device)
    rs = np.random.RandomState(seed)                 # random-init weights of that architecture
    fan = 1
    for k, shp in shapes.items():
        if k.endswith('W_mu') or k.endswith('bias_mu'):
            a = rs.uniform(-1e-4, 1e-4, shp)
        else:
            if k.endswith('.weight'):
                fan = shp[1]
            a = rs.uniform(-1, 1, shp) / np.sqrt(fan)
        arena.p(k).copy_(torch.as_tensor(a, dtype=torch.float32))
    # one seed for all ranks: the Philox draws are keyed by (seed, step, draw, GLOBAL row), so the job's noise does
    # not depend on the number of ranks; this rank owns rows [rank*rows, (rank+1)*rows)
    from drvae_amd import tuning
    eng = E.FusedStep(cfg, arena, seed=1000, concurrent=bool(tuning.get('concurrent')), row0=rank * rows)
    batch = synth.make_batch(kind, rows, cfg.dim_x, cfg.dim_y, seed=1234, row0=rank * rows)
    hx, hy = batch['has_x2'].astype(bool), batch['has_y'].astype(bool)
    # weak scaling: every rank has the same group mix, so the global counts are world * local
    counts = (world * rows, world * int(hx.sum()), world * int(hy.sum()))
    t = lambda k: torch.from_numpy(batch[k]).to(device)
    # (use_s workloads: two nuisance classes, alternating by global row)
    sv = ((rank * rows + np.arange(rows)) // 4 % cfg.dim_s) if cfg.use_s else None
    eng.set_batch(t('x1'), t('x2'), batch['y'], hx, hy, counts=counts, s=sv)
    return cfg, eng, arena, batch, desc


def cpu_baseline(workload, budget_s=12.0, threads=4):
    """The CPU oracle (restatement of the reference's PyTorch-CPU train step, pinned to the
    reference's own outputs by tests/golden) timed on this box's host cores."""
    from oracle import models_ref as M
    kind, rows, L, over, _ = WORKLOADS[workload]
    spec = M.ModelSpec(kind=kind, L=L, **over)
    torch.set_num_threads(threads)                   # the reference's own setting, src/run_drvae.py:40
    tr = M.RefTrainer(spec, M.init_params(spec, 123))
    batch = M.make_batch(spec, rows, seed=1234)
    noises = [M.make_noise(spec, rows, seed=s) for s in range(2)]
    tr.step(batch, noises[0])
    tr.step(batch, noises[1])                        # skip iteration 0 (beta_pert differs) + warm caches
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(batch, noises[n % 2])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 200:
            break
    out = {'value': round(rows * L * n / dt, 1), 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
           'ms_per_step': round(1e3 * dt / n, 2),
           'sample': '%d full train steps of the same workload (%s rows x L=%d) with torch.set_num_threads(%d), '
                     '%.1f s of CPU work; host has %d logical cores' % (n, rows, L, threads, dt, os.cpu_count()),
           'cpu_model': _cpu_model()}
    # SURVEY 8(d) also asks for the all-cores figure: a shorter sample with one thread per physical core
    phys = _physical_cores()
    if phys > threads:
        torch.set_num_threads(phys)
        tr.step(batch, noises[0])
        n2, t0 = 0, time.perf_counter()
        while True:
            tr.step(batch, noises[n2 % 2])
            n2 += 1
            dt2 = time.perf_counter() - t0
            if dt2 >= budget_s / 3 or n2 >= 100:
                break
        out['all_cores'] = {'value': round(rows * L * n2 / dt2, 1), 'cores': phys, 'ms_per_step': round(1e3 * dt2 / n2, 2),
                            'sample': '%d steps, %.1f s' % (n2, dt2)}
        torch.set_num_threads(threads)
    return out


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _physical_cores():
    """distinct (socket, core) pairs the process may run on; falls back to the logical count"""
    try:
        allowed = os.sched_getaffinity(0)
        seen, phys, core, cpu = set(), None, None, None
        with open('/proc/cpuinfo') as f:
            for line in f:
                k = line.split(':', 1)[0].strip()
                v = line.split(':', 1)[1].strip() if ':' in line else ''
                if k == 'processor':
                    cpu = int(v)
                elif k == 'physical id':
                    phys = v
                elif k == 'core id':
                    core = v
                elif not line.strip() and cpu is not None:
                    if cpu in allowed:
                        seen.add((phys, core))
                    cpu = None
        return len(seen) or len(allowed)
    except (OSError, ValueError, AttributeError):
        return os.cpu_count() or 1


def parity_vs_cpu(workload, device, steps=2):
    """'ELBO vs ref delta' of the metric: the same train steps (same parameters, same batch, same injected
    N(0,1) draws) on the HIP path and on the CPU oracle; the largest relative difference over the loss
    scalars (RECL, KLD, PERT, YL, ELBO, CMPL) of every step.  Part of the cpu_baseline leg."""
    from oracle import models_ref as M
    from drvae_amd import engine as E
    from drvae_amd.arena import ParamArena
    kind, rows, L, over, _ = WORKLOADS[workload]
    spec = M.ModelSpec(kind=kind, L=L, **over)
    cfg = E.StepConfig(kind=kind, L=L, **over)
    params = M.init_params(spec, 123, as_numpy=True)
    arena = ParamArena(E.param_shapes(cfg), device, frozen=E.frozen_params(cfg))
    arena.load(params)
    eng = E.FusedStep(cfg, arena)
    batch = M.make_batch(spec, rows, seed=1234)
    t = lambda k: torch.from_numpy(batch[k]).to(device)
    eng.set_batch(t('x1'), t('x2'), batch['y'], batch['has_x2'], batch['has_y'])
    tr = M.RefTrainer(spec, M.init_params(spec, 123))
    worst = 0.0
    for s_ in range(steps):
        noise = M.make_noise(spec, rows, seed=100 + s_)
        eng.train_step(noise)
        got = eng.losses()
        ref, _ = tr.step(batch, noise)
        for k, v in got.items():
            r = float(ref[k].detach()) if torch.is_tensor(ref[k]) else float(ref[k])
            if k != 'MMD':
                worst = max(worst, abs(v - r) / max(abs(r), 1e-6))
    return {'max_rel_diff_losses': float('%.3g' % worst), 'steps': steps, 'tolerance': 1e-4,
            'what': 'HIP train steps vs CPU oracle, identical parameters / batch / injected noise'}


def gemm_roofline(eng, cfg, rows, frac_pair, frac_lab, repeats=20, workload='cfg2', n_params=0, input_bytes=0,
                  ms_per_step=None, brief=False, steady_ms=None):
    """Roofline of the dominant kernel family: the fp32-MFMA GEMM (all tilings/layouts; 32
    launches per cfg-2 step).  Every GEMM launch of one train step is re-issued `repeats`
    times back to back from a small hipGraph and timed with HIP events recorded on the
    launch stream (so host launch latency is not in the bracket); the per-launch times are
    summed over the step.  achieved = algorithmic GEMM FLOPs per step (SURVEY.md 8(d) model)
    / that sum -- equivalently avg FLOPs per launch / avg launch duration."""
    import drvae_amd.kernels as K
    from drvae_amd import synth
    calls = []
    real_gemm, real_pair, real_heads = K.gemm, K.linear_bwd_pair, K.linear_heads
    nbytes = lambda *ts: 4.0 * sum(t.numel() for t in ts if t is not None)

    def rec_gemm(Cm, A, B, a_kc, b_kc, **kw):
        M_, N_ = Cm.shape
        K_ = (A.shape[1] + (kw['A2'].shape[1] if kw.get('A2') is not None else 0)) if a_kc else A.shape[0]
        calls.append((lambda: real_gemm(Cm, A, B, a_kc, b_kc, **kw), [Cm] if kw.get('beta', 0.0) != 0.0 else [],
                      2.0 * M_ * N_ * K_, ('gemm', M_, N_, K_, int(bool(a_kc)), int(bool(b_kc))),
                      nbytes(Cm, A, B, kw.get('A2'))))
        real_gemm(Cm, A, B, a_kc, b_kc, **kw)

    def rec_heads(out, x, W, bias=None, **kw):       # dual-head product with its row work fused in (dv_gemm_heads)
        M_, N_ = out.shape
        K_ = x.shape[1] + (kw['x2'].shape[1] if kw.get('x2') is not None else 0)
        calls.append((lambda: real_heads(out, x, W, bias, **kw), [], 2.0 * M_ * N_ * K_,
                      ('heads ' + ('nll' if kw.get('nll') is not None else 'sample'), M_, N_, K_),
                      nbytes(out, x, W, kw.get('x2'))))
        real_heads(out, x, W, bias, **kw)

    def rec_pair(dW, dbias, dx, dpre, x, W, **kw):      # dW = dpre^T x and dx = dpre W share one launch
        Mb, Nw = dpre.shape
        n_dx = W.shape[1]       # (dx may be None: the DV_EPI_KLQ form writes d/d(mu | logvar) of the input's q rows instead)
        out_dx = dx if dx is not None else kw['klq']['out']
        calls.append((lambda: real_pair(dW, dbias, dx, dpre, x, W, **kw), [dx] if kw.get('beta_x', 0.0) != 0.0 else [],
                      2.0 * Mb * Nw * x.shape[1] + 2.0 * Mb * Nw * n_dx,
                      ('pair dW+dX', Mb, Nw, x.shape[1], n_dx), nbytes(dW, out_dx, dpre, x, W)))
        real_pair(dW, dbias, dx, dpre, x, W, **kw)

    K.gemm, K.linear_bwd_pair, K.linear_heads = rec_gemm, rec_pair, rec_heads
    try:
        eng.training = True
        eng.fuse_bwd = True              # the launch sequence of the TRAIN step (fused heads epilogues included)
        eng.draw_noise()
        eng.forward()
        eng.backward()
        torch.cuda.synchronize()
    finally:
        eng.fuse_bwd = False
        K.gemm, K.linear_bwd_pair, K.linear_heads = real_gemm, real_pair, real_heads
    per_call = []
    gemm_bytes = sum(c[4] for c in calls)
    for (fn, accum, flops, shape, _) in calls:
        keep = [t.clone() for t in accum]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(repeats):
                fn()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / repeats)
        for t, k in zip(accum, keep):
            t.copy_(k)
        per_call.append((best, flops, shape))
    t_step = sum(t for t, _, _ in per_call)
    executed = sum(f for _, f, _ in per_call)
    algorithmic = synth.gemm_flops_per_step(cfg, rows, frac_pair, frac_lab)
    achieved = algorithmic / t_step / 1e12
    top = sorted(per_call, key=lambda r: -r[0])[:3]
    prof, prof_file = profiled_roofline(workload)
    # inputs + parameters read forward and backward + gradient write + Adam's 7 words per parameter
    alg_bytes = input_bytes + 4.0 * n_params * (2 + 1 + 7)
    iso = {'achieved': round(achieved, 3), 'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
           'launches_per_step': len(per_call), 'avg_launch_us': round(1e6 * t_step / len(per_call), 2),
           'gemm_us_per_step': round(1e6 * t_step, 1),
           'how': 'live: every GEMM-family launch of one step re-issued %dx back to back from a hipGraph on an idle, '
                  'un-partitioned chip, HIP events on the launch stream; achieved = algorithmic GEMM FLOPs per step / sum '
                  'of the per-launch times (= avg FLOPs per launch / avg launch duration)' % repeats}
    # ONE definition of the headline figure (round 5): the step's algorithmic GEMM FLOPs over THIS LINE's ms_per_step --
    # the contract's timed region, everything that is not a GEMM included -- over the fp32-MFMA peak; a reader recomputes
    # it from the line: frac = algorithmic_gflop_per_step / ms_per_step / peak (GFLOP per ms = TFLOP/s).  The kernel-level figures keep keys of
    # their own: `isolated` (this run, launches re-issued alone), `in_graph` (the committed profile of the running step)
    a_line = algorithmic / (ms_per_step * 1e-3) / 1e12
    out = {'bound': 'mfma', 'achieved': round(a_line, 3), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(a_line / FP32_MFMA_PEAK_TFLOPS, 4), 'ms_per_step': ms_per_step,
           'definition': 'achieved = algorithmic_gflop_per_step / ms_per_step (this line\'s timed region, whole step); '
                         'frac = achieved / peak',
           'kernel': 'gemm_pipe_kernel / gemm_pair_pipe_kernel / gemm_heads_pipe_kernel<...> (fp32 v_mfma_f32_32x32x2_f32; all tilings/layouts)',
           'algorithmic_gflop_per_step': round(algorithmic / 1e9, 3),
           'executed_gflop_per_step': round(executed / 1e9, 3),
           'isolated': iso, 'traffic': None}
    if steady_ms:        # the longer region of the same replays (>= 3 s): its own key, never the headline
        a3 = algorithmic / (steady_ms * 1e-3) / 1e12
        out['steady'] = {'achieved': round(a3, 3), 'frac': round(a3 / FP32_MFMA_PEAK_TFLOPS, 4), 'ms_per_step': steady_ms}
    def in_graph(ig):
        # (a PROFILE of an earlier run of this command, not a measurement of the build being run: its own key, with the
        # commit it was collected at; `achieved` / `frac` above are this run's live figures)
        a2 = algorithmic / (ig['kernel_us_per_step'] * 1e-6) / 1e12
        return {'gemm_us_per_step': ig['kernel_us_per_step'], 'launches_per_step': ig['launches_per_step'],
                'avg_launch_us': ig['avg_launch_us'], 'achieved': round(a2, 3),
                'frac': round(a2 / FP32_MFMA_PEAK_TFLOPS, 4), 'profiled_at': prof.get('commit'), 'profile': prof_file,
                'what': 'GEMM-family kernel time per step of the RUNNING step (both chains, main chain on its CU partition) '
                        'from the committed rocprofv3 --kernel-trace --stats summary'}
    if brief:
        out['top_launches'] = [{'shape': list(sh), 'us': round(1e6 * t, 2), 'tflops': round(f / t / 1e12, 2)}
                               for t, f, sh in top]
        if prof is not None and 'kernel_us_per_step' in (prof.get('gemm_in_graph') or {}):
            out['in_graph'] = in_graph(prof['gemm_in_graph'])
        if prof is not None and prof.get('traffic'):     # memory-side bytes per GEMM launch from the committed --pmc passes
            out['traffic'] = round(1e6 * prof['traffic']['gemm']['per_launch_corrected_upper_mb'])
            out['traffic_unit'] = 'bytes per GEMM launch (FETCH_SIZE x2 gfx950 wide-load correction + WRITE_SIZE)'
            out['profile'] = prof_file
        return out
    out.update({'algorithmic_mbytes_per_step': round(alg_bytes / 1e6, 2),
                'algorithmic_gemm_mbytes_per_step': round(gemm_bytes / 1e6, 2),
                'top_launches': [{'shape': list(sh), 'us': round(1e6 * t, 2), 'tflops': round(f / t / 1e12, 2)}
                                 for t, f, sh in top]})
    if prof is not None:
        tr, ig = prof.get('traffic'), prof.get('gemm_in_graph')
        out['profile'] = prof_file
        if tr:      # memory-side bytes per GEMM launch (like `achieved`), and per step next to the algorithmic figure
            out['traffic'] = round(1e6 * tr['gemm']['per_launch_corrected_upper_mb'])
            out['traffic_unit'] = 'bytes per GEMM launch (FETCH_SIZE x2 gfx950 wide-load correction + WRITE_SIZE)'
            out['traffic_per_step_mb'] = {k: tr[k] for k in ('fetch_reported', 'fetch_corrected_upper', 'write',
                                                             'total_reported', 'total_corrected_upper') if k in tr}
            out['traffic_per_step_mb']['gemm_total_corrected_upper'] = round(
                tr['gemm']['fetch_corrected_upper'] + tr['gemm']['write'], 1)
            out['wasted_ratio'] = {'step_upper': round(tr['total_corrected_upper'] * 1e6 / alg_bytes, 2),
                                   'step_reported': round(tr['total_reported'] * 1e6 / alg_bytes, 2),
                                   'gemm_upper': round((tr['gemm']['fetch_corrected_upper'] + tr['gemm']['write']) * 1e6
                                                       / max(gemm_bytes, 1.0), 2)}
        if ig and 'kernel_us_per_step' in ig:
            # THE figure of the line: the same FLOPs over the GEMM-family kernel time of the RUNNING step (both chains
            # running, main chain on its CU partition), i.e. avg FLOPs per launch / avg in-situ launch duration, from
            # the committed rocprofv3 --kernel-trace --stats summary of this command
            out['in_graph'] = in_graph(ig)
    return out


def fit_epoch(device, n_train=8192, n_valid=2048, batch=150, epochs=7):
    """One steady-state epoch of the reference's ``fit`` protocol at cfg-2 size (SURVEY.md 8(f) N1 + N3): ``n_train // batch``
    train steps drawn by the graph-resident feed from an HBM-resident training set, then the whole-set evaluation of the
    training and of the validation set (src/DrVAE.py:797,821) -- each ONE captured graph replay and one device->host copy.
    The evaluation's roofline: its forward FLOPs (eval-mode loss pass with L samples + means-only inference,
    src/DrVAE.py:367-543 and 253-311, per row as in ``tools/eval_bench.eval_gflop``) over its wall time."""
    from drvae_amd.DrVAE import DrVAE
    from drvae_amd import data as DD
    from tools.eval_bench import dataset, eval_gflop
    model = DrVAE(dim_x=978, dim_s=1, dim_y=2, dim_h_en_z1=[800], dim_h_de_z1=[200], dim_h_en_z3=[200], dim_h_de_x=[600],
                  dim_h_clf=[], dim_z1=100, dim_z3=100, type_rec='diag_gaussian', nonlinearity='elu', learning_rate=5e-4, L=2,
                  weight_decay=0.05, add_noise_var=0.01, pertloss_rate=0.05, use_MMD=False, random_seed=123, epochs=epochs + 1,
                  batch_size=batch).to(device)
    model.w2log = lambda *a: None
    model.add_noise = True
    tr, va = dataset(n_train, 1, device), dataset(n_valid, 2, device)
    bat = DD.DeviceBatcher(tr, torch.ones(n_train), batch, seed=1)
    ts = []
    for ep in range(epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model._epoch_device(bat, ep + 1, False)        # (epoch numbers as ``fit`` passes them: 1 .. epochs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ptr, _ = model.evaluate_performance_on_dataset(tr)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pva, _ = model.evaluate_performance_on_dataset(va)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ts.append([t1 - t0, t2 - t1, t3 - t2])
    # the median epoch of those behind the first two (which pay for the captures and the CU-split tuning)
    ts = sorted(ts[2:], key=sum)
    t = ts[len(ts) // 2]
    gf = eval_gflop(tr) + eval_gflop(va)
    gfx = eval_gflop(tr, executed=True) + eval_gflop(va, executed=True)
    ev_ms = 1e3 * (t[1] + t[2])
    return {'train_ms': round(1e3 * t[0], 3), 'steps': len(bat), 'ms_per_step': round(1e3 * t[0] / len(bat), 4),
            'eval_train_ms': round(1e3 * t[1], 3), 'eval_valid_ms': round(1e3 * t[2], 3),
            'epoch_ms': round(1e3 * sum(t), 3), 'rows': [n_train, n_valid],
            'finite': bool(np.isfinite(ptr['x1_rmse']) and np.isfinite(pva['x1_rmse'])),
            'eval_roofline': {'bound': 'mfma', 'achieved': round(gf / ev_ms, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(gf / ev_ms / FP32_MFMA_PEAK_TFLOPS, 4), 'gflop': round(gf, 1),
                              'gflop_executed': round(gfx, 1), 'frac_executed': round(gfx / ev_ms / FP32_MFMA_PEAK_TFLOPS, 4),
                              'executed_note': 'the means-only inference takes q(z1|x1) from the loss pass (same rows, same '
                                               'kernels, no input noise in evaluation): its encoder FLOPs are algorithmic, not executed',
                              'what': 'forward FLOPs of both whole-set evaluations (loss pass with L = 2 samples + means-only '
                                      'inference) / their wall time incl. the metrics and the one host copy each'}}


def mmd_addon(device, n=150, Z=100, reps=50):
    """The block-level MMD add-on of cfg 4: ``mmd_objective(z[:n/2], z[n/2:], 'rbf_fourier')`` (Z = 100, 500
    random Fourier features; src/blocks.py:40-76), forward + backward through the HIP-backed block incl. its two RNG draws,
    replayed from a hipGraph."""
    from drvae_amd import blocks as blk
    z = torch.randn(n, Z, device=device, requires_grad=True)

    def step():
        z.grad = None
        m = blk.mmd_objective(z[:n // 2], z[n // 2:], 'rbf_fourier')
        m.backward()
        return m
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return {'us_per_call': round(e0.elapsed_time(e1) * 1e3 / reps, 2), 'rows': [n // 2, n - n // 2], 'dim_z': Z, 'dim_r': 500,
            'finite': bool(torch.isfinite(z.grad).all()),
            'what': "mmd_objective(z[:75], z[75:], 'rbf_fourier') forward + backward incl. the two RNG draws, hipGraph replay"}


class _Watchdog:
    """multi-rank runs: a daemon thread that ends THIS rank (exit code 5) when the measurement makes no progress for
    ``timeout`` seconds -- a peer that died inside a collective, a hung exchange -- so that the launcher (torchrun, or
    ``self_launch``) sees a failure instead of a job that never ends"""

    def __init__(self):
        self.t = None
        self.beat = time.time()

    fallback = None      # (line, exit code): what a stall prints / returns instead of failing (the captured-exchange probe)

    def start(self, timeout, rank):
        import threading
        self.beat = time.time()
        self.on = True

        def run():
            while self.on:
                time.sleep(0.5)
                if time.time() - self.beat > timeout:
                    print('bench.py: rank %d made no progress for %.0f s: giving up' % (rank, timeout), file=sys.stderr,
                          flush=True)
                    if self.fallback is not None:      # the measured headline is out already: print it, say what stalled
                        line, code = self.fallback
                        if rank == 0 and line:
                            print(line, flush=True)
                        os._exit(code)
                    os._exit(5)
        self.t = threading.Thread(target=run, daemon=True)
        self.t.start()

    def kick(self):
        self.beat = time.time()

    def stop(self):
        self.on = False


_WATCHDOG = _Watchdog()


def measure(args, workload, feed, steps, warmup, device, rank, world, steady_s=0.0, exchange_probe=False):
    """Build the workload, capture its train step and time exactly ``steps`` replays (barrier + synchronize on both
    sides, MAX over ranks); with ``steady_s`` a second, longer region of about that many seconds is timed as well.
    Returns (result dict, context for the roofline leg)."""
    from drvae_amd import dist as D
    import torch.distributed as dist
    cfg, eng, arena, batch, desc = build(workload, device, rank, world)
    kind, rows, L = WORKLOADS[workload][:3]
    D.broadcast_params(arena)
    dp = world > 1 or D.force_dp()          # (DRVAE_FORCE_DP=1: the multi-rank step path with a one-rank communicator)
    allreduce = D.allreduce_sum if dp else None

    # iteration 0 runs eagerly (beta_pert = 0.01 only there), then the steady-state step is captured
    eng.train_step(allreduce=allreduce)
    use_graph = not args.no_graph
    dp_mode = None
    bat = None
    n_table = [max(steps + warmup + 8, 1024)]      # batches per index table of the graph-resident feeds (re-drawn in place)
    k_batch = [0]                                    # batch of the table the NEXT replay gathers
    if feed != 'resident':
        from drvae_amd import data as DD, synth
        # (data parallelism: every rank holds the same dataset and batcher seed -- they draw the same global index
        # table and run their own columns of it)
        big = synth.make_batch(kind, args.dataset_rows, cfg.dim_x, cfg.dim_y, seed=77)
        tt = lambda k: torch.from_numpy(big[k]).to(device)
        ds = DD.DrVAEDataset(tt('x1'), tt('x2'), torch.zeros(args.dataset_rows, dtype=torch.int64, device=device),
                             tt('y'), tt('has_x2'), tt('has_y'))
        hx, hy = batch['has_x2'].astype(bool), batch['has_y'].astype(bool)
        gc = [int(((hy == bool(gy)) & (hx == bool(gx))).sum()) for (gy, gx) in DD._GROUPS]
        if feed == 'sampler':
            bat = DD.DeviceBatcher(ds, torch.ones(args.dataset_rows), rows, seed=5, mode='sampler',
                                   pair_bucket=args.pair_bucket or None,
                                   label_bucket=(max(8, rows // 3 // 8 * 8) if args.label_bucket < 0 else args.label_bucket) or None)
            bat.bind(eng, dp=(rank, world) if dp else None)
        else:
            bat = DD.DeviceBatcher(ds, torch.ones(args.dataset_rows), rows, group_counts=gc, seed=5)
            bat.bind(eng, dp=(rank, world) if dp else None)
        if feed in ('epoch', 'sampler'):
            bat.begin_epoch(n_batches=n_table[0])
        else:
            bat.feed()

    def rebase(n=None):
        """graph-resident feeds: a fresh index table, drawn IN PLACE (same number of batches: the captured feed object
        stays the one the graphs were captured with); a region longer than the table re-draws it every ``n_table - 8``
        replays (``run``) instead of re-training on the clamped last batch"""
        if bat is not None and feed in ('epoch', 'sampler'):
            bat.begin_epoch(n_batches=n_table[0])
        k_batch[0] = 0
    if use_graph:
        # one exchange between two graphs by default; --dp-exchange overlap: two overlapped pieces between three
        # graphs (measured with a one-rank RCCL communicator: +49 us of launch/event overhead per step against +24 us)
        overlap = dp and args.dp_exchange == 'overlap'
        dp_mode = 'overlap' if overlap else dp
        if dp and not overlap and args.dp_exchange == 'captured':
            # the exchange captured into the step's graph (needs stream-capturable RCCL: probed here, with the
            # split graphs as the fallback)
            try:
                eng.capture(split_for_allreduce='captured', allreduce=D.allreduce_sum)
                allreduce, dp_mode = None, 'captured'
            except Exception as e:       # noqa: BLE001
                print('bench.py: RCCL capture unavailable (%s); split graphs' % e, file=sys.stderr)
                dp_mode = dp
        if dp_mode != 'captured':
            eng.capture(split_for_allreduce=dp_mode)
        if bat is not None and bat.bucketed:
            # one captured step per number-of-pairs bucket of the table's batches; each replay picks its batch's
            eng.stash_capture()
            bat.prepare_epoch(lambda e: e.capture(split_for_allreduce=dp_mode))
        if overlap and len(eng._graphs) == 3:
            allreduce = D.OverlappedAllReduce()      # decoder block travels while the encoder backward runs
        if feed == 'batcher':
            def step():
                bat.feed()
                eng.replay(allreduce)
        elif bat is not None and bat.bucketed:
            def step():
                # the plan of the batch THIS replay gathers (the device feed takes batch step_dev - epoch base)
                bat.select(k_batch[0])
                k_batch[0] += 1
                eng.replay(allreduce)
                if args.check_feed:      # (tests) the batch the device just gathered is the one whose plan was selected
                    got = int(eng.step_dev) - int(eng.plan.feed.base) - 1
                    assert got == k_batch[0] - 1, 'replay gathered batch %d on the plan of batch %d' % (got, k_batch[0] - 1)
        else:
            step = lambda: eng.replay(allreduce)
    else:
        step = lambda: eng.train_step(allreduce=allreduce)
    import contextlib
    part = contextlib.nullcontext()
    if use_graph:
        eng.tune_partition()             # reserved CUs for the side chain, chosen by timing (state restored)
        part = eng.partition()           # side chain on reserved CUs (dual-graph schedule); no-op otherwise
        rebase()                         # (the tuning replays advanced the step counter past the table's start)
    part.__enter__()
    if world > 1:
        dist.barrier()          # ranks leave capture together: the first exchanges do not sit out capture skew
    def run(n):
        """n replays; graph-resident feeds get a fresh table whenever the current one is used up"""
        chunk = n_table[0] - 8 if (bat is not None and feed in ('epoch', 'sampler')) else n
        done = 0
        while done < n:
            m = min(chunk, n - done)
            if done:
                rebase()
            for _ in range(m):
                step()
            done += m

    run(max(warmup - 1, 0))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        _WATCHDOG.kick()

    def over_ranks(dt):
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item())
        return dt

    barrier()
    t0 = time.perf_counter()
    run(steps)
    t_enq = time.perf_counter() - t0       # host time to enqueue the steps (if ~dt the host is the bound)
    barrier()
    dt = over_ranks(time.perf_counter() - t0)
    # a second, longer region (same step, same barriers) next to the contract's K steps: >= ``steady_s`` seconds of
    # back-to-back replays, so that the driver's utilisation sampler sees the device busy and the number does not
    # hang on a few milliseconds
    steady = None
    if steady_s > 0 and dt < steady_s:
        n2 = int(min(max(200, steady_s * steps / max(dt, 1e-6)), 60000))
        rebase()
        barrier()
        t1 = time.perf_counter()
        run(n2)
        barrier()
        dt2 = over_ranks(time.perf_counter() - t1)
        steady = {'steps': n2, 'seconds': round(dt2, 3), 'ms_per_step': round(1e3 * dt2 / n2, 4),
                  'value': round(world * rows * L * n2 / dt2, 1)}
    if bat is not None and getattr(bat, 'bucketed', False):
        print('bench.py: sampler feed, %d captured plans, %d plan switches' % (len(eng._captures), bat.n_switch), file=sys.stderr)
    losses = eng.losses()                  # (of the last measured step: read before the probe below trains on)
    exchange = None
    if dp and exchange_probe and use_graph and len(getattr(eng, '_graphs', [])) == 2:
        # what the gradient exchange costs a step: the same replays without the collective between the two graphs
        # (every rank's own gradients only -- a timing probe on scratch state, after the measured regions)
        barrier()
        t2 = time.perf_counter()
        for _ in range(steps):
            eng.replay(None)
        barrier()
        dt3 = over_ranks(time.perf_counter() - t2)
        exchange = {'us_per_step': round(1e6 * (dt - dt3) / steps, 2), 'ms_per_step_without': round(1e3 * dt3 / steps, 4),
                    'bytes': int(arena.xchg.numel() * 4), 'mode': str(dp_mode)}
    part.__exit__(None, None, None)
    waits = eng.sync_err.cpu().tolist() if hasattr(eng, 'sync_err') else None
    ok = all(np.isfinite(v) for v in losses.values())
    out = {
        'value': round(world * rows * L * steps / dt, 1), 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(1e3 * dt / steps, 4),
        'config': {'workload': '%s: %s' % (workload, desc), 'global_batch': world * rows, 'L': L,
                   'parallelism': 'dp%d' % world, 'launch': 'hipGraph replay' if use_graph else 'eager', 'feed': feed,
                   'side_chain_cus': getattr(eng, '_side_cus', None),
                   'dist_backend': (dist.get_backend() if dist.is_initialized() else None),
                   'dp_exchange': (dp_mode if dp else None),
                   'rccl_ranks': (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == 'nccl'
                                  else 0),
                   'params': int(sum(int(np.prod(s)) for s in arena.shapes.values()))},
        'losses_last_step': {k: round(v, 4) for k, v in losses.items()}, 'finite': ok,
        'chain_wait_ticks': waits, 'host_enqueue_ms_per_step': round(1e3 * t_enq / steps, 4),
        'steady_state': steady,
    }
    if exchange is not None:
        out['exchange'] = exchange
    return out, dict(eng=eng, cfg=cfg, batch=batch, rows=rows, dp=dp)


def captured_exchange_probe(args, ctx, device, rank, world, single_ms):
    """Data-parallel runs: the SAME step with the gradient all-reduce captured INTO its graph (no graph boundary, no host
    launch between backward and the optimiser sweep; the side chain draws the next step's noise behind the join: DESIGN.md, Multi-GPU), timed
    like the headline region in the same rank processes, after it.  The headline stays the two-graph form (`single`): a
    captured collective that misbehaved on N ranks would hang rather than raise -- the caller arms the stall watchdog with the
    headline line already assembled, so a stalled probe costs nothing but this entry."""
    from drvae_amd import dist as D
    import torch.distributed as dist
    eng = ctx['eng']
    eng.capture(split_for_allreduce='captured', allreduce=D.allreduce_sum)
    steps, warm = max(args.steps, 20), 5

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        _WATCHDOG.kick()
    with eng.partition():
        for _ in range(warm):
            eng.replay(None)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.replay(None)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    losses = eng.losses()
    rows, L = ctx['rows'], ctx['cfg'].L
    return {'ms_per_step': round(1e3 * dt / steps, 4), 'steps': steps, 'value': round(world * rows * L * steps / dt, 1),
            'finite': bool(all(np.isfinite(v) for v in losses.values())), 'noise_drawn_ahead': bool(eng.noise_ahead),
            'vs_single': round(1e3 * dt / steps / single_ms, 4)}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(n, argv, timeout=1500.0):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would) and relay rank 0's JSON line and
    the worst exit code.  This parent never touches the GPU (``import torch`` alone does not initialise
    HIP), and nothing that has is ever re-exec'ed: the ranks are plain child processes."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=os.environ.get('MASTER_PORT') or str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()                                      # (children's stderr goes straight to ours)
    limit = time.time() + timeout
    while any(p.poll() is None for p in procs):
        failed = any(p.poll() not in (None, 0) for p in procs)
        if failed or time.time() > limit:           # a rank died (its peers would sit in a collective for ever)
            time.sleep(2.0 if failed else 0.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()                        # exactly the processes started above
            break
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    rd.join(timeout=10)
    out0 = b''.join(chunks).decode(errors='replace')
    line = None
    for ln in out0.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    else:
        sys.stdout.write(out0)
    bad = [c for c in codes if c != 0]
    return (bad[0] if bad else 0) if line is not None else (bad[0] if bad else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='cfg2', choices=list(WORKLOADS))
    ap.add_argument('--check-feed', action='store_true', help='(tests) bucketed sampler feed: assert after every replay that the '
                    'batch gathered on the device is the one whose plan was selected')
    ap.add_argument('--strict', action='store_true', help='exit code 3 when a leg next to the headline (realistic_feed, other_workloads) raised')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--pair-bucket', type=int, default=16,
                    help="--feed sampler: batches are re-ordered pairs first and run on the plan whose pair slots are "
                         "the batch's number of pairs rounded up to a multiple of this (0: one plan sized for B pairs)")
    ap.add_argument('--label-bucket', type=int, default=-1,
                    help="--feed sampler: the same for the labels -- rows a batch's plan knows to be labeled get one "
                         "fprop row instead of one per class (0: off; default: about a third of the batch)")
    ap.add_argument('--feed', default='resident', choices=['resident', 'batcher', 'epoch', 'sampler'],
                    help='resident: one batch parked in HBM (default); batcher: a fresh stratified minibatch drawn on '
                         'the device from an HBM-resident dataset before every step (host-driven gathers); epoch: the '
                         "same, but the captured step gathers its batch itself from the epoch's index table; sampler: "
                         'as epoch, with the exact WeightedRandomSampler semantics (any group mix per batch: universal plan)')
    ap.add_argument('--dataset-rows', type=int, default=16384)
    ap.add_argument('--gemm-opts', default='', help='tuning: comma list key=value of dv_gemm_tune.opt')
    ap.add_argument('--dp-exchange', default='single', choices=['single', 'overlap', 'captured'],
                    help='data-parallel gradient exchange: one all-reduce between two graphs (default), two overlapped '
                         'pieces between three graphs, or the collective captured into the step graph')
    ap.add_argument('--no-exchange-modes', action='store_true', help='data-parallel runs: skip the captured-exchange probe that follows the headline')
    ap.add_argument('--no-steady', action='store_true', help='skip the second, longer timed region')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the realistic-feed and other-workload measurements that follow the headline (cfg2 only)')
    ap.add_argument('--timeout', type=float, default=1500.0, help='self-launched ranks: seconds before they are ended')
    ap.add_argument('--stall-timeout', type=float, default=300.0,
                    help='multi-rank: a rank that makes no progress for this many seconds exits with code 5')
    ap.add_argument('--fail-rank', type=int, default=-1, help='test hook: this rank exits (code 3) right after joining')
    args = ap.parse_args()
    if args.workload == 'cfg5':
        args.workload = 'wide'

    if args.gpus > 1 and int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return self_launch(args.gpus, sys.argv[1:], args.timeout)   # before anything initialises the GPU in this process

    from drvae_amd import _lib, dist as D
    _lib.load()                                       # fail loudly if the HIP library is missing
    for kv in filter(None, args.gemm_opts.split(',')):     # tuning: dv_gemm_tune.opt keys (include/drvae_hip.h)
        import drvae_amd.kernels as K_
        k, v = kv.split('=')
        K_.gemm_set_option(int(k), int(v))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback for the hot path)')
    rank, world, local = D.init_from_env()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.fail_rank == rank and world > 1:
        os._exit(3)              # (its peers now sit in their first collective: the launcher has to end them)
    local = local % torch.cuda.device_count()        # (one-GPU functional tests of the multi-rank path)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    import torch.distributed as dist

    if world > 1:
        _WATCHDOG.start(args.stall_timeout, rank)
    res, ctx = measure(args, args.workload, args.feed, args.steps, args.warmup, device, rank, world,
                       steady_s=0.0 if args.no_steady else 3.2, exchange_probe=True)
    _WATCHDOG.stop()
    ok, dp = res['finite'], ctx['dp']
    out = {'metric': 'DrVAE ELBO training samples/sec (batch x L)', 'value': res['value'], 'unit': 'samples/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['ms_per_step'],
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic'}
    out.update({k: v for k, v in res.items() if k not in out})
    if rank == 0:
        from drvae_amd import synth
        eng, cfg, batch, rows = ctx['eng'], ctx['cfg'], ctx['batch'], ctx['rows']
        hx, hy = batch['has_x2'].astype(bool), batch['has_y'].astype(bool)
        if not args.no_roofline:
            out['roofline'] = gemm_roofline(eng, cfg, rows, float(hx.mean()), float(hy.mean()),
                                            repeats=20 if args.workload != 'wide' else 3, workload=args.workload,
                                            n_params=out['config']['params'],
                                            input_bytes=4.0 * rows * cfg.dim_x * (2 if cfg.has_pert else 1),
                                            ms_per_step=res['ms_per_step'],
                                            steady_ms=(res['steady_state'] or {}).get('ms_per_step'))
        if not args.no_cpu_baseline and args.workload != 'wide' and world == 1:   # rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(args.workload)
            out['elbo_vs_ref'] = parity_vs_cpu(args.workload, device)
    if dp and not args.no_graph and args.dp_exchange == 'single' and not args.no_exchange_modes and args.feed == 'resident':
        # both exchange forms in one record: the headline (`single`: two graphs, the all-reduce between them) and the
        # captured form, probed AFTER it under the stall watchdog with the headline line as its fallback
        out['exchange_modes'] = {'single': {'ms_per_step': res['ms_per_step'], 'value': res['value']},
                                 'captured': {'error': 'the probe stalled (watchdog): the captured collective did not complete'}}
        if dist.is_initialized() and dist.get_backend() != 'nccl':
            # (gloo -- the one-GPU functional tests of the multi-rank path -- runs its collectives on the host: nothing to capture)
            out['exchange_modes']['captured'] = {'skipped': 'backend %s cannot be captured into a hipGraph' % dist.get_backend()}
        else:
            _WATCHDOG.fallback = (json.dumps(out) if rank == 0 else '', 0 if ok else 1)
            _WATCHDOG.start(min(args.stall_timeout, 90.0), rank)
            try:
                out['exchange_modes']['captured'] = captured_exchange_probe(args, ctx, device, rank, world, res['ms_per_step'])
            except Exception as e:       # noqa: BLE001  (e.g. an RCCL build that cannot be stream-captured)
                print('bench.py: captured-exchange probe failed: %r' % (e,), file=sys.stderr)
                out['exchange_modes']['captured'] = {'error': repr(e)}
            _WATCHDOG.stop()
            _WATCHDOG.fallback = None
    del ctx
    if world == 1 and not args.no_extras and args.workload == 'cfg2' and args.feed == 'resident' and not args.no_graph:
        # next to the headline (inputs resident in HBM, stratified batch): (1) the same step fed the way the
        # reference's own pipeline feeds it -- WeightedRandomSampler batches of any group mix, drawn from an
        # HBM-resident dataset by the captured step itself (universal plan); (2) the other configurations of
        # BASELINE.json, a few steps each, with their own roofline block
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        # (the legs below are additions to the contract line: one that fails is reported in its own entry and on
        # stderr, the headline above still goes out)
        try:
            r2, c2 = measure(args, 'cfg2', 'sampler', 400, 20, device, rank, world)
            out['realistic_feed'] = {'feed': 'sampler', 'ms_per_step': r2['ms_per_step'], 'value': r2['value'],
                                     'steps': r2['steps'], 'finite': r2['finite'],
                                     'plans': 'pairs bucketed by %d, labeled range by %s' % (args.pair_bucket, 'a third of the batch'
                                                                                          if args.label_bucket < 0 else args.label_bucket),
                                     'what': 'exact WeightedRandomSampler batches (any mix of pairs / labels per batch) gathered '
                                             'by the captured step from a %d-row HBM-resident dataset' % args.dataset_rows}
            ok = ok and r2['finite']
            del c2
        except Exception as e:       # noqa: BLE001
            print('bench.py: realistic_feed leg failed: %r' % (e,), file=sys.stderr)
            out['realistic_feed'] = {'feed': 'sampler', 'error': repr(e)}
            out.setdefault('extras_failed', []).append('realistic_feed')
        gc.collect()
        torch.cuda.empty_cache()
        try:
            out['fit_epoch'] = fit_epoch(device)
            ok = ok and out['fit_epoch']['finite']
        except Exception as e:       # noqa: BLE001
            print('bench.py: fit_epoch leg failed: %r' % (e,), file=sys.stderr)
            out['fit_epoch'] = {'error': repr(e)}
            out.setdefault('extras_failed', []).append('fit_epoch')
        out['other_workloads'] = {}
        for wl, (k_, w_) in (('wide', (10, 3)), ('cfg1', (200, 20)), ('cfg4', (200, 20))):
            gc.collect()
            torch.cuda.empty_cache()
            try:
                r3, c3 = measure(args, wl, 'resident', k_, w_, device, rank, world)
            except Exception as e:       # noqa: BLE001
                print('bench.py: other_workloads leg %s failed: %r' % (wl, e), file=sys.stderr)
                out['other_workloads'][{'wide': 'cfg5'}.get(wl, wl)] = {'error': repr(e)}
                out.setdefault('extras_failed', []).append(wl)
                continue
            entry = {'ms_per_step': r3['ms_per_step'], 'value': r3['value'], 'steps': k_, 'warmup': w_,
                     'finite': r3['finite'], 'workload': r3['config']['workload'],
                     'side_chain_cus': r3['config']['side_chain_cus']}
            if not args.no_roofline:
                b3 = c3['batch']
                entry['roofline'] = gemm_roofline(c3['eng'], c3['cfg'], c3['rows'], float(b3['has_x2'].astype(bool).mean()),
                                                  float(b3['has_y'].astype(bool).mean()), repeats=3 if wl == 'wide' else 20,
                                                  workload=wl, n_params=r3['config']['params'],
                                                  input_bytes=4.0 * c3['rows'] * c3['cfg'].dim_x * (2 if c3['cfg'].has_pert else 1),
                                                  ms_per_step=r3['ms_per_step'], brief=True)
            name = {'wide': 'cfg5'}.get(wl, wl)
            out['other_workloads'][name] = entry
            if 'roofline' in entry and 'roofline' in out:
                # the driver keeps `roofline` whole: every configuration's step-level figure rides in it, same definition
                rf = entry['roofline']
                out['roofline'].setdefault('workloads', {})[name] = {
                    'ms_per_step': r3['ms_per_step'], 'steps': k_, 'frac': rf['frac'], 'achieved': rf['achieved'],
                    'algorithmic_gflop_per_step': rf['algorithmic_gflop_per_step'],
                    'isolated_frac': rf['isolated']['frac'], 'value': r3['value'], 'traffic': rf.get('traffic')}
            ok = ok and r3['finite']
            del c3
        # cfg 4 WITH its MMD loss path (BASELINE.json configs[3]): the penalty inside the captured step, and the
        # block-level add-on by itself
        gc.collect()
        torch.cuda.empty_cache()
        try:
            r4, c4 = measure(args, 'cfg4_mmd', 'resident', 20, 5, device, rank, world)
            out['other_workloads']['cfg4_mmd'] = {
                'ms_per_step': r4['ms_per_step'], 'value': r4['value'], 'steps': 20, 'warmup': 5, 'finite': r4['finite'],
                'workload': r4['config']['workload'], 'side_chain_cus': r4['config']['side_chain_cus'],
                'mmd_last_step': r4['losses_last_step'].get('MMD'), 'block_addon': mmd_addon(device)}
            ok = ok and r4['finite']
            del c4
        except Exception as e:       # noqa: BLE001
            print('bench.py: other_workloads leg cfg4_mmd failed: %r' % (e,), file=sys.stderr)
            out['other_workloads']['cfg4_mmd'] = {'error': repr(e)}
            out.setdefault('extras_failed', []).append('cfg4_mmd')
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    if world > 1 or (dp and dist.is_initialized()):
        dist.destroy_process_group()
    if rank == 0:
        try:        # RCCL writes a version banner through C stdio: flush it out FIRST, the JSON line stays the last one
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if args.strict and out.get('extras_failed'):
        return 3        # (the headline line is out; a leg next to it raised: see its 'error' entry)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
