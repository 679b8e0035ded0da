#!/usr/bin/env python3
"""Per-kernel, per-step summary of the rocprofv3 --pmc passes written by tools/pmc_collect.sh."""
import csv
import gzip
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_r1'
wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'      # the workload the passes ran (tools/pmc_collect.sh <workload>)


def load(sub):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    steps = 0
    with gzip.open('%s/%s/p_counter_collection.csv.gz' % (root, sub), 'rt') as fh:
        seen = set()
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            key = (row['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                calls[k] += 1
                steps += 'fill_normal' in k
    return acc, calls, max(steps, 1)


def durations():
    """average kernel duration (us) from the --kernel-trace --stats run of the same command (tools/pmc_collect.sh)"""
    import os
    out = {}
    path = '%s/kernel_stats.csv' % root
    if not os.path.exists(path):
        return out
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            c, t = int(row['Calls']), float(row['TotalDurationNs'])
            if k in out:
                c, t = c + out[k][0], t + out[k][1]
            out[k] = (c, t)
    return {k: v[1] / max(v[0], 1) / 1e3 for k, v in out.items()}


dur = durations()
fetch, calls, steps = load('fetch')
write, _, s2 = load('write')
sq, _, s3 = load('sq')
tcc, _, s4 = load('tcc')
print('# rocprofv3 --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE | 8 SQ counters | TCC_HIT/MISS) over')
print('#   python3 bench.py --workload %s %s --no-cpu-baseline --no-roofline --no-steady --no-extras     (tools/pmc_collect.sh %s;' % (wl, '--steps 3 --warmup 1' if wl == 'wide' else '--steps 20 --warmup 5', wl))
print('#   single-graph schedule: counter collection serializes dispatches).  Per-STEP averages over %d steps.' % steps)
print('# FETCH_SIZE/WRITE_SIZE in KB as reported; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts')
print('# 1/2 of the bytes of wide (16 B/lane) coalesced reads.')
print('# ldsConfl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS-array cycles per LDS cycle); mfma/wave = SQ_VALU_MFMA_BUSY_CYCLES /')
print('# (4 x SQ_WAVE_CYCLES): matrix-pipe cycles per cycle of wave lifetime (SQ_WAVE_CYCLES counts quad-cycles)')
print('# avg_us = average kernel duration of a --kernel-trace --stats run of the same command (no counters attached); GB/s = (2 x FETCH + WRITE)')
print('# per call / avg_us: memory-side bytes with the x2 wide-load correction (an upper bound; exact for 16-B-per-lane streams)')
print('# mfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES per call / (avg_us x 2.4 GHz x 1024 SIMDs): the share of the chip\'s matrix-pipe cycles the kernel')
print('# keeps busy over its duration (nominal 2.4 GHz: under sustained load the clock is lower, so this understates a little)')
print('%-46s %6s %10s %10s %6s %8s %8s %8s %12s %9s %9s %9s %8s %8s' % ('kernel', 'calls', 'FETCH_KB', 'WRITE_KB', 'L2hit', 'waitAny', 'waitInst',
                                                   'active', 'mfmaBusyCyc', 'ldsConfl', 'mfma/wave', 'avg_us', 'GB/s', 'mfmaUtil'))
tot_f = tot_w = gf = gw = gc = 0
for k in sorted(fetch, key=lambda k: -fetch[k]['FETCH_SIZE']):
    f, w = fetch[k]['FETCH_SIZE'] / steps, write[k]['WRITE_SIZE'] / s2
    h, m = tcc[k]['TCC_HIT_sum'], tcc[k]['TCC_MISS_sum']
    wc = max(sq[k]['SQ_WAVE_CYCLES'], 1)
    cps = max(calls[k] / steps, 1e-9)
    d_us = dur.get(k, 0.0)
    gbs = (2 * f + w) / cps * 1e3 / (d_us * 1e-6) / 1e9 if d_us else 0.0
    mutil = (sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / s3 / cps) / (d_us * 1e-6 * 2.4e9 * 1024) if d_us else 0.0
    print('%-46s %6.1f %10.0f %10.0f %6.2f %8.2f %8.2f %8.2f %12.0f %9.3f %9.3f %9.1f %8.0f %8.3f' % (
        k[:46], calls[k] / steps, f, w, h / max(h + m, 1), sq[k]['SQ_WAIT_ANY'] / wc, sq[k]['SQ_WAIT_INST_ANY'] / wc,
        sq[k]['SQ_ACTIVE_INST_ANY'] / wc, sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / s3,
        sq[k]['SQ_LDS_BANK_CONFLICT'] / max(sq[k]['SQ_LDS_IDX_ACTIVE'], 1), sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc), d_us, gbs, mutil))
    tot_f += f
    tot_w += w
    if 'gemm' in k:
        gf += f
        gw += w
        gc += calls[k] / steps
alg = {'cfg2': 'algorithmic ~94 MB (SURVEY 8(d))',
       'wide': 'algorithmic ~5.1 GB (124.5 M parameters x 10 words: read forward and backward, gradient, Adam\'s 7; + 0.12 GB of inputs)'}.get(wl, '')
print('# TOTAL per step: FETCH %.1f MB (reported; <= %.1f MB after the x2 wide-load correction), WRITE %.1f MB; %s'
      % (tot_f / 1e3, 2 * tot_f / 1e3, tot_w / 1e3, alg))
print('# GEMM family per step: %.0f launches, FETCH %.1f MB reported -> %.1f MB corrected, WRITE %.1f MB => %.2f MB '
      'HBM-side traffic per launch' % (gc, gf / 1e3, 2 * gf / 1e3, gw / 1e3, (2 * gf + gw) / 1e3 / max(gc, 1)))
lds = sum(v.get('SQ_LDS_BANK_CONFLICT', 0) for v in sq.values())
print('# SQ_LDS_BANK_CONFLICT summed over all kernels: %.0f' % lds)

# ---- optional memory-side pass (PMC_EXTRA of tools/pmc_collect.sh): L2 -> fabric read requests, how many of them are addressed to
# DRAM (as opposed to GMI / IO), and the average time a read request is outstanding (LEVEL / RDREQ, in L2 clocks): an Infinity-Cache
# hit returns sooner than an HBM access, so a kernel whose operands the Infinity Cache serves shows the latency of the cfg-2 optimiser
# sweep (65 MB arena, resident), one that streams from HBM that of the wide configuration's sweep (3.5 GB)
import os
if os.path.exists('%s/extra0/p_counter_collection.csv.gz' % root):
    ex, ecalls, es = load('extra0')
    print()
    print('# memory-side pass: TCC_EA0_RDREQ (all L2 -> fabric read requests), _32B (of which 32-byte), _DRAM (addressed to the memory')
    print('# controllers), avg outstanding = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ in L2 clocks.  Per-STEP averages over %d steps.' % es)
    print('%-46s %6s %12s %12s %12s %14s' % ('kernel', 'calls', 'RDREQ', 'RDREQ_32B', 'RDREQ_DRAM', 'avg_outst_clk'))
    for k in sorted(ex, key=lambda k: -ex[k].get('TCC_EA0_RDREQ_sum', 0)):
        r = ex[k].get('TCC_EA0_RDREQ_sum', 0)
        print('%-46s %6.1f %12.0f %12.0f %12.0f %14.1f' % (k[:46], ecalls[k] / es, r / es, ex[k].get('TCC_EA0_RDREQ_32B_sum', 0) / es,
                                                    ex[k].get('TCC_EA0_RDREQ_DRAM_sum', 0) / es,
                                                    ex[k].get('TCC_EA0_RDREQ_LEVEL_sum', 0) / max(r, 1)))
