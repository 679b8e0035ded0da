#!/usr/bin/env python3
"""Derive the roofline block of the benchmark from the COMMITTED profiler output, so that every figure in it
can be recomputed from profiles/ by anyone (no constants baked into bench.py):

    python tools/roofline_from_profile.py profiles/r02_cfg2 [--out profiles/r02_cfg2_roofline.json]

reads   <prefix>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-roofline --no-cpu-baseline`
                                    (in situ: both launch chains running, main chain on its CU partition)
        <prefix>_pmc_summary.txt    tools/pmc_summary.py over the separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ | TCC)
        <prefix>_bench.json         the bench line of the same build (algorithmic FLOPs / bytes of the step)
writes  a JSON with
  gemm_in_graph : launches and kernel time of the GEMM family per step as profiled, achieved TFLOP/s = algorithmic
                  GEMM FLOPs per step / that time, frac of the fp32-MFMA peak
  traffic       : memory-side bytes per step (FETCH_SIZE as reported and with the gfx950 x2 wide-load correction as
                  an upper bound, WRITE_SIZE), algorithmic bytes, wasted ratio; the same for the GEMM family alone
                  and per GEMM launch
`bench.py` loads that file for `roofline.traffic` / `roofline.in_graph`.
"""
import argparse
import csv
import json
import os
import re
import sys

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 peak


def kernel_stats(path):
    """-> {short kernel name: (calls, total_ns, avg_ns)}"""
    out = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            short = name.split('(')[0]
            c, t = int(row['Calls']), float(row['TotalDurationNs'])
            if short in out:
                c, t = c + out[short][0], t + out[short][1]
            out[short] = (c, t, t / max(c, 1))
    return out


def _isnum(v):
    try:
        float(v)
        return True
    except ValueError:
        return False


def pmc_table(path):
    """-> ({kernel: dict(calls, fetch_kb, write_kb, l2hit)}, steps) from tools/pmc_summary.py's table"""
    rows, steps = {}, None
    with open(path) as fh:
        for line in fh:
            m = re.search(r'Per-STEP averages over (\d+) steps', line)
            if m:
                steps = int(m.group(1))
            if line.startswith('#') or line.startswith('kernel ') or not line.strip():
                continue
            # the kernel name may contain spaces: the numeric columns are the last eight fields
            parts = line.rstrip('\n').split()
            if len(parts) < 9:
                continue
            # (round 4 added two trailing columns: ldsConfl, mfma/wave; round 6 two more: avg_us, GB/s)
            ncol = 0
            while ncol < len(parts) - 1 and _isnum(parts[-1 - ncol]):
                ncol += 1
            if ncol < 8:
                continue
            try:
                nums = [float(v) for v in parts[-ncol:]][:8]
            except ValueError:
                continue
            name = ' '.join(parts[:-ncol])
            rows[name] = dict(calls=nums[0], fetch_kb=nums[1], write_kb=nums[2], l2hit=nums[3], wait_any=nums[4],
                              wait_inst=nums[5], active=nums[6], mfma_busy_cycles=nums[7])
    return rows, steps


def is_gemm(name):
    return 'gemm' in name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('prefix', help='e.g. profiles/r02_cfg2')
    ap.add_argument('--out', default=None)
    ap.add_argument('--commit', default=None, help='commit the profile was collected at (recorded in the JSON)')
    ap.add_argument('--step-kernel', default='fill_normal_rows_kernel',
                    help='a kernel launched exactly once per train step (counts the profiled steps)')
    args = ap.parse_args()
    pre = args.prefix
    out = {'source': {}, 'commit': args.commit}
    bench = None
    bj = pre + '_bench.json'
    if os.path.exists(bj):
        with open(bj) as fh:
            for line in fh:
                if line.startswith('{'):
                    bench = json.loads(line)
        out['source']['bench'] = os.path.basename(bj)
    rl = (bench or {}).get('roofline', {})
    alg_gflop = rl.get('algorithmic_gflop_per_step')
    alg_mb = rl.get('algorithmic_mbytes_per_step')
    alg_gemm_mb = rl.get('algorithmic_gemm_mbytes_per_step')

    ks = pre + '_kernel_stats.csv'
    if os.path.exists(ks):
        st = kernel_stats(ks)
        step_calls = [v[0] for k, v in st.items() if args.step_kernel in k]
        if not step_calls:      # (profiles of builds before the row-keyed Philox)
            step_calls = [v[0] for k, v in st.items() if k.startswith('fill_normal')]
        steps = max(step_calls) if step_calls else None
        g = {k: v for k, v in st.items() if is_gemm(k)}
        if steps:
            t_us = sum(v[1] for v in g.values()) / steps / 1e3
            n = sum(v[0] for v in g.values()) / steps
            blk = {'steps_profiled': steps, 'launches_per_step': round(n, 2), 'kernel_us_per_step': round(t_us, 2),
                   'avg_launch_us': round(t_us / max(n, 1e-9), 3),
                   'per_kernel': {k: {'calls_per_step': round(v[0] / steps, 2), 'avg_us': round(v[2] / 1e3, 3)}
                                  for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])}}
            if alg_gflop:
                ach = alg_gflop * 1e9 / (t_us * 1e-6) / 1e12
                blk.update({'algorithmic_gflop_per_step': alg_gflop, 'achieved_tflops': round(ach, 2),
                            'peak_tflops': FP32_MFMA_PEAK_TFLOPS, 'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4)})
            out['gemm_in_graph'] = blk
            all_us = sum(v[1] for k, v in st.items() if 'flag_wait' not in k) / steps / 1e3
            out['all_kernels_us_per_step'] = round(all_us, 1)
            out['launches_per_step'] = round(sum(v[0] for k, v in st.items()
                                                 if not k.startswith('at::') and '__amd' not in k) / steps, 1)
        out['source']['kernel_stats'] = os.path.basename(ks)

    ps = pre + '_pmc_summary.txt'
    if os.path.exists(ps):
        rows, psteps = pmc_table(ps)
        f = sum(r['fetch_kb'] for r in rows.values()) / 1e3
        w = sum(r['write_kb'] for r in rows.values()) / 1e3
        gf = sum(r['fetch_kb'] for k, r in rows.items() if is_gemm(k)) / 1e3
        gw = sum(r['write_kb'] for k, r in rows.items() if is_gemm(k)) / 1e3
        gn = sum(r['calls'] for k, r in rows.items() if is_gemm(k))
        tr = {'steps_profiled': psteps, 'unit': 'MB per step',
              'fetch_reported': round(f, 1), 'fetch_corrected_upper': round(2 * f, 1), 'write': round(w, 1),
              'total_reported': round(f + w, 1), 'total_corrected_upper': round(2 * f + w, 1),
              'note': 'FETCH_SIZE / WRITE_SIZE are the L2s\' memory-side request counters (Infinity-Cache hits included); '
                      'on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads, so the true figure lies between '
                      '"reported" and "corrected_upper" (MI355X_MICROARCH.md, HBM)',
              'gemm': {'launches_per_step': round(gn, 1), 'fetch_reported': round(gf, 1),
                       'fetch_corrected_upper': round(2 * gf, 1), 'write': round(gw, 1),
                       'per_launch_corrected_upper_mb': round((2 * gf + gw) / max(gn, 1e-9), 2),
                       'per_launch_reported_mb': round((gf + gw) / max(gn, 1e-9), 2)}}
        if alg_mb:
            tr['algorithmic'] = alg_mb
            tr['wasted_ratio_reported'] = round((f + w) / alg_mb, 2)
            tr['wasted_ratio_upper'] = round((2 * f + w) / alg_mb, 2)
        if alg_gemm_mb:
            tr['gemm']['algorithmic'] = alg_gemm_mb
            tr['gemm']['wasted_ratio_reported'] = round((gf + gw) / alg_gemm_mb, 2)
            tr['gemm']['wasted_ratio_upper'] = round((2 * gf + gw) / alg_gemm_mb, 2)
        out['traffic'] = tr
        out['source']['pmc_summary'] = os.path.basename(ps)

    text = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(text + '\n')
    print(text)


if __name__ == '__main__':
    sys.exit(main())
