#!/usr/bin/env python3
"""Kernel sequence of ONE replayed whole-set evaluation from a rocprofv3 --kernel-trace CSV of tools/eval_bench.py:
eval_timeline.py <..._kernel_trace.csv> [train|valid]  -- a pass starts at its Philox draw
(fill_normal_rows_kernel)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
which = sys.argv[2] if len(sys.argv) > 2 else 'valid'
starts = [i for i, r in enumerate(rows) if 'fill_normal_rows' in r['Kernel_Name']]
# tools/eval_bench.py: per dataset 2 eager passes (plan building, warm-up) + (2 + reps) replays, train set first
half = len(starts) // 2
k = half - 1 if which == 'train' else len(starts) - 2
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]['Start_Timestamp'])
busy = 0
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print('%9.1f us  gap %6.1f  dur %8.1f us  grid=%-9s %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                                                           r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Kernel_Name'][:110]))
    prev_end = e
print('pass: %d kernels, %.1f us wall, %.1f us busy' % (b - a, (prev_end - t0) / 1e3, busy / 1e3))
