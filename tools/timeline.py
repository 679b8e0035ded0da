#!/usr/bin/env python3
"""Timeline of ONE hipGraph-replayed train step from a rocprofv3 --kernel-trace CSV:
timeline.py <..._kernel_trace.csv> [n_kernels_per_step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# one period of the replayed step: from one input gather of the main chain to the next (the Philox draw is no
# anchor any more: it belongs to the previous step's side chain); take the last complete one before the tail
gath = [i for i, n in enumerate(names) if 'batch_feed' in n]            # (graph-resident feeds)
if len(gath) < 8:
    gath = [i for i, n in enumerate(names) if 'rows_gather' in n or 'batch_feed' in n]
grids = [rows[i].get('Grid_Size', rows[i].get('Grid_Size_X', '0')) for i in gath]
big = max(grids, key=lambda g: int(g)) if grids else '0'
starts = [i for i, g in zip(gath, grids) if g == big][:-3]
# (feed-ahead: two feed launches per step -- the side chain's, which feeds, and the main chain's check; a period is from one
# feed launch on the main chain's queue to the next one there)
q_main = rows[gath[0]]['Queue_Id'] if gath else None
if starts and len(set(rows[i]['Queue_Id'] for i in starts)) > 1:
    cnt = {}
    for i in starts:
        cnt[rows[i]['Queue_Id']] = cnt.get(rows[i]['Queue_Id'], 0) + 1
    # the main chain's queue is the one that also runs the optimiser launch right in front of its feed launch
    for q in cnt:
        idx = [i for i in starts if rows[i]['Queue_Id'] == q]
        prev = [r['Kernel_Name'] for r in rows[:idx[-1]] if r['Queue_Id'] == q][-1:]
        if prev and 'adam' in prev[0]:
            starts = idx
            break
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
end = max(int(r['End_Timestamp']) for r in step)
print('# step of %d kernels, span %.1f us' % (len(step), (end - t0) / 1e3))
qs = sorted(set(r['Queue_Id'] for r in step))
busy = 0
for r in step:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%8.1f us  dur=%7.1f us  q=%d  grid=%-8s %s' % (s / 1e3, (e - s) / 1e3, qs.index(r['Queue_Id']),
                                                       r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Kernel_Name'][:90]))
