#!/usr/bin/env python3
"""Timeline of ONE hipGraph-replayed train step from a rocprofv3 --kernel-trace CSV:
timeline.py <..._kernel_trace.csv> [n_kernels_per_step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a step starts at a fill_normal launch; take the last complete one
starts = [i for i, n in enumerate(names) if 'fill_normal' in n]
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
