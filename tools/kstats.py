#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('%s: total kernel time %.3f ms over %d kernels' % (f, tot / 1e6, len(rows)))
for r in rows[:n]:
    print('%-100s calls %6s  total %9.3f ms  avg %9.1f us  %5.1f%%' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                                       float(r['AverageNs']) / 1e3, float(r['Percentage'])))
