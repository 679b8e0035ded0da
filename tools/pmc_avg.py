#!/usr/bin/env python3
"""Average the counters of rocprofv3 counter_collection CSVs per kernel: pmc_avg.py <csv>... [--match substr]"""
import csv
import gzip
import sys
from collections import defaultdict

match = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == '--match':
        match = args.pop(0)
    else:
        files.append(a)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    op = gzip.open if f.endswith('.gz') else open
    with op(f, 'rt') as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].split('(')[0][:70]
            if match and match not in row['Kernel_Name']:
                continue
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]            # skip the cold first half
        print('   %-34s %16.1f   (n=%d)' % (c, sum(v) / len(v), len(v)))
