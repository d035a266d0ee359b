#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per dispatch for
each kernel (only dispatches with the largest grid of that kernel)."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + '/*/*counter_collection.csv') or glob.glob(path + '/*counter_collection.csv')
rows = list(csv.DictReader(open(files[0])))
big = collections.defaultdict(int)
for r in rows:
    big[r['Kernel_Name']] = max(big[r['Kernel_Name']], int(r['Grid_Size']))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if int(r['Grid_Size']) == big[r['Kernel_Name']]:
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if k.startswith('__amd'):
        continue
    print(k[:90], ' grid', big[k])
    for c, v in sorted(d.items()):
        print('    %-26s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
