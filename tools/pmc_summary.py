#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: dispatches, mean and total of each counter.
Usage: pmc_summary.py COUNTER_CSV [OUT.csv]"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
with open(sys.argv[1], newline='') as f:
    rd = csv.DictReader(f)
    for row in rd:
        name = re.sub(r'\(.*', '', row.get('Kernel_Name', row.get('Name', '?')))
        cname, val = row.get('Counter_Name'), row.get('Counter_Value')
        if cname is None:
            continue
        a = acc[name][cname]
        a[0] += 1
        a[1] += float(val)
lines = ['"Kernel","Counter","Dispatches","Total","MeanPerDispatch"']
for k in sorted(acc, key=lambda k: -max(v[1] for v in acc[k].values())):
    for c, (n, t) in acc[k].items():
        lines.append('"%s","%s",%d,%.6g,%.6g' % (k, c, n, t, t / n))
out = '\n'.join(lines) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out)
sys.stdout.write(out[:6000])
