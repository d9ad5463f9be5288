#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel and segment: dispatches, mean and total of each counter.  A launch
whose name contains "BitwiseXor" (tools/pmc_probe.py's marker) starts the next segment.
Usage: pmc_summary.py COUNTER_CSV [OUT.csv]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline='') as f:
    for row in csv.DictReader(f):
        if row.get('Counter_Name') is None:
            continue
        did = row.get('Dispatch_Id') or row.get('Dispatch_ID') or '0'
        rows.append((int(did), row.get('Kernel_Name', row.get('Name', '?')), row['Counter_Name'], float(row['Counter_Value'])))
rows.sort(key=lambda r: r[0])
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
seg, last_marker = 0, None
for did, full, cname, val in rows:
    if 'BitwiseXor' in full or 'bitwise_xor' in full:
        if did != last_marker:
            seg += 1
            last_marker = did
        continue
    a = acc[(seg, re.sub(r'\(.*', '', full))][cname]
    a[0] += 1
    a[1] += val
lines = ['"Kernel","Counter","Dispatches","Total","MeanPerDispatch","Segment"']
for k in sorted(acc, key=lambda k: -max(v[1] for v in acc[k].values())):
    for c, (n, t) in acc[k].items():
        lines.append('"%s","%s",%d,%.6g,%.6g,%d' % (k[1], c, n, t, t / n, k[0]))
out = '\n'.join(lines) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out)
sys.stdout.write(out[:6000])
