"""Summarise a rocprofv3 kernel trace CSV: per kernel name count / mean / median us, and the mean gap between consecutive dispatches."""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
by = {}
prev_end = None
seq = []
for r in rows:
    name = r['Kernel_Name'].split('(')[0][:60]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if pat and pat not in name:
        prev_end = e
        continue
    by.setdefault(name, []).append((e - s) / 1e3)
    seq.append((name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
for k, v in by.items():
    print('%-60s n=%5d mean %.2f median %.2f min %.2f max %.2f us' % (k, len(v), st.mean(v), st.median(v), min(v), max(v)))
if len(sys.argv) > 3:
    for x in seq[-int(sys.argv[3]):]:
        print('%-40s %.2f us  gap %.2f' % x)
