"""Timeline of the LAST tn_site_qr call of tools/pivqr_probe.py from a rocprofv3 kernel trace: every launch with its duration and the gap
in front of it (us).  Usage: piv_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r['Kernel_Name'].startswith(('tn::', 'void tn::', '__amd_rocclr'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call starts at the last pivot_init_kernel (device mode) or at the third-from-last group of colnorm2 launches
idx = [i for i, r in enumerate(rows) if 'pivot_init_kernel' in r['Kernel_Name']]
start = idx[-1] if idx else max(0, len(rows) - 80)
prev = None
for r in rows[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('tn::', '').split('(')[0][:44]
    print('%-46s dur %7.2f  gap %7.2f  grid %s' % (name, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get('Grid_Size', '?')))
    prev = e
print('span %.1f us' % ((int(rows[-1]['End_Timestamp']) - int(rows[start]['Start_Timestamp'])) / 1e3))
