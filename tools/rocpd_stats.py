#!/usr/bin/env python3
"""Per-kernel statistics (calls, total/avg/min/max ns, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats` writes *_results.db by default on ROCm 7.2).  Usage: rocpd_stats.py DB [OUT.csv]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
    scols = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
    name_col = 'display_name' if 'display_name' in scols else ('kernel_name' if 'kernel_name' in scols else scols[1])
    q = 'select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s' % (name_col, kd, ks, name_col)
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for name, n, t, mn, mx in rows:
        short = re.sub(r'\s+', ' ', name)
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (short, n, t, t / n, 100.0 * t / tot, mn, mx))
    out = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out)
    sys.stdout.write(out)


if __name__ == '__main__':
    main()
