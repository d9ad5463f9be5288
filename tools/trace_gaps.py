"""Kernel trace CSV -> busy time, idle time, and idle time attributed to the kernel that FOLLOWS each gap (who is launched late)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
busy = 0; idle = collections.Counter(); cnt = collections.Counter(); dur = collections.Counter()
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][:50]
    if s > prev_end:
        idle[name] += s - prev_end
    busy += max(0, e - max(s, prev_end))
    prev_end = max(prev_end, e)
    cnt[name] += 1; dur[name] += e - s
print('span %.1f ms  busy %.1f ms  idle %.1f ms  kernels %d' % ((t1 - t0) / 1e6, busy / 1e6, sum(idle.values()) / 1e6, len(rows)))
print('%-52s %7s %9s %9s %9s' % ('kernel', 'n', 'dur ms', 'idle-before ms', 'avg gap us'))
for k, v in sorted(dur.items(), key=lambda kv: -(kv[1] + idle[kv[0]]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print('%-52s %7d %9.1f %9.1f %9.2f' % (k, cnt[k], v / 1e6, idle[k] / 1e6, idle[k] / 1e3 / cnt[k]))
