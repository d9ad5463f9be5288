"""Kernel trace CSV of a multi-stream run: wall span, union of busy intervals (some kernel running), sum of durations, per-stream sums."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'], r['Kernel_Name'].split('(')[0][:40]) for r in rows)
lo = int(float(sys.argv[2]) * 1e6) + iv[0][0] if len(sys.argv) > 2 else iv[0][0]
hi = int(float(sys.argv[3]) * 1e6) + iv[0][0] if len(sys.argv) > 3 else max(e for _, e, _, _ in iv)
iv = [x for x in iv if x[0] >= lo and x[1] <= hi]
span = iv[-1][1] - iv[0][0]
union = 0; cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, _, _ in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in iv)
per = collections.Counter()
for s, e, st, _ in iv:
    per[st] += e - s
print('span %.1f ms  union busy %.1f ms (%.0f%%)  sum of durations %.1f ms  overlap factor %.2f' % (span / 1e6, union / 1e6, 100.0 * union / span, tot / 1e6, tot / union))
for k, v in per.most_common(8):
    print('  stream %s: %.1f ms' % (k, v / 1e6))
