#!/usr/bin/env python3
"""per-stream timeline of the LAST factorisation in a rocprofv3 --kernel-trace csv: python3 tools/trace_gaps.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last save_diag_kernel marks the start of the last factorisation
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('save_diag_kernel')]
rows = rows[starts[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
per = collections.defaultdict(list)
for r in rows:
    per[r['Queue_Id']].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name'][:40]))
end = max(e for v in per.values() for _, e, _ in v)
print('factorisation: %.1f us, %d kernels' % (end / 1e3, len(rows)))
for q, v in per.items():
    busy = sum(e - s for s, e, _ in v)
    names = collections.Counter(n for _, _, n in v)
    print('queue %s: %d kernels, busy %.1f us; %s' % (q, len(v), busy / 1e3, ', '.join('%s x%d' % kv for kv in names.most_common(4))))
if len(sys.argv) > 2:
    lim = int(sys.argv[2])
    allk = sorted((s, e, q, n) for q, v in per.items() for s, e, n in v)
    for s, e, q, n in allk[:lim]:
        print('%9.1f %9.1f  q%s  %s' % (s / 1e3, (e - s) / 1e3, q, n))
