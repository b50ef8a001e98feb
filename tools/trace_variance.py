#!/usr/bin/env python3
"""per-queue timeline of the LAST variance solve in a rocprofv3 --kernel-trace csv of bench.py (between the last backward-sweep
launch of the solve and colsq_kernel): python3 tools/trace_variance.py kernel_trace.csv [rows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('colsq_kernel')]
last = ends[-1]
begin = max(i for i, r in enumerate(rows[:last]) if r['Kernel_Name'].startswith('trsv_bwd'))
rows = rows[begin + 1:last + 1]
t0 = int(rows[0]['Start_Timestamp'])
per = collections.defaultdict(list)
for r in rows:
    per[r['Queue_Id']].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name'][:60]))
end = max(e for v in per.values() for _, e, _ in v)
print('variance: %.1f us, %d kernels' % (end / 1e3, len(rows)))
for q, v in per.items():
    busy = sum(e - s for s, e, _ in v)
    gaps = sum(max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1))
    names = collections.Counter(n for _, _, n in v)
    print('queue %s: %d kernels, busy %.1f us, idle between its kernels %.1f us; %s' % (q, len(v), busy / 1e3, gaps / 1e3, ', '.join('%s x%d' % kv for kv in names.most_common(5))))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 0
allk = sorted((s, e, q, n) for q, v in per.items() for s, e, n in v)
for s, e, q, n in allk[:lim]:
    print('%9.1f %9.1f  q%s  %s' % (s / 1e3, (e - s) / 1e3, q, n))
