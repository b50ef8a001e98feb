#!/usr/bin/env python3
"""idle intervals of the GPU (no kernel of any queue running) during the LAST factorisation of a rocprofv3 --kernel-trace csv,
and of its main queue: python3 tools/trace_idle.py kernel_trace.csv [min_gap_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('save_diag_kernel')]
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('logdiag_kernel')]
a, b = starts[-1], [e for e in ends if e > starts[-1]][0]
rows = rows[a:b + 1]
t0 = int(rows[0]['Start_Timestamp'])
iv = sorted((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Queue_Id'], r['Kernel_Name'][:36]) for r in rows)
total = iv[-1][1]
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
# union of busy intervals over all queues
cur_end, idle, gaps = 0, 0, []
for s, e, q, n in iv:
    if s > cur_end:
        idle += s - cur_end
        if s - cur_end >= min_gap:
            gaps.append((cur_end, s - cur_end, n))
    cur_end = max(cur_end, e)
print('factorisation %.2f ms; no kernel running on any queue for %.2f ms in all' % (total / 1e6, idle / 1e6))
for at, g, n in gaps:
    print('  idle %.1f us at %.2f ms (before %s)' % (g / 1e3, at / 1e6, n))
# big-gemm queue: the queue with the most busy time
busy = {}
for s, e, q, n in iv:
    busy[q] = busy.get(q, 0) + e - s
mainq = max(busy, key=busy.get)
mi = [(s, e, n) for s, e, q, n in iv if q == mainq]
idle_main = sum(max(0, mi[i + 1][0] - mi[i][1]) for i in range(len(mi) - 1))
print('queue %s (most busy): busy %.2f ms, idle between its kernels %.2f ms' % (mainq, busy[mainq] / 1e6, idle_main / 1e6))
for i in range(len(mi) - 1):
    g = mi[i + 1][0] - mi[i][1]
    if g >= min_gap:
        print('  gap %.1f us at %.2f ms, between %s and %s' % (g / 1e3, mi[i][1] / 1e6, mi[i][2], mi[i + 1][2]))
