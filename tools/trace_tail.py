#!/usr/bin/env python3
"""the last `ms` milliseconds before the LAST logdiag-free end of a factorisation in a rocprofv3 --kernel-trace csv: per-queue
busy time and the kernels (developer probe): python3 tools/trace_tail.py kernel_trace.csv end_kernel_prefix ms [rows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prefix, ms = sys.argv[2], float(sys.argv[3])
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(prefix)]
last = ends[-1]
t_end = int(rows[last]['End_Timestamp'])
sel = [r for r in rows[:last + 1] if int(r['End_Timestamp']) >= t_end - ms * 1e6]
t0 = t_end - ms * 1e6
per = collections.defaultdict(list)
for r in sel:
    per[r['Queue_Id']].append((max(int(r['Start_Timestamp']), t0) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name'][:50]))
for q, v in per.items():
    busy = sum(e - s for s, e, _ in v)
    names = collections.Counter(n for _, _, n in v)
    tot = collections.Counter()
    for s, e, n in v:
        tot[n] += e - s
    print('queue %s: %d kernels, busy %.2f of %.1f ms; %s' % (q, len(v), busy / 1e6, ms, ', '.join('%s x%d %.2fms' % (n, names[n], tot[n] / 1e6) for n, _ in tot.most_common(8))))
lim = int(sys.argv[4]) if len(sys.argv) > 4 else 0
allk = sorted((s, e, q, n) for q, v in per.items() for s, e, n in v)
for s, e, q, n in allk[-lim:] if lim else []:
    print('%9.1f %9.1f  q%s  %s' % (s / 1e3, (e - s) / 1e3, q, n))
