#!/usr/bin/env python3
"""summarise a rocprofv3 rocpd database (kernel-trace): per-kernel count / average / total, and for the LAST
factorisation in the trace the span against the sum of kernel durations.  usage: prof_db.py results.db [csv_out]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = cur.execute('select k.start, k.end, s.kernel_name, k.grid_size_x, k.workgroup_size_x from %s k join %s s on k.kernel_id=s.id order by k.start' % (kd, ks)).fetchall()
dur = collections.defaultdict(list)
for st, en, nm, gx, wx in rows:
    dur[nm.split('(')[0][:70]].append((en - st) / 1e3)
total = sum(sum(v) for v in dur.values())
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage"']
for nm, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('  %-72s n=%6d avg %9.2f us total %9.2f ms' % (nm, len(v), sum(v) / len(v), sum(v) / 1e3))
    lines.append('"%s",%d,%d,%.1f,%.2f' % (nm, len(v), int(sum(v) * 1e3), sum(v) / len(v) * 1e3, 100 * sum(v) / total))
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write('\n'.join(lines) + '\n')
idx = [i for i, r in enumerate(rows) if 'save_diag' in r[2]]
if idx:
    seq = rows[idx[-1]:]
    end = [i for i, r in enumerate(seq) if 'logdiag' in r[2]]
    if end:
        seq = seq[:end[0] + 1]
        print('  last potrf: %d kernels, span %.1f us, sum of kernel durations %.1f us' % (len(seq), (seq[-1][1] - seq[0][0]) / 1e3, sum(r[1] - r[0] for r in seq) / 1e3))
        for a, b in list(zip(seq, seq[1:]))[:10]:
            print('    %-44s grid %7d dur %7.2f gap-to-next %6.2f' % (a[2][:44], a[3] // max(1, a[4]), (a[1] - a[0]) / 1e3, (b[0] - a[1]) / 1e3))
