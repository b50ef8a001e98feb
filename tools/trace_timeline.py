#!/usr/bin/env python3
"""developer tool: timeline of the LAST factorisation in a rocprofv3 --kernel-trace csv (start, duration, queue, kernel, grid).
usage: trace_timeline.py <..._kernel_trace.csv> [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'save_diag' in r['Kernel_Name']]
seq = rows[idx[-1]:]
end = [i for i, r in enumerate(seq) if 'logdiag' in r['Kernel_Name']][0]
seq = seq[:end + 1]
t0 = int(seq[0]['Start_Timestamp'])
print('span %.1f us, %d kernels' % ((int(seq[-1]['End_Timestamp']) - t0) / 1e3, len(seq)))
for r in seq[:int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    print('%8.1f %8.1f q%s %-44s grid %d' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                              r['Queue_Id'], nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
