#!/usr/bin/env python3
"""developer tool: kernels of a rocprofv3 --kernel-trace csv between the LAST occurrence of kernel name fragment A and the first
occurrence of fragment B after it.  usage: trace_window.py <csv> <fragA> <fragB> [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
a = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']][-1]
seq = rows[a:]
b = [i for i, r in enumerate(seq) if sys.argv[3] in r['Kernel_Name']][0]
seq = seq[:b + 1]
t0 = int(seq[0]['Start_Timestamp'])
print('span %.1f us, %d kernels' % ((int(seq[-1]['End_Timestamp']) - t0) / 1e3, len(seq)))
for r in seq[:int(sys.argv[4]) if len(sys.argv) > 4 else 100]:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
    print('%8.1f %8.1f q%s %-52s grid %d' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                              r['Queue_Id'], nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
