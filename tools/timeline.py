"""One pass of the chain from a rocprofv3 kernel trace:  python tools/timeline.py <dir with *_kernel_trace.csv>
Prints start / duration of every kernel between the last two fin_msg kernels but one (a steady-state pass)."""
import csv
import glob
import sys

import os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)     # (the newest: gpurun merges runs)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('void ookd::', '').replace('ookd::', '') for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('fin_msg')]
last, i0 = idx[-2], idx[-3] + 1
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = None
chain0 = None
for i in range(i0, last + 1):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    if chain0 is None and names[i].startswith('edge_scan_local'):
        chain0 = s
    print("%-34s start %8.1f dur %7.1f gap %6.1f grid %s wg %s" % (names[i][:34], (s - t0) / 1e3, (e - s) / 1e3, gap,
                                                               rows[i].get('Grid_Size_X', rows[i].get('Grid_Size')),
                                                               rows[i].get('Workgroup_Size_X', rows[i].get('Workgroup_Size'))))
    prev_end = e
if chain0:
    print("chain: %.1f us" % ((prev_end - chain0) / 1e3))
