#!/usr/bin/env python3
"""What each kernel of the edges / state machine chain costs a front-end launch running beside it.
Input: a rocprofv3 --kernel-trace CSV of `bench.py` with several contexts in flight.  For every front-end
launch: its duration and the time each chain kernel overlapped it; least squares of
    duration = alone + sum_k slowdown_k * overlap_k
(slowdown_k = extra front-end time per unit of time kernel k runs beside it).
    python tools/overlap_cost.py <kernel_trace.csv>"""
import csv, sys
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
def name(r):
    return r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ookd::', '').split('<')[0]
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), name(r)) for r in rows]
ks.sort()
front = [k for k in ks if k[2].startswith('fir1_')]
others = [k for k in ks if not k[2].startswith('fir1_') and not k[2].startswith('__amd') and not k[2].startswith('synth')]
names = sorted(set(k[2] for k in others))
# steady state only: drop the first and last 15 % of the front-end launches
lo, hi = int(len(front) * 0.15), int(len(front) * 0.85)
front = front[lo:hi]
A, y = [], []
import bisect
starts = [k[0] for k in others]
for s, e, _ in front:
    ov = dict.fromkeys(names, 0.0)
    i = bisect.bisect_left(starts, s - 5_000_000)
    while i < len(others) and others[i][0] < e:
        os_, oe, on = others[i]
        o = min(e, oe) - max(s, os_)
        if o > 0:
            ov[on] += o
        i += 1
    A.append([1.0] + [ov[n] / 1e3 for n in names])
    y.append((e - s) / 1e3)
A, y = np.array(A), np.array(y)
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print("front-end launches used: %d, mean %.1f us, min %.1f us" % (len(y), y.mean(), y.min()))
print("alone (fit): %.1f us per launch" % coef[0])
tot = A[:, 1:].sum(axis=0) / len(y)
print("%-28s %10s %12s %14s" % ("kernel beside it", "slowdown", "overlap/launch", "cost/launch us"))
for n, c, t in sorted(zip(names, coef[1:], tot), key=lambda x: -x[1] * x[2]):
    print("%-28s %10.2f %12.1f %14.1f" % (n, c, t, c * t))
print("sum of costs per launch: %.1f us" % float((coef[1:] * tot).sum()))
