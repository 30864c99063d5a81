#!/usr/bin/env python3
"""Per-scan busy time of the decode kernels in a rocprofv3 kernel trace.

The tiers of one scan run concurrently on several streams, so per-kernel averages in kernel_stats overlap.  A scan is
recognised by its tier-0 dispatch (the rows_kernel launch with the largest grid); its kernels are those between the idle gap
before that dispatch and the idle gap before the next one.  The union of their [start,end] intervals is what has to agree
with bench.py's hipEvent kernel_ms.  (The first scans of a run are the untimed gate and the two skip-index build passes.)
usage: union.py <..._kernel_trace.csv> [last_n]
"""
import csv, sys
L = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'rows_kernel' in r['Kernel_Name'] or 'decode_kernel' in r['Kernel_Name']:
        L.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X']), 'rows_kernel' in r['Kernel_Name']))
L.sort()
gmax = max(g for s, e, g, rk in L if rk)
# union segments
seg = []
for s, e, g, rk in L:
    if seg and s <= seg[-1][1]:
        seg[-1][1] = max(seg[-1][1], e)
    else:
        seg.append([s, e])
t0 = [s for s, e, g, rk in L if rk and g >= 0.9 * gmax]
# boundary of scan k: start of the union segment that contains its tier-0 dispatch
def seg_start(t):
    return max(a for a, b in seg if a <= t)
bounds = [seg_start(t) for t in t0] + [1 << 62]
busy = []
for k in range(len(t0)):
    busy.append(sum(b - a for a, b in seg if bounds[k] <= a < bounds[k + 1]) / 1e6)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print("%d launches, %d scans; busy ms per scan: %s; mean of last %d: %.1f ms" % (len(L), len(t0), ' '.join('%.1f' % x for x in busy), n, sum(busy[-n:]) / n))
