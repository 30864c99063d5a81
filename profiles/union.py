#!/usr/bin/env python3
"""Union of the decode kernels' [start,end] intervals in a rocprofv3 kernel trace.

The tiers of one scan run concurrently on three streams, so per-kernel averages in kernel_stats overlap; the union
divided by the number of scans is what has to agree with bench.py's hipEvent kernel_ms.
usage: union.py <..._kernel_trace.csv> <scans>
"""
import csv, sys
iv = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'rows_kernel' in r['Kernel_Name'] or 'decode_kernel' in r['Kernel_Name']:
        iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
iv.sort()
tot, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        tot += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
tot += ce - cs
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print(f"{len(iv)} launches, union {tot/1e6:.1f} ms, {tot/1e6/scans:.1f} ms per scan")
