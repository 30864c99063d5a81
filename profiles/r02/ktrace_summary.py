#!/usr/bin/env python3
"""Timeline of the LAST scan in a rocprofv3 kernel trace: every decode launch with its start offset, duration, grid, LDS, VGPRs.
usage: ktrace_summary.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
dec = [r for r in rows if any(k in r['Kernel_Name'] for k in ('rows_kernel', 'scan_kernel', 'giant_kernel', 'rows_wg_kernel', 'decode_kernel', 'reduce_acc', 'flat_kernel'))]
dec.sort(key=lambda r: int(r['Start_Timestamp']))
gmax = max(int(r['Grid_Size_X']) for r in dec)
t0 = [i for i, r in enumerate(dec) if int(r['Grid_Size_X']) >= 0.9 * gmax]
# the last scan: from the reduce_acc before the last tier-0 launch (exclusive) to the end
last = t0[-1]
lo = last
while lo > 0 and 'reduce_acc' not in dec[lo - 1]['Kernel_Name']:
    lo -= 1
scan = dec[lo:]
s0 = min(int(r['Start_Timestamp']) for r in scan); e1 = max(int(r['End_Timestamp']) for r in scan)
print("last scan: %d launches, span %.2f ms" % (len(scan), (e1 - s0) / 1e6))
for r in scan:
    kn = r['Kernel_Name']
    short = kn[kn.find('bvg::') + 5:][:60] if 'bvg::' in kn else kn[:60]
    print("  +%8.2f ms  dur %8.2f ms  grid %8s  lds %6s  vgpr %4s  %s" % ((int(r['Start_Timestamp']) - s0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6,
          r['Grid_Size_X'], r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?'), short))
