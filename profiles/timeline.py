#!/usr/bin/env python3
"""Launch timeline of the last scan in a rocprofv3 kernel trace, relative to its tier-0 dispatch. usage: timeline.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'rows_kernel' in r['Kernel_Name'] or 'decode_kernel' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['g'] = int(r['Grid_Size_X'])
g = max(r['g'] for r in rows if 'rows_kernel' in r['Kernel_Name'])
t0 = [r for r in rows if r['g'] >= 0.9 * g and 'rows_kernel' in r['Kernel_Name']]
last = t0[-1]
for r in sorted(rows, key=lambda r: r['s']):
    if r['s'] > last['s'] - 2000000 and r['s'] < last['e'] + 100000000:
        print('%9.3f %9.3f ms  blocks %8d  lds %6s  %s' % ((r['s'] - last['s']) / 1e6, (r['e'] - last['s']) / 1e6, r['g'] // 64, r['LDS_Block_Size'], 'rows' if 'rows_kernel' in r['Kernel_Name'] else 'generic'))
