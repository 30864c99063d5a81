#!/bin/bash
# per-dispatch durations of the last scan's kernels (kernel trace), optionally with phases skipped (BVG_DBG)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BVG_WG=${BVG_WG:-0}
for m in ${MODES:-0 7}; do
  O=gpurun_out/kt_$m; rm -rf $O; mkdir -p $O
  BVG_DBG=$m rocprofv3 --kernel-trace --output-format csv -d $O -- python bench.py --shape ${SHAPE:-eu} --steps 1 --warmup 1 --target-gib 1 --no-cpu-baseline --no-verify > $O/log 2>&1
  python3 - $O $m <<'P'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'rows' in r['Kernel_Name'] or 'decode_kernel' in r['Kernel_Name']]
t0=min(int(r['Start_Timestamp']) for r in rows[-8:])
print('dbg=%s last dispatches:'%sys.argv[2])
for r in rows[-8:]:
    print('   %-40s grid %8s lds %6s start %8.2f dur %8.2f ms'%(r['Kernel_Name'][28:68],r['Grid_Size_X'],r.get('LDS_Block_Size','?'),(int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
P
done
