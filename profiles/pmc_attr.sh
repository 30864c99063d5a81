#!/bin/bash
# instruction attribution of the tier-0 row kernel: VALU/SALU/LDS instruction counts of one steady-state scan with phases skipped (BVG_DBG bits)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BVG_WG=0
for m in ${MODES:-0 128 1 3 7}; do
  O=gpurun_out/pa_$m; rm -rf $O; mkdir -p $O
  BVG_DBG=$m rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/a -- python bench.py --shape ${SHAPE:-eu} --steps 1 --warmup 1 --target-gib 1 --no-cpu-baseline --no-verify > $O/a.log 2>&1
  python3 - $O $m <<'P'
import csv,glob,sys
O,tag=sys.argv[1],sys.argv[2]
res={}
for f in glob.glob(O+'/*/*/*_counter_collection.csv'):
    rows=[r for r in csv.DictReader(open(f)) if 'rows' in r['Kernel_Name']]
    if not rows: continue
    g=max(int(r['Grid_Size']) for r in rows)
    big=[r for r in rows if int(r['Grid_Size'])==g]
    last=max(int(r['Dispatch_Id']) for r in big)
    for r in big:
        if int(r['Dispatch_Id'])==last: res[r['Counter_Name']]=float(r['Counter_Value'])
kt=glob.glob(O+'/*/*/*_kernel_trace.csv')
dur=None
if kt:
    rows=[r for r in csv.DictReader(open(kt[0])) if 'rows' in r['Kernel_Name']]
    g=max(int(r['Grid_Size_X']) for r in rows) if rows and 'Grid_Size_X' in rows[0] else None
    big=[r for r in rows if g is None or int(r['Grid_Size_X'])==g]
    if big: dur=(int(big[-1]['End_Timestamp'])-int(big[-1]['Start_Timestamp']))/1e6
print('dbg=%s: VALU %.3g SALU %.3g LDS %.3g wavecyc %.3g dur_ms %s'%(tag,res.get('SQ_INSTS_VALU',0),res.get('SQ_INSTS_SALU',0),res.get('SQ_INSTS_LDS',0),res.get('SQ_WAVE_CYCLES',0),dur))
P
done
