#!/bin/bash
# two PMC passes of one steady-state scan: issue vs wait breakdown of the tier-0 row kernel. usage: pmc_quick.sh <tag> [env assignments via export]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pq_$tag; rm -rf $O; mkdir -p $O
ARGS="--shape ${SHAPE:-eu} --steps 1 --warmup 1 --target-gib 1 --no-cpu-baseline --no-verify"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -- python bench.py $ARGS > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python bench.py $ARGS > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $O/c -- python bench.py $ARGS > $O/c.log 2>&1
python3 - $O $tag <<'P'
import csv,glob,sys,collections
O,tag=sys.argv[1],sys.argv[2]
res={}
for f in glob.glob(O+'/*/*/*_counter_collection.csv'):
    rows=[r for r in csv.DictReader(open(f)) if 'rows' in r['Kernel_Name']]
    if not rows: continue
    g=max(int(r['Grid_Size']) for r in rows)
    big=[r for r in rows if int(r['Grid_Size'])==g]
    last=max(int(r['Dispatch_Id']) for r in big)
    for r in big:
        if int(r['Dispatch_Id'])==last: res[r['Counter_Name']]=float(r['Counter_Value']); res['_k']=(r['Kernel_Name'][:40],r['Grid_Size'],r['VGPR_Count'],r['LDS_Block_Size'])
print(tag,res.get('_k'))
for k in sorted(res):
    if k!='_k': print('  %-22s %.4g'%(k,res[k]))
wc=res.get('SQ_WAVE_CYCLES')
if wc:
    for k in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_SCA','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_LDS'):
        if k in res: print('  %s / WAVE_CYCLES = %.3f'%(k,res[k]/wc))
P
