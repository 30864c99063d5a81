#!/bin/bash
# r06 evidence at HEAD: the GPU suite, bench lines (default eu15 mosaic; the REAL graph cnr-2000 tiled; the round-3/4 shapes), the RCCL path with one rank, kernel stats +
# timeline of steady-state scans of the default workload.  PMC passes: profiles/r06/pmc.sh + record.py (separate calls; never combined with tracing domains).
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out
if [ "${TESTS:-1}" = "1" ]; then timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gputests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06_gputests.log; fi
run() { tag=$1; shift; timeout -k 10 600 python bench.py "$@" > gpurun_out/r06_${tag}_bench.json 2> gpurun_out/r06_${tag}_bench.err; echo "$tag rc=$? $(python3 -c "import json,sys; d=json.load(open('gpurun_out/r06_${tag}_bench.json')); print('%.1f G edges/s %.1f ms/step roofline %.4f (with index %.4f) no-index %.1f G build %.2f s resident %.1f GB cpu %.2f G' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_with_index'], d.get('value_no_index',0)/1e9, d['index_build_s'], d['hbm_resident_bytes']/1e9, d.get('cpu_baseline',{}).get('value',0)/1e9))" 2>&1)"; }
run eu15
run cnr --shape cnr
if [ "${SHAPES:-1}" = "1" ]; then
run eu15mono --shape eu15mono --no-cpu-baseline --no-index-leg --no-wide-leg
run eu8g --shape eu --no-cpu-baseline --no-index-leg --no-wide-leg
run web8g --shape web --no-cpu-baseline --no-index-leg --no-wide-leg
run uk8g --shape uk --no-cpu-baseline --no-index-leg --no-wide-leg
run w08g --shape w0 --no-cpu-baseline --no-index-leg --no-wide-leg
run eu_u64 --shape eu --tiles 2100 --allow-wide --no-cpu-baseline --no-index-leg --no-wide-leg
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-index-leg --no-wide-leg > gpurun_out/r06_eu15_torchrun1_bench.json 2> gpurun_out/r06_eu15_torchrun1_bench.err; echo "torchrun1 rc=$?"
fi
# kernel stats + timeline of steady-state scans (full-size default workload)
rm -rf gpurun_out/r06_kt; mkdir -p gpurun_out/r06_kt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_kt -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg --no-wide-leg --no-real-leg > $R/gpurun_out/r06_kt/bench.log 2>&1 )
f=$(ls gpurun_out/r06_kt/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f > gpurun_out/r06_eu15_scan_timeline.txt; tail -12 gpurun_out/r06_eu15_scan_timeline.txt
cp $(ls gpurun_out/r06_kt/*/*_kernel_stats.csv | head -1) gpurun_out/r06_eu15_kernel_stats.csv
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$f"))]
red=[i for i,r in enumerate(rows) if 'reduce_acc' in r['Kernel_Name']]
cut=red[-4] if len(red)>=4 else -1
agg=collections.defaultdict(lambda:[0,0])
for r in rows[cut+1:]:
    k=r['Kernel_Name'][:90]; agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
with open("gpurun_out/r06_eu15_kernel_stats_steady.csv","w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs (the last 3 scans only: steady state)\n")
    for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): o.write('"%s",%d,%d,%.0f\n'%(k,c,t,t/c))
print(open("gpurun_out/r06_eu15_kernel_stats_steady.csv").read()[:900])
PY
rm -rf gpurun_out/r06_kt
