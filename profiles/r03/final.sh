#!/bin/bash
# r03 evidence at HEAD: bench lines of every shape, the RCCL path with one rank, kernel stats + timeline of steady-state scans, section timers, occupancy table
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 600 python bench.py "$@" > gpurun_out/r03_${tag}_bench.json 2> gpurun_out/r03_${tag}_bench.err; echo "$tag rc=$? $(python3 -c "import json,sys; d=json.load(open('gpurun_out/r03_${tag}_bench.json')); print('%.1f G edges/s %.1f ms/step roofline %.4f cpu %.2f G' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value',0)/1e9))" 2>&1)"; }
run eu15
run eu15mono --shape eu15mono --no-cpu-baseline
run eu8g --shape eu --no-cpu-baseline
run web8g --shape web --no-cpu-baseline
run w08g --shape w0 --no-cpu-baseline
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > gpurun_out/r03_eu15_torchrun1_bench.json 2> gpurun_out/r03_eu15_torchrun1_bench.err; echo "torchrun1 rc=$?"
# kernel stats + timeline of steady-state scans (full-size default workload)
rm -rf gpurun_out/r03_kt; mkdir -p gpurun_out/r03_kt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_kt -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-verify > $R/gpurun_out/r03_kt/bench.log 2>&1 )
f=$(ls gpurun_out/r03_kt/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f > gpurun_out/r03_eu15_scan_timeline.txt; tail -12 gpurun_out/r03_eu15_scan_timeline.txt
cp $(ls gpurun_out/r03_kt/*/*_kernel_stats.csv | head -1) gpurun_out/r03_eu15_kernel_stats.csv
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f"))]
# steady-state only: drop everything before the third-from-last reduce_acc (index build + warm-up scans carry other launch mixes)
red=[i for i,r in enumerate(rows) if 'reduce_acc' in r['Kernel_Name']]
cut=red[-4] if len(red)>=4 else -1
import collections
agg=collections.defaultdict(lambda:[0,0])
for r in rows[cut+1:]:
    k=r['Kernel_Name'][:90]; agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
with open("gpurun_out/r03_eu15_kernel_stats_steady.csv","w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs (the last 3 scans only: steady state)\n")
    for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): o.write('"%s",%d,%d,%.0f\n'%(k,c,t,t/c))
print(open("gpurun_out/r03_eu15_kernel_stats_steady.csv").read()[:900])
PY
rm -rf gpurun_out/r03_kt
# section timers (prof build), 1 GiB of the default workload, one steady-state scan
export BVG_TEST_KNOBS=1
out=gpurun_out/r03_section_timers.txt; echo "== libbvgraph_hip_prof.so, BVG_DBG=64, 1 GiB of the eu15 mosaic, one steady-state scan; M wave-cycles per section of scan_kernel (slots of the row kernel's report: 'row prep' = levels, 'task set-up' = Z2 set-up, 'seeks' = Z1, 'merge loop' = Z2 loop)" > $out
BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvgraph_hip_prof.so BVG_DEBUG=1 BVG_DBG=64 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --target-gib 1 --no-cpu-baseline --no-verify 2>&1 | grep -E "wave-cycles|phase 1 split|tiers concurrent|scan kernel" | tail -5 | cut -c1-300 >> $out
cat $out
# occupancy: the same pool (the default 16-wavefront configuration), unused LDS added so that fewer wavefronts fit a CU
TAG=occ16 CONFIGS="BVG_SCAN_WAVES=16;BVG_SCAN_WAVES=16 BVG_SCAN_PAD=1400;BVG_SCAN_WAVES=16 BVG_SCAN_PAD=3400;BVG_SCAN_WAVES=16 BVG_SCAN_PAD=6100;BVG_SCAN_WAVES=16 BVG_SCAN_PAD=10200;BVG_SCAN_WAVES=16 BVG_SCAN_PAD=17000" bash profiles/r03/ab.sh | cut -c1-200
