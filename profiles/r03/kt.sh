#!/bin/bash
# r03: section timers of the prof build (scan kernel + row kernel), then a kernel trace of steady-state scans; ARGS / env select the configuration
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
out=gpurun_out/r03_prof_${TAG:-x}.txt; : > $out
echo "== prof (libbvgraph_hip_prof.so, BVG_DBG=64, 1 GiB eu15 shape, one steady-state scan) ${ENVS}" >> $out
env ${ENVS} BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvgraph_hip_prof.so BVG_DEBUG=1 BVG_DBG=64 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --target-gib 1 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|wave-cycles|phase 1 split|tiers concurrent|scan kernel:|^\{" | tail -7 | cut -c1-400 >> $out
cat $out
rm -rf gpurun_out/r03_kt; mkdir -p gpurun_out/r03_kt
cd /tmp && export TMPDIR=/tmp
env ${ENVS} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_kt -- python3 $R/bench.py --steps 3 --warmup 2 --target-gib 4 --no-cpu-baseline --no-verify > $R/gpurun_out/r03_kt/bench.log 2>&1
cd $R
f=$(ls gpurun_out/r03_kt/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f | tee gpurun_out/r03_ktrace_${TAG:-x}.txt
grep '^{' gpurun_out/r03_kt/bench.log | cut -c1-200
s=$(ls gpurun_out/r03_kt/*/*_kernel_stats.csv | head -1); cp $s gpurun_out/r03_kstats_${TAG:-x}.csv; head -8 $s | cut -c1-200
rm -rf gpurun_out/r03_kt
