#!/bin/bash
# kernel trace of the 4.4 G-node workload (lists of block-relative 32-bit ids): every launch of the last scan
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/r03_ktw; mkdir -p gpurun_out/r03_ktw
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_ktw -- python3 $R/bench.py --shape eu --tiles 2100 --allow-wide --steps 2 --warmup 2 --no-cpu-baseline --no-verify > $R/gpurun_out/r03_ktw/bench.log 2>&1 )
f=$(ls gpurun_out/r03_ktw/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f > gpurun_out/r03_eu_u64_scan_timeline.txt; tail -16 gpurun_out/r03_eu_u64_scan_timeline.txt
rm -rf gpurun_out/r03_ktw
