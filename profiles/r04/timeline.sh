#!/bin/bash
# r04: kernel timeline of the last steady-state scan of the default workload (rocprofv3 --kernel-trace only), optionally under extra environment settings.
# usage: [ENVS="GPU_MAX_HW_QUEUES=8 BVG_X=1"] bash profiles/r04/timeline.sh <tag> [bench args]
cd "$(dirname "$0")/../.."; R=$PWD; tag=${1:-tl}; shift
for e in $ENVS; do export $e; done
export BVG_TEST_KNOBS=1
rm -rf gpurun_out/r04_kt_$tag; mkdir -p gpurun_out/r04_kt_$tag
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_kt_$tag -- python3 $R/bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg > $R/gpurun_out/r04_kt_$tag/bench.log 2>&1 )
f=$(ls gpurun_out/r04_kt_$tag/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f > gpurun_out/r04_timeline_$tag.txt; tail -12 gpurun_out/r04_timeline_$tag.txt
rm -rf gpurun_out/r04_kt_$tag
