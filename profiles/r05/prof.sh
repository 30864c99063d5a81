#!/bin/bash
# section cycles / work counts of the flat kernel (make flatprof) on a bench shape:  SHAPE=eu15 GIB=2 bash profiles/r05/prof.sh [env assignments...]
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
env BVG_TEST_KNOBS=1 BVG_FLAT_PROF=1 BVG_DBG=64 BVG_DEBUG=1 BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvg_flatprof.so "$@" timeout -k 10 300 python bench.py --shape ${SHAPE:-eu15} --target-gib ${GIB:-2} --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg --no-real-leg 2> gpurun_out/r05_prof.err > /dev/null
echo "== ${SHAPE:-eu15} ${GIB:-2} GiB $@"; grep -E "flat kernel wave-cycles|flat kernel:|tiers concurrent" gpurun_out/r05_prof.err | tail -3
