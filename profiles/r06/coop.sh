#!/bin/bash
# round 6: the cooperative continuation of long residual lists without index entries (bvg_scan.hip, "COOP").  GPU suite; then the marks-only rate (bench.py's marks_only leg needs the
# default run: here a coarse granularity, BVG_SKIP_GRAN=4096,64, stands in for it) and the default rates, library before | after
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_coop_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r06_coop_tests.log
[ $rc -ne 0 ] && exit $rc
PREV=$PWD/webgraph-big_amd/lib/libbvg_prev.so
for sh in eu15 cnr uk w0; do
TAG=coop_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$PREV BVG_SKIP_GRAN=4096,64;BVG_SKIP_GRAN=4096,64;BVG_HIP_LIB=$PREV;X=1" bash profiles/r06/ab.sh | cut -c1-140
done
