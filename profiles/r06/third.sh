#!/bin/bash
# round 6, third GPU call: the wave-wide list build (WW) -- GPU suite, then same-box A/B: round-5 library | WW off (BVG_DBG=2048) | WW on
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_third_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r06_third_tests.log
[ $rc -ne 0 ] && exit $rc
R05=$PWD/webgraph-big_amd/lib/libbvg_r05.so
for sh in eu15 uk cnr; do
TAG=ww_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$R05;BVG_DBG=2048;X=1;BVG_DBG=2048;X=2" bash profiles/r06/ab.sh || exit 1
done
