#!/bin/bash
# round 6, eighth GPU call: GPU suite on the trimmed set-ups (4-ary dealing, ranked short tails, 32-bit header fields, node ring behind the window), A/B against the library before them
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_eighth_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r06_eighth_tests.log
[ $rc -ne 0 ] && exit $rc
MID=$PWD/webgraph-big_amd/lib/libbvg_mid.so
for sh in eu15 cnr uk; do
TAG=trim_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$MID;X=1;BVG_HIP_LIB=$MID;X=2" bash profiles/r06/ab.sh
done
TAG=w20_eu15 SHAPE=eu15 GIB=4 CONFIGS="BVG_SCAN_WAVES=20 BVG_SCAN_OCC=5;BVG_SCAN_WAVES=14" bash profiles/r06/ab.sh
