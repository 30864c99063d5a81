#!/bin/bash
# round 6, ninth GPU call: GPU suite, then A/B: library before | position loop + branch-free bounds only (one residual code per trip) | + two residual codes per trip
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_ninth_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r06_ninth_tests.log
[ $rc -ne 0 ] && exit $rc
PREV=$PWD/webgraph-big_amd/lib/libbvg_prev.so; SINGLE=$PWD/webgraph-big_amd/lib/libbvg_exp_single.so
for sh in eu15 cnr uk w0; do
TAG=pair_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$PREV;BVG_HIP_LIB=$SINGLE;X=1;BVG_HIP_LIB=$PREV;BVG_HIP_LIB=$SINGLE;X=2" bash profiles/r06/ab.sh
done
