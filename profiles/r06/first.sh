#!/bin/bash
# round 6, first GPU call: parity of the new zeta_3 step loop (scan-kernel tests + fuzz), then same-box A/Bs of the round-5 library against it
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_first_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r06_first_tests.log
[ $rc -ne 0 ] && exit $rc
R05=$PWD/webgraph-big_amd/lib/libbvg_r05.so
for sh in eu15 cnr w0 uk; do
TAG=steps3_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$R05;X=1;BVG_HIP_LIB=$R05;X=2" bash profiles/r06/ab.sh || exit 1
done
