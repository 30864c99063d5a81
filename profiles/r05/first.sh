#!/bin/bash
# round 5, first GPU call: the GPU suite on the tree with the linear checksum, then same-box A/Bs of the round-4 library against it
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_first_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r05_first_tests.log
[ $rc -ne 0 ] && exit $rc
R04=$PWD/webgraph-big_amd/lib/libbvg_r04.so
TAG=chk_eu15 SHAPE=eu15 GIB=4 CONFIGS="BVG_HIP_LIB=$R04;X=1" bash profiles/r05/ab.sh &&
TAG=chk_cnr SHAPE=cnr GIB=4 CONFIGS="BVG_HIP_LIB=$R04;X=1" bash profiles/r05/ab.sh
