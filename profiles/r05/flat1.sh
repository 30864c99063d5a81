#!/bin/bash
# round 5: the flat kernel's first GPU call -- the GPU suite, then same-box A/Bs against scan_kernel (BVG_FLAT=0) on the default shape and the tiled cnr-2000
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_flat1_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r05_flat1_tests.log
[ $rc -ne 0 ] && exit $rc
TAG=flat1_eu15 SHAPE=eu15 GIB=4 CONFIGS="BVG_FLAT=0;BVG_FLAT=1;BVG_FLAT=1 BVG_FLAT_RECS=128" bash profiles/r05/ab.sh &&
TAG=flat1_cnr SHAPE=cnr GIB=4 CONFIGS="BVG_FLAT=0;BVG_FLAT=1 BVG_FLAT_RECS=64;BVG_FLAT=1 BVG_FLAT_RECS=128;BVG_FLAT=1 BVG_FLAT_RECS=256" bash profiles/r05/ab.sh
