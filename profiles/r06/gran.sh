#!/bin/bash
# round 6: skip granularity re-checked on the final kernels (the residual step got cheaper: does a coarser, lighter index cost less than the -6.5 % of round 5?)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
TAG=gran_eu15 SHAPE=eu15 GIB=4 CONFIGS="X=1;BVG_SKIP_GRAN=16,32;BVG_SKIP_GRAN=32,32;BVG_SKIP_GRAN=24,16;BVG_SKIP_GRAN=12,8;X=2" bash profiles/r06/ab.sh | cut -c1-150
TAG=gran_cnr SHAPE=cnr GIB=4 CONFIGS="X=1;BVG_SKIP_GRAN=16,16;BVG_SKIP_GRAN=12,8" bash profiles/r06/ab.sh | cut -c1-150
