#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; L=$PWD/webgraph-big_amd/lib
PREV=$L/libbvg_prev.so
for sh in eu15 uk w0; do
TAG=coop2_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$PREV BVG_SKIP_GRAN=4096,64;BVG_SKIP_GRAN=4096,64;BVG_HIP_LIB=$L/libbvg_exp_coop1.so BVG_SKIP_GRAN=4096,64;BVG_HIP_LIB=$L/libbvg_exp_coop4.so BVG_SKIP_GRAN=4096,64;BVG_HIP_LIB=$PREV;X=1" bash profiles/r06/ab.sh | cut -c1-140
done
