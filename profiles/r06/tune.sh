#!/bin/bash
# round 6: the compile-time task sizes re-checked on the final kernels (leaf chunk 8 -> 12 / 16, shortest position task 1 -> 2, short residual tails 6 -> 4)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; L=$PWD/webgraph-big_amd/lib
for sh in eu15 cnr; do
TAG=tune_$sh SHAPE=$sh GIB=4 CONFIGS="X=1;BVG_HIP_LIB=$L/libbvg_exp_chunk16.so;BVG_HIP_LIB=$L/libbvg_exp_chunk12.so;BVG_HIP_LIB=$L/libbvg_exp_mintask2.so;BVG_HIP_LIB=$L/libbvg_exp_short4.so;X=2" bash profiles/r06/ab.sh | cut -c1-130
done
