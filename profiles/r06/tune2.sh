#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; L=$PWD/webgraph-big_amd/lib
for sh in eu15 uk; do
TAG=tune2_$sh SHAPE=$sh GIB=4 CONFIGS="X=1;BVG_HIP_LIB=$L/libbvg_exp_chunk12.so;BVG_HIP_LIB=$L/libbvg_exp_chunk10.so;BVG_HIP_LIB=$L/libbvg_exp_chunk14.so;X=2;BVG_HIP_LIB=$L/libbvg_exp_chunk12.so" bash profiles/r06/ab.sh | cut -c1-130
done
