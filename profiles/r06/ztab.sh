#!/bin/bash
# round 6 (verdict item 2a, measured): table-driven zeta_3 -- a 4 096-entry {gap, length} table in global memory (L1 / L2 resident), one load per residual step, the arithmetic
# decode only for the lanes of a step whose code is longer than 12 bits -- against the arithmetic decode of the product
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; L=$PWD/webgraph-big_amd/lib
for sh in w0 eu15 cnr; do
TAG=ztab_$sh SHAPE=$sh GIB=4 CONFIGS="X=1;BVG_HIP_LIB=$L/libbvg_exp_ztab.so;X=2;BVG_HIP_LIB=$L/libbvg_exp_ztab.so" bash profiles/r06/ab.sh | cut -c1-130
done
