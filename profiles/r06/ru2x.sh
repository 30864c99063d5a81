#!/bin/bash
# round 6: the lean two-chain zeta_3 step loop (-DBVG_Z3_RU2X, bvg_scan_steps3x2.inc) from 65 / 97 tasks per sub-row on, against one chain per lane
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; L=$PWD/webgraph-big_amd/lib
for sh in eu15 w0 eu; do
TAG=ru2x_$sh SHAPE=$sh GIB=4 CONFIGS="X=1;BVG_HIP_LIB=$L/libbvg_exp_ru2x64.so;BVG_HIP_LIB=$L/libbvg_exp_ru2x96.so;X=2;BVG_HIP_LIB=$L/libbvg_exp_ru2x64.so" bash profiles/r06/ab.sh | cut -c1-130
done
