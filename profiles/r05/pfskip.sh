#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
X=$PWD/webgraph-big_amd/lib/libbvg_exp_pfskip.so
TAG=pfskip_eu15 SHAPE=eu15 GIB=8 STEPS=8 CONFIGS="A=1;BVG_HIP_LIB=$X;A=2;BVG_HIP_LIB=$X" bash profiles/r05/ab.sh
TAG=pfskip_cnr SHAPE=cnr GIB=4 STEPS=8 CONFIGS="A=1;BVG_HIP_LIB=$X" bash profiles/r05/ab.sh
