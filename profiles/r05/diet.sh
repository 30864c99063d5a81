#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
HEAD=$PWD/webgraph-big_amd/lib/libbvg_head.so
TAG=${T:-diet2}_eu15 SHAPE=eu15 GIB=8 STEPS=8 CONFIGS="BVG_HIP_LIB=$HEAD;X=1;BVG_HIP_LIB=$HEAD;X=2" bash profiles/r05/ab.sh
TAG=${T:-diet2}_cnr SHAPE=cnr GIB=4 STEPS=8 CONFIGS="BVG_HIP_LIB=$HEAD;X=1;BVG_HIP_LIB=$HEAD;X=2" bash profiles/r05/ab.sh
