#!/bin/bash
# round 5: tier-0 LDS geometry re-checked on the final kernels (the vector pipe is 91 % busy now, not 96 %)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
TAG=geom_eu15 SHAPE=eu15 GIB=8 STEPS=6 CONFIGS="X=1;BVG_SCAN_WAVES=14;BVG_SCAN_STAGE=448;BVG_SCAN_STAGE=512;BVG_SCAN_SCR=384;BVG_SCAN_SCR=512;BVG_SCAN_STAGE=320;X=2" bash profiles/r05/ab.sh
TAG=geom_cnr SHAPE=cnr GIB=4 STEPS=6 CONFIGS="X=1;BVG_SCAN_WAVES=20;BVG_SCAN_WAVES=16;BVG_SCAN_STAGE=384;BVG_SCAN_SCR=320;X=2" bash profiles/r05/ab.sh
