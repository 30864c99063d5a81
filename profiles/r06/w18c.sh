#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for sh in eu web; do
TAG=w18c_$sh SHAPE=$sh GIB=4 CONFIGS="X=1;BVG_SCAN_WAVES=18;BVG_SCAN_WAVES=20" bash profiles/r06/ab.sh | cut -c1-150
done
TAG=w18c_uk SHAPE=uk GIB=4 CONFIGS="X=1;BVG_SCAN_STAGE=320;BVG_SCAN_STAGE=448;BVG_SCAN_SCR=384;BVG_SCAN_WAVES=20" bash profiles/r06/ab.sh | cut -c1-150
