#!/bin/bash
# round 6, fifth GPU call: WW opt-in (BVG_DBG=4096 + (shortest list << 16)) against the default path on the same library
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
TAG=ww3_eu15 SHAPE=eu15 GIB=4 CONFIGS="X=0;BVG_DBG=4096;BVG_DBG=8392704;BVG_DBG=16781312;BVG_DBG=33558528;X=1" bash profiles/r06/ab.sh
TAG=ww3_uk SHAPE=uk GIB=4 CONFIGS="X=0;BVG_DBG=4096;BVG_DBG=16781312;X=1" bash profiles/r06/ab.sh
TAG=ww3_cnr SHAPE=cnr GIB=4 CONFIGS="X=0;X=1" bash profiles/r06/ab.sh
