#!/bin/bash
# round 6, sixth GPU call: ZE (position tasks by kept element over the extras' bit vectors; opt-in BVG_DBG=8192) against the default on the same library
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for sh in eu15 uk cnr web; do
TAG=ze_$sh SHAPE=$sh GIB=4 CONFIGS="X=0;BVG_DBG=8192;X=1;BVG_DBG=8192" bash profiles/r06/ab.sh
done
