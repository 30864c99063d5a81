#!/bin/bash
# round 6: 18 wavefronts per CU (96-VGPR instantiation) for graphs of 16 ... 48 arcs per node: the rule against the old choice, every bench shape
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for sh in uk eu web eu15 cnr; do
TAG=w18b_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_NO_W18=1;X=1;BVG_NO_W18=1;X=2" bash profiles/r06/ab.sh | cut -c1-150
done
