#!/bin/bash
cd "$(dirname "$0")/../.."
L=$PWD/webgraph-big_amd/lib
TAG=chunk GIB=0 STEPS=3 CONFIGS="BVG_NOP=1;BVG_HIP_LIB=$L/libbvg_exp_chunk12.so;BVG_HIP_LIB=$L/libbvg_exp_chunk16.so;BVG_NOP=1" bash profiles/r04/ab.sh | cut -c1-120
