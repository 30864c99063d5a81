#!/bin/bash
cd "$(dirname "$0")/../.."
L=$PWD/webgraph-big_amd/lib
TAG=t0wait GIB=0 STEPS=3 CONFIGS="BVG_T0WAIT=0;BVG_T0WAIT=1;BVG_T0WAIT=0;BVG_T0WAIT=1;BVG_HIP_LIB=$L/libbvg_exp_ru64.so BVG_T0WAIT=0;BVG_HIP_LIB=$L/libbvg_exp_ru80.so BVG_T0WAIT=0;BVG_HIP_LIB=$L/libbvg_exp_ru128.so BVG_T0WAIT=0;BVG_HIP_LIB=$L/libbvg_exp_ru100000.so BVG_T0WAIT=0" bash profiles/r04/ab.sh
