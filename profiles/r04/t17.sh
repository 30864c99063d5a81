#!/bin/bash
cd "$(dirname "$0")/../.."
L=$PWD/webgraph-big_amd/lib
TAG=ablate_cnr SHAPE=cnr GIB=4 STEPS=5 CONFIGS="BVG_NOP=1;BVG_HIP_LIB=$L/libbvg_exp_abl_Z2.so;BVG_HIP_LIB=$L/libbvg_exp_abl_Z1.so;BVG_HIP_LIB=$L/libbvg_exp_abl_RESLOOP.so;BVG_HIP_LIB=$L/libbvg_exp_abl_LEAF.so" bash profiles/r04/ab.sh | cut -c1-110
