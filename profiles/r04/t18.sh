#!/bin/bash
cd "$(dirname "$0")/../.."
L=$PWD/webgraph-big_amd/lib
TAG=noru2 GIB=0 STEPS=3 CONFIGS="BVG_NOP=1;BVG_HIP_LIB=$L/libbvg_exp_noru2.so;BVG_HIP_LIB=$L/libbvg_exp_noru2.so BVG_SCAN_OCC=5 BVG_SCAN_WAVES=20;BVG_NOP=1" bash profiles/r04/ab.sh | cut -c1-200
TAG=noru2_cnr SHAPE=cnr GIB=4 STEPS=5 CONFIGS="BVG_NOP=1;BVG_HIP_LIB=$L/libbvg_exp_noru2.so;BVG_HIP_LIB=$L/libbvg_exp_noru2.so BVG_SCAN_OCC=50 BVG_SCAN_WAVES=20" bash profiles/r04/ab.sh | cut -c1-200
TAG=noru2_web SHAPE=web GIB=4 STEPS=5 CONFIGS="BVG_NOP=1;BVG_HIP_LIB=$L/libbvg_exp_noru2.so" bash profiles/r04/ab.sh | cut -c1-200
