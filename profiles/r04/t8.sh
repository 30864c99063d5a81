#!/bin/bash
cd "$(dirname "$0")/../.."
TAG=prio GIB=0 STEPS=3 CONFIGS="BVG_NOP=1;BVG_CLASS_STAGE=256,384,512,768 BVG_CLASS_SCR=384,512,1024,2048;BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvg_exp_prio.so;BVG_NOP=1;BVG_CLASS_STAGE=384,512,768,1024 BVG_CLASS_SCR=512,768,1536,3072" bash profiles/r04/ab.sh
