#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
BVG_DEBUG=1 timeout -k 10 500 python profiles/r04/mem_diag.py 128 > gpurun_out/r04_mem_diag.txt 2>&1; grep -v "^\[bvg\] tier\|failures" gpurun_out/r04_mem_diag.txt | tail -25
TAG=xcd GIB=0 STEPS=3 CONFIGS="BVG_XCDS=1;BVG_XCDS=8;BVG_XCDS=1;BVG_XCDS=8" bash profiles/r04/ab.sh
bash profiles/r04/rehearsal.sh
