#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t3_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04_t3_tests.log
TAG=sched GIB=0 STEPS=3 CONFIGS="BVG_NOP=1;BVG_GIANT_PAD=75000;BVG_GIANT_PAD=75000 BVG_ORDER=1;BVG_CLASS_STAGE=512,768,1024,1024;BVG_CLASS_STAGE=384,512,768,1024 BVG_CLASS_SCR=512,768,1536,3072;BVG_CLASS_SCR=512,1024,2048,3072;BVG_GIANT_PAD=75000 BVG_CLASS_STAGE=384,512,768,1024 BVG_CLASS_SCR=512,768,1536,3072" bash profiles/r04/ab.sh
