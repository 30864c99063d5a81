#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
python -m pytest tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | tail -2
TAG=cadmit SHAPE=eu15 GIB=8 STEPS=6 CONFIGS="X=1;BVG_CADMIT=0.75;BVG_CADMIT=0.5;BVG_CADMIT=0.35;X=2" bash profiles/r05/ab.sh
