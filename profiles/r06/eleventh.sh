#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_marks_only.py tests/test_gpu_flat.py tests/test_gpu_long_codes.py -m gpu -x -q 2>&1 | tail -3
TAG=w18_eu15 SHAPE=eu15 GIB=4 CONFIGS="X=1;BVG_SCAN_WAVES=18 BVG_SCAN_OCC=5;BVG_SCAN_WAVES=18;X=2" bash profiles/r06/ab.sh
TAG=w18_uk SHAPE=uk GIB=4 CONFIGS="X=1;BVG_SCAN_WAVES=18 BVG_SCAN_OCC=5" bash profiles/r06/ab.sh
TAG=occ_cnr SHAPE=cnr GIB=4 CONFIGS="X=1;BVG_SCAN_WAVES=20 BVG_SCAN_OCC=5;BVG_SCAN_WAVES=18 BVG_SCAN_OCC=5" bash profiles/r06/ab.sh
