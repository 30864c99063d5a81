#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
BVG_FUZZ=250 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r04_t12_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r04_t12_fuzz.log
