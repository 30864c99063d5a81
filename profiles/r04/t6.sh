#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_giant.py tests/test_gpu_scan_kernel.py tests/test_gpu_shards.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04_t6_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_t6_tests.log
BVG_DEBUG=1 timeout -k 10 500 python profiles/r04/mem_diag.py 128 > gpurun_out/r04_mem_diag.txt 2>&1; grep "in use" gpurun_out/r04_mem_diag.txt
TAG=occ5 GIB=0 STEPS=3 CONFIGS="BVG_NOP=1;BVG_GBATCH=8192;BVG_SCAN_OCC=5 BVG_SCAN_WAVES=20;BVG_NOP=1" bash profiles/r04/ab.sh
bash profiles/r04/rehearsal.sh
bash profiles/r04/pmc.sh eu15full --shape eu15 > gpurun_out/r04_pmc_eu15full.log 2>&1; tail -12 gpurun_out/r04_pmc_eu15full_summary.txt
