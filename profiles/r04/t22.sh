#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
python profiles/r04/idx_diff.py 2>&1 | tail -9
BVG_DEBUG=1 timeout -k 10 500 python profiles/r04/mem_diag.py 128 2>&1 | grep -E "residual skip index|block plan built"
timeout -k 10 500 python profiles/transpose_bench.py 2>&1 | tail -6
