#!/bin/bash
# r04: the materialising lean kernel: parity tests, then the materialise bench (old row kernel via BVG_SCANK=0 vs the lean kernel)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_scan_kernel.py tests/test_gpu_api.py tests/test_malformed_streams.py tests/test_gpu_unknobbed.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04_t2_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_t2_tests.log
export BVG_TEST_KNOBS=1
for cfg in "BVG_SCANK=0" "BVG_NOP=1"; do
  for shape in eu web; do
    echo "[$cfg $shape] $(env $cfg timeout -k 10 300 python profiles/mat_bench.py $shape 2>&1 | tail -2 | head -1)" | tee -a gpurun_out/r04_mat_first.txt
  done
done
