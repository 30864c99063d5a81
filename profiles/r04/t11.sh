#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t11_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04_t11_tests.log
export BVG_TEST_KNOBS=1
for cfg in "BVG_INDEX_WALK=0" "BVG_NOP=1"; do
  echo "== $cfg"
  env $cfg BVG_DEBUG=1 timeout -k 10 500 python profiles/r04/mem_diag.py 128 2>&1 | grep -E "after build_index|block plan built|residual skip index|lean_blocks|failures" | head -12
done
