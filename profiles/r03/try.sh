#!/bin/bash
# scan-kernel parity first, then the whole GPU suite, then a short bench and an occupancy sweep (4 GiB eu15 shape)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
timeout -k 10 600 python -m pytest tests/test_gpu_scan_kernel.py -x -q > gpurun_out/r03_try_scank.txt 2>&1; rc=$?; echo "scan-kernel tests rc=$rc" | tee -a gpurun_out/r03_try_scank.txt
tail -15 gpurun_out/r03_try_scank.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_try_tests.txt 2>&1; rc=$?; echo "tests rc=$rc" | tee -a gpurun_out/r03_try_tests.txt
tail -5 gpurun_out/r03_try_tests.txt
[ $rc -ne 0 ] && exit $rc
for w in "" 8 10 12 16; do
  echo "== BVG_SCAN_WAVES=$w" >> gpurun_out/r03_try_bench.txt
  BVG_DEBUG=1 BVG_SCAN_WAVES=$w timeout -k 10 300 python bench.py --target-gib 4 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r03_try_bench.err | cut -c1-330 >> gpurun_out/r03_try_bench.txt
  grep -E "scan kernel:|tiers concurrent" gpurun_out/r03_try_bench.err | tail -2 >> gpurun_out/r03_try_bench.txt
done
cat gpurun_out/r03_try_bench.txt
