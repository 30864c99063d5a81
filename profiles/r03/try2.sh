#!/bin/bash
# scan-kernel parity, then an occupancy sweep (4 GiB eu15 shape); WAVES="8 10 12" selects
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
timeout -k 10 600 python -m pytest tests/test_gpu_scan_kernel.py tests/test_gpu_modes.py -x -q > gpurun_out/r03_try_scank.txt 2>&1; rc=$?; echo "scan-kernel tests rc=$rc" | tee -a gpurun_out/r03_try_scank.txt
tail -5 gpurun_out/r03_try_scank.txt
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/r03_try_bench.txt
for w in ${WAVES:-8 10 12 14 16}; do
  echo "== BVG_SCAN_WAVES=$w ${ENVS}" >> gpurun_out/r03_try_bench.txt
  env ${ENVS} BVG_DEBUG=1 BVG_DBG=64 BVG_SCAN_WAVES=$w timeout -k 10 300 python bench.py --target-gib 4 --steps 5 --warmup 3 --no-cpu-baseline 2> gpurun_out/r03_try_bench.err | cut -c1-330 >> gpurun_out/r03_try_bench.txt
  grep -E "scan kernel:|scan kernel rows|tiers concurrent" gpurun_out/r03_try_bench.err | tail -3 >> gpurun_out/r03_try_bench.txt
done
cat gpurun_out/r03_try_bench.txt
