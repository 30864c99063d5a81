#!/bin/bash
# r03 baseline at the start of the round: GPU parity suite, a short eu15-shaped bench (4 GiB), the section timers of the prof build
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_base_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r03_base_tests.txt
tail -3 gpurun_out/r03_base_tests.txt
timeout -k 10 300 python bench.py --target-gib 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_base_bench.json 2> gpurun_out/r03_base_bench.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/r03_base_bench.json
out=gpurun_out/r03_base_prof.txt; : > $out
for sh in eu15; do
  echo "== $sh prof (libbvgraph_hip_prof.so, BVG_DBG=64, 1 GiB, one steady-state scan)" >> $out
  BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvgraph_hip_prof.so BVG_DEBUG=1 BVG_DBG=64 timeout -k 10 300 python bench.py --shape $sh --steps 1 --warmup 0 --target-gib 1 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|wave-cycles|phase 1 split|tiers concurrent|^\{" | tail -6 | cut -c1-400 >> $out
done
cat $out
