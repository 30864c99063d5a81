#!/bin/bash
# whole GPU suite, then the default bench (full eu15 mosaic) and round 2's monoculture for comparison
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_full_tests.txt 2>&1; rc=$?; echo "tests rc=$rc" | tee -a gpurun_out/r03_full_tests.txt
tail -4 gpurun_out/r03_full_tests.txt
[ $rc -ne 0 ] && exit $rc
unset BVG_TEST_KNOBS
BVG_DEBUG=1 timeout -k 10 600 python bench.py > gpurun_out/r03_full_bench.json 2> gpurun_out/r03_full_bench.err; echo "bench rc=$?"
cut -c1-600 gpurun_out/r03_full_bench.json; grep -E "scan kernel:|tiers concurrent|skip index" gpurun_out/r03_full_bench.err | tail -3
timeout -k 10 600 python bench.py --shape eu15mono --no-cpu-baseline > gpurun_out/r03_full_bench_mono.json 2> gpurun_out/r03_full_bench_mono.err; echo "bench mono rc=$?"
cut -c1-300 gpurun_out/r03_full_bench_mono.json
