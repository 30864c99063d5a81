#!/bin/bash
# round 6, second GPU call: the new checksum-integrity test, work counts of the current kernel (make work), A/B of the header-loop rewrite
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_checksum_integrity.py tests/test_gpu_scan_kernel.py tests/test_malformed_streams.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r06_second_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r06_second_tests.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/r06_work_counts.txt
for sh in eu15 cnr; do
  env BVG_TEST_KNOBS=1 BVG_DBG=64 BVG_DEBUG=1 BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvgraph_hip_work.so timeout -k 10 300 python bench.py --shape $sh --target-gib 1 --steps 1 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg --no-wide-leg --no-real-leg 2> gpurun_out/r06_work.err > /dev/null
  echo "== $sh 1 GiB (make work, BVG_DBG=64; last scan)" >> gpurun_out/r06_work_counts.txt
  grep -E "tiers concurrent|scan kernel rows|scan kernel work" gpurun_out/r06_work.err | tail -4 >> gpurun_out/r06_work_counts.txt
  grep -E "arcs" gpurun_out/r06_work.err | tail -2 >> gpurun_out/r06_work_counts.txt
done
cat gpurun_out/r06_work_counts.txt
R05=$PWD/webgraph-big_amd/lib/libbvg_r05.so
for sh in eu15 cnr uk; do
TAG=hdr_$sh SHAPE=$sh GIB=4 CONFIGS="BVG_HIP_LIB=$R05;X=1;BVG_HIP_LIB=$R05;X=2" bash profiles/r06/ab.sh || exit 1
done
