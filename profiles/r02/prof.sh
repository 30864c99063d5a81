#!/bin/bash
# r02: section timers (prof build: wave-cycles per section of the row kernel) + work counters, steady-state scans, 1 GiB streams
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r02_prof.txt; : > $out
for sh in ${SHAPES:-eu eu15}; do
  echo "== $sh prof" >> $out; BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/${LIB:-libbvgraph_hip_prof.so} BVG_DEBUG=1 BVG_DBG=${DBG:-64} timeout -k 10 300 python bench.py --shape $sh --steps 1 --warmup 0 --target-gib 1 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|wave-cycles|phase 1 split|^\{" | tail -4 | cut -c1-330 >> $out
done
cat $out
