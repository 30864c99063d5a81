#!/bin/bash
# round 5: the skip index's granularity re-measured on scan_kernel with the linear checksum: rate against entries (index bytes) on the default shape and cnr
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
TAG=skipgran_eu15 SHAPE=eu15 GIB=4 STEPS=6 CONFIGS="X=1;BVG_SKIP_GRAN=16,32;BVG_SKIP_GRAN=32,32;BVG_SKIP_GRAN=24,16;BVG_SKIP_GRAN=8,8" bash profiles/r05/ab.sh
grep -h "index_bytes\|entries" gpurun_out/r05_ab.err | tail -2
TAG=skipgran_cnr SHAPE=cnr GIB=4 STEPS=6 CONFIGS="X=1;BVG_SKIP_GRAN=16,16;BVG_SKIP_GRAN=32,32" bash profiles/r05/ab.sh
