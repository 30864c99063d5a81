#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
SETS=2 bash profiles/r05/pmc.sh flat_cnr --shape cnr --target-gib 2 > gpurun_out/r05_pmc_cnr.log 2>&1; grep -A3 "flat_kernel" gpurun_out/r05_pmc_cnr.log | head -8
export BVG_FLAT_RECS=128
SETS=2 bash profiles/r05/pmc.sh flat_cnr128 --shape cnr --target-gib 2 > gpurun_out/r05_pmc_cnr128.log 2>&1; grep -A3 "flat_kernel" gpurun_out/r05_pmc_cnr128.log | head -8
unset BVG_FLAT_RECS
TAG=flat2_cnr SHAPE=cnr GIB=2 CONFIGS="BVG_FLAT=0;BVG_FLAT_RECS=128 BVG_SCAN_WAVES=16;BVG_FLAT_RECS=128 BVG_SCAN_WAVES=20;BVG_FLAT_RECS=128 BVG_SCAN_WAVES=24;BVG_FLAT_RECS=64 BVG_SCAN_WAVES=24;BVG_FLAT_RECS=128 BVG_SCAN_WAVES=12" bash profiles/r05/ab.sh
TAG=flat2_eu15 SHAPE=eu15 GIB=2 CONFIGS="BVG_FLAT=0;BVG_SCAN_WAVES=16;BVG_SCAN_WAVES=14;BVG_SCAN_WAVES=12;BVG_SCAN_WAVES=20" bash profiles/r05/ab.sh
