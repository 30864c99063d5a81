#!/bin/bash
# round 6, fourth GPU call: what slowed the WW build down?  two-chain residual passes (RU2) on / off x WW on / off, rates and instruction counters
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out
RU2=$R/webgraph-big_amd/lib/libbvg_exp_ru2.so
TAG=ww2_eu15 SHAPE=eu15 GIB=2 CONFIGS="BVG_HIP_LIB=$RU2 BVG_DBG=2048;BVG_HIP_LIB=$RU2;BVG_DBG=2048;X=1" bash profiles/r06/ab.sh
TAG=ww2_uk SHAPE=uk GIB=2 CONFIGS="BVG_HIP_LIB=$RU2 BVG_DBG=2048;BVG_HIP_LIB=$RU2;BVG_DBG=2048;X=1" bash profiles/r06/ab.sh
BVG_TEST_KNOBS=1 BVG_DBG=2048 SETS=2 bash profiles/r06/pmc.sh wwoff --shape eu15 --target-gib 2 > /dev/null 2>&1
BVG_TEST_KNOBS=1 SETS=2 bash profiles/r06/pmc.sh wwon --shape eu15 --target-gib 2 > /dev/null 2>&1
BVG_TEST_KNOBS=1 BVG_DBG=2048 BVG_HIP_LIB=$RU2 SETS=2 bash profiles/r06/pmc.sh wwoff_ru2 --shape eu15 --target-gib 2 > /dev/null 2>&1
for t in wwoff wwon wwoff_ru2; do echo "== $t"; head -4 gpurun_out/r06_pmc_${t}_summary.txt; tail -1 gpurun_out/r06_pmc_${t}_summary.txt; done
