#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
BVG_TEST_KNOBS=1 BVG_DBG=8192 SETS=2 bash profiles/r06/pmc.sh zeon --shape eu15 --target-gib 2 > /dev/null 2>&1
BVG_TEST_KNOBS=1 SETS=2 bash profiles/r06/pmc.sh zeoff --shape eu15 --target-gib 2 > /dev/null 2>&1
for t in zeon zeoff; do echo "== $t"; head -4 gpurun_out/r06_pmc_${t}_summary.txt; tail -1 gpurun_out/r06_pmc_${t}_summary.txt; done
