#!/bin/bash
# section ablation of scan_kernel by instruction counters: one PMC pass (SQ_INSTS_VALU / SALU / LDS ...) per build with a section compiled out
# (-DBVG_ABLATE_x: wrong results, the section's share of the instructions), plus the bench rate of each build.   SHAPE=eu15 GIB=2 bash profiles/r06/ablate.sh
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out
sh=${SHAPE:-eu15}; out=gpurun_out/r06_ablate_$sh.txt; : > $out
for a in base Z1 Z2 Z2LOOP RESLOOP LEAF LPN; do
  lib=$R/webgraph-big_amd/lib/libbvg_exp_abl_$a.so; [ $a = base ] && lib=$R/webgraph-big_amd/lib/libbvgraph_hip.so
  echo "== $a" >> $out
  BVG_HIP_LIB=$lib SETS=2 bash profiles/r06/pmc.sh abl_${sh}_$a --shape $sh --target-gib ${GIB:-2} 2>&1 | grep -E "scan_kernel|per arc|all kernels" | head -4 >> $out
  TAG=abl_${sh}_$a SHAPE=$sh GIB=${GIB:-2} CONFIGS="BVG_HIP_LIB=$lib" bash profiles/r06/ab.sh | cut -c1-120 >> $out
done
cat $out
