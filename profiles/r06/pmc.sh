#!/bin/bash
# r04: PMC passes (each its own run; never combined with tracing domains) over a bench workload; summary per kernel of the LAST scan.
# usage: bash profiles/r06/pmc.sh <tag> [bench args...]     (default workload: --shape eu --target-gib 2)
cd "$(dirname "$0")/../.."; R=$PWD
tag=${1:-pmc}; shift
args="${@:---shape eu --target-gib 2}"
O=$R/gpurun_out/r06_pmc_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
SETS=${SETS:-4}; for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE"; do
  i=$((i+1)); [ $i -gt $SETS ] && break
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py $args --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg --no-wide-leg --no-real-leg > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
cd $R
python3 profiles/r03/pmc_summary.py $O | tee gpurun_out/r06_pmc_${tag}_summary.txt
