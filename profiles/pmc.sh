#!/bin/bash
# usage: profiles/pmc.sh <tag> <bench args...>   — separate rocprofv3 PMC passes (never combined with sys-trace)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/$tag/a -- python bench.py "$@" --no-cpu-baseline > gpurun_out/$tag/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/$tag/b -- python bench.py "$@" --no-cpu-baseline > gpurun_out/$tag/b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/$tag/c -- python bench.py "$@" --no-cpu-baseline > gpurun_out/$tag/c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/$tag/d -- python bench.py "$@" --no-cpu-baseline > gpurun_out/$tag/d.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/e -- python bench.py "$@" --no-cpu-baseline > gpurun_out/$tag/e.log 2>&1
grep -h metric gpurun_out/$tag/a.log | cut -c1-120
