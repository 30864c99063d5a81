#!/bin/bash
# Round-1 measurement batch (run on the GPU box through gpurun): bench lines for the three workload shapes,
# rocprofv3 kernel stats and separate PMC passes for the headline (eu) workload, and a 1-rank RCCL smoke of the N>1 path.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
(time python bench.py) > $O/bench_eu.log 2>&1
python bench.py --shape web > $O/bench_web.log 2>&1
python bench.py --shape w0 > $O/bench_w0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_valu -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > $O/pmc_valu.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_misc -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > $O/pmc_misc.log 2>&1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --target-gib 2 --no-cpu-baseline > $O/dist1.log 2>&1
grep -h metric $O/bench_eu.log $O/bench_web.log $O/bench_w0.log $O/dist1.log | cut -c1-220
