#!/bin/bash
# the four bench lines of final.sh alone (the kernel-stats / counter passes of the last full batch stay valid when only host-side
# reporting changed)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
(time python bench.py) > $O/bench_eu.log 2>&1
python bench.py --shape web > $O/bench_web.log 2>&1
python bench.py --shape w0 > $O/bench_w0.log 2>&1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --target-gib 2 --no-cpu-baseline > $O/dist1.log 2>&1
grep -h metric $O/bench_eu.log $O/bench_web.log $O/bench_w0.log $O/dist1.log | cut -c1-160
