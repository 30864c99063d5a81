#!/bin/bash
# r02: the default bench line, the torchrun-1-rank line (process group + RCCL all-reduce with one rank) and the r01 shape
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r02_bench_default.log 2>&1; tail -c 3000 gpurun_out/r02_bench_default.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --shape eu --target-gib 2 --no-cpu-baseline > gpurun_out/r02_bench_torchrun1.log 2>&1; grep '^{' gpurun_out/r02_bench_torchrun1.log | cut -c1-600 || tail -5 gpurun_out/r02_bench_torchrun1.log
