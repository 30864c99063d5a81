#!/bin/bash
# r02: rehearsal of the N>1 strong-scaling path on a ONE-GPU box: 2 and 4 ranks share cuda:0, the reduction runs over gloo.
# (The RCCL form of the same code path runs with one rank in bench_try.sh; 8-GPU runs belong to the driver.)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for n in 2 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --shape eu --target-gib 2 --no-cpu-baseline --backend gloo --one-device > gpurun_out/r02_strong_$n.log 2>&1
  grep '^{' gpurun_out/r02_strong_$n.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ranks $n:', d['scaling'], '%.1f G edges/s'%(d['value']/1e9), 'arcs', d['arcs'], 'chk', d['checksum'], 'per-gpu arcs', d['config']['arcs_per_gpu'], d['config']['sharding'])" || tail -20 gpurun_out/r02_strong_$n.log
done
python bench.py --steps 3 --warmup 1 --shape eu --target-gib 2 --no-cpu-baseline | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('1 gpu  :', d['scaling'], '%.1f G edges/s'%(d['value']/1e9), 'arcs', d['arcs'], 'chk', d['checksum'])"
