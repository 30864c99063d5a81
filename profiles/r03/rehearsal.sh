#!/bin/bash
# N ranks on the ONE GPU of the box (gloo for the 16-byte reduction: RCCL wants a device per rank): `python bench.py --gpus N` starts its own ranks;
# every rank builds the skip index of its shard only; the reduced {arcs, chk} must equal the one-piece scan (bench.py asserts it)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r03_strong_rehearsal.txt; : > $out
for n in 1 2 4; do
  echo "== python bench.py --gpus $n --backend gloo --one-device --target-gib 2 --steps 3 --warmup 1 --no-cpu-baseline" >> $out
  timeout -k 10 500 python bench.py --gpus $n --backend gloo --one-device --target-gib 2 --steps 3 --warmup 1 --no-cpu-baseline 2>> gpurun_out/r03_strong_rehearsal.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('n_gpus %d scaling %s: %.1f G edges/s, checksum %s, arcs %d, skip entries rank0 %d / all ranks %d, lean blocks all ranks %d, nodes_per_gpu %d' % (d['n_gpus'], d['scaling'], d['value']/1e9, d['checksum'], d['arcs'], d['index']['skip_entries_rank0'], d['index']['skip_entries_all_ranks'], d['index']['lean_blocks_all_ranks'], d['config']['nodes_per_gpu']))
" >> $out
done
cat $out
