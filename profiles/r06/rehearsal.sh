#!/bin/bash
# N ranks on the ONE GPU of the box (gloo for the 16-byte reduction: RCCL wants a device per rank; the pool lets at most 6 processes share
# the card, and the launcher counts: 5 ranks is the largest rehearsal this box holds): `python bench.py --gpus N` starts its own ranks; rank 0 generates the bases
# once and shares them through /dev/shm; every rank builds the skip index of ITS shard only, gates its own shard (tiles against the CPU
# oracle, the shard against the sum of its pieces) and reports its kernel time; the reduced {arcs, chk} must not depend on N.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r06_strong_rehearsal.txt; : > $out
for n in 1 2 4 5; do
  echo "== python bench.py --gpus $n --backend gloo --one-device --target-gib 3 --steps 3 --warmup 1 --no-cpu-baseline --no-real-leg --no-wide-leg [+ --verify-whole at N = 2]" >> $out
  timeout -k 10 500 python bench.py --gpus $n --backend gloo --one-device --target-gib 3 --steps 3 --warmup 1 --no-cpu-baseline --no-real-leg --no-wide-leg $( [ $n = 2 ] && echo --verify-whole ) 2>> gpurun_out/r06_strong_rehearsal.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('n_gpus %d scaling %s: %.1f G edges/s, checksum %s, arcs %d, skip entries per rank %s (sum %d), lean blocks all ranks %d, nodes_per_gpu(rank 0) %d, per-rank kernel ms %s, imbalance %.3f, generate_s %.1f' % (d['n_gpus'], d['scaling'], d['value']/1e9, d['checksum'], d['arcs'], d['per_rank_index_entries'], d['index']['skip_entries_all_ranks'], d['index']['lean_blocks_all_ranks'], d['config']['nodes_per_gpu'], ['%.1f' % v for v in d['per_rank_kernel_ms']['all']], d['imbalance'], d['host']['generate_s']))
" >> $out
done
cat $out
