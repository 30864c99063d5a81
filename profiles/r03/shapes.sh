#!/bin/bash
# every bench shape at ${GIB:-4} GiB, verified against the oracle, one line each
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r03_shapes.txt; : > $out
for sh in ${SHAPES:-eu15 eu15mono eu web w0}; do
  r=$(timeout -k 10 400 python bench.py --shape $sh --target-gib ${GIB:-4} --steps 5 --warmup 3 --no-cpu-baseline 2> gpurun_out/r03_shapes.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G edges/s  %.2f ms/step  lean blocks %s' % (d['value']/1e9, d['ms_per_step'], d['index']['lean_blocks_rank0']))")
  echo "[$sh] $r" | tee -a $out
done
