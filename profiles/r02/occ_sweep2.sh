#!/bin/bash
# r02: pool / window combinations landing exactly on 9, 10 and 12 wavefronts per CU (eu and eu15, 2 GiB)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
out=gpurun_out/r02_occ2.txt; : > $out
run() {
  [ "$2" != "-" ] && export BVG_POOL=$2 || unset BVG_POOL
  [ "$3" != "-" ] && export BVG_STAGE=$3 || unset BVG_STAGE
  timeout -k 10 300 python bench.py --shape $1 --steps 3 --warmup 2 --target-gib 2 --no-cpu-baseline 2>&1 | grep -E '^\{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1 pool=$2 stage=$3: %.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))" >> $out
}
for sh in eu eu15; do for cfg in "- -" "3456 256" "3072 256" "3008 256" "2496 256" "3264 128"; do run $sh $cfg; done; done
cat $out
