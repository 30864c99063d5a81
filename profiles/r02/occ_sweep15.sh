#!/bin/bash
# r02: pool / window sweep on the eu15 shape (4 GiB), looking for the occupancy step that pays
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
out=gpurun_out/r02_occ15.txt; : > $out
run() {
  [ "$1" != "-" ] && export BVG_POOL=$1 || unset BVG_POOL
  [ "$2" != "-" ] && export BVG_STAGE=$2 || unset BVG_STAGE
  timeout -k 10 300 python bench.py --shape eu15 --steps 3 --warmup 2 --target-gib 4 --no-cpu-baseline 2>&1 | grep -E '^\{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pool=$1 stage=$2: %.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))" >> $out
}
for cfg in "- -" "3584 1024" "3584 512" "4096 512" "3072 512" "5120 1024" "2560 512"; do run $cfg; done
cat $out
