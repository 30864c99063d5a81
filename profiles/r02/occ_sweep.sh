#!/bin/bash
# r02: sensitivity of the eu scan to resident waves per CU (LDS footprint per wavefront), 2 GiB eu stream.
# usage: bash profiles/r02/occ_sweep.sh  (on the GPU box; writes gpurun_out/r02_occ.txt)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
out=gpurun_out/r02_occ.txt; : > $out
run() { # pool stage
  local tag="pool=$1 stage=$2"
  local env=""
  [ "$1" != "-" ] && export BVG_POOL=$1 || unset BVG_POOL
  [ "$2" != "-" ] && export BVG_STAGE=$2 || unset BVG_STAGE
  timeout -k 10 300 python bench.py --shape ${SHAPE:-eu} --steps 3 --warmup 2 --target-gib 2 --no-cpu-baseline 2>&1 | grep -E '^\{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$tag: %.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))" >> $out
}
for cfg in "- -" "2752 512" "2048 512" "1536 256" "1024 256" "5000 512" "7000 512" "1536 512" "2048 256"; do run $cfg; done
cat $out
