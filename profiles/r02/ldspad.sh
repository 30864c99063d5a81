#!/bin/bash
# r02: does the SIMD saturate at two wavefronts?  Same kernel, same pool, fewer resident wavefronts (unused LDS padding).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r02_ldspad.txt; : > $out
for pad in 0 3000 7000 13000 21000 34000 62000; do
  BVG_LDSPAD=$pad timeout -k 10 300 python bench.py --shape eu --steps 3 --warmup 2 --target-gib 2 --no-cpu-baseline 2>&1 | grep -E '^\{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pad=$pad: %.1f Gedges/s kernel %.1f ms'%(d['value']/1e9, d['roofline']['kernel_ms']))" >> $out
done
cat $out
