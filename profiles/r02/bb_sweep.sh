#!/bin/bash
# r02: block size (compressed bits per wavefront) on the eu15 and eu shapes, 4 GiB
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r02_bb.txt; : > $out
for sh in eu15 eu; do for bb in 16384 24576 32768 49152 65536; do
  timeout -k 10 300 python bench.py --shape $sh --steps 3 --warmup 2 --target-gib 4 --no-cpu-baseline --block-bits $bb 2>&1 | grep -E '^\{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$sh block_bits=$bb: %.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))" >> $out
done; done
cat $out
