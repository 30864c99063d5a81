#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
: > gpurun_out/r04_ab_blockbits.txt
for bb in 32768 24576 49152 65536; do
  r=$(BVG_DEBUG=1 timeout -k 10 400 python bench.py --block-bits $bb --steps 3 --warmup 3 --no-cpu-baseline --no-verify --no-index-leg 2> gpurun_out/r04_bb.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G edges/s  %.2f ms/step  resident %.1f GB' % (d['value']/1e9, d['ms_per_step'], d['hbm_resident_bytes']/1e9))")
  t=$(grep -E "tiers concurrent" gpurun_out/r04_bb.err | tail -1 | sed 's/.*tiers concurrent: //')
  echo "[block_bits $bb] $r | $t" | tee -a gpurun_out/r04_ab_blockbits.txt
done
