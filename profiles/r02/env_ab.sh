#!/bin/bash
# r02: A/B of run-time switches on one build: bash profiles/r02/env_ab.sh "<shapes>" "VAR=val ..." "VAR=val ..."   ("-" = no switch)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/r02_env_ab.txt; : > $out
shapes="$1"; shift
for cfg in "$@"; do for sh in $shapes; do
  ( [ "$cfg" != "-" ] && export $cfg; timeout -k 10 300 python bench.py --shape $sh --steps ${STEPS:-3} --warmup 2 --target-gib ${GIB:-2} --no-cpu-baseline 2>&1 | grep -E '^\{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('[$cfg] $sh: %.1f Gedges/s kernel %.2f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))" ) >> $out
done; done
cat $out
