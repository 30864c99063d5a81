#!/bin/bash
# r02 A/B of library builds: bash profiles/r02/ab.sh "<shapes>" lib1.so lib2.so ...   (2 GiB streams; writes gpurun_out/r02_ab.txt)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
out=gpurun_out/r02_ab.txt; : > $out
shapes="$1"; shift
for lib in "$@"; do
  for sh in $shapes; do
    BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/$lib timeout -k 10 300 python bench.py --shape $sh --steps ${STEPS:-3} --warmup 2 --target-gib ${GIB:-2} --no-cpu-baseline 2>&1 | grep -E '^\{|rror' | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): print('$lib $sh', l.strip()[:200]); continue
    d=json.loads(l); print('$lib $sh: %.1f Gedges/s kernel %.2f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))" >> $out
  done
done
cat $out
