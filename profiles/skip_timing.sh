#!/bin/bash
# kernel time of one steady-state scan with phases skipped (BVG_DBG bits: 1 emission, 2 residual decode, 4 header parse, 128 position loop)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BVG_WG=0
for sh in ${SHAPES:-eu}; do for m in ${MODES:-0 128 1 2 3 7}; do
  BVG_DBG=$m python bench.py --shape $sh --steps 3 --warmup 2 --target-gib 1 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('$sh dbg=$m kernel %.2f ms'%d['roofline']['kernel_ms'])"
done; done
