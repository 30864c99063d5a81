#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BVG_WG=0
run() { python bench.py --shape ${SHAPE:-w0} --steps 3 --warmup 2 --target-gib 1 --no-cpu-baseline --no-verify "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('kernel %.2f ms  %.1f Gedges/s'%(d['roofline']['kernel_ms'], d['value']/1e9))"; }
echo "base: $(run)"
for st in 512 1024 2048; do echo "stage $st: $(BVG_STAGE=$st run)"; done
for p in 1536 2048 3072; do echo "pool $p: $(BVG_POOL=$p run)"; done
echo "emit0: $(BVG_EMIT=0 run)"
echo "dbg7: $(BVG_DBG=7 run)"
echo "dbg7 stage1024: $(BVG_DBG=7 BVG_STAGE=1024 run)"
echo "noskip: $(BVG_NOSKIP=1 run)"
