#!/bin/bash
# launch order / stream priority of the big-list tiers vs tier 0 at the full 8 GiB size
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 5 --warmup 3 --no-cpu-baseline ${ARGS} 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms'%(d['value']/1e9, d['roofline']['kernel_ms']))"; }
echo "default (classes first, high priority): $(run)"
echo "tier0 first, high prio classes: $(BVG_ORDER=1 run)"
echo "classes first, normal prio: $(BVG_PRIO=0 run)"
echo "tier0 first, normal prio: $(BVG_ORDER=1 BVG_PRIO=0 run)"
echo "tier0 first, low prio: $(BVG_ORDER=1 BVG_PRIO=-1 run)"
echo "classes first, low prio: $(BVG_PRIO=-1 run)"
