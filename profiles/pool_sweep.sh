run() { BVG_POOL=$4 BVG_DBG=$3 timeout 600 python bench.py --shape $1 --steps 3 --warmup 1 --target-gib $2 --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error" | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()[:150]); continue
    d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))"; }
for p in $POOLS; do echo "$SHAPE dbg=$MODE pool=$p: $(run $SHAPE 2 $MODE $p)"; done
