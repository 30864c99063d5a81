run() { BVG_DEBUG=1 BVG_DBG=$2 timeout 600 python bench.py --shape $1 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{|skip index|conc" | tail -3 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[bvg]'): print(l.strip()[:150], end=' | '); continue
    d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d idx %.2f GB'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1), d['roofline']['index_bytes_per_launch']/1e9))"; }
for sh in ${SHAPES:-eu web w0}; do for m in ${MODES:-0 8}; do echo "$sh dbg=$m: $(run $sh $m)"; done; done
