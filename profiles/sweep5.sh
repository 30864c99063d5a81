export BVG_DEBUG=1
run() { timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline $2 2>&1 | grep -E "metric|tiers conc" | tail -2 | python -c "
import sys,json
t=[]
for l in sys.stdin:
    if l.startswith('[bvg]'): t.append(l.strip().replace('[bvg] ',''))
    else:
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']), ' | '.join(t[-1:]))"; }
for sh in eu; do for pool in default 1536 2048 2560 3072; do
  if [ $pool = default ]; then unset BVG_POOL; else export BVG_POOL=$pool; fi
  echo "$sh pool=$pool : $(run $sh)"; done; done
for sh in web; do for pool in default 512 768; do
  if [ $pool = default ]; then unset BVG_POOL; else export BVG_POOL=$pool; fi
  echo "$sh pool=$pool : $(run $sh)"; done; done
