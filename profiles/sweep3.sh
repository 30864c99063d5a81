export BVG_DEBUG=1
for sh in web eu w0; do
 for pool in default 768 1024 1536 2048 3072 4096 5120; do
  if [ $pool = default ]; then unset BVG_POOL; else export BVG_POOL=$pool; fi
  r=$(timeout 300 python bench.py --shape $sh --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline 2>&1 | grep -E "metric|tier" | tail -4 | python -c "
import sys,json
t=[]
for l in sys.stdin:
    if l.startswith('[bvg]'): t.append(l.split(':')[0].replace('[bvg] ','')+'='+l.split(',')[-1].strip())
    else:
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']), ' '.join(t))")
  echo "$sh pool=$pool : $r"
 done
done
