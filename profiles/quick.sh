export BVG_DEBUG=1
run() { timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline $2 2>&1 | grep -E "metric|tier" | tail -6 | python -c "
import sys,json
t=[]
for l in sys.stdin:
    if l.startswith('[bvg]'): t.append(l.strip().replace('[bvg] ',''))
    else:
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']), ' | '.join(t[-3:]))"; }
for sh in web eu w0; do echo "$sh : $(run $sh)"; done
export BVG_NOPREDICT=1
for sh in web eu; do echo "$sh nopredict : $(run $sh)"; done
