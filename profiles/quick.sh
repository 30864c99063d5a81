export BVG_DEBUG=1
run() { timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline $2 2>&1 | grep -E "metric|tier" | tail -6 | python -c "
import sys,json
t=[]
for l in sys.stdin:
    if l.startswith('[bvg]'): t.append(l.split(':')[0].replace('[bvg] ','')+'='+l.split(',')[-1].strip())
    else:
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']), ' '.join(t[-5:]))"; }
for sh in web eu w0; do echo "$sh rows : $(run $sh)"; echo "$sh stream : $(run $sh --stream)"; done
