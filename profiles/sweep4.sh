export BVG_DEBUG=1
run() { timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline $2 2>&1 | grep -E "metric|tier" | tail -4 | python -c "
import sys,json
t=[]
for l in sys.stdin:
    if l.startswith('[bvg]'): t.append(l.split(':')[0].replace('[bvg] ','')+'='+l.split(',')[-1].strip())
    else:
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']), ' '.join(t[-3:]))"; }
for sh in web eu; do
 if [ $sh = eu ]; then export BVG_POOL=3584; else unset BVG_POOL; fi
 for st in 256 512 1024; do export BVG_STAGE=$st; echo "$sh stage=$st : $(run $sh)"; done
 unset BVG_STAGE
 for bb in 4096 8192 16384 65536; do echo "$sh bb=$bb : $(run $sh "--block-bits $bb")"; done
done
