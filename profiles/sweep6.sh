export BVG_DEBUG=1
run() { timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline $2 2>&1 | grep -E "metric|tiers conc" | tail -2 | python -c "
import sys,json
t=[]
for l in sys.stdin:
    if l.startswith('[bvg]'): t.append(l.strip().replace('[bvg] ',''))
    else:
        d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']), ' | '.join(t[-1:]))"; }
for sh in eu web w0; do for bb in 16384 32768 65536 131072 262144; do echo "$sh bb=$bb : $(run $sh "--block-bits $bb")"; done; done
