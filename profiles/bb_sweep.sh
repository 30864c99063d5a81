run() { timeout 600 python bench.py --shape $1 --steps 3 --warmup 1 --target-gib 2 --no-cpu-baseline --no-verify --block-bits $2 2>&1 | grep -E "^\{|Error|error" | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()[:150]); continue
    d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))"; }
for sh in eu web w0; do for bb in 16384 32768 65536 131072; do echo "$sh block_bits=$bb: $(run $sh $bb)"; done; done
