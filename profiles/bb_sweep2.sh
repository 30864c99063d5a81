run() { BVG_DBG=$3 timeout 600 python bench.py --shape $1 --steps 3 --warmup 1 --target-gib 2 --no-cpu-baseline --no-verify --block-bits $2 2>&1 | grep -E "^\{|Error|error" | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()[:150]); continue
    d=json.loads(l); print('kernel %.1f ms'%(d['roofline']['kernel_ms']))"; }
for sh in w0 web; do for bb in 32768 131072 524288; do for m in 7 3; do echo "$sh block_bits=$bb dbg=$m: $(run $sh $bb $m)"; done; done; done
