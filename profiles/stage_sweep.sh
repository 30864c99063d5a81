run() { BVG_STAGE=$3 timeout 600 python bench.py --shape $1 --steps 3 --warmup 1 --target-gib $2 --no-cpu-baseline --no-verify 2>&1 | grep -E "^\{|Error|error" | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()[:150]); continue
    d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms'%(d['value']/1e9, d['roofline']['kernel_ms']))"; }
for sh in ${SHAPES:-w0 web eu}; do for st in 256 512 1024 2048; do echo "$sh stage=$st: $(run $sh 2 $st)"; done; done
