run() { BVG_EMIT=1 BVG_PASSCOST=$3 timeout 600 python bench.py --shape $1 --steps 3 --warmup 2 --target-gib $2 --no-cpu-baseline --no-verify 2>&1 | grep -E "^\{|Error|error" | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()[:150]); continue
    d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms'%(d['value']/1e9, d['roofline']['kernel_ms']))"; }
for sh in eu web w0; do for pc in 4 8 11 14 20 30; do echo "$sh pass_cost=$pc: $(run $sh 2 $pc)"; done; done
