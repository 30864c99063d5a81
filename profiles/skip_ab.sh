# A/B: residual skip index (default) vs BVG_NOSKIP=1, pipelined emission (dbg 0) vs task emission (dbg 8)
run() { BVG_DEBUG=1 BVG_DBG=$3 timeout 300 python bench.py --shape $1 --steps 3 --warmup 1 --target-gib $2 --no-cpu-baseline 2>&1 | grep -E "^\{|skip index" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[bvg]'): print(l.strip(), end=' | '); continue
    d=json.loads(l); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d.get('slow_blocks',-1)))"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
BVG_DBG=8 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for sh in eu web w0; do for m in 0 8; do echo "$sh dbg=$m noskip: $(BVG_NOSKIP=1 run $sh 1 $m)"; echo "$sh dbg=$m skip  : $(run $sh 1 $m)"; done; done
