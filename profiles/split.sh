# time split of the default kernel: BVG_DBG 1 = skip emission, 3 = skip emission + residual decode, 7 = + header parse
run() { BVG_DBG=$3 timeout 600 python bench.py --shape $1 --steps 3 --warmup 1 --target-gib $2 --no-cpu-baseline --no-verify 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('kernel %.1f ms'%(d['roofline']['kernel_ms']))"; }
for sh in ${SHAPES:-eu web w0}; do for m in ${MODES:-0 1 3 7}; do echo "$sh dbg=$m: $(run $sh ${GIB:-2} $m)"; done; done
