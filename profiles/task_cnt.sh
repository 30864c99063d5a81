run() { BVG_DEBUG=1 BVG_DBG=$3 timeout 300 python bench.py --shape $1 --steps 1 --warmup 0 --target-gib $2 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|wave-cycles|phase 1 split" | tail -3 | cut -c1-200; }
for sh in ${SHAPES:-eu web w0}; do for m in ${MODES:-64}; do echo "== $sh dbg=$m:"; BVG_EMIT=${EMIT:-1} run $sh 1 $m; done; done
