run() { BVG_NOPREDICT=$NP BVG_DEBUG=1 BVG_DBG=$3 timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib $2 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|tier0" | tail -${TAILN:-2} | cut -c1-170; }
for sh in ${SHAPES:-eu web}; do for m in $MODES; do echo "== $sh dbg=$m"; run $sh 1 $m; done; done
