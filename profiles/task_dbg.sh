# time split of the task kernel: dbg 8 = tasks, +1 skip emission, +2 skip residual decode
run() { BVG_DEBUG=1 BVG_DBG=$3 timeout 300 python bench.py --shape $1 --steps 2 --warmup 1 --target-gib $2 --no-cpu-baseline 2>&1 | grep -E "^\{|tier|conc" | tail -4 | cut -c1-200; }
for sh in eu web; do for m in 0 1 3 8 9 11; do echo "== $sh dbg=$m"; run $sh 1 $m; done; done
