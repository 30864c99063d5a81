for sh in web eu; do
for pool in 1024 2048 4096 8192; do
 for mode in "--legacy" ""; do
  r=$(BVG_POOL=$pool timeout 300 python bench.py --shape $sh --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline $mode 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']))")
  echo "$sh pool=$pool $mode : $r"
 done
done
done
