for sh in web eu; do
for bb in 8192 32768 131072 524288; do
 for mode in "" "--legacy"; do
  r=$(timeout 300 python bench.py --shape $sh --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline --block-bits $bb $mode 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gedges/s kernel %.1f ms slow %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']))")
  echo "$sh bb=$bb $mode : $r"
 done
done
done
for th in 4 24 40; do r=$(timeout 300 python bench.py --shape web --steps 2 --warmup 1 --target-gib 1 --no-cpu-baseline --grab-threshold $th 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gedges/s kernel %.1f ms'%(d['value']/1e9, d['roofline']['kernel_ms']))"); echo "web th=$th : $r"; done
