cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for w in 4 8 2 0; do for sh in eu15 eu; do BVG_WGC=$w timeout -k 10 300 python bench.py --shape $sh --steps 3 --warmup 2 --target-gib 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('WGC=$w $sh: %.1f Gedges/s kernel %.2f ms'%(d['value']/1e9, d['roofline']['kernel_ms']))"; done; done
