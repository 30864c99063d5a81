export BVG_DEBUG=1
for sh in web; do for bb in 8192 32768 131072 524288; do for dbg in 7 0; do
 export BVG_DBG=$dbg
 t=$(timeout 300 python - <<PY 2>&1 | grep "tier0" | tail -1
import sys; sys.argv=['bench.py','--shape','$sh','--steps','1','--warmup','0','--target-gib','1','--no-cpu-baseline','--block-bits','$bb']
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
except BaseException as e:
    pass
PY
)
 echo "$sh bb=$bb dbg=$dbg : $t"
done; done; done
