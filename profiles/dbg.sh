export BVG_DEBUG=1
for sh in web eu; do for dbg in 0 1 2 3 7; do
 export BVG_DBG=$dbg
 t=$(timeout 300 python - <<PY 2>&1 | grep "tier0" | tail -1
import sys; sys.argv=['bench.py','--shape','$sh','--steps','1','--warmup','0','--target-gib','1','--no-cpu-baseline']
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
except BaseException as e:
    pass
PY
)
 echo "$sh dbg=$dbg : $t"
done; done
