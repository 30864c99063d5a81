#!/bin/bash
# A/B runs of a bench shape: one line per configuration (env assignments separated by spaces, configurations by ';').
# TAG=name CONFIGS="A=1;B=2 C=3" [GIB=4 (0 = the shape's full size)] [SHAPE=eu15] [STEPS=5] bash profiles/r06/ab.sh
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
out=gpurun_out/r06_ab_${TAG:-x}.txt; : > $out
IFS=';' read -ra CFG <<< "$CONFIGS"
size=""; [ "${GIB:-4}" != "0" ] && size="--target-gib ${GIB:-4}"
for c in "${CFG[@]}"; do
  r=$(env $c BVG_DEBUG=1 timeout -k 10 400 python bench.py --shape ${SHAPE:-eu15} $size --steps ${STEPS:-5} --warmup 3 --no-cpu-baseline --no-verify --no-index-leg --no-wide-leg --no-real-leg ${ARGS} 2> gpurun_out/r06_ab.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G edges/s  %.2f ms/step' % (d['value']/1e9, d['ms_per_step']))")
  k=$(grep -E "(scan|flat) kernel:" gpurun_out/r06_ab.err | tail -1 | sed "s/.*kernel: //")
  t=$(grep -E "tiers concurrent" gpurun_out/r06_ab.err | tail -1 | sed 's/.*tiers concurrent: //')
  echo "[$c] $r | $k | $t" | tee -a $out
done
