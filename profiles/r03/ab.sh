#!/bin/bash
# A/B runs of the default bench shape at 4 GiB: one line per configuration (env assignments separated by spaces, configurations by ';')
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1
out=gpurun_out/r03_ab_${TAG:-x}.txt; : > $out
IFS=';' read -ra CFG <<< "$CONFIGS"
for c in "${CFG[@]}"; do
  r=$(env $c BVG_DEBUG=1 timeout -k 10 300 python bench.py --target-gib ${GIB:-4} --steps ${STEPS:-5} --warmup 3 --no-cpu-baseline --no-verify ${ARGS} 2> gpurun_out/r03_ab.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G edges/s  %.2f ms/step' % (d['value']/1e9, d['ms_per_step']))")
  k=$(grep -E "scan kernel:" gpurun_out/r03_ab.err | tail -1 | sed 's/.*scan kernel: //')
  t=$(grep -E "tiers concurrent" gpurun_out/r03_ab.err | tail -1 | sed 's/.*tiers concurrent: //')
  echo "[$c] $r | $k | $t" | tee -a $out
done
