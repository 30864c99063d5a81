#!/bin/bash
# round 6, closing evidence: the default bench line with every leg (it quotes profiles/traffic.json now), fresh fuzz seeds, the 1 / 2 / 4 / 5-rank rehearsal on one GPU
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r06_eu15_bench.json 2> gpurun_out/r06_eu15_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_eu15_bench.json'))
print('%.1f G edges/s, %.1f ms/step; no index %.1f, marks only %.1f, wide %.1f G; real graph %.1f G; valu frac %s' % (d['value']/1e9, d['ms_per_step'], d['value_no_index']/1e9, d['value_marks_only']/1e9, d['value_wide']/1e9, d['real_graph']['value']/1e9, d.get('roofline_valu',{}).get('frac')))
print('wide', d['wide'])"
N=1500 SEEDS="601 602" bash profiles/r06/fuzz.sh 2>&1 | tail -6
bash profiles/r06/rehearsal.sh 2>&1 | tail -6
