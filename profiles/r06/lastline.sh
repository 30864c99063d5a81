#!/bin/bash
# the default bench line at the final tree with profiles/traffic.json in place (roofline.traffic, roofline_valu), then the task-size A/Bs
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r06_eu15_bench.json 2> gpurun_out/r06_eu15_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_eu15_bench.json'))
print('%.1f G edges/s; traffic %s; valu %s' % (d['value']/1e9, d['roofline']['traffic'], {k: d['roofline_valu'][k] for k in ('frac','frac_half_rate_peak','valu_per_arc','active_lanes')} if d.get('roofline_valu') else None))"
bash profiles/r06/tune.sh
