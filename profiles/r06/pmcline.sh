#!/bin/bash
# PMC passes + traffic.json + the default bench line, at the tree as it is (csrc hashes to what the entries say)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash profiles/r06/pmc.sh eu15 --shape eu15 > gpurun_out/r06_eu15_pmc_summary.txt 2>&1; tail -1 gpurun_out/r06_eu15_pmc_summary.txt
python3 profiles/r06/record.py eu15 eu15 128 1048576 r06_eu15_pmc_summary.txt | grep -E "valu_per_arc|src_id"
bash profiles/r06/pmc.sh cnr --shape cnr > gpurun_out/r06_cnr_pmc_summary.txt 2>&1; tail -1 gpurun_out/r06_cnr_pmc_summary.txt
python3 profiles/r06/record.py cnr cnr 6004 325557 r06_cnr_pmc_summary.txt | grep -E "valu_per_arc|src_id"
timeout -k 10 600 python bench.py > gpurun_out/r06_eu15_bench.json 2> gpurun_out/r06_eu15_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_eu15_bench.json'))
print('%.1f G edges/s; traffic %s; valu %s' % (d['value']/1e9, d['roofline']['traffic'], {k: d['roofline_valu'][k] for k in ('frac','frac_half_rate_peak','valu_per_arc','active_lanes')} if d.get('roofline_valu') else None))"
timeout -k 10 300 python -m pytest tests/test_gpu_giant.py tests/test_gpu_scan_kernel.py tests/test_gpu_long_codes.py -m gpu -x -q 2>&1 | tail -2
