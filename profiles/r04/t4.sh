#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_t4_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04_t4_tests.log
timeout -k 10 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/r04_t4_eu15.json 2> gpurun_out/r04_t4_eu15.err; echo "eu15 rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r04_t4_eu15.json')); print('%.1f G, no-index %.1f G, index build %.2f s, resident %.1f GB' % (d['value']/1e9, d['value_no_index']/1e9, d['index_build_s'], d['hbm_resident_bytes']/1e9))"
