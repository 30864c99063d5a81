#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -6
export BVG_TEST_KNOBS=1
: > gpurun_out/r04_mat_bench.txt; : > gpurun_out/r04_mat_bench.jsonl
for cfg in "BVG_SCANK=0" "BVG_NOP=1"; do for shape in eu web; do env $cfg timeout -k 10 300 python profiles/mat_bench.py $shape > gpurun_out/r04_mat_one.log 2>&1; echo "[$cfg $shape] $(grep materialise gpurun_out/r04_mat_one.log)" | tee -a gpurun_out/r04_mat_bench.txt; grep '^JSON ' gpurun_out/r04_mat_one.log | sed 's/^JSON //' >> gpurun_out/r04_mat_bench.jsonl; done; done
python3 -c "
import json; rows=[json.loads(l) for l in open('gpurun_out/r04_mat_bench.jsonl')]
json.dump({'what': 'bvg_decode_range_dev of the whole graph into int64 in HBM (profiles/mat_bench.py: 8 tiles of a 2^21-node base; outdegree pass + prefix sum included; best of calls 3..6)', 'runs': rows}, open('gpurun_out/r04_mat_bench.json', 'w'), indent=1)"
