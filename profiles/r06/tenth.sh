#!/bin/bash
# round 6, tenth GPU call: GPU suite on the split host sources; DESIGN 8 (2b) measured: validation marks WITHOUT skip entries on the lean kernel (BVG_SKIP_GRAN=4096,64: only
# lists of >= 4 096 residuals get entries) against the indexed scan and the index-less checking kernels (bench.py's no_index leg)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_tenth_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r06_tenth_tests.log
[ $rc -ne 0 ] && exit $rc
for sh in eu15 cnr; do
TAG=marks_$sh SHAPE=$sh GIB=4 CONFIGS="X=1;BVG_SKIP_GRAN=4096,64" bash profiles/r06/ab.sh
BVG_TEST_KNOBS=1 timeout -k 10 400 python bench.py --shape $sh --target-gib 4 --steps 5 --warmup 3 --no-cpu-baseline --no-verify --no-wide-leg --no-real-leg 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh: indexed %.1f G, no index (checking kernels) %.1f G edges/s, index entries %d, resident %.2f GB' % (d['value']/1e9, d['value_no_index']/1e9, d['index']['skip_entries_rank0'], d['hbm_resident_bytes']/1e9))" | tee -a gpurun_out/r06_ab_marks_$sh.txt
done
