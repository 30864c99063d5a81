#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
out=gpurun_out/${OUT:-r06_fuzz2.txt}; : > $out
for seed in ${SEEDS:-603 604 605}; do
  echo "== BVG_FUZZ=${N:-1500} BVG_FUZZ_SEED=$seed" | tee -a $out
  BVG_FUZZ=${N:-1500} BVG_FUZZ_SEED=$seed timeout -k 10 ${LIMIT:-540} python -m pytest tests/test_gpu_fuzz.py -x -q -s -m gpu 2>&1 | tee -a $out | grep -E "fuzz: 1500|passed|failed|Error|raised" || exit 1
done
