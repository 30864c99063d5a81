#!/bin/bash
# r06: the randomised end-to-end parity test on fresh seeds.  usage: [N=1500] [SEEDS="41 42"] bash profiles/r06/fuzz.sh
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
out=gpurun_out/r06_fuzz.txt; : > $out
for seed in ${SEEDS:-501 502}; do
  echo "== BVG_FUZZ=${N:-1500} BVG_FUZZ_SEED=$seed" | tee -a $out
  BVG_FUZZ=${N:-1500} BVG_FUZZ_SEED=$seed timeout -k 10 ${LIMIT:-540} python -m pytest tests/test_gpu_fuzz.py -x -q -s -m gpu 2>&1 | tee -a $out | grep -E "fuzz:|passed|failed|Error|assert" || exit 1
done
