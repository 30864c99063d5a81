#!/bin/bash
# r02: repeat the giant-kernel tests (each run bounded) to catch anything timing-dependent; BVG_DEBUG names the last tier launched
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for i in $(seq 1 ${N:-6}); do
  BVG_DEBUG=1 timeout -k 5 120 python -m pytest tests/test_gpu_giant.py -m gpu -x -q -s > gpurun_out/giant_loop_$i.txt 2>&1
  rc=$?; echo "run $i rc=$rc $(tail -1 gpurun_out/giant_loop_$i.txt | cut -c1-120)"
  if [ $rc -ne 0 ]; then tail -30 gpurun_out/giant_loop_$i.txt | cut -c1-200; break; fi
done
