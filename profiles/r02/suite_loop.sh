#!/bin/bash
# r02: the whole GPU suite several times over (anything timing-dependent shows as a failure or a time-out with stacks)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for i in $(seq 1 ${N:-3}); do
  timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/suite_loop_$i.txt 2>&1
  echo "suite run $i rc=$? $(tail -1 gpurun_out/suite_loop_$i.txt | cut -c1-100)"
done
