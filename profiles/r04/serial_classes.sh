#!/bin/bash
# every tier alone on the chip (BVG_SERIAL=1), with two LDS geometries of the lean classes: does their occupancy matter when nothing overlaps them?
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out; export BVG_TEST_KNOBS=1 BVG_SERIAL=1
i=0
for cfg in "BVG_NOP=1" "BVG_CLASS_STAGE=384,512,768,1024 BVG_CLASS_SCR=512,768,1536,3072" "BVG_CLASS_STAGE=256,384,512,768 BVG_CLASS_SCR=384,512,1024,2048"; do
  i=$((i+1)); rm -rf gpurun_out/r04_kt; mkdir -p gpurun_out/r04_kt
  cd /tmp && export TMPDIR=/tmp
  env $cfg rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_kt -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg > $R/gpurun_out/r04_kt/bench.log 2>&1
  cd $R
  f=$(ls gpurun_out/r04_kt/*/*_kernel_trace.csv | head -1)
  echo "== $cfg" >> gpurun_out/r04_serial_classes.txt
  python3 profiles/r02/ktrace_summary.py $f >> gpurun_out/r04_serial_classes.txt 2>&1
done
rm -rf gpurun_out/r04_kt
cat gpurun_out/r04_serial_classes.txt
