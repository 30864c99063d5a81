#!/bin/bash
# r04, first GPU call: instruction-rate micro-benchmark, the whole GPU test suite on the refactored tree, baseline bench lines
# (eu15 default with the new JSON fields; --shape cnr), and every tier alone on the chip (BVG_SERIAL) for the span-vs-sum question.
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rates2 profiles/r04/valu_rates2.hip && /tmp/valu_rates2 > gpurun_out/r04_valu_rates2.txt 2>&1
echo "rates done"; tail -3 gpurun_out/r04_valu_rates2.txt
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_first.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_gputests_first.log
timeout -k 10 600 python bench.py --steps 5 --warmup 3 > gpurun_out/r04_base_eu15_bench.json 2> gpurun_out/r04_base_eu15_bench.err; echo "eu15 rc=$?"; cut -c1-300 gpurun_out/r04_base_eu15_bench.json
timeout -k 10 600 python bench.py --shape cnr --steps 5 --warmup 3 > gpurun_out/r04_base_cnr_bench.json 2> gpurun_out/r04_base_cnr_bench.err; echo "cnr rc=$?"; cut -c1-300 gpurun_out/r04_base_cnr_bench.json
export BVG_TEST_KNOBS=1
for cfg in "BVG_NOP=1" "BVG_SERIAL=1"; do
  env $cfg BVG_DEBUG=1 timeout -k 10 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg 2> gpurun_out/r04_serial.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg] %.1f G edges/s  %.2f ms/step' % (d['value']/1e9, d['ms_per_step']))" | tee -a gpurun_out/r04_serial.txt
  grep -E "tiers concurrent" gpurun_out/r04_serial.err | tail -1 >> gpurun_out/r04_serial.txt
done
rm -rf gpurun_out/r04_kt; mkdir -p gpurun_out/r04_kt
cd /tmp && export TMPDIR=/tmp
BVG_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_kt -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg > $R/gpurun_out/r04_kt/bench.log 2>&1
cd $R
f=$(ls gpurun_out/r04_kt/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f > gpurun_out/r04_serial_ktrace.txt 2>&1; tail -15 gpurun_out/r04_serial_ktrace.txt
rm -rf gpurun_out/r04_kt
