#!/bin/bash
# r02: per-dispatch timeline of one steady-state scan (kernel trace), default bench workload unless ARGS is set
cd "$(dirname "$0")/../.."; R=$PWD; mkdir -p gpurun_out/r02_kt; rm -rf gpurun_out/r02_kt/*
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${ARGS:-} > $R/gpurun_out/r02_kt/bench.log 2>&1
cd $R
f=$(ls gpurun_out/r02_kt/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f | tee gpurun_out/r02_ktrace.txt
grep '^{' gpurun_out/r02_kt/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: %.1f G edges/s, kernel_ms %.1f'%(d['value']/1e9,d['roofline']['kernel_ms']))" | tee -a gpurun_out/r02_ktrace.txt
python3 profiles/union.py $f 3 | tee -a gpurun_out/r02_ktrace.txt
# keep the raw trace small enough to travel back
ls -la gpurun_out/r02_kt/*/
