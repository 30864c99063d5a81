#!/bin/bash
# r02: the batch that produced every profiles/r02_* file (run on the GPU box: gpurun -- 'bash profiles/r02/final.sh')
cd "$(dirname "$0")/../.."; R=$PWD; O=$R/gpurun_out/r02_final; rm -rf $O; mkdir -p $O
say() { echo "[final] $*"; }
say "bench lines"
python bench.py > $O/bench_eu15.json 2> $O/bench_eu15.err; say "eu15 done"
for sh in eu web w0; do python bench.py --shape $sh --no-cpu-baseline > $O/bench_$sh.json 2> $O/bench_$sh.err; say "$sh done"; done
python bench.py --shape eu --tiles 1100 --allow-wide --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_eu_2g3.json 2> $O/bench_eu_2g3.err; say "2.3 G nodes done"
python bench.py --shape eu --tiles 2100 --allow-wide --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_eu_u64.json 2> $O/bench_eu_u64.err; say "u64 (4.4 G nodes) done"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.log 2>&1; grep '^{' $O/bench_torchrun1.log > $O/bench_eu15_torchrun1.json; say "torchrun 1 rank (RCCL) done"
bash profiles/r02/strong_rehearsal.sh > $O/strong_rehearsal.txt 2>&1; say "strong rehearsal done"
python profiles/r02/speedtest.py > $O/speedtest.json 2> $O/speedtest.err; say "speedtest done"
say "kernel trace"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt_bench.log 2>&1 )
f=$(ls $O/kt/*/*_kernel_trace.csv | head -1)
python3 profiles/r02/ktrace_summary.py $f > $O/ktrace.txt; python3 profiles/union.py $f 3 >> $O/ktrace.txt; grep '^{' $O/kt_bench.log > $O/kt_bench.json; say "trace done"
say "PMC passes (eu15, full size)"
bash profiles/r02/pmc.sh final --shape eu15 > $O/pmc_log.txt 2>&1; cp gpurun_out/r02_pmc_final_summary.txt $O/pmc_summary.txt 2>/dev/null; cp gpurun_out/r02_pmc_final/summary.json $O/pmc_summary.json 2>/dev/null; say "pmc done"
SHAPES="eu eu15" bash profiles/r02/prof.sh > $O/prof.txt 2>&1; say "section timers done"
bash profiles/r02/ldspad.sh > $O/ldspad.txt 2>&1; say "ldspad done"
ls -la $O | head -40
