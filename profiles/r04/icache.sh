#!/bin/bash
# r04: instruction-cache counters of the default workload's kernels (one --pmc pass per set, kernel trace only).
cd "$(dirname "$0")/../.."; R=$PWD; O=$R/gpurun_out/r04_icache; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3-avail list 2>/dev/null | tr ' ' '\n' | grep -i -E "icache|ifetch|SQC_|INST_LEVEL|InstrFetch" | sort -u | tr '\n' ' ' > $O/avail.txt; cat $O/avail.txt; echo
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --shape eu15 --target-gib ${GIB:-4} --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; }
done
cd $R
python3 - $O <<'PY' | tee gpurun_out/r04_icache.txt
import csv, glob, sys, collections
O = sys.argv[1]
for p in sorted(glob.glob(O + "/p*/")):
    for f in glob.glob(p + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(f, newline="")):
            if int(r["Grid_Size"]) < 1000000: continue
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])] += float(r["Counter_Value"])
        for k, v in sorted(agg.items()): print("%-42s %-30s %.6g" % (k[0], k[1], v))
PY
find $O -name "*.csv" -size +1M -delete
