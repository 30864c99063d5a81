#!/bin/bash
# r04: derived occupancy / latency counters of one bench workload, one pass per metric (never combined with tracing domains).
# usage: bash profiles/r04/occ_pmc.sh <tag> [bench args...]
cd "$(dirname "$0")/../.."; R=$PWD
tag=${1:-occ}; shift
args="${@:---shape eu15 --target-gib 2}"
O=$R/gpurun_out/r04_occ_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "MeanOccupancyPerCU" "MeanOccupancyPerActiveCU" "OccupancyPercent" "LdsLatency" "MemUnitStalled" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py $args --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; }
done
cd $R
python3 - $O <<'PY' | tee gpurun_out/r04_occ_${tag}.txt
import csv, glob, sys, collections
O = sys.argv[1]
for p in sorted(glob.glob(O + "/p*/")):
    for f in glob.glob(p + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = (r["Kernel_Name"][:60], r["Counter_Name"]); agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
        for (k, c), (v, n) in sorted(agg.items()):
            if "scan_kernel" in k or "giant" in k: print("%-62s %-26s sum %.6g  mean per dispatch %.6g  (%d dispatches)" % (k, c, v, v / n, n))
PY
find $O -name "*.csv" -size +1M -delete
