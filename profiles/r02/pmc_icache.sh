#!/bin/bash
# r02: instruction-cache behaviour of the row kernel (the kernel is ~5000 instructions; 8-10 wavefronts per CU run at different PCs)
cd "$(dirname "$0")/../.."; R=$PWD; O=$R/gpurun_out/r02_pmc_icache; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/p1 -- python3 $R/bench.py --shape eu --target-gib 2 --steps 2 --warmup 0 --no-cpu-baseline > $O/p1.log 2>&1
cd $R
python3 - $O <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p1/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
rows = list(csv.DictReader(open(f)))
red = sorted({int(r["Dispatch_Id"]) for r in rows if "reduce_acc" in r["Kernel_Name"]})
cut = red[-2] if len(red) > 1 else -1
for r in rows:
    if int(r["Dispatch_Id"]) <= cut: continue
    k = "rows_kernel" if "rows_kernel" in r["Kernel_Name"] else ("rows_wg" if "rows_wg" in r["Kernel_Name"] else ("decode" if "decode_kernel" in r["Kernel_Name"] else None))
    if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    print(k, {n: "%.4g" % v for n, v in c.items()})
    if c.get("SQC_ICACHE_REQ"):
        print("   icache hit rate %.4f, misses per 1000 VALU instr %.2f, ifetch per instr %.3f" % (c["SQC_ICACHE_HITS"] / c["SQC_ICACHE_REQ"], 1000 * c["SQC_ICACHE_MISSES"] / max(c["SQ_INSTS_VALU"], 1), c["SQ_IFETCH"] / max(c["SQ_INSTS_VALU"], 1)))
P
