#!/bin/bash
# round 6: PMC passes at the final tree (each its own run, never combined with tracing domains): the default workload at full size and the tiled cnr-2000, recorded into
# profiles/traffic.json (stamped with the hash of csrc); then the default bench line again (it now quotes them), and the fresh fuzz seeds
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash profiles/r06/pmc.sh eu15 --shape eu15 > gpurun_out/r06_eu15_pmc_summary.txt 2>&1; tail -4 gpurun_out/r06_eu15_pmc_summary.txt
python3 profiles/r06/record.py eu15 eu15 128 1048576 r06_eu15_pmc_summary.txt | tail -12
bash profiles/r06/pmc.sh cnr --shape cnr > gpurun_out/r06_cnr_pmc_summary.txt 2>&1; tail -3 gpurun_out/r06_cnr_pmc_summary.txt
python3 profiles/r06/record.py cnr cnr 6004 325557 r06_cnr_pmc_summary.txt | tail -3
bash profiles/r06/last.sh
