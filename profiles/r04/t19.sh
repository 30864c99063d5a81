#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash profiles/r04/final.sh
bash profiles/r04/pmc.sh cnrfull --shape cnr > gpurun_out/r04_pmc_cnrfull.log 2>&1; tail -3 gpurun_out/r04_pmc_cnrfull_summary.txt | cut -c1-200
python3 profiles/r04/pmc_by_launch.py gpurun_out/r04_pmc_cnrfull > gpurun_out/r04_cnr_pmc_by_launch.txt
