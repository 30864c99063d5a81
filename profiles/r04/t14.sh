#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash profiles/r04/pmc.sh eu15full --shape eu15 > gpurun_out/r04_pmc_eu15full.log 2>&1; tail -14 gpurun_out/r04_pmc_eu15full_summary.txt | cut -c1-250
python3 profiles/r04/pmc_by_launch.py gpurun_out/r04_pmc_eu15full > gpurun_out/r04_eu15_pmc_by_launch.txt
bash profiles/r04/pmc.sh cnrfull --shape cnr > gpurun_out/r04_pmc_cnrfull.log 2>&1; tail -5 gpurun_out/r04_pmc_cnrfull_summary.txt | cut -c1-250
python3 profiles/r04/pmc_by_launch.py gpurun_out/r04_pmc_cnrfull > gpurun_out/r04_cnr_pmc_by_launch.txt
bash profiles/r04/rehearsal.sh | cut -c1-400
