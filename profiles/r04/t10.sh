#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
BVG_DEBUG=1 timeout -k 10 500 python profiles/r04/mem_diag.py 128 2>&1 | grep -E "in use|block plan built|residual skip index|plan:" > gpurun_out/r04_mem_diag.txt; cat gpurun_out/r04_mem_diag.txt
bash profiles/r04/pmc.sh cnrfull --shape cnr > gpurun_out/r04_pmc_cnrfull.log 2>&1; tail -12 gpurun_out/r04_pmc_cnrfull_summary.txt
python3 profiles/r04/pmc_by_launch.py gpurun_out/r04_pmc_cnrfull | cut -c1-260
