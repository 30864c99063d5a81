#!/bin/bash
# the reference's SpeedTest protocol on the host path (profiles/r02/speedtest.py): sequential iteration + random access, successors in host memory
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout -k 10 800 python profiles/r02/speedtest.py --shape eu --gib 1 > gpurun_out/r03_speedtest.json 2> gpurun_out/r03_speedtest.err; echo "rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r03_speedtest.json'))
print('sequential host: %.2f G edges/s (%.1f GB/s), unpipelined pageable %.2f G; random access %.2f M nodes/s, %.0f M arcs/s' % (d['sequential_host']['edges_per_s']/1e9, d['sequential_host']['host_GB_per_s'], d['sequential_host_unpipelined_pageable']['edges_per_s']/1e9, d['random_access']['nodes_per_s']/1e6, d['random_access']['arcs_per_s']/1e6))"
