#!/bin/bash
# section timers (prof build) + work counters of the row kernel, steady-state scans only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for sh in ${SHAPES:-eu}; do
  echo "== $sh prof"; BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvgraph_hip_prof.so BVG_DEBUG=1 BVG_DBG=${DBG:-64} timeout 300 python bench.py --shape $sh --steps 1 --warmup 0 --target-gib 1 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|wave-cycles|phase 1 split" | tail -3 | cut -c1-300
done
