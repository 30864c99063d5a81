#!/bin/bash
# HEAD check on the GPU box: GPU test suite, the three bench lines, section timers of the row kernel (prof build).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/head; mkdir -p $O
(time timeout -k 10 900 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
(time python bench.py) > $O/bench_eu.log 2>&1
grep -h metric $O/bench_eu.log | cut -c1-400
for sh in eu web; do
  echo "== $sh prof"; BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/libbvgraph_hip_prof.so BVG_DEBUG=1 BVG_DBG=64 timeout 300 python bench.py --shape $sh --steps 1 --warmup 0 --target-gib 1 --no-cpu-baseline --no-verify 2>&1 | grep -E "counters|wave-cycles|phase 1 split|metric" | cut -c1-600 | tee -a $O/prof_$sh.log
done
