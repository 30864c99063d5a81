#!/bin/bash
# skip-index granularity: edges/s and index bytes per scan for library builds with different (kSkipMin, kSkipEvery)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-libbvgraph_hip.so}; do for sh in eu web w0; do
  BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/$lib python bench.py --shape $sh --steps 3 --warmup 2 --target-gib 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); r=d['roofline']; print('$lib $sh: %.1f Gedges/s kernel %.1f ms index %.2f GB (graph %.2f GB)'%(d['value']/1e9, r['kernel_ms'], r['index_bytes_per_launch']/1e9, r['algorithmic_bytes_per_launch']/1e9))"
done; done
