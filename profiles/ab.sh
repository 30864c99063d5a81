#!/bin/bash
# quick A/B on the GPU box: GPU parity suite, then the three shapes at 2 GiB (3 timed scans after the index build),
# for every library in LIBS (paths relative to webgraph-big_amd/lib; default: the product build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
if [ -z "$NOTEST" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log; fi
for lib in ${LIBS:-libbvgraph_hip.so}; do for sh in ${SHAPES:-eu web w0}; do
  BVG_HIP_LIB=$PWD/webgraph-big_amd/lib/$lib timeout 300 python bench.py --shape $sh --steps ${STEPS:-3} --warmup 2 --target-gib ${GIB:-2} --no-cpu-baseline > $O/$sh.log 2>&1
  python - $O/$sh.log "$lib $sh" <<'P'
import sys,json
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        d=json.loads(l); print('%s: %.1f Gedges/s kernel %.1f ms slow %d' % (sys.argv[2], d['value']/1e9, d['roofline']['kernel_ms'], d['slow_blocks']))
P
done; done
