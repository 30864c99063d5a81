#!/bin/bash
# r04: rate and resident bytes per skip-index granularity (BVG_SKIP_GRAN="min,every") and shape, 4 GiB streams.
cd "$(dirname "$0")/../.."; export BVG_TEST_KNOBS=1
for s in ${SHAPES:-web cnr uk w0 eu eu15}; do for v in ${GRANS:-24,16 16,16 8,8}; do export BVG_SKIP_GRAN=$v
  timeout -k 10 300 python bench.py --shape $s --target-gib 4 --steps 5 --warmup 3 --no-cpu-baseline --no-verify --no-index-leg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$s $v: %.1f G edges/s, resident %.2f GB (stream %.2f GB), build %.2f s' % (d['value']/1e9, d['hbm_resident_bytes']/1e9, d['config']['graph_bytes']/1e9, d['index_build_s']))"
done; done
