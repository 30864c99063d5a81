#!/bin/bash
export NOTEST=1 SHAPES=${SHAPES:-eu}
for cfg in "2 3072" "2 3584" "2 4096" "2 4608" "4 4608" "4 6144" "4 8192"; do set -- $cfg; echo "-- wg $1 pool $2"; BVG_WG=$1 BVG_POOL=$2 bash profiles/ab.sh; done
