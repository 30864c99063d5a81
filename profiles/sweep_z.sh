#!/bin/bash
# knobs of the position-task emission on the eu shape (single-wavefront kernel): launch bounds, pass cost, pool size
export NOTEST=1 SHAPES=${SHAPES:-eu} BVG_WG=0
for lib in libbvgraph_hip.so libbvgraph_hip_tw3.so libbvgraph_hip_tw2.so; do echo "-- $lib"; LIBS=$lib bash profiles/ab.sh; done
for pc in 3 6 16 24; do echo "-- passcost $pc"; BVG_PASSCOST=$pc bash profiles/ab.sh; done
for pool in 2560 3584 5120 6144; do echo "-- pool $pool"; BVG_POOL=$pool bash profiles/ab.sh; done
