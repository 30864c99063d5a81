#!/bin/bash
# PC sampling of one scan (beta feature): where the wavefronts of the row kernel spend their issue slots
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pcs; rm -rf $O; mkdir -p $O
export BVG_WG=0
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit ${UNIT:-time} --pc-sampling-method ${METHOD:-host_trap} --pc-sampling-interval ${INTERVAL:-1000} --kernel-trace --output-format csv -d $O/a -- python bench.py --shape ${SHAPE:-eu} --steps 2 --warmup 1 --target-gib 0.5 --no-cpu-baseline --no-verify > $O/a.log 2>&1
echo "rc=$?"; tail -5 $O/a.log | cut -c1-300; find $O -type f | head; du -sh $O
