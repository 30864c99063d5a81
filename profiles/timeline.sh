cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python bench.py --shape ${SHAPE:-eu} --steps 2 --warmup 1 --no-cpu-baseline --no-verify > /tmp/tl.log 2>&1
python3 profiles/timeline.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1)
