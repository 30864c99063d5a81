# tier-0 kernel duration (longest rows_kernel dispatch) under the phase-skipping debug bits, from rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for sh in ${SHAPES:-eu}; do for m in ${MODES:-0 1 3 7}; do
  rm -rf /tmp/sp; BVG_DBG=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python bench.py --shape $sh --steps 2 --warmup 0 --target-gib ${GIB:-2} --no-cpu-baseline --no-verify > /tmp/sp.log 2>&1
  f=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
  echo "$sh dbg=$m: $(python3 - "$f" <<'P'
import csv,sys
try:
    rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'rows_kernel' in r['Kernel_Name']]
    g=max(int(r['Grid_Size_X']) for r in rows)
    big=[r for r in rows if int(r['Grid_Size_X'])>=0.9*g]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in big]
    print('tier-0 dispatches %d, last %.1f ms (all: %s)' % (len(d), d[-1], ' '.join('%.0f'%x for x in d)))
except Exception as e:
    print('ERR', e)
P
)"
done; done
