cd /root/repo; export BVG_TEST_KNOBS=1; L=/root/repo/webgraph-big_amd/lib
for s in web cnr eu15; do for v in default sk12_8 sk8_8 sk8_4 sk4_4 sk12_4; do
  if [ $v = default ]; then unset BVG_HIP_LIB; else export BVG_HIP_LIB=$L/libbvg_exp_$v.so; fi
  timeout -k 10 300 python bench.py --shape $s --target-gib 4 --steps 5 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$s $v: %.1f G edges/s, no-index %.1f G, index read per scan %.2f GB (stream %.2f GB), resident %.2f GB, build %.2f s, break-even %.2f scans' % (d['value']/1e9, d.get('value_no_index',0)/1e9, d.get('index_bytes_per_launch',0)/1e9, d['config']['graph_bytes']/1e9, d['hbm_resident_bytes']/1e9, d['index_build_s'], d.get('index_break_even_scans') or 0))"
done; done
