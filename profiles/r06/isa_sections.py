#!/usr/bin/env python3
"""Static instruction counts of scan_kernel per BVGMARK section and per loop (hipcc -S -DBVG_MARKS output).
usage: isa_sections.py file.s [mangled-name-substring]   (default: the tier-0 instantiation <true,false,4,true,false>)
Weights: full-rate VALU ops (add/sub/and/or/xor/lshrrev/ashrrev/not/mov, e32) = 1.0, every other VALU op = 1.75 (profiles/r04_valu_rates2.txt)."""
import re, sys
src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else 'scan_kernelILb1ELb0ELi4ELb1ELb0E'
beg = next(i for i, l in enumerate(src) if l.startswith('_ZN') and want in l and l.rstrip().endswith(('E:',)) or (l.startswith('_ZN') and want in l and ':' in l and '@' in l))
end = next(i for i in range(beg, len(src)) if 's_endpgm' in src[i])
FULL = {'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_lshrrev_b32', 'v_ashrrev_i32', 'v_not_b32', 'v_mov_b32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32'}
def kind(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'):
        if op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_barrier', 's_endpgm', 's_setprio', 's_sleep')): return 'ctl'
        if op.startswith(('s_load', 's_buffer_load')): return 'smem'
        return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'
def weight(op):
    base = op.replace('_e32', '').replace('_e64', '').replace('_dpp', '#').replace('_sdwa', '#')
    return 1.0 if base in FULL else 1.75
ins = []      # (line, op, label?)
labels = {}
for i in range(beg, end + 1):
    l = src[i].strip()
    m = re.match(r'^(\.LBB[0-9_]+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    if l.startswith('; BVGMARK'): ins.append((i, l, None)); continue
    if not l or l.startswith((';', '.', '//')): continue
    op = l.split()[0]
    ins.append((i, op, l))
def summarise(a, b):
    c = {'valu': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'smem': 0, 'ctl': 0, 'other': 0}; w = 0.0
    for _, op, l in ins[a:b]:
        if l is None: continue
        k = kind(op); c[k] += 1
        if k == 'valu': w += weight(op)
    return c, w
# consecutive marks (the compiler moves blocks, so begin / end do not always nest in layout order: counts between neighbouring marks)
marks = [idx for idx, x in enumerate(ins) if x[2] is None]
print('kernel', want, 'lines', beg, end, 'instructions', sum(1 for x in ins if x[2]))
prev = 0
for m in marks + [len(ins)]:
    c, w = summarise(prev, m)
    lab = ins[m][1] if m < len(ins) else 'END'
    print('  asm %5d-%5d  VALU %4d (weighted %6.0f)  SALU %4d  LDS %3d  VMEM %2d   -> %s' % (ins[prev][0], ins[m - 1][0] if m else 0, c['valu'], w, c['salu'], c['lds'], c['vmem'], lab))
    prev = m
# loops: backward branches
print('loops (backward branches):')
for idx, (ln, op, l) in enumerate(ins):
    if l and op.startswith(('s_cbranch', 's_branch')):
        t = l.split()[-1]
        if t in labels and labels[t] <= idx:
            c, w = summarise(labels[t], idx + 1)
            print('  loop asm %5d-%5d  VALU %4d (weighted %6.0f) SALU %4d LDS %3d VMEM %2d' % (ins[labels[t]][0], ln, c['valu'], w, c['salu'], c['lds'], c['vmem']))
