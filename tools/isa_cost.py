"""Static issue-cost estimate of a kernel's blocks from a -DMI3D_MARKS listing and the measured cost table of tools/microbench/mix_rates
(profiles/r05/mix_rates_ops*.log: cycles a wave64 instruction holds a SIMD, by wall time at six waves per SIMD):
   python tools/isa_cost.py /tmp/api.s [mangled-name-substring]
fast 2.2: v_fma/fmac/mul/add/sub_f32, v_xor/and/or_b32, v_add/sub_u32 with VGPR / inline / literal operands;  3.1: v_mov v,v, v_bitop3 v,v,v,
v_cndmask_e32 after a compare (2.1) ...;  slow 4.1: any SGPR source operand, compares, v_cndmask_e64, v_min/max, shifts, cvt, floor, the 3-operand
integer forms, v_mul_lo/hi, v_mad_u64_u32, packed math, v_readlane;  8.1: v_rcp/rsq/sqrt/exp/log/sin/cos."""
import re, sys, collections
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else 'k_transport_leanILb0ELb0ELi0ELi0ELi256E'
FAST = ('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_xor_b32', 'v_and_b32', 'v_or_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mac_f32')
TRANS = ('v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32')
def cost(op, args):
    base = op.replace('_e32', '').replace('_e64', '')
    srcs = args.split(',')[1:] if ',' in args else []
    has_s = any(re.search(r'(^|[\s\[|-])s\d+|s\[\d+:\d+\]|\bvcc\b|\bexec\b', a) for a in srcs) and not base.startswith('v_cndmask')
    if base in TRANS: return 8.1, 'trans'
    if base in FAST: return (4.1, 'fast op with an SGPR source') if has_s else (2.2, 'fast')
    if base == 'v_cndmask_b32': return (2.1, 'cndmask (vcc)') if op.endswith('_e32') else (4.1, 'cndmask_e64')
    if base.startswith('v_mov_b32'): return (4.1, 'mov from SGPR / constant') if (has_s or not re.search(r'\bv\d+', ','.join(srcs))) else (3.1, 'mov v,v')
    if base.startswith('v_bitop3'): return (4.1, 'slow') if has_s else (3.1, 'bitop3')
    if base.startswith('v_cmp'): return 4.1, 'compare'
    return 4.1, 'slow'
lines = open(path).read().split('\n')
inside = False; block = 'pre'
C = collections.OrderedDict()
for ln in lines:
    t = ln.strip()
    if not inside and t.startswith('_Z') and want in t.split(':')[0] and ':' in t:
        inside = True; continue
    if not inside: continue
    if t.startswith('.Lfunc_end'): break
    m = re.match(r'; MARK (\w+)', t)
    if m: block = m.group(1); continue
    t = t.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]
    if not op.startswith('v_'): continue
    c, cls = cost(op, t[len(op):])
    d = C.setdefault(block, collections.Counter())
    d['n'] += 1; d['cycles'] += c; d['n:' + cls] += 1; d['c:' + cls] += c
for b, d in C.items():
    print('%-7s %4d vector instructions, %6.0f cycles (%.2f each): ' % (b, d['n'], d['cycles'], d['cycles']/max(d['n'], 1)) +
          ', '.join('%s %d (%.0f)' % (k[2:], d[k], d['c:' + k[2:]]) for k in sorted(d) if k.startswith('n:')))
