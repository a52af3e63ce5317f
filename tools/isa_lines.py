"""Per-source-line instruction counts and issue cost of one kernel, from a listing with line tables:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -gline-tables-only -S --cuda-device-only er3t_amd/csrc/mi3d_api.hip -o /tmp/lines.s
   python tools/isa_lines.py /tmp/lines.s <mangled-name-substring> [file:first-last=label ...]
Every instruction is attributed to the source line its .loc names -- wherever the scheduler has moved it (the `; MARK` comments of
tools/isa_blocks.py are not instructions: code moves across them) --, lines are summed into the labelled ranges; the cost of an
instruction is the measured one of its kind (tools/isa_cost.py, profiles/r05/mix_rates_ops*.log)."""
import re, sys, collections
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
path, want = sys.argv[1], sys.argv[2]
ranges = []
for spec in sys.argv[3:]:
    m = re.match(r'([^:]+):(\d+)-(\d+)=(.+)', spec)
    ranges.append((m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)))
FAST = ('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_xor_b32', 'v_and_b32', 'v_or_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32')
TRANS = ('v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32')
def cost(op, args):
    base = op.replace('_e32', '').replace('_e64', '')
    srcs = args.split(',')[1:] if ',' in args else []
    has_s = any(re.search(r'(^|[\s\[|-])s\d+|s\[\d+:\d+\]|\bvcc\b|\bexec\b', a) for a in srcs) and not base.startswith('v_cndmask')
    if base in TRANS: return 8.1
    if base in FAST: return 4.1 if has_s else 2.2
    if base == 'v_cndmask_b32': return 2.1 if op.endswith('_e32') else 4.1
    if base.startswith('v_mov_b32'): return 4.1 if (has_s or not re.search(r'\bv\d+', ','.join(srcs))) else 3.1
    if base.startswith('v_bitop3'): return 4.1 if has_s else 3.1
    return 4.1
files = {}
inside = False; cur = (None, 0)
per = collections.defaultdict(lambda: collections.Counter())
for ln in open(path):
    t = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', t)
    if m: files[int(m.group(1))] = m.group(2).split('/')[-1]; continue
    if not inside and t.startswith('_Z') and want in t.split(':')[0] and ':' in t:
        inside = True; continue
    if not inside: continue
    if t.startswith('.Lfunc_end'): break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m:
        f = files.get(int(m.group(1)), '?')
        # (a line of the HIP headers -- fminf, __ballot, the atomics' wrappers, inlined where they are used -- counts for the kernel-source line
        #  that was current when it appeared: the call site, as near as the line table says)
        if f.startswith('mi3d_') and int(m.group(2)) > 0: cur = (f, int(m.group(2)))
        continue
    t = t.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]
    c = per[cur]
    if op.startswith('v_'): c['v'] += 1; c['cyc'] += cost(op, t[len(op):])
    elif op.startswith('s_'): c['s'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')): c['mem'] += 1
if ranges:
    tot = collections.defaultdict(collections.Counter)
    for (f, l), c in per.items():
        lab = next((lab for rf, a, b, lab in ranges if rf == f and a <= l <= b), 'other')
        tot[lab].update(c)
    print('%-44s %7s %9s %7s %5s %5s' % ('range', 'vector', 'cycles', 'scalar', 'lds', 'mem'))
    for lab in [r[3] for r in ranges] + ['other']:
        if lab in tot:
            c = tot[lab]; print('%-44s %7d %9.0f %7d %5d %5d' % (lab, c['v'], c['cyc'], c['s'], c['lds'], c['mem']))
else:
    for (f, l), c in sorted(per.items()):
        print('%-22s %5d  v %4d  cyc %6.0f  s %4d  lds %3d  mem %3d' % (f, l, c['v'], c['cyc'], c['s'], c['lds'], c['mem']))
