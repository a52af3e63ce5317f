"""instruction counts per block of a transport kernel from a -DMI3D_MARKS ISA listing:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMI3D_MARKS -S --cuda-device-only er3t_amd/csrc/mi3d_api.hip -o /tmp/api.s
   python tools/isa_blocks.py /tmp/api.s [mangled-name-substring]
per block: vector instructions (of which register-to-register v_mov_b32 / v_mov_b64 copies, and moves of constants), scalar
instructions (of which exec-mask work: s_and_saveexec / s_or_saveexec / s_andn2_saveexec / s_xor / s_or ... exec, s_cbranch), LDS and
global memory instructions"""
import re, sys, collections
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else 'k_transport_leanILb0ELb0ELi0ELb0E'
lines = open(path).read().split('\n')
inside = False; block = 'pre'
C = collections.OrderedDict()
for ln in lines:
    t = ln.strip()
    if not inside and t.startswith('_Z') and want in t.split(':')[0] and ':' in t:
        inside = True; continue
    if not inside:
        continue
    if t.startswith('.Lfunc_end'):
        break
    m = re.match(r'; MARK (\w+)', t)
    if m:
        block = m.group(1); continue
    t = t.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    c = C.setdefault(block, collections.Counter())
    if op.startswith('v_'):
        c['vector'] += 1
        if op.startswith('v_mov_b'):
            args = [a.strip() for a in t[len(op):].split(',')]
            if len(args) == 2 and re.match(r'^v(\d+|\[\d+:\d+\])$', args[1]):
                c['v_mov reg->reg'] += 1
            else:
                c['v_mov constant / scalar'] += 1
    elif op.startswith('s_'):
        c['scalar'] += 1
        if 'exec' in t or op.startswith('s_cbranch') or op.startswith('s_branch'):
            c['scalar: exec masks and branches'] += 1
    elif op.startswith('ds_'):
        c['lds'] += 1
    elif op.startswith('global_') or op.startswith('flat_') or op.startswith('scratch_'):
        c['memory'] += 1
keys = ['vector', 'v_mov reg->reg', 'v_mov constant / scalar', 'scalar', 'scalar: exec masks and branches', 'lds', 'memory']
print('%-8s' % 'block' + ''.join('%12s' % k.split(':')[-1].strip()[:11] for k in keys))
tot = collections.Counter()
for b, c in C.items():
    print('%-8s' % b + ''.join('%12d' % c[k] for k in keys)); tot.update(c)
print('%-8s' % 'total' + ''.join('%12d' % tot[k] for k in keys))
print('columns:', ' | '.join(keys))
