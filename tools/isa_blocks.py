"""instruction counts per block of k_transport from a -DMI3D_MARKS ISA listing:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DMI3D_MARKS -S --cuda-device-only er3t_amd/csrc/mi3d_api.hip -o /tmp/api.s
   python tools/isa_blocks.py /tmp/api.s [mangled-name-substring]"""
import re, sys, collections
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else 'k_transportILb0ELb0ELb0ELb0E'
lines = open(path).read().split('\n')
inside = False; block = 'pre'; counts = collections.OrderedDict(); kinds = collections.defaultdict(collections.Counter)
for ln in lines:
    t = ln.strip()
    if not inside and t.startswith('_Z') and want in t.split(':')[0] and ':' in t:
        inside = True; continue
    if not inside:
        continue
    if t.startswith('.Lfunc_end') or t.startswith('s_endpgm'):
        if t.startswith('.Lfunc_end'):
            break
    m = re.match(r'; MARK (\w+)', t)
    if m:
        block = m.group(1); continue
    t = t.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    counts[block] = counts.get(block, 0) + 1
    kinds[block][op.split('_')[0]] += 1
for b, n in counts.items():
    print('%-5s %5d  %s' % (b, n, dict(kinds[b].most_common(6))))
print('total', sum(counts.values()))
