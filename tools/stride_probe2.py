"""Voxel-record strides: MI3D_VPAD_COL / MI3D_VPAD_ROW (records of padding per column / per row) on a grid of nx columns per row."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
nx = int(sys.argv[1]); nph = int(float(sys.argv[2]))
sol = Mi3dSolver(0)
sc = les_scene(nx=nx, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
sol.load_scene(sc); sol.set_counting(False)
sol.reset(); sol.run(nph // 10, seed=1); sol.sync()
out = []
for r in range(2):
    sol.reset(); sol.run(nph, seed=10 + r); sol.sync(); ms, nl = sol.timing(); out.append(nph / (ms * 1e-3))
print(' '.join('%%.4g' %% v for v in out))
''' % root
nph = sys.argv[1] if len(sys.argv) > 1 else '3e8'
pads = ((0, 0), (1, 0), (2, 0), (4, 0), (8, 0), (12, 0), (28, 0), (0, 8), (0, 16), (0, 64), (0, 136), (4, 8), (28, 16))
if len(sys.argv) > 2 and sys.argv[2] == 'big':   # (row pads of the size of an L2 way and beyond)
    pads = tuple((0, v) for v in (0, 512, 1024, 2048, 3072, 4096, 6144, 8192, 10240, 12288, 14784, 16384))
for nx in ((496,) if len(sys.argv) > 2 else (480, 496)):
    for pc, pr in pads:
        env = dict(os.environ, MI3D_VPAD_COL=str(pc), MI3D_VPAD_ROW=str(pr))
        r = subprocess.run([sys.executable, '-c', code, str(nx), nph], env=env, capture_output=True, text=True)
        col_b = (100 + pc) * 16; row_b = nx * col_b + pr * 16
        print('nx %d  pad col %2d row %5d  column stride %5d B  row stride %7d B (%% 256 KiB = %6d)   %s %s' % (nx, pc, pr, col_b, row_b, row_b % 262144, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ''), flush=True)
