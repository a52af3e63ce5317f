"""What makes a grid of 496 columns per row 13 % slower than one of 480?  The accumulation image (rad_spread), the tiles of the photon order."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
nx = int(sys.argv[1]); nph = int(float(sys.argv[2])); spread = int(sys.argv[3])
sol = Mi3dSolver(0)
sc = les_scene(nx=nx, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
sol.load_scene(sc); sol.set_counting(False); sol.set_tuning(rad_spread=spread)
sol.reset(); sol.run(nph // 10, seed=1); sol.sync()
out = []
for r in range(2):
    sol.reset(); sol.run(nph, seed=10 + r); sol.sync(); ms, nl = sol.timing(); out.append(nph / (ms * 1e-3))
print(' '.join('%%.4g' %% v for v in out))
''' % root
nph = sys.argv[1] if len(sys.argv) > 1 else '3e8'
for nx in (480, 496, 512):
    for spread, tc in ((1, -1), (0, -1), (1, 62), (1, 31), (1, 124), (1, 0)):
        env = dict(os.environ, MI3D_TILE_COLS=str(tc))
        r = subprocess.run([sys.executable, '-c', code, str(nx), nph, str(spread)], env=env, capture_output=True, text=True)
        print('nx %d  rad_spread %d  tile_cols %3d   %s %s' % (nx, spread, tc, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ''), flush=True)
