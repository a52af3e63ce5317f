"""Row padding of the accumulation image against image size: square LES scenes of n x n columns (n x n pixels), MI3D_RAD_ROW_PAD swept."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
n = int(sys.argv[1]); nph = int(float(sys.argv[2]))
sol = Mi3dSolver(0)
sc = les_scene(nx=n, ny=n, nz3=50) if n <= 200 else les_scene(nx=n, ny=n, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
sol.load_scene(sc); sol.set_counting(False)
sol.reset(); sol.run(nph // 10, seed=1); sol.sync()
out = []
for r in range(2):
    sol.reset(); sol.run(nph, seed=10 + r); sol.sync(); ms, nl = sol.timing(); out.append(nph / (ms * 1e-3))
print(' '.join('%%.4g' %% v for v in out))
''' % root
nph = sys.argv[1] if len(sys.argv) > 1 else '2e8'
sizes = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [64, 192, 256, 320, 384]
for n in sizes:
    for pad in (0, 2, 8, 16, 18, 32, 34, 48, 64):
        env = dict(os.environ, MI3D_RAD_ROW_PAD=str(pad))
        r = subprocess.run([sys.executable, '-c', code, str(n), nph], env=env, capture_output=True, text=True)
        row = (n + pad) * 128
        print('n %d  pad %3d  image row %6d B = %7.4f x 4 KiB  image %5.1f MB   %s %s' % (n, pad, row, row / 4096.0, row * n / 1.0e6, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ''), flush=True)
