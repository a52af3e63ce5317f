"""Row padding of the accumulation image (MI3D_RAD_ROW_PAD, pixels of 128 bytes): rate against the image's row stride."""
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
for nx, pads in ((480, (0, 1, 2, 4, 8, 16, 24, 32, 40, 48, 64, 96, 128)), (496, (0, 8, 16, 24, 32, 48, 80)), (500, (0, 12, 28, 44)), (512, (0, 8, 16, 32, 64))):
    for pad in pads:
        env = dict(os.environ, MI3D_RAD_ROW_PAD=str(pad))
        r = subprocess.run([sys.executable, '-c', code, str(nx), nph], env=env, capture_output=True, text=True)
        row = (nx + pad) * 128
        print('nxr %d  pad %3d  image row %6d B = %7.4f x 4 KiB   %s %s' % (nx, pad, row, row / 4096.0, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ''), flush=True)
