"""Photon order sweep: tools/tile_sweep.py <photons> <workload> cols...   (each setting in its own process;
MI3D_TILE_COLS=0 is the unsorted order).  Prints photons/s by wall time around run+sync (binning passes included) and by
the transport kernel's own HIP events."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nph = sys.argv[1]; work = sys.argv[2]; cols = sys.argv[3:]
code = r'''
import os, sys, time
sys.path.insert(0, %r)
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
sol = Mi3dSolver(0); sc = make_scene(%r); sol.load_scene(sc); sol.set_counting(False)
nph = int(float(%r))
sol.reset(); sol.run(nph//10, seed=1); sol.sync(); sol.reset()
out = []
for r in range(3):
    sol.reset(); sol.sync(); t0 = time.perf_counter(); sol.run(nph, seed=1234+r); sol.sync(); t1 = time.perf_counter()
    ms, nl = sol.timing(); out.append((nph/(t1-t0), nph/(ms*1e-3)))
print(' '.join('%%.4g/%%.4g' %% v for v in out), ' mean radiance %%.6f' %% float(sol.radiance(nph)[0].mean()))
''' % (root, work, nph)
for c in cols:
    env = dict(os.environ, MI3D_TILE_COLS=c)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print('tile cols %-4s wall/kernel photons/s: %s %s' % (c, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else ''), flush=True)
