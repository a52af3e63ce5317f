"""A/B of kernel builds: tools/ab.py <photons> lib1.so lib2.so ...  (each build in its own process, 3 rounds interleaved)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nph = sys.argv[1]; libs = sys.argv[2:]
work = os.environ.get('AB_WORKLOAD', 'les480')
code = r'''
import os, sys, time
sys.path.insert(0, %r)
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
sol = Mi3dSolver(0)
if os.environ.get('AB_OWN_STREAM'): sol.set_tuning(own_stream=1)
sc = make_scene(%r); sol.load_scene(sc); sol.set_counting(False)
nph = int(float(%r))
sol.reset(); sol.run(nph//10, seed=1); sol.sync(); sol.reset()
out = []; wall = []
ksteps = int(os.environ.get('AB_STEPS', '1'))   # AB_STEPS=K: K runs back to back before the wait (what a job of several batches does)
for r in range(3):
    sol.reset(); sol.sync(); ms0, nl0 = sol.timing(); t0 = time.perf_counter()
    for q in range(ksteps): sol.run(nph, seed=1234+r, offset=q*nph)
    sol.sync(); t1 = time.perf_counter()
    ms, nl = sol.timing(); out.append(ksteps*nph/((ms-ms0)*1e-3)); wall.append(ksteps*nph/(t1-t0))
print(' '.join('%%.4g' %% v for v in out), '| wall', ' '.join('%%.4g' %% v for v in wall))
''' % (root, work, nph)
for lib in libs:
    env = dict(os.environ, MI3D_LIBRARY=os.path.abspath(lib))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print('%-28s %s %s' % (os.path.basename(lib), r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ''), flush=True)
