"""per-run wall time of small flux jobs read one by one (er3t's pattern: reset, run, read): tools/small_runs.py [photons] [jobs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 6000000
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sol = Mi3dSolver(0); sol.load_scene(make_scene(os.environ.get('AB_WORKLOAD', 'les128_flux'))); sol.set_counting(False)
for q in range(3):
    sol.reset(); sol.run(n, seed=1, offset=q*n); sol.sync()
ts = []
for q in range(jobs):
    t0 = time.perf_counter()
    sol.reset(); sol.run(n, seed=7, offset=q*n); sol.sync()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts)*1e3
print('%d photons per job: %.3f ms per job (min %.3f, max %.3f) = %.4g photons/s; %s' % (n, ts.mean(), ts.min(), ts.max(), n/(ts.mean()*1e-3), sol.kernel_name()))
