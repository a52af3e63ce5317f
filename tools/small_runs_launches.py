"""launches per small flux job read one by one: tools/small_runs_launches.py [photons] [jobs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 6000000
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sol = Mi3dSolver(0); sol.load_scene(make_scene(os.environ.get('AB_WORKLOAD', 'les128_flux'))); sol.set_counting(False)
out = []
for q in range(jobs):
    t0 = time.perf_counter()
    sol.reset(); sol.run(n, seed=7, offset=q*n); sol.sync()
    dt = time.perf_counter() - t0
    ms, nl = sol.timing()
    out.append('%d:%.2fms' % (nl, dt*1e3))
print('%d photons per job, launches:wall of every job: %s' % (n, ' '.join(out)))
