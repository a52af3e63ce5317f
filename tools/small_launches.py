"""rate of SMALL launches (the jobs of a correlated-k loop): tools/small_launches.py [workload] [photons per launch] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1] if len(sys.argv) > 1 else 'les128_aer'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 6250000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 48
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc); sol.set_counting(False)
sol.reset(); sol.run(n, seed=1); sol.sync()
for rep in range(2):
    sol.reset(); t0 = time.perf_counter()
    for j in range(m): sol.run(n, seed=100+j)
    sol.sync(); dt = time.perf_counter()-t0; ms, nl = sol.timing()
    print('%s: %d launches of %d photons: wall %.1f ms, kernels %.1f ms -> %.4g photons/s (wall), %.4g (kernels)' % (work, m, n, dt*1e3, ms, m*n/dt, m*n/(ms*1e-3)))
