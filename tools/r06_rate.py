"""photons/s of `steps` runs back to back, three times: tools/r06_rate.py <workload> <photons> [steps]   (knobs through the MI3D_* environment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1]; n = int(float(sys.argv[2])); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc); sol.set_counting(False)
sol.reset(); sol.run(n, seed=1); sol.sync(); sol.reset(); sol.sync()
rates = []
for r in range(3):
    t0 = time.perf_counter()
    for q in range(steps): sol.run(n, seed=1234+r, offset=q*n)
    sol.sync(); dt = time.perf_counter()-t0
    sol.reset(); sol.sync()
    rates.append(steps*n/dt)
print('%s %s: %s photons/s' % (work, sol.kernel_name(), ' '.join('%.4g' % v for v in rates)), flush=True)
