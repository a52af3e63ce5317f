"""soak: many seeds x workloads x solvers; every launch must end, count its photons and give finite tallies"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
sol = Mi3dSolver(0)
t00 = time.time()
for work, nph, nseed in (('les480', 1e8, 12), ('les480_mv9', 2e7, 6), ('les128_flux', 1e8, 6), ('les128', 1e8, 6), ('les480_flux', 5e7, 3), ('les128_cam', 2e7, 4)):
    for solver in (0, 1, 2):
        if work == 'les128_cam' and solver != 0:      # (cameras need the 3-D solver)
            continue
        sc = make_scene(work); sc.solver = solver
        sol.bind(None, None, None); sol.load_scene(sc); sol.set_counting(False)
        for i in range(nseed if solver == 0 else 2):
            seed = 1000003*(i+1) + solver
            sol.reset(); t0 = time.time(); sol.run(int(nph), seed=seed); sol.sync(); dt = time.time()-t0
            c = sol.counters()
            ok = c['photons'] == int(nph)
            if sc.target & 2:
                r = sol.radiance(int(nph)); ok = ok and bool(np.all(np.isfinite(r))) and r.min() >= 0.0 and r.mean() > 0.0
            if sc.target & 1:
                f = sol.flux(int(nph)); ok = ok and bool(np.all(np.isfinite(f))) and f.min() >= 0.0
            print('%-12s solver %d seed %9d: %.3g photons/s %s' % (work, solver, seed, nph/dt, 'ok' if ok else 'BAD'), flush=True)
            assert ok
print('soak done in %.1f s' % (time.time()-t00))
