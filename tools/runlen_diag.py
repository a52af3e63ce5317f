import os, sys
sys.path.insert(0, '/root/repo')
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
for wl in ('les128_flux', 'les480_flux'):
    sol = Mi3dSolver(0); sc = make_scene(wl); sol.load_scene(sc); sol.set_counting(True)
    n = 2000000
    sol.reset(); sol.run(n, seed=5); sol.sync()
    c = sol.counters()
    print(wl, sol.kernel_name(), {k: round(v/n, 3) for k, v in c.items() if k in ('flux_tally', 'le_rays', 'steps', 'steps3d', 'scatter', 'le_column')})
