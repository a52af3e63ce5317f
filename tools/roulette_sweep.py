"""speed and noise of the nine-view workload against the roulette threshold of local-estimate rays"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
sol = Mi3dSolver(0); sc = make_scene('les480_mv9')
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
for tau1 in (0.0, 4.0, 3.0, 2.0, 1.0):
    sc.le_tau1 = tau1
    sol.load_scene(sc); sol.set_counting(False)
    rads, ms = [], []
    for seed in (11, 12):
        sol.reset(); sol.run(n, seed=seed); sol.sync(); t, _ = sol.timing(); ms.append(t)
        rads.append(sol.radiance(n).astype(np.float64))
    d = (rads[0]-rads[1])/np.sqrt(2.0)
    noise = np.sqrt((d**2).mean(axis=(1, 2)))/np.mean(rads, axis=0).mean(axis=(1, 2))        # per view: relative per-pixel noise
    t = np.mean(ms)*1e-3
    print('tau1 %.1f: %.3g photons/s; mean radiance per view %s' % (tau1, n/t, np.round(np.mean(rads, axis=0).mean(axis=(1, 2)), 5)))
    print('          per-pixel noise per view %s ; efficiency 1/(noise^2 t) of the slant views %.3g' % (np.round(noise, 4), 1.0/(np.mean(noise[1:]**2)*t)), flush=True)
