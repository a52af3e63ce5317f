"""speed and noise of the nine-view workload against the weight threshold of the local-estimate roulette (mi3d_set_le_weight_roulette)
   tools/weight_roulette_sweep.py [photons] [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
work = sys.argv[2] if len(sys.argv) > 2 else 'les480_mv9'
sol = Mi3dSolver(0); sc = make_scene(work)
print('# %s, %d photons per run, two seeds per threshold; noise = rms over pixels of the difference of the two runs / sqrt(2) / mean' % (work, n))
for cmin in (0.0, 0.01, 0.02, 0.04, 0.08, 0.16, 0.32):
    sc.le_cmin = cmin
    sol.load_scene(sc); sol.set_counting(False)
    sol.reset(); sol.run(n//4, seed=5); sol.sync()
    rads, ms = [], []
    for seed in (11, 12):
        sol.reset(); sol.run(n, seed=seed); sol.sync(); t, _ = sol.timing(); ms.append(t)
        rads.append(sol.radiance(n).astype(np.float64))
    d = (rads[0]-rads[1])/np.sqrt(2.0)
    noise = np.sqrt((d**2).mean(axis=(1, 2)))/np.mean(rads, axis=0).mean(axis=(1, 2))
    t = np.mean(ms)*1e-3
    print('cmin %.2f: %.4g photons/s; mean radiance per view %s' % (cmin, n/t, np.round(np.mean(rads, axis=0).mean(axis=(1, 2)), 5)))
    print('           per-pixel noise per view %s ; efficiency 1/(noise^2 t) of the marched views %.3g' % (np.round(noise, 4), 1.0/(np.mean(noise[1:]**2)*t)), flush=True)
