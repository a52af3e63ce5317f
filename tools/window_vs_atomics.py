"""The tally window against atomics alone at full size: the same 1e9 photon ids on the bench scene, image against image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
nph = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
for wl in ('les480', 'les128_aer'):
    sol = Mi3dSolver(0); sc = make_scene(wl); sol.load_scene(sc); sol.set_counting(False)
    img = {}
    for win in (1, 0):
        sol.set_tuning(tally_window=win)
        sol.reset(); sol.run(nph, seed=2026); sol.sync()
        ms, nl = sol.timing()
        img[win] = sol.radiance(nph).astype(np.float64)
        print('%s tally_window=%d  %.4g photons/s (kernels)  mean radiance %.9f' % (wl, win, nph/(ms*1e-3), img[win].mean()))
    d = img[1] - img[0]
    print('%s: mean difference %.3e relative; largest pixel difference %.3e of the pixel, %.3e of the brightest pixel; pixels %d'
          % (wl, d.mean()/img[0].mean(), np.abs(d/np.maximum(img[0], 1e-30)).max(), np.abs(d).max()/img[0].max(), d.size))
