import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene
from er3t_amd.scene import TARGET_FLUX, TARGET_HEAT
sol = Mi3dSolver(0)
sc = les_scene(nx=16, ny=16, nz3=50, target='flux', aerosol=True)
n = 300000
sol.load_scene(sc); sol.set_counting(False)
sol.set_tuning(tlcap_log2=17)
def seq(mode, nruns):
    sol.set_tuning(overlap_sort=mode, tl_split=4)
    sol.reset()
    for q in range(nruns): sol.run(n, seed=7, offset=q*n)
    raw = sol.flux(nruns*n).astype(np.float64)*nruns*n
    return raw
for nruns in (1, 2, 3):
    a = seq(0, nruns); b = seq(1, nruns)
    d = b-a
    print('runs %d: raw sum mode0 %.6e mode1 %.6e diff %.4e; by plane %s' % (nruns, a.sum(), b.sum(), d.sum(), ' '.join('%.3e' % d[p].sum() for p in range(3))))
    lev = d.sum(axis=(2, 3))
    print('   levels with |diff| > 0: plane0 %s plane1 %s plane2 %s' % tuple(str(np.nonzero(np.abs(lev[p]) > 1e-9*max(a.sum(), 1))[0][:12]) for p in range(3)))
