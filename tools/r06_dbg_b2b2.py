import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene
from er3t_amd.scene import TARGET_FLUX, TARGET_HEAT
sol = Mi3dSolver(0)
sc = les_scene(nx=16, ny=16, nz3=50, target='flux', aerosol=True)
sc.target = TARGET_FLUX | TARGET_HEAT
sc.abs1d = sc.abs1d*30.0 + 2.0e-5
n = 300000
sol.load_scene(sc); sol.set_counting(False)
sol.set_tuning(tlcap_log2=17)
sol.set_tuning(overlap_sort=int(sys.argv[1]), tl_split=4)
sol.reset()
for q in range(3): sol.run(n, seed=7, offset=q*n)
f = sol.flux(3*n).astype(np.float64)
print('mode', sys.argv[1], 'sum %.9e' % f.sum())
