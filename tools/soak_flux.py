"""Flux at scale, the record route against an atomic per crossing: tools/soak_flux.py [photons]
Same scene, same photon ids: every record written, sorted and summed must show up exactly once -- level sums of the three planes
equal to float32 output precision -- over some 10^10 records, many launches and list refills."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000000
sol = Mi3dSolver(0); sc = make_scene('les128_flux'); sol.load_scene(sc); sol.set_counting(False)
out = {}
for name, lists in (('records', 1), ('atomics', 0)):
    sol.set_tuning(tally_lists=lists)
    sol.reset(); t0 = time.time(); sol.run(n, seed=2026); sol.sync(); dt = time.time()-t0
    f = sol.flux(n).astype(np.float64)
    out[name] = f
    print('%-8s %s  %.3g photons/s  sum of all cells %.9e' % (name, sol.kernel_name(), n/dt, f.sum()), flush=True)
a, b = out['records'], out['atomics']
lev = np.abs(a.sum(axis=(2, 3))-b.sum(axis=(2, 3)))/np.maximum(b.sum(axis=(2, 3)), 1e-30)
cell = np.abs(a-b).max()/b.max()
print('largest relative difference of a level sum %.2e, of a cell (relative to the largest cell) %.2e' % (lev.max(), cell))
assert lev.max() < 2e-6 and cell < 2e-6
print('ok')
