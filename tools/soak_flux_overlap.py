"""Flux at scale with the sort on its own stream: runs queued back to back (the next run's photon loops beside this run's last sort, nothing read in
between), small and large runs mixed so that one-stream and two-stream runs follow each other, against an atomic per crossing on the same photon ids:
tools/soak_flux_overlap.py [photons per large run] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sol = Mi3dSolver(0); sc = make_scene(os.environ.get('AB_WORKLOAD', 'les128_flux')); sol.load_scene(sc); sol.set_counting(False)
sizes = []
for r in range(rounds):
    sizes += [n, 3000000, n//3, 40000000, 5000000]
tot = sum(sizes)
out = {}
for name, lists in (('records (first pass: the lists are allocated)', 1), ('records', 1), ('atomics', 0)):
    sol.set_tuning(tally_lists=lists)
    sol.reset(); t0 = time.time(); off = 0
    for i, m in enumerate(sizes):
        sol.run(m, seed=2026, offset=off); off += m
        if lists and i % 7 == 6: sol.sync()          # (now and then somebody looks: the next small run takes one stream)
    sol.sync(); dt = time.time()-t0
    f = sol.flux(tot).astype(np.float64)
    out[name.split()[0]] = f
    print('%-48s %s  %d runs, %.3g photons, %.3g photons/s  sum of all cells %.9e' % (name, sol.kernel_name(), len(sizes), tot, tot/dt, f.sum()), flush=True)
a, b = out['records'], out['atomics']
lev = np.abs(a.sum(axis=(2, 3))-b.sum(axis=(2, 3)))/np.maximum(b.sum(axis=(2, 3)), 1e-30)
cell = np.abs(a-b).max()/b.max()
print('largest relative difference of a level sum %.2e, of a cell (relative to the largest cell) %.2e' % (lev.max(), cell))
assert lev.max() < 2e-6 and cell < 2e-6
print('ok')
