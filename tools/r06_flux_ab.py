"""Run records (round 6) against a record per level and against an atomic per crossing: tools/r06_flux_ab.py <workload> <photons> [steps]
Same photon ids on every route: flux fields equal to float32 output precision, flux_tally counters equal; photons/s of `steps` runs back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1]; n = int(float(sys.argv[2])); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc)
out, cnt = {}, {}
nchk = min(n, 20000000)
for name, knobs in (('runs', dict(tally_lists=1, tally_runs=1)), ('levels', dict(tally_lists=1, tally_runs=0)), ('atomics', dict(tally_lists=0, tally_runs=0))):
    sol.set_tuning(**knobs)
    sol.set_counting(True); sol.reset(); sol.run(2000000, seed=2026); sol.sync(); cnt[name] = sol.counters()
    sol.set_counting(False)
    sol.reset(); sol.run(nchk, seed=2026); sol.sync()
    out[name] = sol.flux(nchk).astype(np.float64)
    if name == 'atomics':
        print('%-8s %s' % (name, sol.kernel_name()), flush=True)
        continue
    sol.reset(); sol.run(n, seed=1); sol.sync(); sol.reset(); sol.sync()
    rates = []
    for r in range(3):
        t0 = time.perf_counter()
        for q in range(steps): sol.run(n, seed=1234+r, offset=q*n)
        sol.sync(); dt = time.perf_counter()-t0
        sol.reset(); sol.sync()
        rates.append(steps*n/dt)
    print('%-8s %s  back to back: %s photons/s   flux_tally/photon %.4f' % (name, sol.kernel_name(), ' '.join('%.4g' % v for v in rates), cnt[name]['flux_tally']/cnt[name]['photons']), flush=True)
for k in ('photons', 'scatter', 'surface', 'escaped', 'killed', 'flux_tally'):
    assert cnt['runs'][k] == cnt['levels'][k] == cnt['atomics'][k], (k, cnt['runs'][k], cnt['levels'][k], cnt['atomics'][k])
for a in ('runs', 'levels'):
    d = out[a]-out['atomics']
    lev = np.abs(d.sum(axis=(2, 3)))/np.maximum(out['atomics'].sum(axis=(2, 3)), 1e-30)
    cell = np.abs(d).max()/out['atomics'].max()
    print('%-8s against atomics: largest relative difference of a level sum %.2e, of a cell (relative to the largest cell) %.2e' % (a, lev.max(), cell), flush=True)
    assert lev.max() < 2e-6 and cell < 2e-6, a
print('ok')
