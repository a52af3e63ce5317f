"""Flux jobs by grid size, the record route against an atomic per crossing: tools/flux_grid_sizes.py [photons]
(128 / 160 / 256 columns square at 50 cloud layers: 210 / 329 / 840 bins of 16 384 tally cells; 480 columns at 100 layers: 5000 bins,
beyond what the record route takes -- atomics either way)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50000000
sol = Mi3dSolver(0)
for nx, kw in ((100, dict(nz3=50)), (128, dict(nz3=50)), (160, dict(nz3=50)), (200, dict(nz3=50)), (256, dict(nz3=50)), (400, dict(nz3=50)), (480, dict(nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004))):
    sc = les_scene(nx=nx, ny=nx, target='flux', aerosol=(nx != 480), **kw)
    sol.load_scene(sc); sol.set_counting(True); sol.reset(); sol.run(1000000, seed=3); sol.sync()
    per = sol.counters()['flux_tally']/1.0e6
    sol.set_counting(False)
    line = '%3d x %3d x %3d voxels, %3d levels, %.1f crossings per photon:' % (nx, nx, sc.nz3, sc.nz+1, per)
    for name, lists in (('records', 1), ('atomics', 0)):
        sol.set_tuning(tally_lists=lists)
        sol.reset(); sol.run(n//5, seed=1); sol.sync()
        sol.reset(); sol.run(n, seed=2); sol.sync(); ms, _ = sol.timing()
        line += '  %s %.3g photons/s (%s)' % (name, n/(ms*1e-3), sol.kernel_name().replace('k_transport_flux', 'flux').replace(' + k_tl_scatter + k_tl_sum', ' + sort + sum'))
    sol.set_tuning(tally_lists=1)
    print(line, flush=True)
