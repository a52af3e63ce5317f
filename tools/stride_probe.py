"""Does the row stride of the voxel records matter (L2 / HBM channel conflicts)?  The same synthetic cloud generator on grids of
nx = 472 ... 512 columns per row (row stride nx * nz3 * 16 bytes), ny = 480: photons/s of the plain radiance loop (two runs), then voxel steps and collisions per photon."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
nph = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3 * 10**8
sol = Mi3dSolver(0)
for nx in (472, 476, 480, 484, 488, 496, 504, 512):
    sc = les_scene(nx=nx, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
    sol.load_scene(sc); sol.set_counting(False)
    sol.reset(); sol.run(nph // 10, seed=1); sol.sync()
    out = []
    for r in range(2):
        sol.reset(); sol.run(nph, seed=10 + r); sol.sync(); ms, nl = sol.timing(); out.append(nph / (ms * 1e-3))
    sol.set_counting(True); sol.reset(); sol.run(2000000, seed=3); sol.sync(); c = sol.counters(); sol.set_counting(False)
    out.append(c['steps3d'] / c['photons']); out.append(c['scatter'] / c['photons'])
    print('nx %4d  row stride %8d B (%% 4096 = %4d)  %s' % (nx, nx * 1600, (nx * 1600) % 4096, ' '.join('%.4g' % v for v in out)), flush=True)
