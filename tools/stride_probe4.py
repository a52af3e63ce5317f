"""Is it the radiance image?  Rad_nxr set apart from Atm_nx: (nx, nxr) in {480, 496}^2 (a pixel grid other than the columns costs the
position -> pixel arithmetic of every tally, the same for both off-diagonal cases)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
nph = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3 * 10**8
sol = Mi3dSolver(0)
for nx in (480, 496):
    for nxr in (480, 496, 500, 512):
        sc = les_scene(nx=nx, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
        sc.nxr = nxr
        sol.load_scene(sc); sol.set_counting(False)
        sol.reset(); sol.run(nph // 10, seed=1); sol.sync()
        out = []
        for r in range(2):
            sol.reset(); sol.run(nph, seed=10 + r); sol.sync(); ms, nl = sol.timing(); out.append(nph / (ms * 1e-3))
        print('nx %d  nxr %d  image row %6d B   %s' % (nx, nxr, nxr * 128, ' '.join('%.4g' % v for v in out)), flush=True)
