"""GPU first-light diagnostics: GPU vs oracle on small scenes (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
from er3t_amd.scene import Scene
from oracle import oracle

sol = Mi3dSolver(0)
a = sol.philox(1234, 5, 3, 1000)
b = np.stack([oracle.philox(1234, 5+i, 3) for i in range(1000)])
print('philox bit-exact:', np.array_equal(a, b))

def compare(scene, nph, tag, column_le=True):
    sol.load_scene(scene, column_le=column_le)
    sol.set_counting(True)
    sol.reset()
    t0 = time.time(); sol.run(nph, seed=7); sol.sync(); t1 = time.time()
    cnt = sol.counters()
    ref = oracle.run(scene, nph, seed=7, nthreads=16)
    print('== %s: GPU %.3fs (%.3g ph/s counting build)' % (tag, t1-t0, nph/(t1-t0)))
    for k in ('steps', 'steps3d', 'scatter', 'surface', 'le_rays', 'le_steps', 'le_column', 'flux_tally', 'roulette', 'killed', 'escaped', 'absorbed'):
        print('   %-10s gpu %12d  oracle %12d' % (k, cnt[k], ref['counters'][k]))
    if scene.target & 2:
        rad = sol.radiance(nph).astype(np.float64)
        for iv in range(scene.nview):
            r0 = ref['rad'][iv]
            print('   view %d mean gpu %.6f oracle %.6f reldiff %.2e ; pixel rms rel diff %.3e' %
                  (iv, rad[iv].mean(), r0.mean(), rad[iv].mean()/r0.mean()-1, np.sqrt(np.mean((rad[iv]-r0)**2))/r0.mean()))
    if scene.target & 1:
        fl = sol.flux(nph).astype(np.float64)
        for v, name in enumerate(('direct', 'down', 'up')):
            g = fl[v].mean(axis=(1, 2)); o = ref['flux'][v].mean(axis=(1, 2))
            print('   flux %-6s max |gpu-oracle| over levels %.3e (toa %.5f / %.5f, sfc %.5f / %.5f)' % (name, np.abs(g-o).max(), g[-1], o[-1], g[0], o[0]))

sc = les_scene(nx=16, ny=16, nz3=50)
compare(sc, 200000, 'small nadir radiance (column LE)')
compare(sc, 200000, 'small nadir radiance (marched LE)', column_le=False)
sc = les_scene(nx=16, ny=16, nz3=50, vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
compare(sc, 200000, 'small 3 views')
sc = les_scene(nx=16, ny=16, nz3=50, target='flux')
compare(sc, 200000, 'small flux')
sc = les_scene(nx=16, ny=16, nz3=50, lsrt=True, aerosol=True, vza=(30.0,), vaa=(100.0,))
compare(sc, 200000, 'small lsrt+aerosol slant')

# throughput on config 2 (timing build)
sc = les_scene()
sol.load_scene(sc)
sol.set_counting(False)
for nph in (1000000, 10000000):
    sol.reset(); t0 = time.time(); sol.run(nph, seed=1234); sol.sync(); t1 = time.time()
    ms, nl = sol.timing()
    print('config2 %d photons: wall %.3fs kernel %.1f ms -> %.3g photons/s' % (nph, t1-t0, ms, nph/(ms*1e-3)))
