"""The tally-window path against the oracle with more photons than the tests spend: paired batches on the config-2 and config-4 grids
(nadir view, the lean loop with its window), domain means and per-pixel / per-block z-scores."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene, parity_stats
from oracle import oracle
for wl, nb, nper in (('les128', 48, 500000), ('les480', 32, 500000)):
    sc = make_scene(wl)
    sol = Mi3dSolver(0); sol.load_scene(sc); sol.set_counting(False)
    g, o = [], []
    for b in range(nb):
        sol.reset(); sol.run(nper, seed=4242, offset=b*nper); sol.sync(); g.append(sol.radiance(nper).astype(np.float64))
        o.append(oracle.run(sc, nper, seed=4242, offset=b*nper, nthreads=16)['rad'])
    g = np.stack(g); o = np.stack(o)
    st = parity_stats(g, o)
    v = st[0]
    print('%s (%s): %d x %d photons, the same ids on both sides' % (wl, sol.kernel_name(), nb, nper))
    print('   domain mean GPU %.7f oracle %.7f: %+.2f sigma of two independent estimates, paired %+.3e relative (%+.2f of its own standard error)'
          % (v['mean_gpu'], v['mean_oracle'], v['domain_mean_diff_sigma'], v['paired_rel_diff'], v['paired_diff_in_paired_se']))
    print('   16 x 16 blocks: z mean %+.3f std %.3f, largest |z| %.2f, share beyond 2: %.4f' % (v['block_z_mean'], v['block_z_std'], v['block_abs_z_max'], v['frac_abs_z_gt_2']))
