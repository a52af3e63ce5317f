"""the failing part of test_record_sort_beside_the_next_photon_loop_changes_no_result, by itself, with numbers"""
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
sol.load_scene(sc); sol.set_counting(True)
sol.set_tuning(tlcap_log2=int(os.environ.get('TLCAP', '17')))
def three(mode, split):
    sol.set_tuning(overlap_sort=mode, tl_split=split)
    sol.reset()
    for q in range(3): sol.run(n, seed=7, offset=q*n)
    return sol.flux(3*n).astype(np.float64), sol.heating(3*n).astype(np.float64), sol.counters()
ref = None
for rnd in range(3):
    for mode, split in ((0, 4), (2, 4), (1, 4), (1, 4), (2, 1)):
        f, hh, c = three(mode, split)
        if ref is None: ref = (f, hh, c)
        d = np.abs(f-ref[0]); dh = np.abs(hh-ref[1])
        print('round %d overlap_sort %d split %d: flux_tally %d (ref %d) photons %d  max |d flux| %.3e (sum %.9e vs %.9e)  max |d heat| %.3e  cells off %d'
              % (rnd, mode, split, c['flux_tally'], ref[2]['flux_tally'], c['photons'], d.max(), f.sum(), ref[0].sum(), dh.max(), int((d > 1e-6*np.abs(ref[0]) + 1e-9).sum())), flush=True)
