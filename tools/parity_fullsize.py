"""GPU vs oracle on the full config-2 grid (128x128x50), several views, many photons: domain-mean agreement in sigma"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene
from oracle import oracle
sc = les_scene(vza=(0.0, 45.6, 70.5), vaa=(0.0, 0.0, 180.0), aerosol=True, lsrt=True)
import sys as _s
SEED = int(_s.argv[1]) if len(_s.argv) > 1 else 42
nb, nper = 24, 250000
sol = Mi3dSolver(0); sol.load_scene(sc); sol.set_counting(False)
g = []
for b in range(nb):
    sol.reset(); sol.run(nper, seed=SEED, offset=b*nper); sol.sync(); g.append(sol.radiance(nper).astype(np.float64))
g = np.stack(g)
o = np.stack([oracle.run(sc, nper, seed=SEED, offset=b*nper, nthreads=16)['rad'] for b in range(nb)])
for iv in range(sc.nview):
    gm, om = g[:, iv].mean(axis=(1, 2)), o[:, iv].mean(axis=(1, 2))
    d = gm-om                                  # paired batches (same photon ids): the difference is what matters
    print('view %d: GPU %.6f  oracle %.6f  paired diff %+.2e +- %.2e (%.2f sigma)  | batch sigma of one estimate %.2e' %
          (iv, gm.mean(), om.mean(), d.mean(), d.std(ddof=1)/np.sqrt(nb), d.mean()/(d.std(ddof=1)/np.sqrt(nb)), om.std(ddof=1)/np.sqrt(nb)))
    z = (g[:, iv].mean(0)-o[:, iv].mean(0))/np.sqrt(g[:, iv].var(0, ddof=1)/nb+o[:, iv].var(0, ddof=1)/nb+1e-30)
    print('        per-pixel z: mean %+.3f std %.3f  frac|z|>2 %.3f  frac|z|>3 %.4f' % (z.mean(), z.std(), np.mean(np.abs(z) > 2), np.mean(np.abs(z) > 3)))
