"""
The CPU oracle against the deterministic plane-parallel answer K16 (tests/k16_adding_doubling.py) on the matrix of
tests/test_k16.py -- the long form of tests/test_k16.py::test_oracle_against_k16 (the CPU suite has minutes; this takes an hour).

    python tools/k16_oracle_matrix.py [--photons 2000000] [--quick] > profiles/r03/k16_oracle_matrix.log
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from er3t_amd.scene import TARGET_FLUX, TARGET_RADIANCE      # noqa: E402
from oracle import oracle                                      # noqa: E402
from tests import test_k16 as T                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--photons', type=float, default=2.0e6)
    ap.add_argument('--quick', action='store_true', help='a diagonal of the matrix instead of all of it')
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    nb = 16
    nper = int(a.photons)//nb
    cases = [c + (0.0,) for c in itertools.product(T.GS, T.OMEGAS, T.TAUS, T.MU0S, T.ALBEDOS)]
    if a.quick:
        cases = cases[::7]
    cases += [(0.85, 1.0, 8.0, 0.5, 0.3, 0.3), (0.85, 1.0, 2.0, 1.0, 0.0, 0.3)]
    allrel, allz = [], []
    print('# oracle (oracle/mi3d_oracle.c, %d photons per case in %d batches, %d threads) against K16; columns: relative difference in %% (z)' % (nb*nper, nb, a.threads))
    print('# views: vza %s' % T.VZA)
    print('#        vaa %s' % T.VAA)
    for ic, (g, omega, tau, mu0, albedo, tray) in enumerate(cases):
        t0 = time.time()
        want = T.k16_answer(g, omega, tau, mu0, albedo, tray)
        sc = T.slab(g, omega, tau, mu0, albedo, tray, grid=(ic % 2 == 0), target=TARGET_RADIANCE | TARGET_FLUX)
        rad, up, dn = [], [], []
        for b in range(nb):
            r = oracle.run(sc, nper, seed=16, offset=b*nper, nthreads=a.threads)
            rad.append(r['rad'].mean(axis=(1, 2))); up.append(r['flux'][2, -1].mean()/mu0); dn.append(r['flux'][1, 0].mean()/mu0)
        got = np.concatenate([np.stack(rad).mean(0), [np.mean(up), np.mean(dn)]])
        se = np.concatenate([np.stack(rad).std(0, ddof=1), [np.std(up, ddof=1), np.std(dn, ddof=1)]])/np.sqrt(nb)
        ref = np.concatenate([want['radiance'], [want['albedo'], want['transmittance']]])
        rel = (got-ref)/ref
        z = (got-ref)/np.maximum(se, 1e-300)
        keep = (np.abs(ref) > 1.0e-5) & (se > 0.0)       # (the light that gets through an absorbing slab of optical thickness 32: no relative difference)
        allrel.append(rel[keep]); allz.append(z[keep])
        print('g %.2f w %.1f tau %4.1f mu0 %.1f A %.1f ray %.1f %-4s | ' % (g, omega, tau, mu0, albedo, tray, 'grid' if ic % 2 == 0 else '1d') +
              ' '.join('%+.2f(%+.1f)' % (100*r_, z_) for r_, z_ in zip(rel, z)) + ' | %.0f s' % (time.time()-t0), flush=True)
    rel = np.concatenate(allrel); z = np.concatenate(allz)
    print('# %d comparisons: mean relative difference %+.4f %%, rms %.3f %%, max |.| %.2f %%; mean z %+.3f, std z %.2f, beyond 3 se: %.1f %%'
          % (rel.size, 100*rel.mean(), 100*np.sqrt(np.mean(rel**2)), 100*np.abs(rel).max(), z.mean(), z.std(), 100*np.mean(np.abs(z) > 3)))


if __name__ == '__main__':
    main()
