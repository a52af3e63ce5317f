"""Where does the spread of the flux z-scores against K16 come from?  (std z 1.17 / 1.19 over 248 / 224 albedos and transmittances at
1.6e8 photons per case, where 32 batches allow 1.03.)  The lean flux loop on the K16 matrix at K16_SCALE x the test's photons, 64 batches, two
seeds; per value: z with the batch standard error, the deterministic solver's OWN error estimated from 48 against 64 streams (delta), z with
delta added in quadrature, and the Poisson character of the value (photons that make it up per batch).
    K16_SCALE=10 python tools/k16_flux_zspread.py > profiles/r04/k16_flux_zspread.log"""
import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.scene import TARGET_FLUX
from tests import test_k16 as T
from tests import k16_adding_doubling as k16

scale = int(os.environ.get('K16_SCALE', '10'))
nb, nper = 64, 125000*scale
sol = Mi3dSolver(0)


def answer(g, omega, tau, mu0, albedo, tau_ray, nstream):
    old = T.NSTREAM
    T.NSTREAM = nstream
    try:
        return T.k16_answer.__wrapped__.__wrapped__(g, omega, tau, mu0, albedo, tau_ray)
    finally:
        T.NSTREAM = old


for tau_ray in (0.0, 0.3):
    for seed_add in (0, 1000):
        rows, exact = [], []
        for ic, ((g, omega), (tau, mu0, albedo)) in enumerate(itertools.product(itertools.product(T.GS, T.OMEGAS), itertools.product(T.TAUS, T.MU0S, T.ALBEDOS))):
            w48 = answer(g, omega, tau, mu0, albedo, tau_ray, 48)
            w64 = answer(g, omega, tau, mu0, albedo, tau_ray, 64) if seed_add == 0 else None
            sol.bind(None, None, None); sol.load_scene(T.slab(g, omega, tau, mu0, albedo, tau_ray=tau_ray, target=TARGET_FLUX, views=False)); sol.set_counting(False)
            up, dn, dr = [], [], []
            for b in range(nb):
                sol.reset(); sol.run(nper, seed=21+37*ic+seed_add, offset=b*nper)
                f = sol.flux(nper).astype(np.float64)
                up.append(f[2, -1].mean()/mu0); dn.append(f[1, 0].mean()/mu0); dr.append(f[0, 0].mean()/mu0)
            for kind, vals, key in (('albedo', up, 'albedo'), ('transmittance', dn, 'transmittance'), ('direct', dr, 'transmittance_direct')):
                want = float(w48[key])
                if want <= 1e-5:
                    continue
                a = np.asarray(vals); got, se = a.mean(), a.std(ddof=1)/np.sqrt(nb)
                delta = abs(float(w64[key])-want) if w64 is not None else float('nan')
                if se <= 0.0:      # (the same number in every batch: the direct beam where the HIP path adds it analytically -- not a Monte-Carlo value)
                    exact.append((got-want)/want)
                    continue
                rows.append((kind, g, omega, tau, mu0, albedo, got, want, se, delta))
        R = np.array([(r[6], r[7], r[8], r[9]) for r in rows])
        z = (R[:, 0]-R[:, 1])/R[:, 2]
        print('# Rayleigh optical thickness %g, seeds + %d: %d values, %d batches of %d photons; mean z %+.3f, std z %.3f (Student-t, %d d.o.f.: %.3f), beyond 3 se %.2f %%'
              % (tau_ray, seed_add, z.size, nb, nper, z.mean(), z.std(), nb-1, np.sqrt((nb-1.0)/(nb-3.0)), 100*np.mean(np.abs(z) > 3)))
        if exact:
            print('#   %d values are the same in every batch (the analytic direct beam): relative difference to K16 at most %.1e' % (len(exact), np.max(np.abs(exact))))
        # values of one case share its histories (a conservative case: albedo + transmittance = 1, their z mirror each other): the spread per CASE
        cases = {}
        for r, zz in zip(rows, z):
            cases.setdefault(r[1:6], []).append(zz)
        zcase = np.array([np.sqrt(np.mean(np.square(v))) for v in cases.values()])
        print('#   %d cases; rms z per case: median %.2f, cases above 2: %d; std of std z expected from that many independent cases: +-%.2f' % (len(cases), np.median(zcase), int(np.sum(zcase > 2.0)), 1.0/np.sqrt(2.0*len(cases))))
        if seed_add == 0:
            zc = (R[:, 0]-R[:, 1])/np.sqrt(R[:, 2]**2 + R[:, 3]**2)
            print('#   with the solver\'s own error (|48 - 64 streams|) added in quadrature: std z %.3f, beyond 3: %.2f %%; median delta / se %.4f, max %.3f'
                  % (zc.std(), 100*np.mean(np.abs(zc) > 3), np.median(R[:, 3]/R[:, 2]), np.max(R[:, 3]/R[:, 2])))
        for kind in ('albedo', 'transmittance', 'direct'):
            for tau in T.TAUS:
                m = np.array([r[0] == kind and r[3] == tau for r in rows])
                if m.sum() > 2:
                    print('#   %-13s tau %4g: n %2d  std z %.2f  mean z %+.2f  median relative se %.1e  median delta/se %s'
                          % (kind, tau, m.sum(), z[m].std(), z[m].mean(), np.median(R[m, 2]/R[m, 1]), ('%.4f' % np.median(R[m, 3]/R[m, 2])) if seed_add == 0 else '-'))
        for r, zz in zip(rows, z):
            if abs(zz) > 2.5:
                print('%-13s g %.2f omega %.1f tau %4g mu0 %.1f A %.1f: got %.6f want %.6f se %.1e z %+.2f delta %.1e' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], zz, r[9]))
        sys.stdout.flush()
