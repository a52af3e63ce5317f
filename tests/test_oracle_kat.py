"""
Pins the CPU oracle (oracle/mi3d_oracle.c).  The reference holds no golden vector for solver results
("parity unpinned"), so the oracle is anchored on
  * Random123's published Philox4x32-10 known answers,
  * analytic radiative-transfer results (K1 Beer's law, K2 Lambert surface, K4 single scattering,
    K5 energy conservation, K6 homogeneous 3-D == 1-D, two-stream band K3),
  * moments of the phase-function samplers and hand-computed Ross-Li kernel values.
CPU only; sized to run in well under a minute.
"""

import numpy as np
import pytest

from er3t_amd.scene import Scene, TARGET_FLUX, TARGET_RADIANCE, TARGET_HEAT, SOLVER_3D, SOLVER_P3D, SOLVER_IPA
from tests.util import slab_scene, batch_stats, block_scene, block_expectations


# ---------------------------------------------------------------------------------------------
def test_philox_random123_known_answers(oracle):
    # kat_vectors of the Random123 distribution, philox4x32 with 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff]*4, [0xffffffff]*2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert [int(x) for x in oracle.philox_raw(ctr, key)] == want


def test_philox_keying(oracle):
    # counter = (id lo, id hi, draw, 0), key = (seed lo, seed hi)
    seed = 0x0123456789abcdef; ident = 0xfedcba9876543210; draw = 77
    a = oracle.philox(seed, ident, draw)
    b = oracle.philox_raw([ident & 0xffffffff, ident >> 32, draw, 0], [seed & 0xffffffff, seed >> 32])
    assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------
def test_k1_beer_law_direct_beam(oracle, nthreads):
    sza = 40.0; mu0 = np.cos(np.deg2rad(sza)); tau_abs = 0.7
    sc = slab_scene(tau=0.0, abs_tau=tau_abs, albedo=0.0, sza=sza, nz=5, target=TARGET_FLUX)
    n = 400000
    r = oracle.run(sc, n, seed=3, nthreads=nthreads)
    z = sc.zgrd
    want = mu0*np.exp(-tau_abs*(z[-1]-z)/z[-1]/mu0)
    got = r['flux'][0, :, 0, 0]
    sigma = np.sqrt(want/mu0*(1.0-want/mu0)/n)*mu0 + 1e-12
    assert np.all(np.abs(got-want) < 4.5*sigma + 1e-9)
    assert np.allclose(r['flux'][1], r['flux'][0])          # nothing scatters: total down == direct
    assert np.all(r['flux'][2] == 0.0)                      # black surface: no upward flux


def test_k2_lambert_surface_exact(oracle, nthreads):
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); a = 0.3
    sc = slab_scene(tau=0.0, albedo=a, sza=sza, nx=4, ny=3, vza=(0.0, 50.0), vaa=(0.0, 77.0))
    n = 120000
    r = oracle.run(sc, n, seed=5, nthreads=nthreads)
    # every photon reflects once with weight a; the local estimate is deterministic per photon
    assert np.allclose(r['rad'].mean(axis=(1, 2)), np.float32(a)*mu0/np.pi, rtol=1e-12)
    assert np.isclose(r['flux'][2, -1].mean(), np.float32(a)*mu0, rtol=1e-12)
    assert np.isclose(r['flux'][1, 0].mean(), mu0, rtol=1e-12)
    assert r['counters']['surface'] == n and r['counters']['scatter'] == 0


@pytest.mark.parametrize('apf', [-2.0, -1.0, 0.0, 0.6, 0.85])
def test_k4_single_scattering_limit(oracle, nthreads, apf):
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); tau = 0.02; omega = 0.5
    vza = np.array([0.0, 40.0]); vaa = np.array([0.0, 135.0])
    sc = slab_scene(tau=tau, omega=omega, apf=apf, albedo=0.0, sza=sza, nz=2, vza=vza, vaa=vaa, target=TARGET_RADIANCE)
    nb, nper = 8, 100000
    mean, se = batch_stats(lambda n, s, off: oracle.run(sc, n, seed=s, offset=off, nthreads=nthreads)['rad'][:, 0, 0], nb, nper, 9)
    sdir = np.array([np.sin(np.deg2rad(sc.src_the))*np.cos(np.deg2rad(sc.src_phi)),
                     np.sin(np.deg2rad(sc.src_the))*np.sin(np.deg2rad(sc.src_phi)), np.cos(np.deg2rad(sc.src_the))])
    for iv in range(2):
        t = np.deg2rad(sc.view_the[iv]); p = np.deg2rad(sc.view_phi[iv])
        v = -np.array([np.sin(t)*np.cos(p), np.sin(t)*np.sin(p), np.cos(t)])
        mu = float(sdir @ v); muv = v[2]
        P = oracle.phase_eval(apf, mu)
        want = omega*P/(4.0*np.pi)*mu0/(mu0+muv)*(1.0-np.exp(-tau*(1.0/mu0+1.0/muv)))
        # higher orders of scattering add O(omega*tau) relative
        assert abs(mean[iv]-want) < 4.0*se[iv] + 0.03*want, (apf, iv, mean[iv], want, se[iv])


@pytest.mark.parametrize('apf', [-1.0, 0.6])
def test_k11_up_looking_sensor_single_scattering(oracle, nthreads, apf):
    """a pixel sensor looking UP from the ground (er3t: sensor_zenith_angle > 90, Rad_the < 90): diffusely transmitted
    single scattering of a thin slab, L = omega P/(4 pi) mu0/(mu0-muv) (exp(-tau/mu0) - exp(-tau/muv)); a second sensor
    half way up sees only the upper half of the slab; the black surface and everything below a sensor are invisible"""
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); tau = 0.02; omega = 0.5; ztop = 4000.0
    sc = slab_scene(tau=tau, omega=omega, apf=apf, albedo=0.0, sza=sza, nz=4, ztop=ztop, vza=(180.0, 140.0, 140.0), vaa=(0.0, 135.0, 135.0),
                    target=TARGET_RADIANCE)
    sc.view_zloc = [0.0, 0.0, 0.5*ztop]
    sc.le_tau1 = 0.0
    nb, nper = 8, 100000
    mean, se = batch_stats(lambda n, s, off: oracle.run(sc, n, seed=s, offset=off, nthreads=nthreads)['rad'][:, 0, 0], nb, nper, 9)
    sdir = np.array([np.sin(np.deg2rad(sc.src_the))*np.cos(np.deg2rad(sc.src_phi)),
                     np.sin(np.deg2rad(sc.src_the))*np.sin(np.deg2rad(sc.src_phi)), np.cos(np.deg2rad(sc.src_the))])
    for iv, frac in ((0, 1.0), (1, 1.0), (2, 0.5)):
        t = np.deg2rad(sc.view_the[iv]); p = np.deg2rad(sc.view_phi[iv])
        v = -np.array([np.sin(t)*np.cos(p), np.sin(t)*np.sin(p), np.cos(t)])        # travelling DOWN to the sensor
        assert v[2] < 0.0
        muv = -v[2]; P = oracle.phase_eval(apf, float(sdir @ v)); tv = tau*frac
        want = omega*P/(4.0*np.pi)*mu0/(mu0-muv)*(np.exp(-tv/mu0)-np.exp(-tv/muv)) if abs(mu0-muv) > 1e-9 else \
            omega*P/(4.0*np.pi)*tv/mu0*np.exp(-tv/mu0)
        assert abs(mean[iv]-want) < 4.0*se[iv] + 0.03*want, (apf, iv, mean[iv], want, se[iv])
    # the surface itself is never seen by an up-looking sensor: bright surface under a purely absorbing slab -> exactly nothing
    sc2 = slab_scene(tau=tau, omega=0.0, apf=apf, albedo=0.8, sza=sza, nz=4, ztop=ztop, vza=(180.0, 140.0), vaa=(0.0, 135.0), target=TARGET_RADIANCE)
    sc2.view_zloc = [0.0, 0.0]
    assert np.all(oracle.run(sc2, 50000, seed=3, nthreads=nthreads)['rad'] == 0.0)


def _chandrasekhar_h(omega, mu, n=96):
    """H function of isotropic scattering: 1/H(mu) = sqrt(1-omega) + (omega/2) int_0^1 mu' H(mu')/(mu+mu') dmu' (iterated)"""
    x, wq = np.polynomial.legendre.leggauss(n)
    x = 0.5*(x+1.0); wq = 0.5*wq
    h = np.ones(n)
    for _ in range(500):
        hn = 1.0/(np.sqrt(1.0-omega) + 0.5*omega*np.array([np.sum(wq*x*h/(xi+x)) for xi in x]))
        if np.max(np.abs(hn-h)) < 1e-13:
            h = hn
            break
        h = hn
    return np.array([1.0/(np.sqrt(1.0-omega) + 0.5*omega*np.sum(wq*x*h/(m+x))) for m in np.atleast_1d(mu)])


def test_k12_semi_infinite_isotropic_atmosphere_chandrasekhar(oracle, nthreads):
    """multiple scattering to all orders against an exact result: diffuse reflection by a semi-infinite isotropically
    scattering atmosphere, I(mu) = omega/(4 pi) mu0/(mu+mu0) H(mu) H(mu0) per unit flux normal to the beam
    (Chandrasekhar 1950, Radiative Transfer, par. 33); a slab of optical thickness 40 with omega = 0.9 stands in for it"""
    omega, sza = 0.9, 40.0
    mu0 = np.cos(np.deg2rad(sza))
    vza = np.array([0.0, 35.0, 65.0])
    sc = slab_scene(tau=40.0, omega=omega, apf=-2.0, albedo=0.0, sza=sza, nz=8, vza=vza, vaa=(0.0, 90.0, 200.0), target=TARGET_RADIANCE)
    sc.le_tau1 = 0.0
    nb, nper = 10, 40000
    mean, se = batch_stats(lambda n, s, off: oracle.run(sc, n, seed=s, offset=off, nthreads=nthreads)['rad'][:, 0, 0], nb, nper, 21)
    mu = np.cos(np.deg2rad(vza))
    want = omega/(4.0*np.pi)*mu0/(mu+mu0)*_chandrasekhar_h(omega, mu)*_chandrasekhar_h(omega, mu0)
    assert np.all(np.abs(mean-want) < 4.0*se + 2e-3*want), (mean, want, se)
    # the same atmosphere's plane albedo is 1 - H(mu0) sqrt(1 - omega): upward flux at the top
    scf = slab_scene(tau=40.0, omega=omega, apf=-2.0, albedo=0.0, sza=sza, nz=8, target=TARGET_FLUX)
    fb = np.stack([oracle.run(scf, 40000, seed=5, offset=b*40000, nthreads=nthreads)['flux'][2, -1, 0, 0] for b in range(10)])
    want_up = mu0*(1.0-_chandrasekhar_h(omega, mu0)[0]*np.sqrt(1.0-omega))
    assert abs(fb.mean()-want_up) < 4.0*fb.std(ddof=1)/np.sqrt(10) + 1e-3*want_up, (fb.mean(), want_up)
    # the H function itself: its zeroth moment obeys (omega/2) int_0^1 H dmu = 1 - sqrt(1 - omega) exactly
    x, wq = np.polynomial.legendre.leggauss(64)
    assert abs(0.5*omega*np.sum(0.5*wq*_chandrasekhar_h(omega, 0.5*(x+1.0))) - (1.0-np.sqrt(1.0-omega))) < 1e-8


def test_k5_energy_conservation_and_two_stream_band(oracle, nthreads):
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); a = 0.2; tau = 10.0
    sc = slab_scene(tau=tau, omega=1.0, apf=0.85, albedo=a, sza=sza, nz=6, target=TARGET_FLUX)
    nb, nper = 8, 25000
    mean, se = batch_stats(lambda n, s, off: oracle.run(sc, n, seed=s, offset=off, nthreads=nthreads)['flux'][:, :, 0, 0], nb, nper, 21)
    net = mean[1]-mean[2]
    net_se = np.sqrt(se[1]**2+se[2]**2)
    # conservative medium: the net flux is the same at every level
    assert np.all(np.abs(net-net[-1]) < 4.0*np.sqrt(net_se**2+net_se[-1]**2) + 1e-9)
    assert np.isclose(mean[1, -1], mu0, rtol=1e-12)
    assert abs(mean[2, 0]-a*mean[1, 0]) < 4.0*np.sqrt(se[2, 0]**2+(a*se[1, 0])**2)
    # K3: flux albedo of the cloud over a black surface.  The reference's own yardstick is the two-stream value
    # (er3t/util/util.py:1135-1151), a sanity band; the deterministic plane-parallel answer (K16, tests/k16_adding_doubling.py)
    # holds it to Monte-Carlo noise
    from tests import k16_adding_doubling as k16
    sc0 = slab_scene(tau=tau, omega=1.0, apf=0.85, albedo=0.0, sza=sza, nz=6, target=TARGET_FLUX)
    fb = np.array([oracle.run(sc0, 25000, seed=2, offset=b*25000, nthreads=nthreads)['flux'][2, -1, 0, 0]/mu0 for b in range(8)])
    x = 2.0*mu0/(1.0-0.85)
    assert abs(fb.mean() - tau/(tau+x)) < 0.06
    want = k16.solve([(tau, 1.0, k16.hg_moments(0.85, 95))], mu0, 0.0, nstream=48)['albedo']
    assert abs(fb.mean()-want) < max(3.0e-3*want, 4.0*fb.std(ddof=1)/np.sqrt(8)), (fb.mean(), want)


def test_k6_homogeneous_3d_equals_1d_and_ipa(oracle, nthreads):
    kw = dict(tau=4.0, omega=0.98, apf=0.7, albedo=0.25, sza=50.0, nz=4, vza=(0.0, 35.0), vaa=(0.0, 250.0), abs_tau=0.1)
    n = 60000
    r1 = oracle.run(slab_scene(**kw), n, seed=4, nthreads=nthreads)
    # same photons, same random numbers: cell faces consume no draws, so the histories are the same up to
    # rounding; compare domain means
    r3 = oracle.run(slab_scene(nx=5, ny=4, nz3=3, **kw), n, seed=4, nthreads=nthreads)
    ri = oracle.run(slab_scene(nx=5, ny=4, nz3=3, solver=SOLVER_IPA, **kw), n, seed=4, nthreads=nthreads)
    for r in (r3, ri):
        assert np.allclose(r['rad'].mean(axis=(1, 2)), r1['rad'][:, 0, 0], rtol=2e-3)
        assert np.allclose(r['flux'].mean(axis=(2, 3)), r1['flux'][:, :, 0, 0], rtol=2e-3, atol=1e-4)
        assert abs(r['counters']['scatter']-r1['counters']['scatter']) < 2e-3*r1['counters']['scatter']


def test_k8_partial_3d_solver_limits(oracle, nthreads):
    """partial 3-D (solver 1, mcarats.py:450-454): the direct beam is transported in 3-D, everything scattered in
    independent columns.  Same photon ids => (a) the direct-beam tallies are those of the 3-D solver bit for bit,
    (b) with an exactly vertical beam nothing ever leaves its column: identical to IPA,
    (c) the scattered light differs from both on a broken cloud field under a slant sun."""
    from er3t_amd.synth import les_scene
    n = 40000
    run = lambda sc: oracle.run(sc, n, seed=3, nthreads=nthreads)
    scs = {sv: les_scene(nx=12, ny=12, nz3=50, sza=60.0, saa=30.0, target='flux', solver=sv) for sv in (SOLVER_3D, SOLVER_P3D, SOLVER_IPA)}
    f = {sv: run(sc)['flux'] for sv, sc in scs.items()}
    assert np.array_equal(f[SOLVER_P3D][0], f[SOLVER_3D][0])                       # (a)
    assert not np.array_equal(f[SOLVER_IPA][0], f[SOLVER_3D][0])
    assert not np.allclose(f[SOLVER_P3D][2], f[SOLVER_3D][2]) and not np.allclose(f[SOLVER_P3D][2], f[SOLVER_IPA][2])    # (c)
    for sv in (SOLVER_P3D, SOLVER_IPA):                                            # domain-mean albedo: a 3-D effect of tens of percent at most
        assert abs(f[sv][2, -1].mean()/f[SOLVER_3D][2, -1].mean()-1.0) < 0.3
    v = {}
    for sv in (SOLVER_P3D, SOLVER_IPA):
        sc = les_scene(nx=12, ny=12, nz3=50, sza=0.0, target='radiance', vza=(0.0, 30.0), vaa=(0.0, 70.0), solver=sv)
        sc.src_qmax = 0.0
        sc.target = TARGET_FLUX | TARGET_RADIANCE
        v[sv] = run(sc)
    for key in ('flux', 'rad'):                                                    # (b) same histories; threads only reorder the sums
        assert np.allclose(v[SOLVER_P3D][key], v[SOLVER_IPA][key], rtol=1e-12, atol=0.0)
    assert v[SOLVER_P3D]['counters'] == v[SOLVER_IPA]['counters']


def test_k9_ipa_columns_are_their_own_plane_parallel_problems(oracle, nthreads):
    """independent-pixel approximation: the radiance and fluxes of a column equal those of the plane-parallel atmosphere
    made of that column alone, for clear and cloudy columns alike (this pins where events inside the 1-D layers above and
    below the 3-D region are registered: in the photon's column, not where its unfolded position points)"""
    import copy
    from er3t_amd.synth import les_scene
    sc = les_scene(nx=4, ny=4, nz3=50, sza=40.0, saa=20.0, vza=(0.0, 30.0), vaa=(0.0, 100.0), solver=SOLVER_IPA, surface_albedo=0.03)
    sc.target = TARGET_FLUX | TARGET_RADIANCE
    cot = (sc.extp[0]*40.0).sum(axis=0)                                       # (ny, nx) cloud optical thickness
    nb, nper = 8, 80000
    full = [oracle.run(sc, nper, seed=11, offset=b*nper, nthreads=nthreads) for b in range(nb)]
    rad = np.stack([r['rad'] for r in full]); fup = np.stack([r['flux'][2, -1] for r in full])
    for (iy, ix) in (np.unravel_index(np.argmin(cot), cot.shape), np.unravel_index(np.argmax(cot), cot.shape)):
        col = copy.copy(sc)
        col.nx = col.ny = col.nxr = col.nyr = 1
        for name in ('abst', 'extp', 'omgp', 'apfp'):
            a = getattr(sc, name)
            setattr(col, name, np.ascontiguousarray(a[..., iy:iy+1, ix:ix+1]))
        one = [oracle.run(col, nper//4, seed=12, offset=b*nper, nthreads=nthreads) for b in range(nb)]
        r1 = np.stack([r['rad'][:, 0, 0] for r in one]); f1 = np.stack([r['flux'][2, -1, 0, 0] for r in one])
        for iv in range(2):
            a, b = rad[:, iv, iy, ix], r1[:, iv]
            se = np.sqrt(a.var(ddof=1)/nb + b.var(ddof=1)/nb)
            assert abs(a.mean()-b.mean()) < 4.0*se + 1e-3*b.mean(), (iy, ix, iv, a.mean(), b.mean(), se)
        a, b = fup[:, iy, ix], f1
        se = np.sqrt(a.var(ddof=1)/nb + b.var(ddof=1)/nb)
        assert abs(a.mean()-b.mean()) < 4.0*se + 1e-3*b.mean(), (iy, ix, a.mean(), b.mean(), se)
    assert cot.min() < 0.5 and cot.max() > 5.0


def test_k10_roulette_on_local_estimate_rays_is_unbiased(oracle, nthreads):
    """Russian roulette on local-estimate rays (le_tau1 > 0): same expectation as marching every ray to the sensor, for
    slant views through an optically thick slab; the vertical view of a sensor above the atmosphere is left alone"""
    kw = dict(tau=12.0, omega=0.999, apf=0.85, albedo=0.1, sza=40.0, nz=12, vza=(0.0, 55.0, 70.0), vaa=(0.0, 30.0, 200.0))
    nb, nper = 12, 30000
    res = {}
    for tau1 in (0.0, 1.5):
        sc = slab_scene(**kw); sc.le_tau1 = tau1
        res[tau1] = np.stack([oracle.run(sc, nper, seed=4, offset=b*nper, nthreads=nthreads)['rad'][:, 0, 0] for b in range(nb)])
    a, b = res[0.0], res[1.5]
    assert np.allclose(a[:, 0], b[:, 0], rtol=1e-12)                           # vertical view: untouched
    assert not np.allclose(a[:, 1:], b[:, 1:], rtol=1e-6)                     # slant views: the roulette is being played
    se = np.sqrt(a.var(axis=0, ddof=1)/nb + b.var(axis=0, ddof=1)/nb)
    assert np.all(np.abs(a.mean(axis=0)-b.mean(axis=0))[1:] < 3.5*se[1:]), (a.mean(axis=0), b.mean(axis=0), se)
    assert np.all(b.std(axis=0, ddof=1)[1:] < 2.0*a.std(axis=0, ddof=1)[1:])   # and costs little extra noise


def test_k13_geometry_shadow_and_parallax(oracle, nthreads):
    """pure geometry, exact: (a) the shadow one opaque voxel casts on the ground under a slant sun (3-D walk, cyclic boundary,
    flux columns); (b) where a small cloud appears in the image of a slant view (local-estimate ray, pixel registration at
    Rad_zref): tests/util.py block_scene / block_expectations"""
    shadow, image = block_expectations()
    mu0 = np.cos(np.deg2rad(45.0))
    n = 960000
    f = oracle.run(block_scene('absorber'), n, seed=2, nthreads=nthreads)['flux'][0, 0]/mu0          # direct beam at the ground
    per_col = n/48.0
    assert np.all(np.abs(f[3]-shadow) < 5.0*np.sqrt(np.maximum(shadow, 1e-4)/per_col) + 0.01), (f[3], shadow)
    assert np.all(np.abs(np.delete(f, 3, axis=0)-1.0) < 5.0/np.sqrt(per_col))                          # the other rows: untouched
    r = oracle.run(block_scene('scatterer'), 24000000, seed=2, nthreads=nthreads)['rad']   # 1/48 of them meet the voxel, 1 % of those scatter
    for iv, want in ((0, None), (1, image[-1.0]), (2, image[1.0])):
        img = r[iv]/r[iv].sum()
        assert np.all(np.delete(img, 3, axis=0) == 0.0)                                               # nothing off the voxel's row
        if want is None:
            assert img[3, 2] == 1.0                                                                    # nadir: the voxel's own pixel
        else:
            assert np.all(img[3][want == 0.0] == 0.0) and np.all(np.abs(img[3]-want) < 0.03), (iv, img[3], want)       # 5e3 events: sigma 0.007


def test_photon_ranges_add_up(oracle):
    # linearity in the photon-id range: [0,N) == [0,N/2) + [N/2,N)  (what photon sharding relies on)
    sc = slab_scene(tau=2.0, apf=0.5, albedo=0.1, nx=3, ny=2, nz3=2)
    n = 6000
    full = oracle.run_raw(sc, n, seed=8, offset=0, nthreads=1)
    a = oracle.run_raw(sc, n//2, seed=8, offset=0, nthreads=1)
    b = oracle.run_raw(sc, n-n//2, seed=8, offset=n//2, nthreads=1)
    assert np.allclose(full[0], a[0]+b[0], rtol=1e-12, atol=0)
    assert np.allclose(full[1], a[1]+b[1], rtol=1e-12, atol=0)
    assert np.array_equal(full[2], a[2]+b[2])


def test_threads_do_not_change_the_answer(oracle, nthreads):
    sc = slab_scene(tau=2.0, apf=0.5, albedo=0.1, nx=3, ny=2, nz3=2)
    a = oracle.run_raw(sc, 5000, seed=8, nthreads=1)
    b = oracle.run_raw(sc, 5000, seed=8, nthreads=max(2, nthreads))
    assert np.allclose(a[0], b[0], rtol=1e-11) and np.allclose(a[1], b[1], rtol=1e-11) and np.array_equal(a[2], b[2])


# ---------------------------------------------------------------------------------------------
def test_phase_samplers_moments(oracle):
    u = (np.arange(10000)+0.5)/10000.0
    for g in (-0.5, 0.3, 0.85):
        mu = np.array([oracle.phase_sample(g, x) for x in u])
        assert abs(mu.mean()-g) < 2e-4
    mu = np.array([oracle.phase_sample(-1.0, x) for x in u])          # Rayleigh
    assert abs(mu.mean()) < 1e-6 and abs((mu**2).mean()-0.4) < 1e-4
    mu = np.array([oracle.phase_sample(-2.0, x) for x in u])          # isotropic
    assert abs((mu**2).mean()-1.0/3.0) < 1e-4
    # normalisation: (1/2) * integral of P over mu is 1
    x = np.linspace(-1, 1, 20001)
    for apf in (-2.0, -1.0, 0.0, 0.85):
        p = np.array([oracle.phase_eval(apf, m) for m in x])
        assert abs(0.5*np.trapezoid(p, x)-1.0) < 2e-4


def test_tabulated_phase_function(oracle):
    g = 0.85
    ang = np.linspace(0.0, 180.0, 1801)
    pha = (1-g*g)/(1+g*g-2*g*np.cos(np.deg2rad(ang)))**1.5
    sc = slab_scene(tau=1.0, apf=1.0, ang=ang, pha=3.7*pha[None])         # arbitrary scale: the table is renormalised
    mu = np.linspace(-1.0, 0.999, 4001)
    u = (np.arange(4001)+0.5)/4001.0
    p, m = oracle.phase_table(sc, 0, mu, u)
    exact = (1-g*g)/(1+g*g-2*g*mu)**1.5
    assert np.allclose(p, exact, rtol=2e-3)
    assert abs(m.mean()-g) < 2e-3
    # sampling inverts the table's own CDF: the sampled mu ascend with u and span the whole range
    assert np.all(np.diff(m) >= 0.0) and m[0] < -0.9 and m[-1] > 0.999
    # apf = 1 selects table 1; the analytic HG must agree with its table in a radiance run
    kw = dict(tau=1.0, omega=1.0, albedo=0.0, nz=2, target=TARGET_RADIANCE)
    ra = oracle.run(slab_scene(apf=g, **kw), 60000, seed=3, nthreads=4)['rad'][0, 0, 0]
    rt = oracle.run(slab_scene(apf=1.0, ang=ang, pha=pha[None], **kw), 60000, seed=3, nthreads=4)['rad'][0, 0, 0]
    assert abs(ra-rt) < 0.02*ra


def test_lsrt_kernels(oracle):
    def dirs(sza, vza, phi):
        si, sv = np.sin(np.deg2rad(sza)), np.sin(np.deg2rad(vza))
        din = np.array([-si, 0.0, -np.cos(np.deg2rad(sza))])                 # photon travelling away from the sun at azimuth 0
        dout = np.array([sv*np.cos(np.deg2rad(phi)), sv*np.sin(np.deg2rad(phi)), np.cos(np.deg2rad(vza))])
        return din, dout
    # both kernels vanish for sun and viewer at the zenith
    din, dout = dirs(0.0, 0.0, 0.0)
    assert abs(oracle.lsrt(0.0, 1.0, 0.0, din, dout)) < 1e-9 and abs(oracle.lsrt(0.3, 0.0, 1.0, din, dout)-0.3) < 1e-9
    # hand-computed values for sza = 30 deg, nadir view: Kvol = -0.03143, Kgeo = -0.69820 (negative: use an offset)
    din, dout = dirs(30.0, 0.0, 0.0)
    assert abs(oracle.lsrt(1.0, 0.0, 1.0, din, dout)-(1.0-0.03143)) < 2e-4
    assert abs(oracle.lsrt(1.0, 1.0, 0.0, din, dout)-(1.0-0.69820)) < 2e-4
    # reciprocity
    for sza, vza, phi in ((20.0, 55.0, 40.0), (60.0, 10.0, 170.0)):
        din, dout = dirs(sza, vza, phi)
        din2, dout2 = -dout, -din
        assert abs(oracle.lsrt(0.2, 0.05, 0.1, din, dout)-oracle.lsrt(0.2, 0.05, 0.1, din2, dout2)) < 1e-12
    # hot spot: the volumetric kernel peaks in the backscatter direction
    din, dout = dirs(40.0, 40.0, 0.0)
    hot = oracle.lsrt(0.0, 0.0, 1.0, din, dout)
    din, dout = dirs(40.0, 40.0, 60.0)
    assert hot > oracle.lsrt(0.0, 0.0, 1.0, din, dout)


def test_lsrt_surface_in_transport(oracle, nthreads):
    # empty atmosphere over an LSRT surface: nadir radiance = R(sun, nadir) * mu0 / pi exactly
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); f = (0.25, 0.03, 0.12)
    sc = slab_scene(tau=0.0, sza=sza, nx=2, ny=2, target=TARGET_RADIANCE)
    sc.jsfc = np.full((2, 2), 4.0, dtype=np.float32)
    sc.psfc = np.zeros((5, 2, 2), dtype=np.float32); sc.psfc[0] = f[0]; sc.psfc[1] = f[1]; sc.psfc[2] = f[2]
    r = oracle.run(sc, 20000, seed=1, nthreads=nthreads)
    din = np.array([0.0, -np.sin(np.deg2rad(sza)), -mu0]); dout = np.array([0.0, 0.0, 1.0])
    R = oracle.lsrt(np.float32(f[0]), np.float32(f[1]), np.float32(f[2]), din, dout)
    # (per-pixel values carry the launch-position noise; the domain mean is exact)
    assert np.isclose(r['rad'][0].mean(), R*mu0/np.pi, rtol=1e-6)


# ---------------------------------------------------------------------------------------------
# K14  diffuse-specular mixture (jsfc = 2, Cox-Munk): the pieces have closed forms
# ---------------------------------------------------------------------------------------------
def _dirs(sza, vza, phi):
    si, sv = np.sin(np.deg2rad(sza)), np.sin(np.deg2rad(vza))
    din = np.array([si, 0.0, -np.cos(np.deg2rad(sza))])                 # photon travelling away from a sun at azimuth 180
    dout = np.array([sv*np.cos(np.deg2rad(phi)), sv*np.sin(np.deg2rad(phi)), np.cos(np.deg2rad(vza))])
    return din, dout


def test_dsm_fresnel_closed_forms(oracle):
    # normal incidence: ((n-1)^2 + k^2) / ((n+1)^2 + k^2)
    for n, k in ((1.333, 0.0), (1.5, 0.0), (1.34, 0.2), (0.2, 3.4)):
        assert abs(oracle.fresnel(n, k, 1.0)-((n-1)**2+k*k)/((n+1)**2+k*k)) < 1e-12
    # Brewster angle of a dielectric: the parallel component vanishes, what is left is half of sin^2(ti - tt)
    n = 1.5
    ti = np.arctan(n); tt = np.arcsin(np.sin(ti)/n)
    assert abs(oracle.fresnel(n, 0.0, np.cos(ti))-0.5*np.sin(ti-tt)**2) < 1e-12
    # a general angle against the Fresnel equations written with the refraction angle
    ti = np.deg2rad(63.0); tt = np.arcsin(np.sin(ti)/1.34)
    rs = (np.sin(ti-tt)/np.sin(ti+tt))**2; rp = (np.tan(ti-tt)/np.tan(ti+tt))**2
    assert abs(oracle.fresnel(1.34, 0.0, np.cos(ti))-0.5*(rs+rp)) < 1e-12
    assert abs(oracle.fresnel(1.34, 0.0, 1e-9)-1.0) < 1e-6                # grazing incidence reflects everything


def test_dsm_reflectance_properties(oracle):
    water = [0.0, 0.0, 1.34, 0.0, 0.02]
    # all diffuse: a Lambert surface of the diffuse albedo, whatever the rest
    din, dout = _dirs(35.0, 50.0, 70.0)
    assert abs(oracle.dsm([0.22, 1.0, 1.34, 0.0, 0.02], din, dout)[0]-0.22) < 1e-12
    # mixture: linear in the diffuse fraction
    r0 = oracle.dsm(water, din, dout)[0]
    assert abs(oracle.dsm([0.22, 0.3, 1.34, 0.0, 0.02], din, dout)[0]-(0.3*0.22+0.7*r0)) < 1e-12
    # reciprocity
    for sza, vza, phi in ((20.0, 55.0, 40.0), (60.0, 10.0, 170.0), (45.0, 45.0, 0.0)):
        din, dout = _dirs(sza, vza, phi)
        assert abs(oracle.dsm(water, din, dout)[0]-oracle.dsm(water, -dout, -din)[0]) < 1e-12
    # the glint peaks around the mirror direction of the mean surface (phi = 0 here) and is negligible in the backscatter direction
    din, d_spec = _dirs(40.0, 40.0, 0.0)
    _, d_side = _dirs(40.0, 40.0, 60.0)
    _, d_back = _dirs(40.0, 40.0, 180.0)
    rs, rside, rback = (oracle.dsm(water, din, d)[0] for d in (d_spec, d_side, d_back))
    assert rs > 10.0*rside and rback < 1e-6*rs
    # at the mirror direction of a nearly flat surface the value is pi F P(0) S / (4 mu^2) with P(0) = 1 / (pi sigma^2)
    s2 = 0.004
    mu = np.cos(np.deg2rad(40.0))
    want = oracle.fresnel(1.34, 0.0, mu)/(4.0*mu*mu*s2)
    got = oracle.dsm([0.0, 0.0, 1.34, 0.0, s2], din, d_spec)[0]
    assert abs(got/want-1.0) < 1e-3                                        # (shadowing at 40 degrees and sigma = 0.06: none)


def test_dsm_conserves_energy_with_mirror_facets(oracle):
    """facets that reflect everything (|m| -> infinity): the directional-hemispherical reflectance (1/pi) int R mu dOmega is 1
    up to the light the shadowing function removes (facets hidden from the sun or from the viewer), which vanishes for a
    smooth surface and near-vertical incidence"""
    n = 300
    th = (np.arange(n)+0.5)*(0.5*np.pi)/n; ph = (np.arange(2*n)+0.5)*np.pi/n
    T, P = np.meshgrid(th, ph, indexing='ij')
    d = np.stack([np.sin(T)*np.cos(P), np.sin(T)*np.sin(P), np.cos(T)], axis=-1).reshape(-1, 3)
    w = (np.cos(T)*np.sin(T)).ravel()*(0.5*np.pi/n)*(np.pi/n)/np.pi

    def albedo(s2, sza):
        din, _ = _dirs(sza, 0.0, 0.0)
        return float((oracle.dsm([0.0, 0.0, 1.0e4, 0.0, s2], din, d)*w).sum())
    assert abs(albedo(0.003, 0.0)-1.0) < 2e-3 and abs(albedo(0.003, 60.0)-1.0) < 2e-3
    assert abs(albedo(0.02, 30.0)-1.0) < 2e-3
    a = albedo(0.05, 72.0)
    assert 0.85 < a < 1.0                                                  # rough and grazing: shadowing takes a few per cent
    # water: the albedo at normal incidence is the Fresnel value to within the spread of facet tilts
    din, _ = _dirs(0.0, 0.0, 0.0)
    aw = float((oracle.dsm([0.0, 0.0, 1.34, 0.0, 0.003], din, d)*w).sum())
    assert abs(aw/oracle.fresnel(1.34, 0.0, 1.0)-1.0) < 0.02


def test_dsm_surface_in_transport(oracle, nthreads):
    # empty atmosphere over a diffuse-specular surface: radiance of a slant view = R(sun, view) * mu0 / pi exactly
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); p = (0.1, 0.2, 1.34, 0.0, 0.03)
    sc = slab_scene(tau=0.0, sza=sza, nx=2, ny=2, target=TARGET_RADIANCE, vza=(25.0,), vaa=(0.0,))
    sc.jsfc = np.full((2, 2), 2.0, dtype=np.float32)
    sc.psfc = np.zeros((5, 2, 2), dtype=np.float32)
    for q in range(5):
        sc.psfc[q] = p[q]
    r = oracle.run(sc, 20000, seed=1, nthreads=nthreads)
    th = np.deg2rad(sc.src_the); ph = np.deg2rad(sc.src_phi)
    din = np.array([np.sin(th)*np.cos(ph), np.sin(th)*np.sin(ph), np.cos(th)])
    tv = np.deg2rad(sc.view_the[0]); pv = np.deg2rad(sc.view_phi[0])
    dout = -np.array([np.sin(tv)*np.cos(pv), np.sin(tv)*np.sin(pv), np.cos(tv)])
    R = oracle.dsm(np.float32(p), din, dout)[0]
    assert np.isclose(r['rad'][0].mean(), R*mu0/np.pi, rtol=1e-6)
    # uniform surface given by Sfc_mtype / Sfc_param: the same
    sc2 = slab_scene(tau=0.0, sza=sza, nx=2, ny=2, target=TARGET_RADIANCE, vza=(25.0,), vaa=(0.0,))
    sc2.sfc_mtype = 2; sc2.sfc_param = np.array(p, dtype=np.float32)
    r2 = oracle.run(sc2, 20000, seed=1, nthreads=nthreads)
    assert np.isclose(r2['rad'][0].mean(), R*mu0/np.pi, rtol=1e-6)


# ---------------------------------------------------------------------------------------------
# K15  all-sky camera (Rad_mrkind = 1): a point sensor with the 1/r^2 local estimate
# ---------------------------------------------------------------------------------------------
def _camera(sc, the, zloc, nxr, nyr, qmax=120.0, umax=120.0, xpos=0.5, ypos=0.5, apsize=0.0, phi=0.0, psi=0.0):
    sc.rad_kind = 1
    sc.view_the = [float(the)]; sc.view_phi = [float(phi)]; sc.view_zloc = [float(zloc)]
    sc.cam_psi = [float(psi)]; sc.cam_xpos = [float(xpos)]; sc.cam_ypos = [float(ypos)]
    sc.cam_qmax = [float(qmax)]; sc.cam_umax = [float(umax)]; sc.cam_vmax = [float(umax)]; sc.cam_apsize = [float(apsize)]
    sc.nxr = nxr; sc.nyr = nyr
    return sc


def _pixel_theta(nxr, nyr, umax):
    """largest angle from the camera axis inside each pixel of the polar map"""
    du = np.deg2rad(umax)/nxr; dv = np.deg2rad(umax)/nyr
    ue = (np.arange(nxr+1)-0.5*nxr)*du; ve = (np.arange(nyr+1)-0.5*nyr)*dv
    umx = np.maximum(np.abs(ue[:-1]), np.abs(ue[1:])); vmx = np.maximum(np.abs(ve[:-1]), np.abs(ve[1:]))
    return np.sqrt(vmx[:, None]**2+umx[None, :]**2)


def test_camera_above_a_lambert_plane(oracle, nthreads):
    """no atmosphere, Lambertian ground of albedo A under a sun at mu0: every line of sight that meets the ground reads
    A mu0 / pi, whatever its direction and the camera's height -- closed form for the point estimator, its 1/r^2, the solid
    angle of the polar pixel map and the normalisation.  (Lines of sight beyond the nearest periodic image of the domain
    are not complete: the cone of view is kept inside it.)"""
    A, sza = 0.4, 35.0
    mu0 = np.cos(np.deg2rad(sza))
    sc = slab_scene(tau=0.0, albedo=A, sza=sza, nx=40, ny=40, dx=200.0, dy=200.0, target=TARGET_RADIANCE)
    _camera(sc, the=180.0, zloc=600.0, nxr=8, nyr=8, qmax=120.0, umax=120.0, xpos=0.3, ypos=0.6)
    nb, nper = 8, 250000
    runs = np.stack([oracle.run(sc, nper, seed=3, offset=b*nper, nthreads=nthreads)['rad'][0] for b in range(nb)])
    img, se = runs.mean(axis=0), runs.std(axis=0, ddof=1)/np.sqrt(nb)
    inside = _pixel_theta(8, 8, 120.0) < np.deg2rad(60.0)                 # pixels wholly inside the cone of view
    assert inside.sum() >= 24
    want = A*mu0/np.pi
    assert np.all(np.abs(img[inside]-want) < 5.0*se[inside] + 0.01*want), (img[inside], want)
    assert abs(img[inside].mean()-want) < 0.01*want
    # pixels wholly outside the cone see nothing
    du = np.deg2rad(120.0)/8
    ue = (np.arange(9)-4)*du
    umn = np.minimum(np.abs(ue[:-1]), np.abs(ue[1:]))
    outside = np.sqrt(umn[:, None]**2+umn[None, :]**2) > np.deg2rad(60.0)
    assert np.all(img[outside] == 0.0)


def test_camera_sees_the_periodic_images_of_the_domain(oracle, nthreads):
    """the domain is cyclic (er3t's cameras look 89 degrees off their axis, mcarats.py:291-296): a camera 600 m above a Lambertian
    plane of 2 km x 2 km sees its nearest image out to atan(1000 / 600) = 59 degrees only.  With `cam_images` = 2 an event
    contributes to the 25 images of the camera within two domain lengths, the ground out to 5 km is there, and every line of sight
    inside 75 degrees reads A mu0 / pi; with the nearest image alone the ring between 60 and 75 degrees is incomplete (the number
    below is the bias VERDICT r3 asked for: a fifth of the light at those angles on this geometry)."""
    A, sza = 0.4, 35.0
    mu0 = np.cos(np.deg2rad(sza))
    want = A*mu0/np.pi
    imgs = {}
    for nimg in (0, 2):
        sc = slab_scene(tau=0.0, albedo=A, sza=sza, nx=10, ny=10, dx=200.0, dy=200.0, target=TARGET_RADIANCE)
        _camera(sc, the=180.0, zloc=600.0, nxr=8, nyr=8, qmax=160.0, umax=160.0, xpos=0.3, ypos=0.6)
        sc.cam_images = nimg
        nb, nper = 8, 200000
        runs = np.stack([oracle.run(sc, nper, seed=3, offset=b*nper, nthreads=nthreads)['rad'][0] for b in range(nb)])
        imgs[nimg] = (runs.mean(axis=0), runs.std(axis=0, ddof=1)/np.sqrt(nb))
    th = _pixel_theta(8, 8, 160.0)
    inner, ring = th < np.deg2rad(55.0), (th > np.deg2rad(62.0)) & (th < np.deg2rad(75.0))
    assert inner.sum() >= 4 and ring.sum() >= 8
    img, se = imgs[2]
    assert np.all(np.abs(img[inner | ring]-want) < 5.0*se[inner | ring] + 0.015*want), (img/want)
    assert abs(img[ring].mean()-want) < 0.01*want
    img0, se0 = imgs[0]
    assert np.all(np.abs(img0[inner]-want) < 5.0*se0[inner] + 0.015*want)       # inside the nearest image nothing changes
    assert img0[ring].mean() < 0.9*want, img0[ring].mean()/want                # beyond it the nearest image alone reads low


def test_camera_equals_the_plane_parallel_radiance(oracle, nthreads):
    """horizontally homogeneous atmosphere: the radiance a camera on the ground records in the direction (theta, phi) is the
    radiance field of the plane-parallel problem, which the pixel-area estimator (Rad_mrkind = 2, up-looking sensor) gives
    for the same direction -- two estimators with different geometry (point sensor with 1/r^2 and solid angles against a plane
    of sensors with 1/|mu| and pixel areas) on multiple scattering in a conservative Henyey-Greenstein slab"""
    kw = dict(tau=0.8, omega=1.0, apf=0.6, albedo=0.2, sza=40.0, nx=30, ny=30, dx=400.0, dy=400.0, nz=4, ztop=2000.0, target=TARGET_RADIANCE)
    cam = _camera(slab_scene(**kw), the=0.0, zloc=0.0, nxr=5, nyr=5, qmax=150.0, umax=150.0, apsize=20.0)
    cam.le_tau1 = 0.0
    nb, nper = 8, 200000
    runs = np.stack([oracle.run(cam, nper, seed=9, offset=b*nper, nthreads=nthreads)['rad'][0] for b in range(nb)])
    img, se = runs.mean(axis=0), runs.std(axis=0, ddof=1)/np.sqrt(nb)
    du = np.deg2rad(150.0)/5
    for (jr, ir) in ((2, 2), (2, 3), (1, 2), (3, 1)):
        U, V = (ir-2)*du, (jr-2)*du                                      # pixel centre
        theta, phi = np.hypot(U, V), np.arctan2(V, U)
        # the camera looks along (sin(theta) cos(phi), sin(theta) sin(phi), cos(theta)) (its axes are the world's: the = phi = psi = 0);
        # the light it sees travels the opposite way: an up-looking sensor of zenith angle 180 - theta ... in this build's view
        # convention (view_the, view_phi) is the direction the sensor looks in
        sat = slab_scene(**kw)
        sat.view_the = [float(np.rad2deg(theta))]; sat.view_phi = [float(np.rad2deg(phi))]; sat.view_zloc = [0.0]; sat.nxr = 1; sat.nyr = 1
        sat.le_tau1 = 0.0
        if theta == 0.0:
            sat.view_the = [0.0]
        ref = np.stack([oracle.run(sat, nper, seed=21, offset=b*nper, nthreads=nthreads)['rad'][0].mean() for b in range(nb)])
        rm, rse = ref.mean(), ref.std(ddof=1)/np.sqrt(nb)
        # the pixel averages the field over 30 x 30 degrees around its centre: allow 6 % for the curvature of the field
        assert abs(img[jr, ir]-rm) < 4.0*np.hypot(se[jr, ir], rse) + 0.06*rm, ((jr, ir), img[jr, ir], rm, se[jr, ir], rse)


# ---------------------------------------------------------------------------------------------
# K17: heating rates (Flx_mhrt = 1, er3t/rtm/mca/mcarats.py:279-283)
# ---------------------------------------------------------------------------------------------
def test_k17_heating_of_a_purely_absorbing_slab_is_beers_law(oracle, nthreads):
    """nothing scatters: a photon leaves all its weight where its first collision is, so the power absorbed in a layer is the
    drop of the direct beam across it, mu0 [exp(-tau_top/mu0) - exp(-tau_bottom/mu0)], per unit volume: / dz"""
    sza = 40.0; mu0 = np.cos(np.deg2rad(sza)); tau_abs = 1.3
    sc = slab_scene(tau=0.0, abs_tau=tau_abs, albedo=0.0, sza=sza, nz=5, nx=3, ny=2, target=TARGET_FLUX | TARGET_HEAT)
    n = 600000
    r = oracle.run(sc, n, seed=3, nthreads=nthreads)
    z = sc.zgrd
    beam = mu0*np.exp(-tau_abs*(z[-1]-z)/z[-1]/mu0)                # direct flux at the levels
    want = np.diff(beam)/np.diff(z)
    got = r['heat'].mean(axis=(1, 2))
    assert r['heat'].shape == (5, 2, 3)
    sigma = np.sqrt(np.diff(beam)/mu0/n)*mu0/np.diff(z)
    assert np.all(np.abs(got-want) < 4.5*sigma), (got, want, sigma)
    # a conservative atmosphere absorbs nothing
    sc0 = slab_scene(tau=3.0, omega=1.0, albedo=0.5, sza=sza, nz=4, target=TARGET_FLUX | TARGET_HEAT)
    assert np.all(oracle.run(sc0, 20000, seed=1, nthreads=nthreads)['heat'] == 0.0)


def test_k17_energy_budget_closes_in_every_column_under_ipa(oracle, nthreads):
    """without Russian roulette (Pho_wmin = 0) a photon's weight goes three ways only: out through the top, into the surface, into
    the cells where it collided.  Under the independent-column approximation a photon never leaves its column, so per column
        F_down(TOA) - F_up(TOA) - [F_down(surface) - F_up(surface)] = sum over layers of heating x thickness
    to rounding, history by history -- gas absorption, an absorbing aerosol and a cloud field in the columns"""
    from er3t_amd.synth import les_scene
    sc = les_scene(nx=6, ny=5, nz3=50, target='flux', aerosol=True, solver=SOLVER_IPA, surface_albedo=0.3)
    sc.target = TARGET_FLUX | TARGET_HEAT
    sc.wmin = 0.0
    sc.abs1d = sc.abs1d*30.0
    r = oracle.run(sc, 60000, seed=8, nthreads=nthreads)
    f = r['flux']
    absorbed = (r['heat']*np.diff(sc.zgrd)[:, None, None]).sum(axis=0)
    budget = (f[1, -1]-f[2, -1]) - (f[1, 0]-f[2, 0])
    assert absorbed.min() > 0.0 and absorbed.mean() > 0.02*sc.mu0
    assert np.allclose(absorbed, budget, rtol=1e-9, atol=1e-12), np.abs(absorbed-budget).max()


# ---------------------------------------------------------------------------------------------
# K18: the Russian roulette on the weight of local-estimate rays is unbiased and thins the rays
# ---------------------------------------------------------------------------------------------
def test_k18_weight_roulette_of_local_estimates_keeps_the_mean(oracle, nthreads):
    """Scene.le_cmin (include/mi3d.h: mi3d_set_le_weight_roulette): estimates that would carry less than c_min are marched with
    probability c / c_min at weight c_min.  Same photons with the roulette off, at the default 1/(4 pi) and at four times that:
    the radiances of six slant views agree within their noise (the histories are the same: only which rays are marched differs),
    far fewer cells are walked by the rays, and the view answered without marching in the HIP path (nadir from above the
    atmosphere) is never touched."""
    vza = (0.0, 26.1, 45.6, 60.0, 70.5, 45.6, 60.0); vaa = (0.0, 0.0, 180.0, 0.0, 180.0, 60.0, 235.0)
    kw = dict(tau=8.0, omega=1.0, apf=0.85, albedo=0.2, sza=40.0, nz=8, nx=4, ny=4, vza=vza, vaa=vaa, target=TARGET_RADIANCE)
    nb, nper = 12, 25000
    res = {}
    for cmin in (0.0, 0.0796, 0.32):
        sc = slab_scene(le_cmin=cmin, **kw)
        runs = [oracle.run(sc, nper, seed=9, offset=b*nper, nthreads=nthreads) for b in range(nb)]
        rad = np.stack([r['rad'].mean(axis=(1, 2)) for r in runs])
        res[cmin] = (rad.mean(0), rad.std(0, ddof=1)/np.sqrt(nb), sum(r['counters']['le_steps'] for r in runs))
    m0, se0, steps0 = res[0.0]
    for cmin in (0.0796, 0.32):
        m, se, steps = res[cmin]
        assert np.isclose(m[0], m0[0], rtol=1e-12, atol=0.0)               # the column view: no roulette, the same contributions (sum order aside)
        z = (m[1:]-m0[1:])/np.sqrt(se[1:]**2+se0[1:]**2)
        assert np.all(np.abs(z) < 3.0) and abs(z.mean()) < 1.0, (cmin, z)
        assert steps < (0.75 if cmin < 0.1 else 0.5)*steps0, (cmin, steps, steps0)
