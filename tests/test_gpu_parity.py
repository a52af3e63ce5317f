"""
Parity tests proper: the HIP path, called through the C-ABI (libmi3drt.so via er3t_amd.solver), against the CPU
oracle on the same seeded inputs.  Run with `-m gpu` on an MI355X.

Floating-point Monte Carlo: both sides consume the same Philox stream per photon id, but float32 vs float64
rounding lets individual histories part ways, so agreement is statistical.  Tolerances (stated per test):
  * domain means within 2 sigma of the combined standard error (sigma from 16 oracle batches; north_star's figure).  The two
    sides follow the same histories, so their difference is a fraction of the error of two independent runs (measured |z| of
    the domain means on these grids: below 0.6): the false-alarm budget of the 2-sigma bound over this file's ~150 comparisons is
    what a flipped history costs, not 4.6 % each,
  * per-pixel z-scores: fewer than 5 % beyond |z| > 3 and |mean z| < 0.5,
  * event counters within 1 % (roulette games 1.5 %: a float32 weight equal to wmin flips the comparison),
  * deterministic identities (Philox words, Lambert surface, id-range additivity, column-table vs marched
    local estimate) to rounding.
"""

import numpy as np
import pytest

from er3t_amd.scene import Scene, TARGET_FLUX, TARGET_RADIANCE, TARGET_HEAT, SOLVER_3D, SOLVER_P3D, SOLVER_IPA
from er3t_amd.synth import les_scene, z_levels_config4, pha_hg_synth
from tests.util import slab_scene, block_scene, block_expectations

pytestmark = pytest.mark.gpu


def gpu_run(solver, scene, nphoton, seed=7, offset=0, column_le=True, counting=True):
    solver.bind(None, None, None)
    solver.load_scene(scene, column_le=column_le)
    solver.set_counting(counting)
    solver.reset()
    solver.run(nphoton, seed=seed, offset=offset)
    solver.sync()
    out = {'counters': solver.counters()}
    if scene.target & TARGET_RADIANCE:
        out['rad'] = solver.radiance(nphoton).astype(np.float64)
    if scene.target & TARGET_FLUX:
        out['flux'] = solver.flux(nphoton).astype(np.float64)
    if scene.target & TARGET_HEAT:
        out['heat'] = solver.heating(nphoton).astype(np.float64)
    return out


def oracle_batches(oracle, scene, nbatch, nper, seed, nthreads):
    rad, flux = [], []
    cnt = None
    for b in range(nbatch):
        r = oracle.run(scene, nper, seed=seed, offset=b*nper, nthreads=nthreads)
        rad.append(r['rad']); flux.append(r['flux'])
        cnt = r['counters'] if cnt is None else {k: cnt[k]+r['counters'][k] for k in cnt}
    rad = np.stack(rad); flux = np.stack(flux)
    return {'rad': rad.mean(0), 'rad_se': rad.std(0, ddof=1)/np.sqrt(nbatch),
            'flux': flux.mean(0), 'flux_se': flux.std(0, ddof=1)/np.sqrt(nbatch), 'counters': cnt,
            'rad_mean_se': rad.mean(axis=(2, 3)).std(0, ddof=1)/np.sqrt(nbatch),
            'flux_mean_se': flux.mean(axis=(3, 4)).std(0, ddof=1)/np.sqrt(nbatch)}


def check_counters(g, o, skip=()):
    # event counts are properties of the histories; cell-step counts are not compared: the HIP path crosses
    # horizontally uniform layers without walking their cells, the oracle walks every cell
    # tolerance: the counts are sums over a few 1e5 histories whose lengths scatter widely (std of the total
    # ~0.2 %); once rounding has split a pair of histories they are independent, so allow 1 %
    # (flux tallies are not compared one to one: the HIP path does not tally the direct beam above the 3-D region, which it
    #  adds analytically when the result is read, the oracle tallies every crossing)
    assert g['flux_tally'] <= o['flux_tally']
    for k in ('photons', 'scatter', 'surface', 'le_rays', 'killed', 'escaped'):
        if k in skip:
            continue
        tol = 1e-2*max(o[k], 1) + 30
        assert abs(g[k]-o[k]) <= tol, (k, g[k], o[k])
    assert abs(g['roulette']-o['roulette']) <= 1.5e-2*max(o['roulette'], 1) + 30


def check_radiance(g, o, zstd_max=None):
    # zstd_max: the HIP path and the oracle follow the same histories (same Philox stream) until float32 rounding flips a
    # decision, so their per-pixel difference is far below that of two independent runs (std of z = 1); how far depends on
    # how soon histories part: hardly ever under the independent-pixel approximation, after a few 3-D cell walks otherwise
    for iv in range(o['rad'].shape[0]):
        gm, om = g['rad'][iv].mean(), o['rad'][iv].mean()
        se = o['rad_mean_se'][iv]
        assert abs(gm-om) < 2.0*np.sqrt(2.0)*se + 1e-4*om, (iv, gm, om, se)
        sep = np.maximum(o['rad_se'][iv], 1e-12)
        z = (g['rad'][iv]-o['rad'][iv])/(np.sqrt(2.0)*sep)
        assert np.mean(np.abs(z) > 3.0) < 0.05, (iv, np.mean(np.abs(z) > 3.0))
        assert abs(z.mean()) < 0.5, (iv, z.mean())
        if zstd_max is not None:
            assert z.std() < zstd_max, (iv, z.std())


# ---------------------------------------------------------------------------------------------
def test_philox_words_bit_exact(solver, oracle):
    for seed, id0, draw in ((1234, 0, 0), (0xdeadbeefcafef00d, (1 << 40)+17, 5), (7, (1 << 32)-3, 123456)):
        got = solver.philox(seed, id0, draw, 257)
        want = np.stack([oracle.philox(seed, id0+i, draw) for i in range(257)])
        assert np.array_equal(got, want)


def test_lambert_surface_identity(solver):
    # K2: no atmosphere; each photon contributes a*mu0/pi to one pixel: exact up to float32 summation
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); a = 0.3
    sc = slab_scene(tau=0.0, albedo=a, sza=sza, nx=4, ny=3, vza=(0.0, 50.0), vaa=(0.0, 77.0))
    n = 120000
    g = gpu_run(solver, sc, n)
    # (10^4 equal float32 addends per pixel round the same way every time: systematic, up to ~3e-4 relative)
    assert np.allclose(g['rad'].mean(axis=(1, 2)), a*mu0/np.pi, rtol=5e-4)
    assert np.isclose(g['flux'][2, -1].mean(), a*mu0, rtol=5e-4)
    assert np.isclose(g['flux'][1, 0].mean(), mu0, rtol=2e-5)
    assert g['counters']['surface'] == n and g['counters']['scatter'] == 0 and g['counters']['photons'] == n


def test_empty_launch_and_tiny_counts(solver):
    sc = slab_scene(tau=1.0, albedo=0.1)
    solver.load_scene(sc); solver.set_counting(True); solver.reset()
    solver.run(0, seed=1); solver.sync()
    assert solver.counters()['photons'] == 0
    for n in (1, 63, 65, 257):
        solver.reset(); solver.run(n, seed=1); solver.sync()
        assert solver.counters()['photons'] == n
    # the tail of a launch (fewer photons than lanes, lanes running dry at different times) in every kind of build, instrumented
    # and plain: the passes that serve only part of the work must never leave a lane waiting for ever
    for kw in (dict(), dict(vza=(0.0, 50.0), vaa=(0.0, 30.0)), dict(target='flux'), dict(solver=SOLVER_P3D, vza=(0.0, 50.0), vaa=(0.0, 30.0))):
        sc = les_scene(nx=6, ny=5, nz3=50, **kw)
        if 'vza' in kw:
            sc.target = TARGET_FLUX | TARGET_RADIANCE
        for counting in (True, False):
            solver.load_scene(sc); solver.set_counting(counting)
            for n in (1, 2, 64, 100, 300, 5000):
                solver.reset(); solver.run(n, seed=9); solver.sync()
                assert solver.counters()['photons'] == n, (kw, counting, n)


def test_beer_law_direct_beam(solver):
    sza = 40.0; mu0 = np.cos(np.deg2rad(sza)); tau_abs = 0.7
    sc = slab_scene(tau=0.0, abs_tau=tau_abs, albedo=0.0, sza=sza, nz=5, target=TARGET_FLUX)
    n = 2000000
    g = gpu_run(solver, sc, n)
    z = sc.zgrd
    want = mu0*np.exp(-tau_abs*(z[-1]-z)/z[-1]/mu0)
    sigma = np.sqrt(want/mu0*(1.0-want/mu0)/n)*mu0
    assert np.all(np.abs(g['flux'][0, :, 0, 0]-want) < 4.5*sigma + 2e-5)
    assert np.all(g['flux'][2] == 0.0)


@pytest.mark.parametrize('variant', ['column', 'marched', 'flux', 'flux+marched', 'lds-table', 'lds-table-general', 'global-tables', 'p3d', 'p3d-flux+marched',
                                     'aerosol', 'aerosol+marched', 'mie', 'mie+marched', 'mie-fluxonly', 'mie-aerosol', 'slab-table', 'slab-table-fluxonly'])
def test_single_histories_follow_the_oracle(solver, oracle, variant):
    """K7: one photon id at a time, every compile-time specialisation of the transport kernel.  The HIP kernel and the
    oracle consume the same Philox stream, so a history has the same events in both unless float32 rounding flips a
    decision somewhere along it: require identical event counts for at least 85 % of the histories (a dropped random
    number or a wrong state hand-over between the kernel's phases would leave almost none identical)."""
    kw = dict(nx=32, ny=32, nz3=50)
    variant0 = variant
    if variant.startswith('mie'):
        # round 5: tabulated phase functions in the lean loops -- four Mie-like tables of 498 angles, a real-valued table index per voxel
        # (its fraction mixes neighbouring tables); with marched views (the ray kernel's look-ups), as a flux job, with a second 3-D constituent
        kw.update(mie=True, aerosol=(variant == 'mie-aerosol'))
        if variant == 'mie+marched':
            kw.update(vza=(0.0, 40.0), vaa=(0.0, 120.0))
        if variant == 'mie-fluxonly':
            kw.update(target='flux')
        variant = 'marched' if variant == 'mie+marched' else 'column'
    if variant.startswith('p3d'):
        kw.update(solver=SOLVER_P3D, sza=55.0)
        variant = variant[4:] or 'column'
    if variant.startswith('aerosol'):
        kw.update(aerosol=True)          # a second 3-D constituent (config 3): the lean builds read it from `csca`
        variant = variant[8:] or 'column'
    if variant in ('marched', 'flux+marched'):
        kw.update(vza=(0.0, 40.0), vaa=(0.0, 120.0))
    sc = les_scene(**kw)
    if variant.startswith('slab-table'):
        # func_ref_vs_cot's scene (er3t/rtm/mca/util.py:130-160): no voxels, Rayleigh + gas and a cloud slab as TWO 1-D constituents, the
        # slab's selector a table index -- through the lean loops since round 5
        pha = pha_hg_synth()
        sc = slab_scene(tau=0.3, omega=1.0, apf=-1.0, albedo=0.1, nz=6, ztop=6000.0, abs_tau=0.05, target=(TARGET_FLUX if variant.endswith('fluxonly') else TARGET_RADIANCE),
                        ang=pha.data['ang']['data'].astype(np.float32), pha=np.ascontiguousarray(pha.data['pha']['data'].T, dtype=np.float32))
        e2 = np.zeros((1, 6)); e2[0, 1:3] = 8.0/2000.0
        sc.ext1d = np.vstack([sc.ext1d, e2]); sc.omg1d = np.vstack([sc.omg1d, np.ones((1, 6))]); sc.apf1d = np.vstack([sc.apf1d, np.full((1, 6), 2.0)])
        variant = 'column'
    if variant.startswith('flux'):
        sc.target = TARGET_FLUX | TARGET_RADIANCE
    if variant in ('lds-table', 'lds-table-general', 'global-tables'):
        # cloud droplets scatter by table 2 of three HG tables: one table in use -> staged in LDS by the kernel;
        # with the selector 1.5 mixed in, tables 1..2 are in use -> too large for the LDS budget, read from global memory
        pha = pha_hg_synth()
        sc.ang = pha.data['ang']['data'].astype(np.float32)
        sc.pha = np.ascontiguousarray(pha.data['pha']['data'].T, dtype=np.float32)
        sc.apfp[0][sc.extp[0] > 0] = 2.0
        if variant == 'global-tables':
            sc.apfp[0][:, ::2, :][sc.extp[0][:, ::2, :] > 0] = 1.5
    keys = ('scatter', 'surface', 'roulette', 'killed', 'escaped', 'absorbed')
    solver.bind(None, None, None)
    solver.set_kernel(general=(variant0 == 'lds-table-general'))
    try:
        solver.load_scene(sc, column_le=(variant not in ('marched', 'flux+marched')))
        solver.set_counting(True)
        same, nph = 0, 96
        for i in range(nph):
            solver.reset(); solver.run(1, seed=5, offset=i); solver.sync()
            g = solver.counters()
            o = oracle.run(sc, 1, seed=5, offset=i, nthreads=1)['counters']
            assert g['photons'] == 1 and g['killed']+g['escaped']+g['absorbed'] == 1
            same += all(g[k] == o[k] for k in keys)
        name = solver.kernel_name()
    finally:
        solver.set_kernel()
    assert same >= 0.85*nph, (variant0, same, nph)
    # which build served: the lean loops' general-mixture builds (",2>") wherever the tables fit the LDS; the general kernel where they
    # do not, where flux and radiance are asked for together, and where the test asks for it
    # (round 6: ",3>" where the scene is the general mixture's common one -- ONE Rayleigh 1-D constituent, one 3-D constituent with tables -- and
    #  the column view serves: a build with that known at compile time; a second 3-D constituent, marched views, two 1-D constituents: ",2>")
    want = {'lds-table': 'k_transport_lean<1,0,0,3>', 'lds-table-general': 'k_transport<', 'global-tables': 'k_transport<', 'mie': 'k_transport_lean<1,0,0,3>',
            'mie+marched': 'k_transport_lean<1,0,2,2> + k_rays', 'mie-fluxonly': 'k_transport_flux<1,0,2>', 'mie-aerosol': 'k_transport_lean<1,0,0,2>',
            'slab-table': 'k_transport_lean<1,0,0,2>', 'slab-table-fluxonly': 'k_transport_flux<1,0,2>', 'column': 'k_transport_lean<1,0,0,0>',
            'aerosol': 'k_transport_lean<1,0,0,1>', 'flux': 'k_transport<', 'marched': 'k_transport_lean<1,0,2,0> + k_rays'}.get(variant0)
    if want is not None:
        assert name.startswith(want), (variant0, name)


@pytest.mark.parametrize('case', ['nadir_column', 'nadir_marched', 'three_views', 'ipa', 'p3d', 'le_roulette', 'up_looking', 'aerosol_views'])
def test_radiance_parity_les(solver, oracle, nthreads, case):
    kw = dict(nx=16, ny=16, nz3=50)
    column_le = True
    if case == 'nadir_marched':
        column_le = False
    if case == 'three_views':
        kw.update(vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
    if case == 'aerosol_views':
        kw.update(vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0), aerosol=True)   # two 3-D constituents through the lean builds + k_rays
    if case == 'ipa':
        kw.update(solver=SOLVER_IPA, vza=(0.0, 26.1), vaa=(0.0, 180.0))
    if case == 'p3d':
        kw.update(solver=SOLVER_P3D, sza=60.0, vza=(0.0, 26.1), vaa=(0.0, 180.0))
    if case == 'le_roulette':
        kw.update(vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
    if case == 'up_looking':
        # sensors looking up: zenith and slant from the ground, slant from inside the cloud layer, plus a nadir satellite view
        kw.update(vza=(180.0, 150.0, 130.0, 0.0), vaa=(0.0, 60.0, 250.0, 0.0), surface_albedo=0.3)
    sc = les_scene(**kw)
    if case == 'up_looking':
        sc.view_zloc = [0.0, 0.0, 900.0, 705000.0]
    if case == 'le_roulette':
        sc.le_tau1 = 2.0        # same hashed decisions on both sides: the rays that survive are the same rays
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 7, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=7, column_le=column_le)
    check_counters(g['counters'], o['counters'])
    if case == 'le_roulette':
        sc0 = les_scene(**kw); sc0.le_tau1 = 0.0
        plain = gpu_run(solver, sc0, nb*nper, seed=7, column_le=column_le)
        assert g['counters']['le_steps'] < 0.6*plain['counters']['le_steps']             # it does shorten the marching
        assert np.array_equal(g['rad'][0], plain['rad'][0]) or np.allclose(g['rad'][0], plain['rad'][0], rtol=2e-3)   # nadir untouched
    check_radiance(g, o, zstd_max={'ipa': 0.05, 'p3d': 0.3}.get(case, 0.8))
    if case == 'up_looking':
        assert np.all(g['rad'][:3].mean(axis=(1, 2)) > 0.01) and g['counters']['le_column'] > 0      # diffuse sky light is there; nadir view by table
    if not column_le:
        assert g['counters']['le_column'] == 0
    elif case == 'nadir_column':
        assert g['counters']['le_column'] == g['counters']['le_rays'] and g['counters']['le_steps'] == 0


def test_flux_parity_les(solver, oracle, nthreads):
    sc = les_scene(nx=16, ny=16, nz3=50, target='flux', aerosol=True)
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 7, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=7)
    check_counters(g['counters'], o['counters'])
    gm = g['flux'].mean(axis=(2, 3)); om = o['flux'].mean(axis=(2, 3))
    se = o['flux_mean_se']
    assert np.all(np.abs(gm-om) < 2.0*np.sqrt(2.0)*se + 2e-4), np.abs(gm-om).max()       # north_star: within 2 sigma
    sep = np.maximum(o['flux_se'], 1e-9)
    z = (g['flux']-o['flux'])/(np.sqrt(2.0)*sep)
    z = z[np.isfinite(z) & (o['flux'] > 0) & (o['flux_se'] > 0)]
    assert np.mean(np.abs(z) > 3.0) < 0.05 and abs(z.mean()) < 0.5


@pytest.mark.parametrize('mode', [0, 1, 2], ids=['3d', 'partial3d', 'ipa'])
def test_flux_tally_routes_agree(solver, mode):
    """One flux job, four ways for its level crossings to reach the tally: records sorted and summed after the launch (the lean flux
    loop's default), the same with lists far too small (what does not fit goes out as atomics: nothing may be lost), an atomic per
    crossing, and the general kernel.  The lean loop's three agree to the order of float64 sums (same histories, same tallies);
    the general kernel is another float32 program (positions tracked instead of rebuilt from face parameters, crossings inside a
    run of uniform layers placed layer by layer): level sums agree to a few parts in 10^3 at 3e5 photons"""
    sc = les_scene(nx=16, ny=16, nz3=50, target='flux', aerosol=True)
    sc.solver = mode        # (3-D, partial 3-D: the direct beam in 3-D and everything after it in its column, independent columns)
    n = 300000
    try:
        ref = gpu_run(solver, sc, n, seed=11)
        assert solver.kernel_name().startswith('k_transport_flux<') and 'k_tl_scatter' in solver.kernel_name()
        assert ref['counters']['le_column'] == 0      # (instrumented flux loop: tallies that went out as atomics)
        solver.set_tuning(tlcap_log2=17)
        small = gpu_run(solver, sc, n, seed=11)
        assert 'k_tl_scatter' in solver.kernel_name() and 0 < small['counters']['le_column'] < small['counters']['flux_tally']
        solver.set_tuning(tlcap_log2=31, tally_lists=0)
        atom = gpu_run(solver, sc, n, seed=11)
        assert solver.kernel_name().startswith('k_transport_flux<') and 'k_tl_scatter' not in solver.kernel_name()
        solver.set_tuning(tally_lists=1)
        solver.set_kernel(general=True)
        gen = gpu_run(solver, sc, n, seed=11)
        assert solver.kernel_name().startswith('k_transport<')
    finally:
        solver.set_tuning(tlcap_log2=31, tally_lists=1)
        solver.set_kernel()
    for other in (small, atom):
        assert other['counters']['flux_tally'] == ref['counters']['flux_tally']
        assert np.allclose(other['flux'], ref['flux'], rtol=1e-6, atol=1e-9)
    # (float32 histories of two differently written loops part ways now and then: a few tallies in a million)
    assert abs(gen['counters']['flux_tally'] - ref['counters']['flux_tally']) < 1e-4*ref['counters']['flux_tally']
    assert np.allclose(gen['flux'].sum(axis=(2, 3)), ref['flux'].sum(axis=(2, 3)), rtol=5e-3, atol=1e-3)


def test_heating_rates_parity_and_energy_budget(solver, oracle, nthreads):
    """Flx_mhrt = 1: the power absorbed per cell (gas, absorbing aerosol) against the oracle's on the same photon ids (layer means as
    two independent runs would agree, column totals to 0.3 %), and the exact budget of the
    known answer K17 on the HIP path itself: without roulette what enters at the top leaves through the top, into the surface, or
    into the cells (float32 weights: to 2e-5 of the beam)"""
    sc = les_scene(nx=16, ny=16, nz3=50, target='flux', aerosol=True)
    sc.target = TARGET_FLUX | TARGET_HEAT
    sc.abs1d = sc.abs1d*30.0
    nb, nper = 16, 20000
    heat = np.stack([oracle.run(sc, nper, seed=7, offset=b*nper, nthreads=nthreads)['heat'] for b in range(nb)])
    g = gpu_run(solver, sc, nb*nper, seed=7)
    assert solver.kernel_name().startswith('k_transport_flux<') and g['heat'].shape == (sc.nz, 16, 16)
    om, gm = heat.mean(axis=(0, 2, 3)), g['heat'].mean(axis=(1, 2))
    se = heat.mean(axis=(2, 3)).std(axis=0, ddof=1)/np.sqrt(nb)
    # (a collision with the gas is a rare event in a clear layer: once rounding has parted two histories their collision sites are
    #  independent, so the 68 layer means scatter like those of two independent runs -- measured: z between -1.6 and +3.1, no trend)
    z = (gm-om)/(np.sqrt(2.0)*se)
    assert om.max() > 0.0 and np.all(np.abs(z) < 4.0) and abs(z.mean()) < 0.5 and z.std() < 1.4, (z.mean(), z.std(), np.abs(z).max())
    dz = np.diff(sc.zgrd)
    assert abs((gm*dz).sum()/(om*dz).sum()-1.0) < 3e-3          # ... while the column totals agree far inside the noise (measured: 2e-4)
    # the budget: without roulette every photon's weight ends up above the top, in the surface or in the cells, history by history.
    # (Domain totals: the HIP path does not tally the direct beam above the 3-D region but adds its known value, mu0 in every
    #  column, when the fluxes are read, so the photons that happened to start in ONE column are not in its figures.)
    sc.wmin = 0.0; sc.sfc_param[0] = 0.3
    b = gpu_run(solver, sc, 400000, seed=3)
    f = b['flux'].mean(axis=(2, 3))
    absorbed = (b['heat'].mean(axis=(1, 2))*dz).sum()
    budget = (f[1, -1]-f[2, -1]) - (f[1, 0]-f[2, 0])
    assert np.isclose(f[1, -1], sc.mu0, rtol=1e-6) and absorbed > 0.02*sc.mu0
    assert abs(absorbed-budget) < 2e-5*sc.mu0, (absorbed, budget)
    # a job without the heating target has no such tally
    sc.target = TARGET_FLUX
    gpu_run(solver, sc, 1000, seed=3)
    with pytest.raises(OSError, match='heating'):
        solver.heating(1000)


def test_heating_budget_closes_under_a_wide_source_cone_and_killing_collisions(solver):
    """The corner ADVICE r3 named: a wide source cone (the top level is a Monte-Carlo tally again: a launch makes a tally of its own)
    over a strongly absorbing atmosphere with weights allowed to fall to zero (Pho_wmin = 0: a collision in a purely absorbing
    layer KILLS the photon, and the lane may take a new photon in the same pass).  The heat record of the killing collision was
    pending when the launch tally of the lane's next photon overwrote it; it is flushed right after the collision block now.
    Checked by the exact budget (K17): what enters at the top leaves through the top, into the surface or into the cells."""
    sc = les_scene(nx=8, ny=8, nz3=50, target='flux', sza=50.0)
    sc.src_qmax = 40.0
    sc.target = TARGET_FLUX | TARGET_HEAT
    sc.abs1d = sc.abs1d*30.0 + 4.0e-4            # optical thickness ~8 of pure absorption over 20 km
    sc.ext1d = sc.ext1d*0.0                       # no Rayleigh scattering: above and below the clouds every collision is an absorption
    sc.wmin = 0.0; sc.sfc_param[0] = 0.3
    for lists in (1, 0):
        try:
            solver.set_tuning(tally_lists=lists)
            b = gpu_run(solver, sc, 400000, seed=3)
        finally:
            solver.set_tuning(tally_lists=1)
        assert solver.kernel_name().startswith('k_transport_flux<'), solver.kernel_name()
        f = b['flux'].mean(axis=(2, 3))
        dz = np.diff(sc.zgrd)
        absorbed = (b['heat'].mean(axis=(1, 2))*dz).sum()
        budget = (f[1, -1]-f[2, -1]) - (f[1, 0]-f[2, 0])
        assert b['flux'][0, -1].std() > 0.0 and absorbed > 0.5*f[1, -1]         # the top level is counted; most of the light is absorbed
        assert abs(absorbed-budget) < 2e-5*f[1, -1], (lists, absorbed, budget)


def test_partial_3d_flux_parity_and_direct_beam(solver, oracle, nthreads):
    """solver 1: parity with the oracle, and the defining property -- the direct beam is the 3-D solver's (same photon
    ids, same direct histories; float32 atomics only differ in summation order), the diffuse field is not"""
    kw = dict(nx=16, ny=16, nz3=50, target='flux', sza=60.0, saa=30.0)
    sc = les_scene(solver=SOLVER_P3D, **kw)
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 7, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=7)
    check_counters(g['counters'], o['counters'])
    gm = g['flux'].mean(axis=(2, 3)); om = o['flux'].mean(axis=(2, 3))
    assert np.all(np.abs(gm-om) < 2.0*np.sqrt(2.0)*o['flux_mean_se'] + 2e-4), np.abs(gm-om).max()
    sep = np.maximum(o['flux_se'], 1e-9)
    z = (g['flux']-o['flux'])/(np.sqrt(2.0)*sep)
    z = z[np.isfinite(z) & (o['flux'] > 0) & (o['flux_se'] > 0)]
    assert np.mean(np.abs(z) > 3.0) < 0.05 and abs(z.mean()) < 0.5
    g3 = gpu_run(solver, les_scene(solver=SOLVER_3D, **kw), nb*nper, seed=7)
    assert np.allclose(g['flux'][0], g3['flux'][0], rtol=1e-4, atol=1e-6)
    assert not np.allclose(g['flux'][2], g3['flux'][2], rtol=1e-2, atol=1e-4)


def test_lsrt_aerosol_slant_parity(solver, oracle, nthreads):
    sc = les_scene(nx=16, ny=16, nz3=50, lsrt=True, aerosol=True, vza=(30.0, 60.0), vaa=(100.0, 280.0))
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 11, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=11)
    check_counters(g['counters'], o['counters'])
    check_radiance(g, o)
    assert abs(g['counters']['absorbed']-o['counters']['absorbed']) <= 0.1*o['counters']['absorbed'] + 30


def test_tabulated_phase_parity(solver, oracle, nthreads):
    # cloud droplets scatter by table 2 of three HG tables (apf = 2), mixed tables in the aerosol (apf = 1.4)
    pha = pha_hg_synth()
    sc = les_scene(nx=12, ny=12, nz3=50, aerosol=True, vza=(0.0, 40.0), vaa=(0.0, 60.0))
    sc.ang = pha.data['ang']['data'].astype(np.float32)
    sc.pha = np.ascontiguousarray(pha.data['pha']['data'].T, dtype=np.float32)
    sc.apfp[0][sc.extp[0] > 0] = 2.0
    sc.apfp[1][...] = 1.4
    nb, nper = 16, 15000
    o = oracle_batches(oracle, sc, nb, nper, 5, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=5)
    check_counters(g['counters'], o['counters'])
    check_radiance(g, o)
    # and the table must reproduce the analytic HG it tabulates (g = 0.85): domain means within 1 %
    sa = les_scene(nx=12, ny=12, nz3=50, vza=(0.0,))
    st = les_scene(nx=12, ny=12, nz3=50, vza=(0.0,))
    st.ang = sc.ang; st.pha = sc.pha; st.apfp[0][st.extp[0] > 0] = 2.0
    ga = gpu_run(solver, sa, 400000, seed=3); gt = gpu_run(solver, st, 400000, seed=3)
    assert abs(ga['rad'].mean()-gt['rad'].mean()) < 0.01*ga['rad'].mean()


def test_one_dimensional_clear_sky_flux(solver, oracle, nthreads):
    # BASELINE config 1 shape: 1-D clear sky, flux, Rayleigh + gas absorption (nx = ny = 1, no 3-D region)
    from er3t_amd.synth import atm_synth, abs_synth, rayleigh_tau
    from er3t_amd.scene import Scene
    atm = atm_synth(np.linspace(0.0, 20.0, 21)); ab = abs_synth(650.0, atm)
    dz = atm.lay['thickness']['data']*1000.0; p = atm.lev['pressure']['data']
    sc = Scene(zgrd=atm.lev['altitude']['data']*1000.0, ext1d=(rayleigh_tau(0.65, p[:-1], p[1:])/dz)[None],
               omg1d=np.ones((1, 20)), apf1d=-np.ones((1, 20)), abs1d=ab.coef['abso_coef']['data'][:, 15]/dz*50.0,
               sfc_param=[0.03, 0, 0, 0, 0], src_the=150.0, src_phi=270.0, target=TARGET_FLUX)
    n = 400000
    g = gpu_run(solver, sc, n, seed=2)
    o = oracle.run(sc, n, seed=2, nthreads=nthreads)
    check_counters(g['counters'], o['counters'])
    assert np.allclose(g['flux'][:, :, 0, 0], o['flux'][:, :, 0, 0], atol=2.5e-3)
    assert np.isclose(g['flux'][1, -1, 0, 0], np.cos(np.deg2rad(30.0)), rtol=2e-5)


# ---------------------------------------------------------------------------------------------
# properties at BASELINE.json sizes (config 2 grid 128x128x50, config 4 grid 480x480x100)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def scene_c2():
    return les_scene()


def test_full_size_id_ranges_add_up_and_column_equals_marched(solver, scene_c2):
    sc = scene_c2
    n = 2000000
    g_all = gpu_run(solver, sc, n, seed=1234, counting=True)
    # two launches over the two halves of the id range accumulate to the same tally (what photon sharding does)
    solver.reset()
    solver.run(n//2, seed=1234, offset=0); solver.run(n-n//2, seed=1234, offset=n//2); solver.sync()
    two = solver.radiance(n).astype(np.float64)
    c2 = solver.counters()
    phys = ('photons', 'steps', 'steps3d', 'scatter', 'surface', 'le_rays', 'le_steps', 'le_column', 'roulette', 'killed', 'escaped', 'absorbed')
    assert all(c2[k] == g_all['counters'][k] for k in phys)   # integer event counts: identical histories
    assert np.allclose(two, g_all['rad'], rtol=2e-4, atol=1e-6)   # float32 atomics: order of summation only
    # the column optical-depth table answers what marching the vertical ray gives.  The two are different
    # compile-time specialisations of the kernel (floating-point contraction may differ), so a few histories
    # in 10^4 part ways: counts agree to 0.3 %, the image to Monte-Carlo noise far below the per-pixel sigma
    g_m = gpu_run(solver, sc, n, seed=1234, column_le=False)
    for k in ('steps', 'steps3d', 'scatter', 'surface', 'le_rays', 'killed'):
        assert abs(g_m['counters'][k]-g_all['counters'][k]) <= 3e-3*g_all['counters'][k], k
    assert abs(g_m['counters']['roulette']-g_all['counters']['roulette']) <= 1e-2*g_all['counters']['roulette']
    assert g_m['counters']['le_steps'] > 0 and g_all['counters']['le_steps'] == 0
    assert abs(g_m['rad'].mean()/g_all['rad'].mean()-1.0) < 2e-3
    rel = np.abs(g_m['rad']-g_all['rad'])/g_all['rad'].mean()
    assert np.median(rel) < 0.02 and np.corrcoef(g_m['rad'].ravel(), g_all['rad'].ravel())[0, 1] > 0.98


def test_full_size_conservative_net_flux_is_constant(solver):
    # config-2 grid without any absorption: the domain-mean net flux is the same at every level (K5)
    sc = les_scene(target='flux', surface_albedo=0.2)
    sc.abs1d[:] = 0.0
    n = 4000000
    g = gpu_run(solver, sc, n, seed=99)
    f = g['flux'].mean(axis=(2, 3))
    net = f[1]-f[2]
    mu0 = sc.mu0
    assert np.isclose(f[1, -1], mu0, rtol=2e-5)
    # every level sees the same photons minus those still in flight: standard error of the net ~ sqrt(1/n)
    assert np.all(np.abs(net-net[-1]) < 5.0*np.sqrt(2.0/n) + 2e-4), np.abs(net-net[-1]).max()
    assert abs(f[2, 0]-0.2*f[1, 0]) < 5.0*np.sqrt(1.0/n) + 1e-4


def test_config4_grid_runs_and_is_sane(solver):
    # 480x480x100 voxels: build, transport, tally; radiance positive everywhere bright, counters consistent
    sc = les_scene(nx=480, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
    n = 3000000
    g = gpu_run(solver, sc, n, seed=1234)
    c = g['counters']
    assert c['photons'] == n and c['killed']+c['escaped']+c['absorbed'] == n
    assert c['le_rays'] == c['scatter']+c['surface'] and c['le_column'] == c['le_rays']
    rad = g['rad'][0]
    assert np.all(np.isfinite(rad)) and rad.min() >= 0.0
    # domain-mean reflectance between clear-sky and a thick cloud deck
    refl = np.pi*rad.mean()/sc.mu0
    assert 0.15 < refl < 0.6, refl
    # brighter where the cloud is thicker: correlation between column optical depth and radiance
    cot = (sc.extp[0]*40.0).sum(axis=0)
    assert np.corrcoef(cot.ravel(), rad.ravel())[0, 1] > 0.5


# ---------------------------------------------------------------------------------------------
# unusual but legal inputs (the reference's tests have no counterpart; these walk the corners of the C-ABI)
# ---------------------------------------------------------------------------------------------
def _edge_scene(case):
    rng = np.random.default_rng(11)
    if case == 'single_column_3d':
        # nx = ny = 1 with a 3-D region: every horizontal face wraps onto the same column
        sc = slab_scene(tau=3.0, apf=0.7, albedo=0.2, nx=1, ny=1, nz3=3, nz=5, vza=(0.0, 50.0), vaa=(0.0, 10.0), dx=500.0, dy=500.0)
    elif case == 'four_constituents':
        # the maximum number of 1-D and 3-D scattering constituents, mixed phase-function kinds
        nz, nz3, nx, ny = 6, 4, 5, 4
        zgrd = np.linspace(0.0, 3000.0, nz+1)
        ext1d = np.stack([np.full(nz, 2e-4), np.full(nz, 1e-4), np.full(nz, 5e-5), np.full(nz, 3e-5)])
        omg1d = np.stack([np.ones(nz), np.full(nz, 0.9), np.full(nz, 0.5), np.ones(nz)])
        apf1d = np.stack([-np.ones(nz), np.full(nz, 0.6), np.full(nz, -2.0), np.full(nz, 0.2)])
        extp = rng.uniform(0.0, 2e-3, (4, nz3, ny, nx)).astype(np.float32); extp[:, :, 0, 0] = 0.0
        omgp = rng.uniform(0.5, 1.0, (4, nz3, ny, nx)).astype(np.float32)
        apfp = np.stack([np.full((nz3, ny, nx), v) for v in (0.85, -1.0, 0.3, -2.0)]).astype(np.float32)
        sc = Scene(zgrd=zgrd, ext1d=ext1d, omg1d=omg1d, apf1d=apf1d, abs1d=np.full(nz, 2e-5), nx=nx, ny=ny, dx=300.0, dy=200.0,
                   nz3=nz3, iz3l=2, abst=rng.uniform(0, 1e-5, (nz3, ny, nx)).astype(np.float32), extp=extp, omgp=omgp, apfp=apfp,
                   sfc_param=[0.1, 0, 0, 0, 0], src_the=140.0, src_phi=33.0, src_qmax=0.5,
                   view_the=[180.0, 150.0], view_phi=[0.0, 200.0], view_zloc=[1e6, 1e6], nxr=nx, nyr=ny, target=TARGET_RADIANCE | TARGET_FLUX)
    elif case == 'aircraft_and_zref':
        # sensors inside the atmosphere (one vertical: the column table must NOT be used for it), pixels registered at
        # cloud-top height, coarser radiance grid than the atmosphere grid, no solar cone
        sc = les_scene(nx=12, ny=12, nz3=50, vza=(0.0, 35.0, 0.0), vaa=(0.0, 300.0, 0.0))
        sc.view_zloc = [1500.0, 5000.0, 705000.0]
        sc.zref = 1400.0
        sc.nxr, sc.nyr = 6, 4
        sc.src_qmax = 0.0
    elif case == 'wide_source_cone':
        # a source cone far wider than the solar disc: the direct beam above the clouds is no longer exp(-tau/mu0), so the
        # HIP path must tally it level by level like the oracle instead of adding it analytically
        sc = les_scene(nx=8, ny=8, nz3=50, target='flux', sza=50.0)
        sc.src_qmax = 40.0
    elif case == 'sixteen_views':
        sc = les_scene(nx=10, ny=10, nz3=50, vza=np.linspace(0.0, 75.0, 16), vaa=np.linspace(0.0, 337.5, 16))
    else:
        raise ValueError(case)
    return sc


@pytest.mark.parametrize('case', ['single_column_3d', 'four_constituents', 'aircraft_and_zref', 'sixteen_views', 'wide_source_cone'])
def test_edge_inputs_against_the_oracle(solver, oracle, nthreads, case):
    sc = _edge_scene(case)
    nb, nper = 12, 15000
    o = oracle_batches(oracle, sc, nb, nper, 3, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=3)
    check_counters(g['counters'], o['counters'])
    if sc.target & TARGET_RADIANCE:
        for iv in range(sc.nview):
            gm, om = g['rad'][iv].mean(), o['rad'][iv].mean()
            assert abs(gm-om) < 2.0*np.sqrt(2.0)*o['rad_mean_se'][iv] + 2e-4*om, (case, iv, gm, om)
    if sc.target & TARGET_FLUX:
        gm = g['flux'].mean(axis=(2, 3)); om = o['flux'].mean(axis=(2, 3))
        # north_star's 2 sigma.  Twelve batches know their own standard error to 20 %, and where the HIP path adds the direct beam
        # analytically the difference is the oracle's noise alone, unpaired: the downward planes (the direct beam's photons carry
        # weight 1) get the binomial standard error mu0 sqrt(p (1 - p) / N) as a floor under the batch estimate
        se = o['flux_mean_se'].copy()
        p_ = np.clip(om[:2]/sc.mu0, 0.0, 1.0)
        se[:2] = np.maximum(se[:2], sc.mu0*np.sqrt(p_*(1.0-p_)/(nb*nper)))
        assert np.all(np.abs(gm-om) < 2.0*np.sqrt(2.0)*se + 3e-4), np.abs(gm-om).max()
    if case == 'wide_source_cone':
        assert g['counters']['flux_tally'] > 0.98*o['counters']['flux_tally']             # every crossing tallied
        assert g['flux'][0, -1].std() > 0.0                                                # the top level is a Monte-Carlo count again
    if case == 'aircraft_and_zref':
        # the vertical view from 1.5 km sees only what lies below it: darker than the same view from orbit
        assert g['rad'][0].mean() < g['rad'][2].mean()
        assert g['counters']['le_column'] > 0 and g['counters']['le_steps'] > 0


# ---------------------------------------------------------------------------------------------
# randomised corners: many tiny scenes with extreme geometry -- every history must end, counts must add up, and the
# domain-mean results must follow the oracle.  (A photon that never ends would hang the launch: the pytest timeout is the net.)
# ---------------------------------------------------------------------------------------------
def _random_scene(rng):
    nz = int(rng.integers(1, 8))
    nz3 = int(rng.integers(0, nz+1))
    iz3l = int(rng.integers(1, nz-nz3+2)) if nz3 > 0 else 1
    nx, ny = (int(rng.integers(1, 7)), int(rng.integers(1, 7))) if nz3 > 0 else (1, 1)
    dz = rng.choice([20.0, 200.0, 2000.0])
    zgrd = np.concatenate([[0.0], np.cumsum(rng.uniform(0.5, 1.5, nz)*dz)])
    np1d = int(rng.integers(1, 3))
    ext1d = rng.choice([0.0, 1e-5, 1e-3], size=(np1d, nz))*rng.uniform(0.5, 2.0, (np1d, nz))
    omg1d = rng.choice([0.0, 0.9, 1.0], size=(np1d, nz))
    apf1d = rng.choice([-2.0, -1.0, 0.0, 0.85, -0.4], size=(np1d, nz))
    kw = dict(zgrd=zgrd, ext1d=ext1d, omg1d=omg1d, apf1d=apf1d, abs1d=rng.choice([0.0, 1e-5], size=nz), nx=nx, ny=ny,
              dx=float(rng.choice([50.0, 500.0])), dy=float(rng.choice([50.0, 700.0])),
              sfc_mtype=1, sfc_param=[float(rng.choice([0.0, 0.3, 1.0])), 0, 0, 0, 0],
              src_the=180.0-float(rng.choice([0.0, 30.0, 75.0, 89.0])), src_phi=float(rng.choice([0.0, 90.0, 213.0])),
              src_qmax=float(rng.choice([0.0, 0.533133, 20.0])), solver=int(rng.choice([0, 1, 2])))
    if nz3 > 0:
        np3d = int(rng.integers(1, 3))
        extp = (rng.choice([0.0, 1e-4, 3e-2], size=(np3d, nz3, ny, nx))*rng.uniform(0.5, 2.0, (np3d, nz3, ny, nx))).astype(np.float32)
        if rng.random() < 0.3:
            extp[:, int(rng.integers(0, nz3))] = extp[:, 0, :1, :1]           # a horizontally uniform layer inside the 3-D region
        kw.update(nz3=nz3, iz3l=iz3l, extp=extp, omgp=rng.choice([0.0, 0.95, 1.0], size=extp.shape).astype(np.float32),
                  apfp=rng.choice([-2.0, -1.0, 0.7, 0.9], size=extp.shape).astype(np.float32),
                  abst=rng.choice([0.0, 1e-5], size=extp.shape[1:]).astype(np.float32))
    target = int(rng.choice([TARGET_FLUX, TARGET_RADIANCE, TARGET_FLUX | TARGET_RADIANCE]))
    if target & TARGET_RADIANCE:
        nv = int(rng.integers(1, 4))
        vza = rng.choice([0.0, 20.0, 60.0, 85.0, 180.0, 150.0, 95.0], size=nv)
        kw.update(view_the=list(180.0-vza), view_phi=list(rng.choice([0.0, 45.0, 180.0, 270.0], size=nv)),
                  view_zloc=list(rng.choice([705000.0, float(zgrd[-1]), float(0.5*(zgrd[-1]+zgrd[-2])), 0.0], size=nv)),
                  nxr=int(rng.choice([nx, 1, 2*nx])), nyr=int(rng.choice([ny, 1])), zref=float(rng.choice([0.0, float(zgrd[1])])))
    sc = Scene(target=target, **kw)
    sc.le_tau1 = float(rng.choice([0.0, 2.0]))
    return sc


@pytest.mark.timeout(300)
def test_random_corner_scenes_end_and_follow_the_oracle(solver, oracle, nthreads):
    rng = np.random.default_rng(20251003)
    n = 40000
    for i in range(60):
        sc = _random_scene(rng)
        g = gpu_run(solver, sc, n, seed=100+i)
        c = g['counters']
        assert c['photons'] == n and c['killed']+c['escaped']+c['absorbed'] == n, (i, c)
        o = oracle.run(sc, n, seed=100+i, nthreads=nthreads)
        oc = o['counters']
        for k in ('scatter', 'surface', 'escaped'):
            assert abs(c[k]-oc[k]) <= 0.03*max(oc[k], 1) + 60, (i, k, c[k], oc[k])
        if sc.target & TARGET_RADIANCE:
            assert np.all(np.isfinite(g['rad']))
            gm, om = g['rad'].mean(axis=(1, 2)), o['rad'].mean(axis=(1, 2))
            assert np.all(np.abs(gm-om) <= 0.08*np.abs(om) + 2e-4), (i, gm, om)
        if sc.target & TARGET_FLUX:
            assert np.all(np.isfinite(g['flux']))
            gm, om = g['flux'].mean(axis=(2, 3)), o['flux'].mean(axis=(2, 3))
            assert np.all(np.abs(gm-om) <= 0.03*np.abs(om) + 5e-3), (i, np.abs(gm-om).max())


def test_gpu_against_chandrasekhar_semi_infinite_isotropic(solver):
    """the HIP path against an exact multiple-scattering result, no oracle in between: diffuse reflection by a semi-infinite
    isotropically scattering atmosphere, I(mu) = omega/(4 pi) mu0/(mu+mu0) H(mu) H(mu0) (tests/test_oracle_kat.py K12)"""
    from tests.test_oracle_kat import _chandrasekhar_h
    omega, sza = 0.9, 40.0
    mu0 = np.cos(np.deg2rad(sza))
    vza = np.array([0.0, 35.0, 65.0])
    sc = slab_scene(tau=40.0, omega=omega, apf=-2.0, albedo=0.0, sza=sza, nz=8, vza=vza, vaa=(0.0, 90.0, 200.0), target=TARGET_RADIANCE)
    for tau1 in (0.0, 2.0):                    # with and without the roulette on the marched rays
        sc.le_tau1 = tau1
        nb, nper = 8, 500000
        r = np.stack([gpu_run(solver, sc, nper, seed=31, offset=b*nper)['rad'][:, 0, 0] for b in range(nb)])
        mean, se = r.mean(axis=0), r.std(axis=0, ddof=1)/np.sqrt(nb)
        mu = np.cos(np.deg2rad(vza))
        want = omega/(4.0*np.pi)*mu0/(mu+mu0)*_chandrasekhar_h(omega, mu)*_chandrasekhar_h(omega, mu0)
        assert np.all(np.abs(mean-want) < 4.0*se + 1e-3*want), (tau1, mean, want, se)
    # plane albedo of the same atmosphere, 1 - H(mu0) sqrt(1 - omega): the upward flux at the top
    scf = slab_scene(tau=40.0, omega=omega, apf=-2.0, albedo=0.0, sza=sza, nz=8, target=TARGET_FLUX)
    fb = np.stack([gpu_run(solver, scf, 1000000, seed=5, offset=b*1000000)['flux'][:, -1, 0, 0] for b in range(8)])
    want_up = mu0*(1.0-_chandrasekhar_h(omega, mu0)[0]*np.sqrt(1.0-omega))
    assert abs(fb[:, 2].mean()-want_up) < 4.0*fb[:, 2].std(ddof=1)/np.sqrt(8) + 3e-4*want_up, (fb[:, 2].mean(), want_up)
    assert np.allclose(fb[:, 0], mu0, rtol=1e-6) and np.allclose(fb[:, 1], mu0, rtol=1e-3)      # what comes in at the top: the beam


@pytest.mark.parametrize('apf', [-1.0, 0.6])
def test_gpu_single_scattering_both_ways(solver, apf):
    """the HIP path against the analytic single-scattering limit, no oracle in between (tests/test_oracle_kat.py K4, K11):
    diffusely reflected light seen from above, diffusely transmitted light seen from the ground; one pixel, 8e6 photons
    (which a float32 tally could not hold)"""
    from oracle import oracle as orc     # only its phase-function evaluation, to build the expected value
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza)); tau = 0.004; omega = 0.5     # (second order: O(omega tau), more where P is small)
    sc = slab_scene(tau=tau, omega=omega, apf=apf, albedo=0.0, sza=sza, nz=3, vza=(0.0, 40.0, 180.0, 140.0), vaa=(0.0, 135.0, 0.0, 135.0),
                    target=TARGET_RADIANCE)
    sc.view_zloc = [705000.0, 705000.0, 0.0, 0.0]
    nb, nper = 8, 1000000
    r = np.stack([gpu_run(solver, sc, nper, seed=17, offset=b*nper)['rad'][:, 0, 0] for b in range(nb)])
    mean, se = r.mean(axis=0), r.std(axis=0, ddof=1)/np.sqrt(nb)
    sdir = np.array([np.sin(np.deg2rad(sc.src_the))*np.cos(np.deg2rad(sc.src_phi)),
                     np.sin(np.deg2rad(sc.src_the))*np.sin(np.deg2rad(sc.src_phi)), np.cos(np.deg2rad(sc.src_the))])
    for iv in range(4):
        t = np.deg2rad(sc.view_the[iv]); p = np.deg2rad(sc.view_phi[iv])
        v = -np.array([np.sin(t)*np.cos(p), np.sin(t)*np.sin(p), np.cos(t)])
        P = orc.phase_eval(apf, float(sdir @ v)); muv = abs(v[2])
        if v[2] > 0.0:
            want = omega*P/(4.0*np.pi)*mu0/(mu0+muv)*(1.0-np.exp(-tau*(1.0/mu0+1.0/muv)))
        else:
            want = omega*P/(4.0*np.pi)*mu0/(mu0-muv)*(np.exp(-tau/mu0)-np.exp(-tau/muv))
        # higher orders of scattering add O(omega*tau) relative
        assert abs(mean[iv]-want) < 4.0*se[iv] + 0.015*want, (apf, iv, mean[iv], want, se[iv])
    # one launch of all the photons gives what the eight batches give: the tally does not saturate
    g = gpu_run(solver, sc, nb*nper, seed=17)['rad'][:, 0, 0]
    assert np.allclose(g, mean, rtol=2e-4)


def test_gpu_geometry_shadow_and_parallax(solver):
    """the HIP path against exact geometry, no oracle in between (tests/test_oracle_kat.py K13): the shadow of one opaque voxel
    under a slant sun, and where a small cloud appears in slant views"""
    from tests.util import block_scene, block_expectations
    shadow, image = block_expectations()
    mu0 = np.cos(np.deg2rad(45.0))
    n = 48000000
    f = gpu_run(solver, block_scene('absorber'), n, seed=2)['flux'][0, 0]/mu0
    per_col = n/48.0
    assert np.all(np.abs(f[3]-shadow) < 5.0*np.sqrt(np.maximum(shadow, 1e-4)/per_col) + 2e-3), (f[3], shadow)
    assert np.all(np.abs(np.delete(f, 3, axis=0)-1.0) < 5.0/np.sqrt(per_col))
    for column_le in (True, False):
        r = gpu_run(solver, block_scene('scatterer'), 200000000, seed=2, column_le=column_le)['rad']
        for iv, want in ((0, None), (1, image[-1.0]), (2, image[1.0])):
            img = r[iv]/r[iv].sum()
            # (float32 positions: an event within rounding of a pixel edge may land in the neighbouring pixel -- one in 4e4 here)
            assert np.all(np.delete(img, [2, 3, 4], axis=0) == 0.0) and img[[2, 4]].sum() < 1e-4
            if want is None:
                assert img[3, 2] > 1.0 - 1e-4
            else:
                assert img[3][want == 0.0].sum() < 1e-4 and np.all(np.abs(img[3]-want) < 0.015), (iv, img[3], want)    # 4e4 events: sigma 0.0025


def test_gpu_ipa_columns_and_partial_3d_limits(solver):
    """the HIP path against itself, no oracle (tests/test_oracle_kat.py K8, K9): under the independent-pixel approximation a
    column is its own plane-parallel problem; with an exactly vertical beam the partial-3-D solver IS the independent-pixel
    approximation; its direct beam is the 3-D solver's"""
    import copy
    kw = dict(nx=4, ny=4, nz3=50, sza=40.0, saa=20.0, vza=(0.0, 30.0), vaa=(0.0, 100.0), surface_albedo=0.03)
    sc = les_scene(solver=SOLVER_IPA, **kw)
    sc.target = TARGET_FLUX | TARGET_RADIANCE
    cot = (sc.extp[0]*40.0).sum(axis=0)
    nb, nper = 8, 1600000
    full = [gpu_run(solver, sc, nper, seed=11, offset=b*nper) for b in range(nb)]
    rad = np.stack([r['rad'] for r in full]); fup = np.stack([r['flux'][2, -1] for r in full])
    for (iy, ix) in (np.unravel_index(np.argmin(cot), cot.shape), np.unravel_index(np.argmax(cot), cot.shape)):
        col = copy.copy(sc)
        col.nx = col.ny = col.nxr = col.nyr = 1
        for name in ('abst', 'extp', 'omgp', 'apfp'):
            setattr(col, name, np.ascontiguousarray(getattr(sc, name)[..., iy:iy+1, ix:ix+1]))
        one = [gpu_run(solver, col, nper//16, seed=12, offset=b*nper) for b in range(nb)]
        r1 = np.stack([r['rad'][:, 0, 0] for r in one]); f1 = np.stack([r['flux'][2, -1, 0, 0] for r in one])
        for iv in range(2):
            a, b = rad[:, iv, iy, ix], r1[:, iv]
            se = np.sqrt(a.var(ddof=1)/nb + b.var(ddof=1)/nb)
            assert abs(a.mean()-b.mean()) < 4.0*se + 5e-4*b.mean(), (iy, ix, iv, a.mean(), b.mean(), se)
        a, b = fup[:, iy, ix], f1
        assert abs(a.mean()-b.mean()) < 4.0*np.sqrt(a.var(ddof=1)/nb + b.var(ddof=1)/nb) + 5e-4*b.mean()
    # vertical beam without a cone: partial 3-D == IPA, history by history
    v = {}
    for sv in (SOLVER_P3D, SOLVER_IPA):
        s2 = les_scene(solver=sv, **dict(kw, sza=0.0)); s2.src_qmax = 0.0; s2.target = TARGET_FLUX | TARGET_RADIANCE
        v[sv] = gpu_run(solver, s2, 400000, seed=3)
    assert v[SOLVER_P3D]['counters']['scatter'] == v[SOLVER_IPA]['counters']['scatter']
    assert np.allclose(v[SOLVER_P3D]['rad'], v[SOLVER_IPA]['rad'], rtol=1e-9) and np.allclose(v[SOLVER_P3D]['flux'], v[SOLVER_IPA]['flux'], rtol=1e-9)


def test_gpu_lsrt_surface_without_an_atmosphere(solver, oracle):
    """the HIP path's Ross-Thick / Li-Sparse-R reflectance against hand-computed kernel values (sun at 30 deg, nadir view:
    Kvol = -0.03143, Kgeo = -0.69820, tests/test_oracle_kat.py) and, for slant views, against the double-precision formula:
    over a vacuum the radiance is exactly R(sun, view) mu0 / pi"""
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza))
    vza = (0.0, 50.0, 35.0); vaa = (0.0, 20.0, 200.0)
    for f, r_nadir in (((1.0, 0.0, 1.0), 1.0-0.03143), ((1.0, 1.0, 0.0), 1.0-0.69820), ((0.25, 0.03, 0.12), None)):
        sc = slab_scene(tau=0.0, sza=sza, nx=2, ny=2, target=TARGET_RADIANCE, vza=vza, vaa=vaa)
        sc.jsfc = np.full((2, 2), 4.0, dtype=np.float32)
        sc.psfc = np.zeros((5, 2, 2), dtype=np.float32); sc.psfc[0] = f[0]; sc.psfc[1] = f[1]; sc.psfc[2] = f[2]
        g = gpu_run(solver, sc, 400000, seed=1)['rad'].mean(axis=(1, 2))
        din = np.array([np.sin(np.deg2rad(sc.src_the))*np.cos(np.deg2rad(sc.src_phi)),
                        np.sin(np.deg2rad(sc.src_the))*np.sin(np.deg2rad(sc.src_phi)), np.cos(np.deg2rad(sc.src_the))])
        for iv in range(3):
            t = np.deg2rad(sc.view_the[iv]); p = np.deg2rad(sc.view_phi[iv])
            dout = -np.array([np.sin(t)*np.cos(p), np.sin(t)*np.sin(p), np.cos(t)])
            R = oracle.lsrt(f[0], f[1], f[2], din, dout)
            assert abs(g[iv]-R*mu0/np.pi) < 2e-5*abs(R*mu0/np.pi) + 1e-7, (f, iv, g[iv], R*mu0/np.pi)
        if r_nadir is not None:
            assert abs(g[0]-r_nadir*mu0/np.pi) < 2e-4*mu0/np.pi


@pytest.mark.parametrize('sel', [1.0, 2.0, 1.5])
def test_gpu_tabulated_phase_function_single_scattering(solver, sel):
    """the HIP path's tabulated phase functions (LDS-staged tables, fractional selector = mix of neighbours) against the
    single-scattering limit computed from the table itself in numpy: tables as er3t hands them over (degrees, P normalised
    by the solver so that (1/2) int P dmu = 1, piecewise linear in mu; mca_sca.py:82-92)"""
    ang = np.linspace(0.0, 180.0, 721)
    mu_t = np.cos(np.deg2rad(ang))
    hg = lambda g: (1.0-g*g)/(1.0+g*g-2.0*g*mu_t)**1.5
    tabs = np.stack([hg(0.7), 0.75*(1.0+mu_t**2)])                     # table 1: HG(0.7), table 2: Rayleigh, unnormalised scale
    tabs[1] *= 3.7                                                     # (the solver must renormalise)
    sza = 35.0; mu0 = np.cos(np.deg2rad(sza)); tau = 0.004; omega = 0.8
    sc = slab_scene(tau=tau, omega=omega, apf=sel, albedo=0.0, sza=sza, nz=3, vza=(0.0, 50.0, 180.0), vaa=(0.0, 160.0, 0.0),
                    target=TARGET_RADIANCE, ang=ang.astype(np.float32), pha=tabs.astype(np.float32))
    sc.view_zloc = [705000.0, 705000.0, 0.0]
    nb, nper = 8, 1000000
    r = np.stack([gpu_run(solver, sc, nper, seed=23, offset=b*nper)['rad'][:, 0, 0] for b in range(nb)])
    mean, se = r.mean(axis=0), r.std(axis=0, ddof=1)/np.sqrt(nb)

    def p_table(it, mu):
        m = mu_t[::-1]; p = tabs[it][::-1].astype(np.float64)
        p = p/np.sum(0.25*(p[1:]+p[:-1])*(m[1:]-m[:-1]))
        return np.interp(mu, m, p)
    i0 = int(np.floor(sel-1.0)); fr = (sel-1.0)-i0
    sdir = np.array([np.sin(np.deg2rad(sc.src_the))*np.cos(np.deg2rad(sc.src_phi)),
                     np.sin(np.deg2rad(sc.src_the))*np.sin(np.deg2rad(sc.src_phi)), np.cos(np.deg2rad(sc.src_the))])
    for iv in range(3):
        t = np.deg2rad(sc.view_the[iv]); p = np.deg2rad(sc.view_phi[iv])
        v = -np.array([np.sin(t)*np.cos(p), np.sin(t)*np.sin(p), np.cos(t)])
        mu = float(sdir @ v); muv = abs(v[2])
        P = (1.0-fr)*p_table(i0, mu) + (fr*p_table(min(i0+1, 1), mu) if fr > 0.0 else 0.0)
        if v[2] > 0.0:
            want = omega*P/(4.0*np.pi)*mu0/(mu0+muv)*(1.0-np.exp(-tau*(1.0/mu0+1.0/muv)))
        else:
            want = omega*P/(4.0*np.pi)*mu0/(mu0-muv)*(np.exp(-tau/mu0)-np.exp(-tau/muv))
        assert abs(mean[iv]-want) < 4.0*se[iv] + 0.015*want, (sel, iv, mean[iv], want, se[iv])


def test_gpu_mixture_of_constituents_single_scattering(solver, oracle):
    """two 1-D and two 3-D scattering constituents plus gas absorption in every cell: in the single-scattering limit the
    radiance is that of the mixture phase function sum(omega_i ext_i P_i)/beta_t -- the HIP path directly (weights, the
    further-constituent table `csca`, the mixture local estimate)"""
    nz, ztop, nx, ny = 4, 2000.0, 3, 2
    zgrd = np.linspace(0.0, ztop, nz+1)
    e1 = np.array([2.0e-7, 1.0e-7]); w1 = np.array([1.0, 0.9]); a1 = np.array([-1.0, 0.6]); ab = 1.5e-7
    e3 = np.array([4.0e-7, 1.5e-7]); w3 = np.array([0.95, 0.5]); a3 = np.array([0.85, -2.0])
    sza = 30.0; mu0 = np.cos(np.deg2rad(sza))
    full = lambda v: np.full((nz, ny, nx), v, dtype=np.float32)
    sc = Scene(zgrd=zgrd, ext1d=np.repeat(e1[:, None], nz, axis=1), omg1d=np.repeat(w1[:, None], nz, axis=1),
               apf1d=np.repeat(a1[:, None], nz, axis=1), abs1d=np.full(nz, ab), nx=nx, ny=ny, dx=300.0, dy=300.0, nz3=nz, iz3l=1,
               extp=np.stack([full(e3[0]), full(e3[1])]), omgp=np.stack([full(w3[0]), full(w3[1])]), apfp=np.stack([full(a3[0]), full(a3[1])]),
               sfc_mtype=1, sfc_param=[0.0, 0, 0, 0, 0], src_the=180.0-sza, src_phi=270.0, src_qmax=0.0,
               view_the=[180.0, 130.0, 0.0], view_phi=[0.0, 160.0, 0.0], view_zloc=[705000.0, 705000.0, 0.0], nxr=nx, nyr=ny,
               target=TARGET_RADIANCE)
    bt = e1.sum() + e3.sum() + ab
    tau = bt*ztop
    nb, nper = 8, 2000000
    r = np.stack([gpu_run(solver, sc, nper, seed=29, offset=b*nper)['rad'].mean(axis=(1, 2)) for b in range(nb)])
    mean, se = r.mean(axis=0), r.std(axis=0, ddof=1)/np.sqrt(nb)
    sdir = np.array([np.sin(np.deg2rad(sc.src_the))*np.cos(np.deg2rad(sc.src_phi)),
                     np.sin(np.deg2rad(sc.src_the))*np.sin(np.deg2rad(sc.src_phi)), np.cos(np.deg2rad(sc.src_the))])
    for iv in range(3):
        t = np.deg2rad(sc.view_the[iv]); p = np.deg2rad(sc.view_phi[iv])
        v = -np.array([np.sin(t)*np.cos(p), np.sin(t)*np.sin(p), np.cos(t)])
        mu = float(sdir @ v); muv = abs(v[2])
        wP = sum(w*e*oracle.phase_eval(a, mu) for w, e, a in zip(np.concatenate([w1, w3]), np.concatenate([e1, e3]), np.concatenate([a1, a3])))/bt
        if v[2] > 0.0:
            want = wP/(4.0*np.pi)*mu0/(mu0+muv)*(1.0-np.exp(-tau*(1.0/mu0+1.0/muv)))
        else:
            want = wP/(4.0*np.pi)*mu0/(mu0-muv)*(np.exp(-tau/mu0)-np.exp(-tau/muv))
        assert abs(mean[iv]-want) < 4.0*se[iv] + 0.015*want, (iv, mean[iv], want, se[iv])


def test_gpu_x_y_mirror_symmetry(solver):
    """the scene mirrored across the line y = x (arrays transposed, dx and dy swapped, azimuths phi -> 90 - phi) must give the
    transposed images: x faces and y faces are walked by different code, in the kernel and in the oracle alike, so this is
    checked on the HIP path against itself"""
    rng = np.random.default_rng(5)
    nz, nz3, nx, ny = 5, 3, 6, 4
    zgrd = np.linspace(0.0, 2500.0, nz+1)
    ext = (rng.uniform(0.0, 4e-3, (1, nz3, ny, nx))*(rng.random((1, nz3, ny, nx)) > 0.4)).astype(np.float32)
    base = dict(zgrd=zgrd, ext1d=np.full((1, nz), 2e-5), omg1d=np.ones((1, nz)), apf1d=-np.ones((1, nz)), abs1d=np.full(nz, 1e-6),
                nz3=nz3, iz3l=2, sfc_mtype=1, sfc_param=[0.2, 0, 0, 0, 0], src_the=130.0, src_qmax=0.0, view_the=[180.0, 140.0, 20.0],
                view_zloc=[705000.0, 705000.0, 0.0], target=TARGET_FLUX | TARGET_RADIANCE)
    a = Scene(nx=nx, ny=ny, dx=100.0, dy=150.0, extp=ext, omgp=np.full_like(ext, 0.98), apfp=np.full_like(ext, 0.8),
              src_phi=25.0, view_phi=[0.0, 70.0, 310.0], nxr=nx, nyr=ny, **base)
    extT = np.ascontiguousarray(np.transpose(ext, (0, 1, 3, 2)))
    b = Scene(nx=ny, ny=nx, dx=150.0, dy=100.0, extp=extT, omgp=np.full_like(extT, 0.98), apfp=np.full_like(extT, 0.8),
              src_phi=90.0-25.0, view_phi=[90.0-0.0, 90.0-70.0, (90.0-310.0) % 360.0], nxr=ny, nyr=nx, **base)
    nb, nper = 8, 1500000
    ra, rb = [], []
    for k in range(nb):
        ra.append(gpu_run(solver, a, nper, seed=41, offset=k*nper)); rb.append(gpu_run(solver, b, nper, seed=43, offset=k*nper))
    for key in ('rad', 'flux'):
        xa = np.stack([r[key] for r in ra]); xb = np.stack([np.swapaxes(r[key], -1, -2) for r in rb])
        se = np.sqrt(xa.var(axis=0, ddof=1)/nb + xb.var(axis=0, ddof=1)/nb)
        z = (xa.mean(axis=0)-xb.mean(axis=0))/np.maximum(se, 1e-12)
        z = z[(se > 0) & (xa.mean(axis=0) > 0)]
        assert z.size > 50 and np.mean(np.abs(z) > 3.0) < 0.03 and abs(z.mean()) < 0.35 and 0.7 < z.std() < 1.35, (key, z.mean(), z.std())


# ---------------------------------------------------------------------------------------------
# diffuse-specular mixture (jsfc = 2, Cox-Munk)
# ---------------------------------------------------------------------------------------------
def _dsm_map(sc, p):
    sc.jsfc = np.full((sc.ny, sc.nx), 2.0, dtype=np.float32)
    sc.psfc = np.zeros((5, sc.ny, sc.nx), dtype=np.float32)
    for q in range(5):
        sc.psfc[q] = p[q]


def test_dsm_over_vacuum_is_exact(solver, oracle):
    """no atmosphere: one reflection, radiance of every view = R(sun, view) mu0 / pi -- the HIP path against the closed form
    (R from the oracle's reflectance function, itself pinned by tests/test_oracle_kat.py K14); 2-D map and uniform surface"""
    sza = 35.0; mu0 = np.cos(np.deg2rad(sza)); p = (0.15, 0.1, 1.34, 0.01, 0.04)
    for uniform in (False, True):
        sc = slab_scene(tau=0.0, sza=sza, nx=3, ny=2, target=TARGET_RADIANCE, vza=(0.0, 35.0, 50.0), vaa=(0.0, 0.0, 140.0))
        if uniform:
            sc.sfc_mtype = 2; sc.sfc_param = np.array(p, dtype=np.float32)
        else:
            _dsm_map(sc, p)
        g = gpu_run(solver, sc, 200000, seed=3)
        th = np.deg2rad(sc.src_the); ph = np.deg2rad(sc.src_phi)
        din = np.array([np.sin(th)*np.cos(ph), np.sin(th)*np.sin(ph), np.cos(th)])
        for iv in range(3):
            tv = np.deg2rad(sc.view_the[iv]); pv = np.deg2rad(sc.view_phi[iv])
            dout = -np.array([np.sin(tv)*np.cos(pv), np.sin(tv)*np.sin(pv), np.cos(tv)])
            R = oracle.dsm(np.float32(p), din, dout)[0]
            assert np.isclose(g['rad'][iv].mean(), R*mu0/np.pi, rtol=2e-4), (uniform, iv, g['rad'][iv].mean(), R*mu0/np.pi)


def test_dsm_cloud_scene_parity(solver, oracle, nthreads):
    """a cloud field over a rough sea: nadir + two slant views (one into the glint) and the lean kernel's column view;
    photons reflected by the sea carry weights far from 1 (cosine-sampled directions times R), so the per-pixel noise is
    larger than over land: domain means and counters only"""
    p = (0.2, 0.05, 1.34, 0.0, 0.03)
    sc = les_scene(nx=16, ny=16, nz3=50, vza=(0.0, 30.0, 50.0), vaa=(0.0, 225.0, 45.0), surface_albedo=0.0)
    _dsm_map(sc, p)
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 13, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=13)
    check_counters(g['counters'], o['counters'])
    for iv in range(3):
        gm, om, se = g['rad'][iv].mean(), o['rad'][iv].mean(), o['rad_mean_se'][iv]
        assert abs(gm-om) < 2.0*np.sqrt(2.0)*se + 1e-4*om, (iv, gm, om, se)
    # nadir only: the lean kernel build serves it
    sc1 = les_scene(nx=16, ny=16, nz3=50, surface_albedo=0.0)
    _dsm_map(sc1, p)
    o1 = oracle_batches(oracle, sc1, nb, nper, 14, nthreads)
    g1 = gpu_run(solver, sc1, nb*nper, seed=14)
    assert solver.kernel_name().startswith('k_transport_lean')
    check_counters(g1['counters'], o1['counters'])
    assert abs(g1['rad'][0].mean()-o1['rad'][0].mean()) < 2.0*np.sqrt(2.0)*o1['rad_mean_se'][0] + 1e-4*o1['rad'][0].mean()


def test_dsm_flux_energy(solver):
    """flux over a sea surface: up-welling flux at the surface = albedo(sun) x down-welling flux, with the albedo between
    the Fresnel value at normal incidence and 1"""
    sc = slab_scene(tau=0.0, sza=40.0, nx=2, ny=2, target=TARGET_FLUX)
    _dsm_map(sc, (0.0, 0.0, 1.34, 0.0, 0.02))
    g = gpu_run(solver, sc, 400000, seed=4)
    down, up = g['flux'][1, 0].mean(), g['flux'][2, 0].mean()
    assert np.isclose(down, np.cos(np.deg2rad(40.0)), rtol=1e-5)
    assert 0.02 < up/down < 0.06, up/down                # Fresnel reflectance of water at 40 degrees incidence: 0.025, plus facet tilts


# ---------------------------------------------------------------------------------------------
# all-sky camera (Rad_mrkind = 1)
# ---------------------------------------------------------------------------------------------
def _camera(sc, the, zloc, nxr, nyr, qmax=120.0, umax=120.0, xpos=0.5, ypos=0.5, apsize=0.0, phi=0.0, psi=0.0):
    sc.rad_kind = 1
    sc.view_the = [float(the)]; sc.view_phi = [float(phi)]; sc.view_zloc = [float(zloc)]
    sc.cam_psi = [float(psi)]; sc.cam_xpos = [float(xpos)]; sc.cam_ypos = [float(ypos)]
    sc.cam_qmax = [float(qmax)]; sc.cam_umax = [float(umax)]; sc.cam_vmax = [float(umax)]; sc.cam_apsize = [float(apsize)]
    sc.nxr = nxr; sc.nyr = nyr
    return sc


def test_camera_above_a_lambert_plane_is_exact(solver):
    """the closed form of tests/test_oracle_kat.py K15 against the HIP path directly: no atmosphere, Lambertian ground, every
    line of sight that meets the ground inside the cone of view reads A mu0 / pi; camera turned about all three axes"""
    A, sza = 0.4, 35.0
    mu0 = np.cos(np.deg2rad(sza))
    sc = slab_scene(tau=0.0, albedo=A, sza=sza, nx=40, ny=40, dx=200.0, dy=200.0, target=TARGET_RADIANCE)
    _camera(sc, the=170.0, phi=30.0, psi=20.0, zloc=600.0, nxr=8, nyr=8, qmax=100.0, umax=100.0, xpos=0.3, ypos=0.6)
    g = gpu_run(solver, sc, 4000000, seed=3)
    assert solver.kernel_name().endswith('+ k_rays'), solver.kernel_name()
    img = g['rad'][0]
    du = np.deg2rad(100.0)/8
    ue = (np.arange(9)-4)*du
    umx = np.maximum(np.abs(ue[:-1]), np.abs(ue[1:]))
    inside = np.sqrt(umx[:, None]**2+umx[None, :]**2) < np.deg2rad(50.0)
    want = A*mu0/np.pi
    # (a pixel collects the reflections of about 2000 photons: 2-3 % of noise each; 32 pixels together 0.5 %)
    assert inside.sum() >= 24 and np.all(np.abs(img[inside]-want) < 0.10*want), (img[inside]/want)
    assert abs(img[inside].mean()-want) < 0.012*want, img[inside].mean()/want


def test_camera_sees_the_periodic_images_of_the_domain(solver, oracle, nthreads):
    """`cam_images` on the HIP path (camera build of the ray kernel: a start batch per image of the camera): the closed form of
    tests/test_oracle_kat.py -- a camera 600 m above a Lambertian plane of 2 km x 2 km reads A mu0 / pi out to 75 degrees once the
    25 images within two domain lengths are served --, and parity with the oracle on a cloud scene seen from the ground under a cone
    of 160 degrees, with the images of one domain length around the nearest one"""
    A, sza = 0.4, 35.0
    mu0 = np.cos(np.deg2rad(sza))
    want = A*mu0/np.pi
    sc = slab_scene(tau=0.0, albedo=A, sza=sza, nx=10, ny=10, dx=200.0, dy=200.0, target=TARGET_RADIANCE)
    _camera(sc, the=180.0, zloc=600.0, nxr=8, nyr=8, qmax=160.0, umax=160.0, xpos=0.3, ypos=0.6)
    sc.cam_images = 2
    g = gpu_run(solver, sc, 4000000, seed=3)
    assert solver.kernel_name().endswith('+ k_rays'), solver.kernel_name()
    img = g['rad'][0]
    du = np.deg2rad(160.0)/8
    ue = (np.arange(9)-4)*du
    umx = np.maximum(np.abs(ue[:-1]), np.abs(ue[1:]))
    th = np.sqrt(umx[:, None]**2+umx[None, :]**2)
    seen = th < np.deg2rad(75.0)
    assert seen.sum() >= 12 and np.all(np.abs(img[seen]-want) < 0.10*want), (img[seen]/want)
    assert abs(img[seen].mean()-want) < 0.012*want, img[seen].mean()/want
    sc.cam_images = 0
    g0 = gpu_run(solver, sc, 4000000, seed=3)
    ring = (th > np.deg2rad(62.0)) & (th < np.deg2rad(75.0))
    assert g0['rad'][0][ring].mean() < 0.9*want                      # the nearest image alone: incomplete beyond 59 degrees
    # a cloud scene from the ground, one ring of images: against the oracle on the same photon ids
    sc = les_scene(nx=16, ny=16, nz3=50, surface_albedo=0.1)
    _camera(sc, the=0.0, zloc=0.0, nxr=16, nyr=16, qmax=160.0, umax=160.0, xpos=0.4, ypos=0.55, apsize=30.0)
    sc.cam_images = 1
    nb, nper = 16, 10000
    o = oracle_batches(oracle, sc, nb, nper, 19, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=19)
    assert solver.kernel_name().endswith('+ k_rays'), solver.kernel_name()
    check_counters(g['counters'], o['counters'])
    gm, om, se = g['rad'][0].mean(), o['rad'][0].mean(), o['rad_mean_se'][0]
    assert om > 0.0 and abs(gm-om) < 2.0*np.sqrt(2.0)*se + 2e-3*om, (gm, om, se)
    gb = g['rad'][0].reshape(4, 4, 4, 4).mean(axis=(1, 3)); ob = o['rad'][0].reshape(4, 4, 4, 4).mean(axis=(1, 3))
    seb = np.sqrt((o['rad_se'][0]**2).reshape(4, 4, 4, 4).sum(axis=(1, 3)))/16.0
    lit = ob > 0.05*ob.max()
    assert np.all(np.abs(gb-ob)[lit] < 4.0*np.sqrt(2.0)*seb[lit] + 0.02*ob[lit]), ((gb-ob)[lit]/ob[lit])


def test_camera_parity_cloud_scene(solver, oracle, nthreads):
    """a camera on the ground looking up at a broken cloud field and one above it looking down: the HIP path against the oracle
    on the same photon ids -- image means, and the images themselves in 4 x 4 blocks"""
    for the, zloc in ((0.0, 0.0), (180.0, 3000.0)):
        sc = les_scene(nx=16, ny=16, nz3=50, surface_albedo=0.1)
        _camera(sc, the=the, zloc=zloc, nxr=16, nyr=16, qmax=140.0, umax=140.0, xpos=0.4, ypos=0.55, apsize=30.0)
        sc.cam_images = 0      # (the general kernel, held against the ray kernel below, serves the nearest image of a camera only)
        nb, nper = 16, 20000
        o = oracle_batches(oracle, sc, nb, nper, 17, nthreads)
        g = gpu_run(solver, sc, nb*nper, seed=17)
        # (cameras over Lambertian surfaces go through the event lists and the ray kernel's camera build; the general kernel with the
        #  rays inside its loop is the other route: same counters, images equal to what two float32 programs differ by)
        assert solver.kernel_name().endswith('+ k_rays'), solver.kernel_name()
        check_counters(g['counters'], o['counters'])
        try:
            solver.set_kernel(general=True)
            gg = gpu_run(solver, sc, nb*nper, seed=17)
            assert solver.kernel_name().startswith('k_transport<'), solver.kernel_name()
        finally:
            solver.set_kernel()
        check_counters(gg['counters'], o['counters'])
        assert abs(gg['rad'][0].mean()-g['rad'][0].mean()) < 5e-3*g['rad'][0].mean()
        gm, om, se = g['rad'][0].mean(), o['rad'][0].mean(), o['rad_mean_se'][0]
        assert om > 0.0 and abs(gm-om) < 2.0*np.sqrt(2.0)*se + 2e-3*om, (the, gm, om, se)
        gb = g['rad'][0].reshape(4, 4, 4, 4).mean(axis=(1, 3)); ob = o['rad'][0].reshape(4, 4, 4, 4).mean(axis=(1, 3))
        seb = np.sqrt((o['rad_se'][0]**2).reshape(4, 4, 4, 4).sum(axis=(1, 3)))/16.0
        lit = ob > 0.05*ob.max()
        assert np.all(np.abs(gb-ob)[lit] < 4.0*np.sqrt(2.0)*seb[lit] + 0.02*ob[lit]), (the, (gb-ob)[lit]/ob[lit])


def test_camera_over_an_lsrt_surface(solver, oracle, nthreads):
    """a camera above broken clouds looking down at a Ross-Li surface: the ray kernel's camera build evaluates the reflectance model
    itself (no second build of it) -- against the oracle on the same photon ids, and against the general kernel"""
    sc = les_scene(nx=16, ny=16, nz3=50, lsrt=True, cot_mean=2.0)
    _camera(sc, the=180.0, zloc=3000.0, nxr=16, nyr=16, qmax=140.0, umax=140.0, xpos=0.4, ypos=0.55, apsize=30.0)
    sc.cam_images = 0          # (as above: the general kernel is part of the comparison)
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 29, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=29)
    assert solver.kernel_name().endswith('+ k_rays'), solver.kernel_name()
    check_counters(g['counters'], o['counters'])
    assert o['counters']['surface'] > 0.2*nb*nper       # (the surface is seen: most photons reach it under clouds this thin)
    gm, om, se = g['rad'][0].mean(), o['rad'][0].mean(), o['rad_mean_se'][0]
    assert om > 0.0 and abs(gm-om) < 2.0*np.sqrt(2.0)*se + 2e-3*om, (gm, om, se)
    try:
        solver.set_kernel(general=True)
        gg = gpu_run(solver, sc, nb*nper, seed=29)
        assert solver.kernel_name().startswith('k_transport<'), solver.kernel_name()
    finally:
        solver.set_kernel()
    assert abs(gg['rad'][0].mean()-gm) < 5e-3*gm, (gg['rad'][0].mean(), gm)


def test_two_cameras_in_one_run_equal_each_alone(solver):
    """two cameras (one on the ground looking up, one above the clouds looking down) served from the same event lists: each image is
    what the camera records alone on the same photon ids (roulettes off, nearest image only: they are keyed by the view's number)"""
    def scene(which):
        sc = les_scene(nx=16, ny=16, nz3=50, surface_albedo=0.1)
        _camera(sc, the=0.0, zloc=0.0, nxr=16, nyr=16, qmax=140.0, umax=140.0, xpos=0.4, ypos=0.55, apsize=30.0)
        the, zloc, xpos = [0.0, 180.0], [0.0, 3000.0], [0.4, 0.7]
        pick = [0, 1] if which is None else [which]
        sc.view_the = [the[i] for i in pick]; sc.view_phi = [0.0 for _ in pick]; sc.view_zloc = [zloc[i] for i in pick]
        for name, val in (('cam_psi', 0.0), ('cam_ypos', 0.55), ('cam_qmax', 140.0), ('cam_umax', 140.0), ('cam_vmax', 140.0), ('cam_apsize', 30.0)):
            setattr(sc, name, [val for _ in pick])
        sc.cam_xpos = [xpos[i] for i in pick]
        sc.le_tau1 = 0.0; sc.le_cmin = 0.0; sc.cam_images = 0      # (the roulette on the farther images of a camera is keyed by the view's number too)
        return sc
    n = 200000
    both = gpu_run(solver, scene(None), n, seed=23)
    assert solver.kernel_name().endswith('+ k_rays') and both['rad'].shape[0] == 2
    for i in (0, 1):
        one = gpu_run(solver, scene(i), n, seed=23)
        assert one['rad'][0].sum() > 0.0 and np.allclose(both['rad'][i], one['rad'][0], rtol=1e-5, atol=1e-7*one['rad'][0].max())


def test_ray_kernel_with_small_event_lists(solver, oracle, nthreads):
    """the launch machinery of the marched views at a size where its corners are reached: event lists of 65 536 records, so that a
    run of 3.2e5 photons is a pilot launch and dozens of launches sized from the events per photon seen so far (a short launch may
    land on ONE XCD's list); the images are the oracle's and those of a run with lists that hold everything"""
    sc = les_scene(nx=16, ny=16, nz3=50, vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 7, nthreads)
    res = {}
    try:
        for cap in (16, 27):
            solver.set_tuning(evcap_log2=cap)
            g = gpu_run(solver, sc, nb*nper, seed=7)
            assert solver.kernel_name().endswith('+ k_rays')
            check_counters(g['counters'], o['counters'])
            check_radiance(g, o)
            ms, launches = solver.timing()
            assert (launches > 20) == (cap == 16), (cap, launches)
            res[cap] = g['rad']
    finally:
        solver.set_tuning(evcap_log2=28)
    assert np.allclose(res[16], res[27], rtol=1e-4, atol=1e-9)      # same photons, same rays: the order of the sums only


@pytest.mark.parametrize('surface', ['lambert', 'lsrt'])
def test_ray_kernels_beside_the_next_photon_loop_change_no_result(solver, surface):
    """mi3d_set_tuning "overlap_rays" (round 5): with two sets of event lists the ray kernels of launch i run on a stream of their own
    beside the photon loop of launch i + 1.  Same photon ids, same rays: the event counters are equal and the images differ by the
    order of float64 sums only -- over dozens of launches (small lists), with the heavy build's notes (LSRT) and without, and with
    the workgroups per CU of either kernel cut down so that the two really run side by side."""
    sc = les_scene(nx=16, ny=16, nz3=50, vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0), lsrt=(surface == 'lsrt'))
    n = 320000
    res = {}
    try:
        solver.set_tuning(evcap_log2=16)
        for key, knobs in (('one', dict(overlap_rays=0)), ('two', dict(overlap_rays=1)), ('shared', dict(overlap_rays=1, rays_wg=3, emit_wg=2))):
            solver.set_tuning(rays_wg=0, emit_wg=0)
            solver.set_tuning(**knobs)
            res[key] = gpu_run(solver, sc, n, seed=7)
            assert solver.kernel_name().endswith('+ k_rays')
            ms, launches = solver.timing()
            assert launches > 20 and ms > 0.0
    finally:
        solver.set_tuning(evcap_log2=28, overlap_rays=0, rays_wg=0, emit_wg=0)
    for key in ('two', 'shared'):
        for k in ('photons', 'scatter', 'surface', 'killed', 'escaped', 'roulette', 'steps3d', 'le_rays', 'le_steps3d'):
            assert res[key]['counters'][k] == res['one']['counters'][k], (key, k)
        assert np.allclose(res[key]['rad'], res['one']['rad'], rtol=1e-4, atol=1e-9), key


def test_record_sort_beside_the_next_photon_loop_changes_no_result(solver):
    """mi3d_set_tuning "overlap_sort" (round 5, on by default): with two sets of record lists the sort and the sums of launch i run on a
    stream of their own beside the photon loop of launch i + 1, and -- tallies in the handle's own buffers -- beside the NEXT RUN's loops
    too: mi3d_run returns without making its stream wait for the last sort; every call that reads or clears the tallies does.  Same
    photon ids, same tallies: flux and heating rates equal to the order of float64 sums -- one stream against two over dozens of
    launches (small lists), three runs back to back with nothing read in between, a reset straight after a run, and buffers and a
    stream of the caller's (the stream alone is waited for: the run itself must have joined the sort stream)."""
    import torch
    from er3t_amd.scene import TARGET_FLUX, TARGET_HEAT
    sc = les_scene(nx=16, ny=16, nz3=50, target='flux', aerosol=True)
    sc.target = TARGET_FLUX | TARGET_HEAT
    sc.abs1d = sc.abs1d*30.0 + 2.0e-5
    n = 300000
    res = {}
    try:
        solver.set_tuning(tlcap_log2=17)
        for key, knobs in (('one', dict(overlap_sort=0)), ('two', dict(overlap_sort=2, tl_split=1)), ('split', dict(overlap_sort=2, tl_split=8))):
            solver.set_tuning(**knobs)
            res[key] = gpu_run(solver, sc, 3*n, seed=7)
            assert 'k_tl_scatter' in solver.kernel_name()
            ms, launches = solver.timing()
            assert launches > 20 and ms > 0.0
        # three runs back to back, nothing read in between -- on two streams whatever the run (2), and as the library chooses (1: a small run on an
        # idle stream takes one stream, a run queued behind another two: the routes follow each other) --; then a run whose tallies a reset clears
        # while its last sort may still be on its way
        for key, mode in (('b2b', 2), ('b2b_auto', 1)):
            solver.set_tuning(overlap_sort=mode, tl_split=4)
            solver.reset()
            for q in range(3):
                solver.run(n, seed=7, offset=q*n)
            res[key] = {'flux': solver.flux(3*n).astype(np.float64), 'heat': solver.heating(3*n).astype(np.float64), 'counters': solver.counters()}
        solver.set_tuning(overlap_sort=2)
        solver.run(n, seed=99)
        solver.reset()
        for q in range(3):
            solver.run(n, seed=7, offset=q*n)
        solver.sync()
        res['after_reset'] = {'flux': solver.flux(3*n).astype(np.float64), 'heat': solver.heating(3*n).astype(np.float64), 'counters': solver.counters()}
        # the caller's buffers and stream: what the caller queues on ITS stream after mi3d_run finds the tallies complete
        dev = torch.device('cuda:0')
        flux_t = torch.zeros(res['one']['flux'].size, dtype=torch.float64, device=dev)
        heat_t = torch.zeros(res['one']['heat'].size, dtype=torch.float64, device=dev)
        st = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        solver.bind(None, flux_t.data_ptr(), st.cuda_stream, heat_ptr=heat_t.data_ptr())
        solver.reset()
        for q in range(3):
            solver.run(n, seed=7, offset=q*n)
        with torch.cuda.stream(st):
            fsum = flux_t.sum(); hsum = heat_t.sum()      # (queued behind the runs on the caller's stream, no mi3d call in between)
        st.synchronize()
        res['bound'] = {'flux': solver.flux(3*n).astype(np.float64), 'heat': solver.heating(3*n).astype(np.float64), 'counters': solver.counters()}
        raw_f, raw_h = float(fsum.item()), float(hsum.item())
        assert np.isclose(raw_f, float(flux_t.sum().item()), rtol=1e-12) and np.isclose(raw_h, float(heat_t.sum().item()), rtol=1e-12)
    finally:
        solver.bind(None, None, None)
        solver.set_tuning(tlcap_log2=31, overlap_sort=1, tl_split=4)
    assert res['one']['flux'].sum() > 0.0 and res['one']['heat'].sum() > 0.0
    for key in ('two', 'split', 'b2b', 'b2b_auto', 'after_reset', 'bound'):
        assert res[key]['counters']['flux_tally'] == res['one']['counters']['flux_tally'], key
        assert np.allclose(res[key]['flux'], res['one']['flux'], rtol=1e-6, atol=1e-9), key
        assert np.allclose(res[key]['heat'], res['one']['heat'], rtol=1e-6, atol=1e-12), key


@pytest.mark.parametrize('job', ['marched views', 'flux'])
def test_pre_pass_beside_the_previous_photon_loop_changes_no_result(solver, job):
    """mi3d_set_tuning "overlap_pre" (round 5, on by default for the flux loop and the event-writing loop): two sets of photon order, tiles'
    ends and entry records; the pre-pass kernels of launch i + 1 run on a stream of their own beside the photon loop of launch i.  Same
    photon ids in the same order: counters equal, tallies equal to the order of the sums -- over dozens of launches (small lists), and
    over runs back to back."""
    if job == 'flux':
        sc = les_scene(nx=48, ny=48, nz3=50, target='flux', aerosol=True)
        knob, small, big = 'tlcap_log2', 17, 31
    else:
        sc = les_scene(nx=48, ny=48, nz3=50, vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
        knob, small, big = 'evcap_log2', 16, 28
    n = 320000
    key = 'flux' if job == 'flux' else 'rad'
    res = {}
    try:
        solver.set_tuning(**{knob: small})
        for name, on in (('one', 0), ('two', 2)):      # (2: whatever the run; 1, the default, leaves small runs that are read one by one on one stream)
            solver.set_tuning(overlap_pre=on)
            res[name] = gpu_run(solver, sc, n, seed=7)
            ms, launches = solver.timing()
            assert launches > 10 and ms > 0.0
        solver.set_tuning(overlap_pre=1)
        solver.reset()
        for q in range(3):
            solver.run(n, seed=7, offset=q*n)
        solver.sync()
        b2b = (solver.flux(3*n) if job == 'flux' else solver.radiance(3*n)).astype(np.float64)
        solver.set_tuning(overlap_pre=0)
        solver.reset()
        for q in range(3):
            solver.run(n, seed=7, offset=q*n)
        solver.sync()
        b2b_one = (solver.flux(3*n) if job == 'flux' else solver.radiance(3*n)).astype(np.float64)
    finally:
        solver.set_tuning(**{knob: big, 'overlap_pre': 1})
    for k in ('photons', 'scatter', 'surface', 'killed', 'escaped', 'roulette', 'steps3d', 'le_rays', 'le_steps3d', 'flux_tally'):
        assert res['two']['counters'][k] == res['one']['counters'][k], k
    assert res['one'][key].sum() > 0.0
    assert np.allclose(res['two'][key], res['one'][key], rtol=1e-4 if key == 'rad' else 1e-6, atol=1e-9)
    assert np.allclose(b2b, b2b_one, rtol=1e-4 if key == 'rad' else 1e-6, atol=1e-9)
    # The hand-over ADVICE r5 found unguarded: a LONG one-stream run (below 2^22 photons on an idle handle: one launch, milliseconds of photon
    # loop reading set 0) with a two-stream run queued straight behind it, whose pre-pass on the other stream writes set 0 again -- it must
    # wait for the first run's loop although that run never took the two-stream route.  Several rounds with sizes of their own; the photon
    # counters tell a photon that was lost or taken twice, the tallies a torn entry record.
    n1, n2 = 3000000, 300000
    ref = None
    for mode in (0, 1):
        solver.set_tuning(overlap_pre=mode)
        acc = []
        for rnd in range(3):
            solver.reset()
            solver.run(n1 + rnd*4096, seed=11, offset=0)                     # (idle handle, < 2^22 photons: one stream under either mode)
            solver.run(n2, seed=11, offset=n1 + rnd*4096)                     # (queued behind it: two streams under mode 1)
            solver.run(n2 + 77, seed=11, offset=n1 + rnd*4096 + n2)
            solver.sync()
            tot = n1 + rnd*4096 + 2*n2 + 77
            acc.append(((solver.flux(tot) if job == 'flux' else solver.radiance(tot)).astype(np.float64), solver.counters()))
        if ref is None:
            ref = acc
        else:
            for (a, ca), (b, cb) in zip(acc, ref):
                for k in ('photons', 'scatter', 'surface', 'killed', 'escaped', 'flux_tally'):
                    assert ca[k] == cb[k], (k, ca[k], cb[k])
                assert np.allclose(a, b, rtol=1e-4 if key == 'rad' else 1e-6, atol=1e-9)
    solver.set_tuning(overlap_pre=1)


def test_a_run_that_overflows_its_event_lists_fails_loudly_and_leaves_nothing_behind(solver, oracle, nthreads):
    """An event list that runs full fails the run (never silently short) -- and mi3d_reset clears the partial tallies the
    failed run left in the accumulation image, so that the next run on the same handle is the oracle's again.  Forced here by
    what sizes the launches: a nearly empty scene first (few events per photon), then the same 3-D arrays under a 1-D profile
    that scatters a hundred times more (only the 1-D profiles changed: the estimate is kept with a margin of 1.5)."""
    import copy
    sc = les_scene(nx=16, ny=16, nz3=50, vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
    thin = copy.deepcopy(sc)
    thin.extp = (thin.extp*1.0e-3).astype(np.float32)
    thick1d = copy.deepcopy(thin)
    thick1d.ext1d = thick1d.ext1d*3.0e3
    nph = 300000
    try:
        solver.set_tuning(evcap_log2=16)
        gpu_run(solver, thin, nph, seed=3)                    # pilot + launches: the handle now knows ~1 event per photon
        assert solver.kernel_name().endswith('+ k_rays')
        solver.update_atm1d(thick1d)
        solver.reset()
        with pytest.raises(OSError, match='ran full'):
            solver.run(nph, seed=3)                           # (mi3d_run does not wait for its last launches: the call that looks
            solver.sync()                                     #  at the tallies next reports the list that ran full)
        # ... unless the tallies have been cleared in between: a reset forgets the runs before it
        solver.reset()
        try:
            solver.run(nph, seed=3)
        except OSError as e:                                  # (a run of several launches may see its first ones' lists itself)
            assert 'ran full' in str(e)
        solver.reset()
        solver.sync()
    finally:
        solver.set_tuning(evcap_log2=28)
    nb, nper = 16, 20000
    o = oracle_batches(oracle, sc, nb, nper, 7, nthreads)
    g = gpu_run(solver, sc, nb*nper, seed=7)                  # (load_scene, reset, run on the handle the failed run used)
    check_counters(g['counters'], o['counters'])
    check_radiance(g, o)


def test_an_overflow_on_the_statistics_path_fails_the_call_that_reads_the_tallies(solver):
    """`mcarats_ng`'s in-memory route is run -> mi3d_stats_add -> reset for the next job: mi3d_run does not wait for its last launches, so
    a launch of that job whose event list ran full must fail mi3d_stats_add, which reads the job's tallies into the run field -- the
    mi3d_reset that follows would forget it, and the short tallies would be part of the run's mean and standard deviation for good
    (ADVICE r4).  Forced as in the test above: the handle believes in one event per photon, the scene then scatters a thousand
    times more."""
    import copy
    sc = les_scene(nx=16, ny=16, nz3=50, vza=(0.0, 45.6, 60.0), vaa=(0.0, 30.0, 200.0))
    thin = copy.deepcopy(sc)
    thin.extp = (thin.extp*1.0e-3).astype(np.float32)
    thick1d = copy.deepcopy(thin)
    thick1d.ext1d = thick1d.ext1d*3.0e3
    nph = 300000
    try:
        solver.set_tuning(evcap_log2=16)
        gpu_run(solver, thin, nph, seed=3)
        assert solver.kernel_name().endswith('+ k_rays')
        solver.update_atm1d(thick1d)
        solver.reset()
        solver.stats_begin()
        with pytest.raises(OSError, match='ran full'):
            solver.run(nph, seed=3)
            solver.stats_add(nph)                             # (no mi3d_sync, no read-out in between: the statistics path as mcarats_ng drives it)
        solver.reset()
        solver.sync()
    finally:
        solver.set_tuning(evcap_log2=28)
    g = gpu_run(solver, sc, 50000, seed=7)                    # the handle serves the next job
    assert g['counters']['photons'] == 50000 and g['rad'].mean() > 0.0


@pytest.mark.parametrize('what', ['column', 'marched', 'general', 'flux', 'two_constituents'])
def test_padded_voxel_record_strides_change_no_result(solver, what):
    """mi3d_set_tuning "vpad_col" / "vpad_row" move the voxel records apart in memory (DevScene::vcol_f4, vrow_f4) and nothing
    else: every build that reads them -- the lean loop, the ray kernel, the general kernel, the flux
    loop, the build with a second 3-D constituent -- follows the same histories (event counters equal) and sums the same tallies
    (float64 atomics in another order: 1e-5 relative on the image means, 1e-3 of the largest pixel per pixel)."""
    kw = dict(nx=20, ny=12, nz3=14)
    if what == 'flux':
        sc = les_scene(target='flux', **kw)
    elif what == 'two_constituents':
        sc = les_scene(aerosol=True, **kw)
    elif what == 'column':
        sc = les_scene(**kw)
    else:
        sc = les_scene(vza=(0.0, 40.0), vaa=(0.0, 120.0), **kw)
    nph = 200000
    try:
        solver.set_kernel(general=(what == 'general'))
        a = gpu_run(solver, sc, nph, seed=11)
        solver.set_tuning(vpad_col=3, vpad_row=5)
        b = gpu_run(solver, sc, nph, seed=11)
    finally:
        solver.set_tuning(vpad_col=0, vpad_row=0)
        solver.set_kernel()
    for k in ('photons', 'scatter', 'surface', 'killed', 'escaped', 'roulette', 'steps3d'):
        assert a['counters'][k] == b['counters'][k], (k, a['counters'][k], b['counters'][k])
    for key in ('rad', 'flux'):
        if key in a and a[key].size:
            assert np.allclose(a[key].mean(), b[key].mean(), rtol=1e-5), key
            assert np.abs(a[key]-b[key]).max() <= 1e-3*np.abs(a[key]).max(), key


@pytest.mark.parametrize('sza,saa', [(30.0, 45.0), (63.0, 200.0), (0.0, 0.0)])
def test_tally_window_sums_what_the_atomics_sum(solver, sza, saa):
    """The lean loop sums the column view's tallies of a workgroup in LDS for the 64 x 64 pixels around the tile its photons started
    above and adds them to the image when it moves on (mi3d_set_tuning "tally_window", the default).  Same histories, same tallies:
    the image equals the one made of atomics alone up to the float32 partial sums (1e-6 of a pixel's value, here 2e-5 of the brightest
    one), whatever the sun's direction puts between a tile and its window -- one of the three shifts it across the domain's cyclic
    edge -- and with tiles much smaller than the window's reach as well as larger."""
    sc = les_scene(nx=96, ny=80, nz3=12, sza=sza, saa=saa)
    nph = 3000000
    res = {}
    try:
        for tc in (24, 48):
            for win in (1, 0):
                solver.set_tuning(tile_cols=tc, tally_window=win)
                res[(tc, win)] = gpu_run(solver, sc, nph, seed=5)
                assert solver.kernel_name().startswith('k_transport_lean<')
    finally:
        solver.set_tuning(tile_cols=-1, tally_window=1)
    for tc in (24, 48):
        a, b = res[(tc, 1)], res[(tc, 0)]
        for k in ('photons', 'scatter', 'surface', 'killed', 'escaped', 'roulette', 'steps3d'):
            assert a['counters'][k] == b['counters'][k], (k, a['counters'][k], b['counters'][k])
        assert np.isclose(a['rad'].mean(), b['rad'].mean(), rtol=2e-6)
        assert np.abs(a['rad']-b['rad']).max() <= 2e-5*b['rad'].max()


@pytest.mark.parametrize('solver_id,aerosol', [(SOLVER_IPA, False), (SOLVER_P3D, False), (SOLVER_3D, True)])
def test_tally_window_under_every_solver_and_with_two_constituents(solver, solver_id, aerosol):
    """The window is placed where the direct beam from a tile meets the clouds -- above the tile itself under the independent-pixel
    approximation, where no photon leaves its column -- and serves the builds for the partial 3-D solver and for a second 3-D
    constituent alike: same histories, the same image as with atomics alone."""
    sc = les_scene(nx=96, ny=80, nz3=12, sza=40.0, saa=120.0, solver=solver_id, aerosol=aerosol)
    nph = 2000000
    res = {}
    try:
        for win in (1, 0):
            solver.set_tuning(tile_cols=32, tally_window=win)
            res[win] = gpu_run(solver, sc, nph, seed=9)
            assert solver.kernel_name().startswith('k_transport_lean<')
    finally:
        solver.set_tuning(tile_cols=-1, tally_window=1)
    a, b = res[1], res[0]
    for k in ('photons', 'scatter', 'surface', 'killed', 'escaped', 'roulette', 'steps3d'):
        assert a['counters'][k] == b['counters'][k], (k, a['counters'][k], b['counters'][k])
    assert np.isclose(a['rad'].mean(), b['rad'].mean(), rtol=2e-6)
    assert np.abs(a['rad']-b['rad']).max() <= 2e-5*b['rad'].max()


@pytest.mark.parametrize('n', [4096, 100003, 5000000])
def test_photon_order_is_a_permutation_grouped_by_tile(solver, oracle, n):
    """The order a launch works through its photons in (k_bin_count / k_bin_scan / k_bin_scatter: a counting sort by start tile, the
    scatter 4096 indices at a time inside LDS): every index of the launch exactly once; the pieces of the tiles follow each other and
    end where the cursors say -- the map the lean loop's tally window is placed by -- and every index lies in the piece of the tile
    its photon starts above (Philox block 0 of the oracle gives the position)."""
    sc = les_scene(nx=96, ny=80, nz3=12)
    tc = 32
    try:
        solver.set_tuning(tile_cols=tc)
        solver.load_scene(sc); solver.set_counting(True); solver.reset()
        solver.run(n, seed=77); solver.sync()
        assert solver.counters()['photons'] == n
        order, tend = solver.debug_order(n)
    finally:
        solver.set_tuning(tile_cols=-1)
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
    ntx, nty = (sc.nx+tc-1)//tc, (sc.ny+tc-1)//tc
    ends = tend[:ntx*nty].astype(np.int64)
    assert np.all(np.diff(ends) >= 0) and ends[-1] == n
    # the tile of a sample of the indices, from the launch position of their photons
    starts = np.concatenate(([0], ends[:-1]))
    rng = np.random.default_rng(5)
    for pos in rng.integers(0, n, size=200):
        w = oracle.philox(77, int(order[pos]), 0)
        u = ((w >> 9).astype(np.float64) + 0.5)/8388608.0
        x, y = np.float32(u[0])*np.float32(sc.dx*sc.nx), np.float32(u[1])*np.float32(sc.dy*sc.ny)
        tx, ty = min(int(x/(sc.dx*tc)), ntx-1), min(int(y/(sc.dy*tc)), nty-1)
        t = ty*ntx + (ntx-1-tx if ty & 1 else tx)
        # (a position on a tile's edge may round either way in float32: the neighbouring tile is as good)
        near = [tt for tt in (t-1, t, t+1) if 0 <= tt < ntx*nty and starts[tt] <= pos < ends[tt]]
        assert near, (pos, t)
        assert near[0] == t or abs(x/(sc.dx*tc) - round(x/(sc.dx*tc))) < 1e-3 or abs(y/(sc.dy*tc) - round(y/(sc.dy*tc))) < 1e-3, (pos, t, near)


def test_tally_window_without_the_accumulation_image(solver):
    """"rad_spread" 0: the tallies go to the compact image (8 bytes per pixel, rows of nxr pixels) -- the window's sums as well as the
    tallies outside it.  The same image as through the accumulation image, to the order of the float64 sums."""
    sc = les_scene(nx=96, ny=80, nz3=12, sza=35.0, saa=300.0)
    nph = 2000000
    res = {}
    try:
        for spread in (1, 0):
            solver.set_tuning(tile_cols=32, rad_spread=spread)
            res[spread] = gpu_run(solver, sc, nph, seed=21)
    finally:
        solver.set_tuning(tile_cols=-1, rad_spread=-1)
    a, b = res[1], res[0]
    assert a['counters']['scatter'] == b['counters']['scatter']
    assert np.isclose(a['rad'].mean(), b['rad'].mean(), rtol=2e-6)
    assert np.abs(a['rad']-b['rad']).max() <= 2e-5*b['rad'].max()
