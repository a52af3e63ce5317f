"""
Oracle parity at BASELINE.json's FULL grid sizes, every configuration bench.py can run (bench.make_scene):

    les128       config 2   128 x 128 x 50,  nadir radiance
    les128_flux  config 3   128 x 128 x 50 + 3-D aerosol, flux, 16 g through `mcarats_ng` / `mca_out_ng` (the drop-in route)
    les128_aer   config 3   the same scene's radiance leg (two 3-D constituents), 16 g x 4 runs through `mcarats_ng` / `mca_out_ng`
    les480       config 4   480 x 480 x 100, nadir radiance (the bench workload; lean AND general kernel build)
    les480_mv9   config 5   480 x 480 x 100, nine views + LSRT surface

The HIP path (through the C-ABI) and the CPU oracle transport the SAME photon ids, batch by batch.  Tolerances:
  * domain mean per view: |GPU - oracle| < 2 sigma, sigma = Monte-Carlo standard error of the difference of two independent
    estimates of this size (sqrt(2) x the oracle's batch-to-batch standard error) -- north_star's "within 2 sigma";
  * the same difference PAIRED (same ids in both, so most of the noise cancels): < 4 standard errors of the paired
    difference + 0.03 % of the mean (float32 against float64 rounding) -- catches a bias far below the Monte-Carlo noise;
  * 16 x 16 block means of the image: |z| < 4 (one of the 256 blocks of a view may reach 6: eight batches give a Student-t tail,
    and the typical |z| of these paired runs is 0.05), |mean z| < 0.5, std z < 1 (1 would be two independent runs);
  * flux (config 3): per level and variable, the domain means of the two g-summed results within 2 sigma + 0.1 %.
Sizes are chosen so that the oracle (16 threads) needs 10-30 s per configuration.
"""

import contextlib
import io
import os

import numpy as np
import pytest

from bench import make_scene

pytestmark = pytest.mark.gpu


def _paired(solver, oracle, sc, nb, nper, seed, nthreads, general=False):
    solver.bind(None, None, None)
    solver.set_kernel(general=general)
    solver.load_scene(sc)
    solver.set_counting(False)
    g, o = [], []
    for b in range(nb):
        solver.reset(); solver.run(nper, seed=seed, offset=b*nper); solver.sync()
        g.append(solver.radiance(nper).astype(np.float64))
        o.append(oracle.run(sc, nper, seed=seed, offset=b*nper, nthreads=nthreads)['rad'])
    name = solver.kernel_name()
    solver.set_kernel(general=False)
    return np.stack(g), np.stack(o), name


def _check_images(g, o, nblk=16, min_block_rel=0.0):
    """the assertions of this file on the numbers bench.py prints as its `parity` object (bench.parity_stats)"""
    from bench import parity_stats
    for q in parity_stats(g, o, nblk, min_block_rel):
        iv = q['view']
        assert abs(q['diff']) < 2.0*q['se_independent'], ('view %d: domain means differ by more than 2 sigma' % iv, q)
        assert abs(q['diff']) < 4.0*q['se_paired'] + 3.0e-4*q['mean_oracle'], ('view %d: paired difference' % iv, q)
        assert q['n_abs_z_ge_4'] <= 1 and q['block_abs_z_max'] < 6.0, (iv, q)
        assert abs(q['block_z_mean']) < 0.5 and q['block_z_std'] < 1.0, (iv, q)


def test_config2_les128_nadir(solver, oracle, nthreads):
    g, o, name = _paired(solver, oracle, make_scene('les128'), nb=8, nper=250000, seed=31, nthreads=nthreads)
    assert name.startswith('k_transport_lean')
    _check_images(g, o)


@pytest.mark.parametrize('seed', [41, 20260105])
@pytest.mark.parametrize('views', ['nadir', 'marched'])
def test_config2_les128_mie(solver, oracle, nthreads, seed, views):
    """Config 2's grid with TABULATED phase functions in the cloud -- four Mie-like tables of 498 angles, a real-valued table index per voxel
    (er3t's `apf = index + 1`, rtm/mca/util.py:153; the table-index line of mca_atm.py:275-277) -- held to the PAIRED criterion at full size
    (the bench leg's paired difference read +0.29 % and nothing asserted it): the lean loop's bucket-indexed LDS look-ups
    (`k_transport_lean<.,.,.,2>`) against the oracle's bisection over the whole table, same photon ids, eight batches, two seeds; once through the
    column table (nadir) and once with two marched views (the ray kernel's table look-ups for the local estimate)."""
    import dataclasses
    sc = make_scene('les128_mie')
    if views == 'marched':
        sc = dataclasses.replace(sc, view_the=[180.0-26.1, 180.0-60.0], view_phi=[270.0, 90.0], view_zloc=[sc.view_zloc[0]]*2)
    nper = 300000 if views == 'nadir' else 120000
    g, o, name = _paired(solver, oracle, sc, nb=8, nper=nper, seed=seed, nthreads=nthreads)
    assert name.startswith('k_transport_lean<0,0,0,3>' if views == 'nadir' else 'k_transport_lean<0,0,2,2> + k_rays'), name
    _check_images(g, o)
    from bench import parity_stats
    for q in parity_stats(g, o):
        print('les128_mie %s seed %d view %d: paired %+.3e relative = %+.2f paired se; %+.2f sigma' % (views, seed, q['view'], q['paired_rel_diff'], q['paired_diff_in_paired_se'], q['domain_mean_diff_sigma']))


@pytest.mark.parametrize('general', [False, True])
def test_config4_les480_nadir(solver, oracle, nthreads, general):
    """the bench workload on its own grid, through the lean kernel build and through the general one"""
    g, o, name = _paired(solver, oracle, make_scene('les480'), nb=8, nper=250000, seed=32, nthreads=nthreads, general=general)
    assert name.startswith('k_transport<' if general else 'k_transport_lean<')
    _check_images(g, o)


@pytest.mark.parametrize('build', ['rays', 'general'])
def test_config5_les480_nine_views_lsrt(solver, oracle, nthreads, build):
    """one view from the column table, eight marched with the local-estimate roulette, LSRT surface: through the event lists and
    the ray kernel (default) and through the general kernel"""
    g, o, name = _paired(solver, oracle, make_scene('les480_mv9'), nb=8, nper=40000, seed=33, nthreads=nthreads,
                         general=(build == 'general'))
    assert g.shape[1] == 9
    assert name.startswith({'rays': 'k_transport_lean<0,0,2,0> + k_rays', 'general': 'k_transport<'}[build])
    _check_images(g, o)


@pytest.mark.parametrize('nx', [128, 160, 256, 480, 481], ids=['flux+heating 279 bins', 'flux 329 bins', 'flux 840 bins', 'flux 5000 bins', 'flux+heating 6600 bins'])
def test_tally_records_equal_an_atomic_per_crossing_beyond_256_bins(solver, nx):
    """The record route of flux jobs (sort into bins of 16 384 tally cells, LDS sums) against an atomic per crossing, same photon
    ids, on tallies of more than 256 bins -- more bins than a workgroup of the sort has threads, so that every thread owns two:
    the config-3 grid with heating rates (the heating cells follow the flux cells in the records' index space: 279 bins), a
    160 x 160 grid (329 bins), a 256 x 256 one (840 bins: four per thread) and config 4's 480 x 480 x 100 (5000 bins).  Every record must arrive exactly once: all cells equal to the float32 precision of the output, and
    twice the same.  (A race between the end of one tile and the start of the next in the sort -- a thread zeroing counters another
    thread was still reading -- lost a few tallies in 10^5 on exactly such tallies and went unseen at 210 bins.)"""
    from er3t_amd.synth import les_scene
    from er3t_amd.scene import TARGET_FLUX, TARGET_HEAT
    if nx == 128:
        sc = make_scene('les128_flux')
        sc.target = TARGET_FLUX | TARGET_HEAT
        sc.abs1d = sc.abs1d*30.0 + 2.0e-5
    elif nx >= 480:      # config 4's grid as a flux job: beyond 1024 bins the four waves of a workgroup of the photon loop share one histogram
        from er3t_amd.synth import z_levels_config4
        sc = les_scene(nx=480, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004, target='flux')
        if nx == 481:    # ... with heating rates on top
            sc.target = TARGET_FLUX | TARGET_HEAT
            sc.abs1d = sc.abs1d*30.0 + 2.0e-5
    else:
        sc = les_scene(nx=nx, ny=nx, nz3=50, target='flux', aerosol=True)
    n = 20000000
    solver.bind(None, None, None); solver.load_scene(sc); solver.set_counting(False)
    out = []
    try:
        for lists in (1, 1, 0):
            solver.set_tuning(tally_lists=lists)
            solver.reset(); solver.run(n, seed=7); solver.sync()
            assert ('k_tl_scatter' in solver.kernel_name()) == bool(lists), solver.kernel_name()
            out.append((solver.flux(n).astype(np.float64), solver.heating(n).astype(np.float64) if sc.target & TARGET_HEAT else np.zeros(1)))
        # ... and with the sort and the sums on a stream of their own beside the next launch's photon loop whatever the run's size
        # ("overlap_sort" 2; four launches), two runs of the same photon ids back to back with nothing read in between: the tallies of one run twice, normalised by twice the photons
        if nx in (128, 480, 481):
            solver.set_tuning(tally_lists=1, overlap_sort=2, tl_split=4)
            solver.reset(); solver.run(n, seed=7); solver.run(n, seed=7); solver.sync()
            assert 'k_tl_scatter' in solver.kernel_name()
            out.append((solver.flux(2*n).astype(np.float64), solver.heating(2*n).astype(np.float64) if sc.target & TARGET_HEAT else np.zeros(1)))
    finally:
        solver.set_tuning(tally_lists=1, overlap_sort=1, tl_split=4)
    for f, hh in out[:2] + out[3:]:
        assert np.abs(f-out[2][0]).max() <= 2e-6*out[2][0].max() and np.abs(hh-out[2][1]).max() <= 2e-6*max(out[2][1].max(), 1e-30)
    assert out[2][0].sum() > 0.0 and (nx not in (128, 481) or out[2][1].sum() > 0.0)


def test_config4_grid_flux_job_against_the_oracle(solver, oracle, nthreads):
    """config 4's 480 x 480 x 100 grid as a flux job -- 117 levels, tally indices up to 8e7, 5000 bins: the record route with one
    histogram per workgroup of the photon loop and the 1024-thread sort (bench workload les480_flux) -- against the ORACLE on the same
    photon ids, batch by batch (until round 4 this route was only held against atomics of the same loop).  Per variable and level:
    the domain means within 2 sigma of the difference of two independent estimates + 0.1 % (above the clouds the HIP path adds the
    direct beam analytically where the oracle counts photons: there the difference IS the oracle's noise); paired, the level means
    of the diffuse fluxes within 4 standard errors of the paired difference + 0.03 %; 16 x 16 block means at the surface, the
    cloud top and the top of the atmosphere as in the radiance tests."""
    from bench import parity_stats
    sc = make_scene('les480_flux')
    nb, nper = 8, 250000
    solver.bind(None, None, None); solver.load_scene(sc); solver.set_counting(False)
    lev = (0, 40, sc.nz)                                      # surface, cloud top (1.6 km), top of the atmosphere
    gm, om, gimg, oimg = [], [], [], []
    for b in range(nb):
        solver.reset(); solver.run(nper, seed=34, offset=b*nper); solver.sync()
        g = solver.flux(nper).astype(np.float64)
        o = oracle.run(sc, nper, seed=34, offset=b*nper, nthreads=nthreads)['flux']
        gm.append(g.mean(axis=(2, 3))); om.append(o.mean(axis=(2, 3)))
        gimg.append(g[:, lev].reshape(-1, sc.ny, sc.nx)); oimg.append(o[:, lev].reshape(-1, sc.ny, sc.nx))
    name = solver.kernel_name()
    assert name.startswith('k_transport_flux<') and 'k_tl_scatter' in name, name
    gm, om = np.stack(gm), np.stack(om)                       # (batches, 3 variables, levels)
    d = gm-om
    se_ind = np.sqrt(2.0)*om.std(axis=0, ddof=1)/np.sqrt(nb)
    se_pair = d.std(axis=0, ddof=1)/np.sqrt(nb)
    assert om.shape[1:] == (3, sc.nz+1) and om.mean(axis=0)[2, -1] > 0.05
    assert np.all(np.abs(d.mean(axis=0)) < 2.0*se_ind + 1.0e-3*np.abs(om.mean(axis=0))), np.abs(d.mean(axis=0)/np.maximum(se_ind, 1e-30)).max()
    assert np.all(np.abs(d.mean(axis=0)) < 4.0*se_pair + 3.0e-4*np.abs(om.mean(axis=0)) + 2.0*se_ind*(se_pair > 0.5*se_ind)), \
        (np.abs(d.mean(axis=0)).max(), se_pair.max())          # (last term: the analytic direct beam is not paired with anything)
    for q in parity_stats(np.stack(gimg), np.stack(oimg)):
        if q['mean_oracle'] > 1.0e-3:                          # (the direct beam at the top is the same constant in every block)
            assert q['n_abs_z_ge_4'] <= 1 and q['block_abs_z_max'] < 6.0 and abs(q['block_z_mean']) < 0.5 and q['block_z_std'] < 1.0, q


def test_all_sky_camera_at_the_reference_image_size_against_the_oracle(solver, oracle, nthreads):
    """er3t's all-sky camera (mcarats.py:291-296: on the ground, 178 degree cone) at the reference's image size, 500 x 500 pixels, on
    the config-2 grid (bench workload les128_cam: event lists + the camera build of the ray kernel), against the oracle on the same
    photon ids.  The image is sparse at any affordable photon count (250 000 pixels) and a pixel next to the horizon gets a
    contribution in a thousand photons: the comparison is in 25 x 25 blocks of 20 x 20 pixels, as the satellite images are compared
    in 16 x 16 blocks."""
    sc = make_scene('les128_cam')
    assert (sc.nyr, sc.nxr) == (500, 500)
    g, o, name = _paired(solver, oracle, sc, nb=8, nper=100000, seed=35, nthreads=nthreads)
    assert name.startswith('k_transport_lean<') and name.endswith('+ k_rays'), name
    assert g.shape == (8, 1, 500, 500) and o.mean() > 0.0
    _check_images(g, o, nblk=25, min_block_rel=0.02)      # (the corners of the round image are empty: blocks below 2 % of the image mean are left out)


def test_config4_single_histories(solver, oracle):
    """K7 on the 480 x 480 x 100 grid: one photon id per launch, identical event counts in the HIP path and the oracle for
    at least 85 % of the histories (float32 rounding flips a decision in the others) -- 32-bit voxel offsets, the column
    table and the photon order are exercised at full size; both kernel builds"""
    sc = make_scene('les480')
    keys = ('scatter', 'surface', 'roulette', 'killed', 'escaped', 'absorbed')
    ref = [oracle.run(sc, 1, seed=5, offset=i, nthreads=1)['counters'] for i in range(64)]
    for general in (False, True):
        solver.bind(None, None, None)
        solver.set_kernel(general=general)
        solver.load_scene(sc)
        solver.set_counting(True)
        same = 0
        for i in range(64):
            solver.reset(); solver.run(1, seed=5, offset=i); solver.sync()
            c = solver.counters()
            assert c['photons'] == 1 and c['killed']+c['escaped']+c['absorbed'] == 1
            same += all(c[k] == ref[i][k] for k in keys)
        solver.set_kernel(general=False)
        assert same >= 0.85*64, (general, same)


def test_config3_les128_flux_16g_through_the_dropin(tmp_path, oracle, nthreads):
    """config 3 at full size through the reference's own interface: `mcarats_ng` writes 16 job files (cloud + aerosol side
    file of 26 MB, flux target), every job runs on the GPU; the oracle runs the same 16 jobs from the same files; both sets
    of outputs go through `mca_out_ng`"""
    import er3t_amd.rtm.mca as mca
    from er3t_amd.rtm.mca.mca_exe import get_runner
    from er3t_amd.rtm.mca.mca_out import mca_out_write
    from er3t_amd.scene import Scene
    from er3t_amd.synth import atm_synth, abs_synth, cld_synth, z_levels_config2
    from tests.golden import inputs as gin

    def quiet(fn, *a, **k):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(*a, **k)

    atm = atm_synth(z_levels_config2())
    ab = abs_synth(650.0, atm, Ng=16)
    cld = cld_synth(atm, nx=128, ny=128, nz=50, z_base=0.6, z_top=1.4, cot_mean=10.0, seed=20251003)
    a1 = quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    aer = np.zeros((128, 128, 50)); aer[:, :, 0] = 1.2e-4; aer[:, :, 1] = 0.8e-4      # examples/00_er3t_mca.py:763-772
    a3.add_mca_3d_atm(ext3d=aer, omg3d=np.full_like(aer, 0.85), apf3d=np.full_like(aer, 0.6))
    quiet(a3.gen_mca_3d_atm_file, str(tmp_path/'atm3d.bin'))
    nph = 2000000
    kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=16, target='flux', surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=45.0,
              Nrun=1, photons=nph, weights=ab.coef['weight']['data'], solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    m = quiet(mca.mcarats_ng, fdir=str(tmp_path/'gpu'), **kw)
    kname = get_runner().sol.kernel_name()        # (what served the last job: the lean flux loop with its tally records, not a silent fall-back)
    assert kname.startswith('k_transport_flux<') and 'k_tl_scatter' in kname, kname
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data
    # the oracle on the same job files, its results written in the solver's output format next to them
    mo = quiet(mca.mcarats_ng, fdir=str(tmp_path/'orc'), **dict(kw, mp_mode='sh'))      # job files only ('sh': nothing is run)
    names = [('fdnd', 'direct downward flux density'), ('fdn', 'total downward flux density'), ('fup', 'upward flux density')]
    se2 = np.zeros((3, a1.nml[0]['Atm_nz']['data']+1))
    for ig in range(16):
        nml = mca.mca_inp_read(m.fnames_inp[0][ig])
        sc = Scene.from_nml(nml, os.path.dirname(m.fnames_inp[0][ig]), solver=0)
        n = int(m.photons[ig])
        r = oracle.run(sc, n, seed=int(nml['Wld_jseed']), nthreads=nthreads)
        f = r['flux']
        mca_out_write(mo.fnames_out[0][ig], [(nm, d, np.transpose(f[i], (2, 1, 0))) for i, (nm, d) in enumerate(names)])
    outo = mca.mca_out_ng(mca_obj=mo, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data
    toa = out['toa']['data']
    for v in ('f_down_direct', 'f_down', 'f_up'):
        a = out[v]['data'].mean(axis=(0, 1)); b = outo[v]['data'].mean(axis=(0, 1))       # per level
        # Monte-Carlo error of a domain-mean flux from n photons of weight <= 1: below toa*mu0/sqrt(n) per job; the g-sum
        # of 16 jobs of 2e6/16 photons each cannot be noisier than one job of 2e6/16 photons
        sigma = toa*np.cos(np.deg2rad(30.0))/np.sqrt(nph/16.0)
        assert a.shape == b.shape == (69,)
        assert np.all(np.abs(a-b) < 2.0*np.sqrt(2.0)*sigma + 1.0e-3*np.abs(b)), (v, np.abs(a-b).max(), sigma)
    # energy: what goes down at the top is the source
    assert np.isclose(out['f_down']['data'].mean(axis=(0, 1))[-1], toa*np.cos(np.deg2rad(30.0)), rtol=1e-3)


def test_config3_les128_radiance_16g_through_the_dropin(tmp_path, oracle, nthreads):
    """config 3's radiance leg at full size through the reference's own interface: cloud + 3-D aerosol (np3d = 2, a side file of
    26 MB), nadir view, 16 g x 4 runs = 64 jobs of `mcarats_ng` on the GPU (lean kernel build for two 3-D constituents); the
    oracle runs the same 64 job files; both sets of outputs go through `mca_out_ng(mode='all')`, and the four runs take the place
    of the batches of the other full-size tests (same seeds on both sides: the differences are paired)"""
    import er3t_amd.rtm.mca as mca
    from er3t_amd.rtm.mca.mca_exe import get_runner
    from er3t_amd.rtm.mca.mca_out import mca_out_write
    from er3t_amd.scene import Scene
    from er3t_amd.synth import atm_synth, abs_synth, cld_synth, z_levels_config2
    from tests.golden import inputs as gin

    def quiet(fn, *a, **k):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(*a, **k)

    atm = atm_synth(z_levels_config2())
    ab = abs_synth(650.0, atm, Ng=16)
    cld = cld_synth(atm, nx=128, ny=128, nz=50, z_base=0.6, z_top=1.4, cot_mean=10.0, seed=20251003)
    a1 = quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    aer = np.zeros((128, 128, 50)); aer[:, :, 0] = 1.2e-4; aer[:, :, 1] = 0.8e-4      # examples/00_er3t_mca.py:763-772
    a3.add_mca_3d_atm(ext3d=aer, omg3d=np.full_like(aer, 0.85), apf3d=np.full_like(aer, 0.6))
    quiet(a3.gen_mca_3d_atm_file, str(tmp_path/'atm3d.bin'))
    nrun, nph = 4, 500000
    kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=16, target='radiance', surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=45.0,
              sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, Nrun=nrun, photons=nph, weights=ab.coef['weight']['data'], solver='3D',
              mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    m = quiet(mca.mcarats_ng, fdir=str(tmp_path/'gpu'), **kw)
    assert get_runner().sol.kernel_name().startswith('k_transport_lean<'), get_runner().sol.kernel_name()
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='all', squeeze=True, quiet=True).data
    mo = quiet(mca.mcarats_ng, fdir=str(tmp_path/'orc'), **dict(kw, mp_mode='sh'))      # the same job files ('sh': nothing is run)
    for ir in range(nrun):
        for ig in range(16):
            nml = mca.mca_inp_read(m.fnames_inp[ir][ig])                                  # (the GPU run's files: its seeds)
            sc = Scene.from_nml(nml, os.path.dirname(m.fnames_inp[ir][ig]), solver=0)
            assert sc.np3d == 2
            r = oracle.run(sc, int(m.photons[ir*16+ig]), seed=int(nml['Wld_jseed']), nthreads=nthreads)
            mca_out_write(mo.fnames_out[ir][ig], [('rad', 'pixel-averaged radiance', np.transpose(r['rad'], (2, 1, 0)))])
    outo = mca.mca_out_ng(mca_obj=mo, abs_obj=ab, mode='all', squeeze=True, quiet=True).data
    g = np.transpose(out['rad']['data'], (2, 1, 0))[:, None]          # (nx, ny, nrun) -> (nrun, 1, ny, nx)
    o = np.transpose(outo['rad']['data'], (2, 1, 0))[:, None]
    assert g.shape == o.shape == (nrun, 1, 128, 128)
    _check_images(g.astype(np.float64), o.astype(np.float64))
