"""
Host layer against the golden fixtures recorded from the REFERENCE's own code (tests/golden/make_golden.py):
scalars, adapter arrays, byte images of the side files, namelist text, and the g/run reduction of outputs.
Text and bytes must be identical; float32 results must be equal.
"""

import datetime
import io
import json
import os
import contextlib

import numpy as np
import pytest

import er3t_amd.rtm.mca as mca
import er3t_amd.util as util
from er3t_amd.scene import Scene
from tests.golden import inputs as gin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def scalars():
    with open(os.path.join(GOLD, 'golden_scalars.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def built(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp('adapters'))
    inp = gin.make_inputs()
    ad = gin.build_adapters(mca, inp, tmp)
    return tmp, inp, ad


# ---------------------------------------------------------------------------------------------
def test_cal_mca_azimuth(scalars):
    g = scalars['cal_mca_azimuth']
    assert [mca.cal_mca_azimuth(a) for a in g['in']] == g['out']


def test_distribute_photon(scalars):
    for case in scalars['distribute_photon']:
        out = mca.distribute_photon(case['N'], np.array(case['w']), base_ratio=case['base_ratio'])
        assert [int(v) for v in out] == case['out']
        assert out.sum() == case['N']
    # the vector embedded in the reference's own test script (tests/00_test_util.py:249-252)
    assert scalars['distribute_photon'][0]['out'] == [14824075, 14483931, 13811633, 12822922, 11540979, 9995858, 8223786, 6266341,
                                                      4349287, 752063, 676158, 599970, 523397, 446830, 363135, 319635]


def test_rearrange_jobs(scalars):
    for case in scalars['rearrange_jobs']:
        out = mca.rearrange_jobs(case['Ncpu'], np.array(case['w']))
        assert [int(v) for v in out] == case['out']
        assert sorted(int(v) for v in out) == list(range(len(case['w'])))


def test_small_helpers(scalars):
    g = scalars['cal_sol_fac']
    assert [float(util.cal_sol_fac(datetime.datetime.strptime(d, '%Y-%m-%d'))) for d in g['in']] == g['out']
    for case in scalars['cal_r_twostream']:
        got = util.cal_r_twostream(np.array(case['tau']), a=case['a'], g=case['g'], mu=case['mu'])
        assert [float(v) for v in got] == case['out']
    g = scalars['get_lay_index']
    assert [int(v) for v in util.get_lay_index(np.array(g['lay']), np.array(g['lay_ref']))] == g['out']
    with pytest.raises(ValueError):
        util.get_lay_index(np.array([30.0]), np.array(g['lay_ref']))
    for case in scalars['nice_array_str']:
        assert util.nice_array_str(np.array(case['in'])) == case['out']
    with pytest.raises(ValueError):
        util.nice_array_str(np.zeros((2, 2)))


def test_rayleigh_extinction(scalars, built):
    tmp, inp, ad = built
    tau = np.asarray(ad['a1'].nml[0]['Atm_ext1d(1:, 1)']['data'])*inp['atm'].lay['thickness']['data']*1000.0
    assert np.array_equal(tau, np.array(scalars['cal_mol_ext']['out']))


def test_adapter_arrays(built):
    tmp, inp, ad = built
    got = gin.adapter_arrays(ad)
    with np.load(os.path.join(GOLD, 'adapters.npz')) as want:
        assert sorted(want.files) == sorted(got.keys())
        for key in want.files:
            w, g = want[key], np.asarray(got[key])
            assert w.shape == g.shape and w.dtype == g.dtype, key
            assert np.array_equal(w, g), key
    # the reference's quirk is reproduced: first cloudy layer is 0-based index 1, handed over as 3
    assert ad['a3'].nml['Atm_iz3l']['data'] == 3


def test_side_file_bytes(built):
    tmp, inp, ad = built
    for src, dst in gin.SIDE_FILES.items():
        with open(os.path.join(tmp, src), 'rb') as f, open(os.path.join(GOLD, dst), 'rb') as g:
            assert f.read() == g.read(), src


def test_adapter_errors():
    with pytest.raises(OSError):
        mca.mca_atm_1d(atm_obj=None, abs_obj=None)
    with pytest.raises(OSError):
        mca.mca_atm_3d(atm_obj=None, cld_obj=None)
    with pytest.raises(OSError):
        mca.mca_sca(pha_obj=None)
    with pytest.raises(OSError):
        mca.mca_sfc_2d(atm_obj=None, sfc_obj=None)


# ---------------------------------------------------------------------------------------------
def _mask(text, tmp):
    lines = [' Wld_jseed       = <masked>' if l.startswith(' Wld_jseed') else l for l in text.split('\n')]
    return '\n'.join(lines).replace(tmp, '<fdir>')


@pytest.fixture(scope='module')
def simulations(built, scalars):
    tmp, inp, ad = built
    cases = gin.simulation_cases(ad['a1'], ad['a1b'], ad['a3'], ad['a3b'], ad['sca'], ad['s_l'], ad['s_b'], inp['abs'].coef['weight']['data'])
    objs = {}
    cwd = os.getcwd()
    os.chdir(tmp)             # 'sh' mode drops its batch script into the working directory
    try:
        for name, kw in cases.items():
            with contextlib.redirect_stdout(io.StringIO()):
                objs[name] = mca.mcarats_ng(fdir='%s/%s' % (tmp, name), Nrun=2, Ncpu=2, mp_mode='sh', overwrite=True, date=gin.DATE, quiet=True, **kw)
    finally:
        os.chdir(cwd)
    return objs


def test_namelist_text_identical(simulations, built, scalars):
    tmp = built[0]
    for name, m in simulations.items():
        for ig in (0, 15):
            got = _mask(open(m.fnames_inp[1][ig]).read(), tmp)
            want = open(os.path.join(GOLD, 'nml_%s_g%02d.txt' % (name, ig))).read()
            assert got == want, (name, ig)
        g = scalars['mcarats_ng_%s' % name]
        assert [int(v) for v in m.photons] == g['photons'] and int(m.Nx) == g['Nx'] and int(m.Ny) == g['Ny']
        assert m.solver == g['solver'] and m.target == g['target'] and int(m.photons_per_set) == g['photons_per_set'] and m.np_mode == g['np_mode']
        # every job has its own seed
        seeds = {mca.mca_inp_read(f)['Wld_jseed'] for row in m.fnames_inp for f in row}
        assert len(seeds) == m.Nrun*m.Ng


def test_namelist_round_trip_to_scene(simulations, built):
    """what the writer emits, the reader and Scene.from_nml take back: the solver side of the file interface"""
    tmp, inp, ad = built
    m = simulations['rad_3d_sca_sfc']
    nml = mca.mca_inp_read(m.fnames_inp[0][3])
    assert nml['Atm_np1d'] == 2 and nml['Atm_iz3l'] == 3 and nml['Sca_npf'] == 3 and nml['Sfc_nxb'] == 3
    assert nml['Sca_inpfile'] == '../sca.bin' and isinstance(nml['Atm_zgrd0'], np.ndarray) and nml['Atm_zgrd0'].size == 21
    sc = Scene.from_nml(nml, os.path.dirname(m.fnames_inp[0][3]), solver=2)
    assert (sc.nx, sc.ny, sc.nz3, sc.iz3l, sc.np3d, sc.np1d, sc.npf) == (3, 2, 2, 3, 2, 2, 3)
    assert sc.jsfc is not None and np.all(sc.jsfc == 4.0) and sc.psfc.shape == (5, 2, 3)
    # file layout: x fastest -> Scene arrays are the transposes of the adapter's (nx, ny, nz3, np) arrays
    ext = ad['a3b'].nml['Atm_extp3d']['data']
    assert np.array_equal(sc.extp, np.transpose(ext, (3, 2, 1, 0)).astype(np.float32))
    assert np.array_equal(sc.apfp, np.transpose(ad['a3b'].nml['Atm_apfp3d']['data'], (3, 2, 1, 0)).astype(np.float32))
    assert np.allclose(sc.ext1d[1], np.float32(ad['a1b'].nml[3]['Atm_ext1d(1:, 2)']['data']), rtol=1e-6)
    assert sc.view_the == [153.9] and sc.view_phi == [90.0] and sc.src_the == 138.5 and sc.src_phi == 70.0
    assert np.allclose(sc.pha, inp['pha'].data['pha']['data'].T.astype(np.float32))


def test_unknown_namelist_key_is_rejected(tmp_path):
    with pytest.raises(OSError):
        mca.mca_inp_file(str(tmp_path/'x.txt'), {'Atm_nonsense': 1}, comment=False)
    with pytest.raises(ValueError):
        mca.mca_inp_file(str(tmp_path/'y.txt'), {'Atm_nx': [1, 2]}, comment=False)


def test_constructor_errors(built):
    tmp, inp, ad = built
    kw = dict(atm_1ds=[ad['a1']], fdir='%s/err' % tmp, Nrun=1, mp_mode='sh', quiet=True)
    with pytest.raises(OSError):
        mca.mcarats_ng(solver='4d', **kw)
    with pytest.raises(OSError):
        mca.mcarats_ng(target='colour', **kw)
    with pytest.raises(OSError):
        mca.mcarats_ng(Ncpu=0, **kw)
    with pytest.raises(OSError):
        mca.mcarats_ng(atm_1ds=[], fdir='%s/err' % tmp, Nrun=1, mp_mode='sh', quiet=True)
    with pytest.raises(ValueError):
        mca.mcarats_ng(surface_albedo='green', **kw)
    # reading mode with nothing to read: the reference's "Missing some output files"
    with pytest.raises(OSError):
        mca.mcarats_ng(atm_1ds=[ad['a1']], fdir='%s/empty' % tmp, Nrun=1, overwrite=False, quiet=True)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['rad_3d_hg', 'flux_1d', 'flux0_3d'])
def test_output_reduction_equal(simulations, built, name):
    tmp, inp, ad = built
    m = simulations[name]
    with np.load(os.path.join(GOLD, 'out_%s.npz' % name)) as want:
        # lay down the synthetic solver outputs the reference read
        for ir in range(m.Nrun):
            for ig in range(m.Ng):
                want['bin_r%d_g%d' % (ir, ig)].astype('<f4').tofile(m.fnames_out[ir][ig])
                with open(m.fnames_out[ir][ig]+'.ctl', 'w') as f:
                    f.write(str(want['ctl']))
        raw = mca.mca_out_raw(m.fnames_out[1][3])
        for i, d in enumerate(raw.data):
            assert np.array_equal(d['data'], want['raw_%d' % i])
            assert d['name'] == str(want['raw_names'][i]) and d['dims_info'] == ['Nx', 'Ny', 'Nz', 'Nt']
        fn = mca.read_radiance_mca_out if name.startswith('rad') else mca.read_flux_mca_out
        for mode in ('mean', 'all'):
            for squeeze in (True, False):
                d = fn(m, inp['abs'], mode=mode, squeeze=squeeze)
                keys = [k[len('%s_sq%d_' % (mode, int(squeeze))):] for k in want.files
                        if k.startswith('%s_sq%d_' % (mode, int(squeeze))) and not k.endswith('_dims')]
                assert sorted(keys) == sorted(d.keys())
                for key in keys:
                    w = want['%s_sq%d_%s' % (mode, int(squeeze), key)]
                    g = np.asarray(d[key]['data'])
                    assert w.shape == g.shape and w.dtype == g.dtype, (mode, squeeze, key)
                    assert np.array_equal(w, g), (mode, squeeze, key)
                    if 'dims_info' in d[key]:
                        assert list(want['%s_sq%d_%s_dims' % (mode, int(squeeze), key)]) == list(d[key]['dims_info'])
        with pytest.raises(OSError):
            fn(m, inp['abs'], mode='median')
    # mca_out_ng: objects only / objects + cache / cache only
    o1 = mca.mca_out_ng(mca_obj=m, abs_obj=inp['abs'], mode='mean', squeeze=True, quiet=True)
    fcache = '%s/%s_cache.npz' % (tmp, name)
    o2 = mca.mca_out_ng(fname=fcache, mca_obj=m, abs_obj=inp['abs'], mode='mean', squeeze=True, quiet=True, overwrite=True)
    o3 = mca.mca_out_ng(fname=fcache, mode='mean', quiet=True)
    for key in o1.data:
        assert np.array_equal(np.asarray(o1.data[key]['data']), np.asarray(o2.data[key]['data']))
        assert np.array_equal(np.asarray(o1.data[key]['data']), np.asarray(o3.data[key]['data']))
    with pytest.raises(OSError):
        mca.mca_out_ng()


def test_hdf5_cache_layout_is_the_references(tmp_path):
    """`mca_out_ng.dump()` to an .h5 name writes what the reference writes (er3t/rtm/mca/mca_out.py:209-233: one group named after the
    mode; a gzip-compressed, chunked dataset per array, a scalar dataset per number; every other item of the key as an attribute,
    `dims_info` as a byte string), so that the reference's own readers (`f['mean/rad'][...]`, er3t/rtm/mca/util.py:82-88) take
    it.  Needs h5py (not in this image: skipped there); without it an .h5 name must fail loudly, never fall back to another format."""
    data = {'rad': {'data': np.arange(12, dtype=np.float32).reshape(3, 4), 'name': 'Radiance', 'units': 'W/m^2/nm/sr', 'dims_info': ['Nx', 'Ny']},
            'rad_std': {'data': np.ones((3, 4), dtype=np.float32), 'name': 'Radiance standard deviation', 'units': 'W/m^2/nm/sr', 'dims_info': ['Nx', 'Ny']},
            'toa': {'data': 1.5, 'name': 'TOA without SZA', 'units': 'W/m^2/nm'},
            'N_photon': {'data': np.array([100, 200]), 'name': 'Number of photons', 'units': 'N/A'}}
    o = mca.mca_out_ng.__new__(mca.mca_out_ng)
    o.mode, o.quiet, o.verbose, o.fname, o.data = 'mean', True, False, str(tmp_path/'out.h5'), data

    class _T:
        target = 'radiance'
    o.mca = _T()
    try:
        import h5py
    except ImportError:
        with pytest.raises(OSError, match='h5py'):
            o.dump()
        assert not os.path.exists(o.fname)
        pytest.skip('h5py is not installed: the HDF5 layout cannot be written here')
    o.dump()
    with h5py.File(o.fname, 'r') as f:
        assert list(f.keys()) == ['mean'] and sorted(f['mean'].keys()) == sorted(data.keys())
        for key, item in data.items():
            d = f['mean/%s' % key]
            assert np.array_equal(d[...], item['data'])
            if isinstance(item['data'], np.ndarray):
                assert d.compression == 'gzip' and d.compression_opts == 9 and d.chunks is not None
            else:
                assert d.shape == ()
            assert sorted(d.attrs.keys()) == sorted(k for k in item if k != 'data')
            assert d.attrs['name'] == item['name'] and d.attrs['units'] == item['units']
            if 'dims_info' in item:
                assert isinstance(d.attrs['dims_info'], (bytes, np.bytes_)) and d.attrs['dims_info'] == np.bytes_(str(item['dims_info']))
        assert float(f['mean/rad'][...].mean()) == 5.5            # (what func_ref_vs_cot.load_all of the reference reads)
    back = mca.mca_out_ng(fname=o.fname, mode='mean', quiet=True)
    assert np.array_equal(back.data['rad']['data'], data['rad']['data']) and float(back.data['toa']['data']) == 1.5


def test_output_writer_reader_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    a = rng.uniform(size=(4, 3, 5)).astype(np.float32); b = rng.uniform(size=(4, 3, 1)).astype(np.float32)
    f = str(tmp_path/'x.out.bin')
    mca.mca_out_write(f, [('fdn', 'total downward flux density', a), ('rad', 'pixel-averaged radiance', b)])
    raw = mca.mca_out_raw(f)
    assert raw.Nvar == 2 and raw.data[0]['dims'] == [4, 3, 5, 1] and raw.data[1]['dims'] == [4, 3, 1, 1]
    assert np.array_equal(raw.data[0]['data'][..., 0], a) and np.array_equal(raw.data[1]['data'][..., 0], b)
    with pytest.raises(OSError):
        mca.mca_out_raw(str(tmp_path/'missing.bin'))


def test_ab_input_sets_parse_and_are_complete():
    """the committed input sets of the MCARaTS A/B run (tests/golden/ab/, tools/ab_mcarats.sh): every job file parses back
    into a scene together with its side files, three runs per case, fixed and distinct seeds"""
    from er3t_amd.rtm.mca.mca_inp import mca_inp_read
    from er3t_amd.scene import Scene
    ab = os.path.join(GOLD, 'ab')
    cases = sorted(d for d in os.listdir(ab) if os.path.isdir(os.path.join(ab, d)))
    assert cases == ['c2_nadir', 'c2_slant', 'c3_flux', 'c4_absorb', 'c5_lsrt', 'c6_sea', 'c7_allsky']
    for case in cases:
        d = os.path.join(ab, case)
        assert sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) <= 1 << 20
        seeds = []
        for ir in range(3):
            nml = mca_inp_read(os.path.join(d, 'r%02d.g000.inp.txt' % ir))
            sc = Scene.from_nml(nml, d, solver=0)
            assert (sc.nx, sc.ny, sc.nz3) == (32, 32, 20)
            seeds.append(int(nml['Wld_jseed']))
        assert len(set(seeds)) == 3
        assert (sc.np3d == 2) == (case == 'c3_flux')
        assert (sc.rad_kind == 1) == (case == 'c7_allsky')
        if case == 'c6_sea':
            assert np.all(sc.jsfc == 2.0)
