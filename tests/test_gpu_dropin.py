"""
The drop-in boundary end to end on the GPU: `mcarats_ng` writes the reference's input files, runs every (run, g) job
through libmi3drt.so, leaves MCARaTS-format outputs, and `mca_out_ng` reduces them -- checked against the oracle run on
the very same input files (parsed back with mca_inp_read / Scene.from_nml).
"""

import contextlib
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import er3t_amd.rtm.mca as mca
from er3t_amd.scene import Scene
from er3t_amd.synth import atm_synth, abs_synth, cld_synth, pha_hg_synth, sfc_lsrt_synth, sfc_dsm_synth
from er3t_amd.util import cal_sol_fac
from tests.golden import inputs as gin

pytestmark = pytest.mark.gpu


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _atm(levels):
    atm = atm_synth(levels)
    atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
    atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
    return atm


def _oracle_job(oracle, fname_inp, nphoton, solver, nthreads):
    nml = mca.mca_inp_read(fname_inp)
    sc = Scene.from_nml(nml, os.path.dirname(fname_inp), solver=solver)
    return sc, oracle.run(sc, nphoton, seed=int(nml['Wld_jseed']), nthreads=nthreads)


def test_config1_clear_sky_flux_16g(tmp_path, oracle, nthreads):
    """BASELINE config 1: the examples/00_er3t_mca.py flux case shape -- 1-D clear sky, 16 g, 3 runs, 1e5 photons"""
    atm = _atm(np.linspace(0.0, 20.0, 21))
    ab = abs_synth(650.0, atm, Ng=16)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    date = gin.DATE
    m = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[], Ng=16, target='flux', surface_albedo=0.03, solar_zenith_angle=30.0,
               solar_azimuth_angle=0.0, fdir=str(tmp_path/'c1'), Nrun=3, weights=ab.coef['weight']['data'], photons=1e5,
               solver='3D', Ncpu=12, mp_mode='py', overwrite=True, date=date, quiet=True)
    assert m.photons.sum() == 3*100000 and len(m.fnames_out) == 3 and len(m.fnames_out[0]) == 16
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True)
    f_down = out.data['f_down']['data']; f_up = out.data['f_up']['data']; f_dir = out.data['f_down_direct']['data']
    assert f_down.shape == (21,) and out.data['f_down']['dims_info'] == ['Nz']
    toa = out.data['toa']['data']
    mu0 = np.cos(np.deg2rad(30.0))
    assert np.isclose(toa, cal_sol_fac(date)*np.sum(ab.coef['solar']['data']*ab.coef['weight']['data']))
    assert np.isclose(f_down[-1], toa*mu0, rtol=1e-4)                       # every photon enters through the top
    assert np.all(f_dir <= f_down+1e-6) and np.all(np.diff(f_dir) >= -1e-6)  # the direct beam only attenuates downwards
    assert np.allclose(out.data['f_down_diffuse']['data'], f_down-f_dir, atol=1e-6)
    # one job against the oracle on the same file
    raw = mca.mca_out_raw(m.fnames_out[1][9])
    sc, o = _oracle_job(oracle, m.fnames_inp[1][9], int(m.photons[16+9]), 0, nthreads)
    n = int(m.photons[16+9])
    for iv in range(3):
        got = raw.data[iv]['data'][0, 0, :, 0]
        assert np.all(np.abs(got-o['flux'][iv, :, 0, 0]) < 6.0*np.sqrt(1.0/n) + 1e-3), iv
    # reading mode re-uses the files without touching the GPU; the batch-script mode writes the reference's command shape
    m2 = _quiet(mca.mcarats_ng, atm_1ds=[a1], Ng=16, target='flux', fdir=str(tmp_path/'c1'), Nrun=3, photons=1e5,
                weights=ab.coef['weight']['data'], overwrite=False, date=date, quiet=True)
    out2 = mca.mca_out_ng(mca_obj=m2, abs_obj=ab, mode='mean', squeeze=True, quiet=True)
    assert np.array_equal(out2.data['f_up']['data'], f_up)


def test_3d_radiance_through_the_api_and_cli(tmp_path, oracle, nthreads):
    """config-3 shape at toy size: 3-D cloud + aerosol component, tabulated phase functions, LSRT surface, slant view, 2 g"""
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=2)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
    pha = pha_hg_synth()
    fdir = str(tmp_path/'sim')
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, pha_obj=pha, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    aer = np.zeros((12, 10, 10)); aer[:, :, 0] = 1.2e-4; aer[:, :, 1] = 0.8e-4
    a3.add_mca_3d_atm(ext3d=aer, omg3d=np.full_like(aer, 0.85), apf3d=np.full_like(aer, 0.6))
    _quiet(a3.gen_mca_3d_atm_file, str(tmp_path/'atm3d.bin'))
    sca = _quiet(mca.mca_sca, pha_obj=pha, fname=str(tmp_path/'sca.bin'), quiet=True)
    sfc = _quiet(mca.mca_sfc_2d, atm_obj=atm, sfc_obj=sfc_lsrt_synth(12, 10), fname=str(tmp_path/'sfc.bin'), quiet=True)
    nph = 400000
    m = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], sca=sca, Ng=2, target='radiance', surface_albedo=sfc,
               solar_zenith_angle=35.0, solar_azimuth_angle=120.0, sensor_zenith_angle=26.1, sensor_azimuth_angle=180.0,
               fdir=fdir, Nrun=2, photons=nph, solver='3D', Ncpu=2, mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    assert (m.Nx, m.Ny) == (12, 10) and m.dx == 100.0
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True)
    rad = out.data['rad']['data']
    assert rad.shape == (12, 10) and out.data['rad']['dims_info'] == ['Nx', 'Ny'] and np.all(rad > 0.0)
    assert np.all(out.data['rad_std']['data'] < 0.5*rad.max())
    # one job against the oracle on the same input file
    job = (1, 1)
    n = int(m.photons[2+1])
    raw = mca.mca_out_raw(m.fnames_out[job[0]][job[1]]).data[0]['data'][:, :, 0, 0]        # (nx, ny)
    sc, o = _oracle_job(oracle, m.fnames_inp[job[0]][job[1]], n, 0, nthreads)
    orad = o['rad'][0].T
    assert abs(raw.mean()-orad.mean()) < 0.02*orad.mean()
    assert np.corrcoef(raw.ravel(), orad.ravel())[0, 1] > 0.8
    # the same job through the solver's command line (a separate process, like the reference's os.system call)
    fout = str(tmp_path/'cli.out.bin')
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-m', 'er3t_amd.rtm.mca.mca_exe', str(n), '0', m.fnames_inp[job[0]][job[1]], fout],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cli = mca.mca_out_raw(fout).data[0]['data'][:, :, 0, 0]
    assert np.allclose(cli, raw, rtol=2e-3, atol=1e-6)      # same seed, same photons: float32 atomics order only
    # unsupported requests surface as the reference's error type
    r = subprocess.run([sys.executable, '-m', 'er3t_amd.rtm.mca.mca_exe', '1000', '7', m.fnames_inp[0][0], fout],
                       env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'solver=7' in r.stderr
    with pytest.raises(OSError):
        _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], Ng=2, target='actinic flux', fdir=str(tmp_path/'hr'), Nrun=1,
               photons=1000, mp_mode='py', quiet=True)


def test_several_views_in_one_simulation(tmp_path):
    """Not in the reference (one view per simulation, mcarats.py:301): sequences of sensor angles -> Rad_nrad views in every job, one set of
    photon histories for all of them, the views as the third axis of the radiance.  Each view must be what a simulation of its own gives:
    the nadir view (answered from the column table either way, same seeds -> same histories) to float32 rounding, the slant views (their
    rays' roulettes are drawn per view number) within the Monte-Carlo noise; through the files and through the fused statistics alike."""
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=2)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    vza, vaa = [0.0, 26.1, 45.6], [0.0, 0.0, 180.0]
    kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=2, target='radiance', surface_albedo=0.05, solar_zenith_angle=35.0, solar_azimuth_angle=120.0,
              Nrun=2, photons=600000, solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    m = _quiet(mca.mcarats_ng, fdir=str(tmp_path/'all'), sensor_zenith_angle=vza, sensor_azimuth_angle=vaa, **kw)
    assert m.Nview == 3 and m.nml[0]['Rad_nrad'] == 3 and np.allclose(m.nml[0]['Rad_the'], 180.0-np.array(vza))
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data
    rad = out['rad']['data']
    assert rad.shape == (12, 10, 3) and out['rad']['dims_info'] == ['Nx', 'Ny', 'Nz'] and np.all(rad > 0.0)
    for iv in range(3):
        m1 = _quiet(mca.mcarats_ng, fdir=str(tmp_path/('one%d' % iv)), sensor_zenith_angle=vza[iv], sensor_azimuth_angle=vaa[iv], **kw)
        assert m1.Nview == 1 and 'Rad_nrad' in m1.nml[0] and m1.nml[0]['Rad_nrad'] == 1
        r1 = mca.mca_out_ng(mca_obj=m1, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data['rad']['data']
        assert r1.shape == (12, 10)
        if iv == 0:
            assert np.allclose(rad[:, :, 0], r1, rtol=1e-4, atol=1e-7*r1.max())          # same seeds, same histories, the column table
        else:
            assert abs(rad[:, :, iv].mean()/r1.mean()-1.0) < 0.02 and np.corrcoef(rad[:, :, iv].ravel(), r1.ravel())[0, 1] > 0.9
    # the fused statistics give what the files give (same seeds)
    mf = _quiet(mca.mcarats_ng, fdir=str(tmp_path/'fused'), sensor_zenith_angle=vza, sensor_azimuth_angle=vaa, abs_obj=ab, keep_files=False, **kw)
    rf = mca.mca_out_ng(mca_obj=mf, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data['rad']['data']
    assert rf.shape == rad.shape and np.allclose(rf[:, :, 0], rad[:, :, 0], rtol=1e-4, atol=1e-7*rad.max())
    assert np.all(np.abs(rf.mean(axis=(0, 1))/rad.mean(axis=(0, 1))-1.0) < 0.02)
    with pytest.raises(OSError):
        _quiet(mca.mcarats_ng, fdir=str(tmp_path/'bad'), sensor_zenith_angle=[0.0, 10.0], sensor_azimuth_angle=[0.0, 1.0, 2.0], **kw)


@pytest.mark.real_clock          # (identities between two routes through the same job files: they hold under any seed -- the wall clock's, as in production)
def test_fused_g_loop_and_run_statistics_equal_the_file_route(tmp_path):
    """row f3: sum over g per run and mean / std over runs gathered on the device while the jobs run
    (mca_out.py:313-352, 438-500 semantics) against the reference's route through Nrun*Ng output files"""
    import copy
    from er3t_amd.scene import TARGET_FLUX, TARGET_RADIANCE
    from er3t_amd.rtm.mca.mca_exe import get_runner
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=4)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    for target, keys in (('radiance', ['rad']), ('flux', ['f_up', 'f_down', 'f_down_direct', 'f_down_diffuse'])):
        m = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], Ng=4, target=target, surface_albedo=0.05, solar_zenith_angle=40.0,
                   fdir=str(tmp_path/target), Nrun=3, photons=2e5, weights=ab.coef['weight']['data'], solver='3D', mp_mode='py',
                   overwrite=True, date=gin.DATE, quiet=True, abs_obj=ab, keep_files=True)
        assert m.fused is not None and all(os.path.exists(f) for row in m.fnames_out for f in row)
        files = copy.copy(m); files.fused = None
        for mode in ('mean', 'all'):
            a = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode=mode, squeeze=True, quiet=True).data
            b = mca.mca_out_ng(mca_obj=files, abs_obj=ab, mode=mode, squeeze=True, quiet=True).data
            assert sorted(a.keys()) == sorted(b.keys())
            for k in keys + ([k+'_std' for k in keys] if mode == 'mean' else []):
                assert a[k]['dims_info'] == b[k]['dims_info'] and a[k]['data'].shape == b[k]['data'].shape, (k, mode)
                assert np.array_equal(a[k]['data'], b[k]['data']), (target, mode, k)       # same float32 operations in the same order
            assert a['toa']['data'] == b['toa']['data']
        # the device's own sum / sum-of-squares statistics (float64) against numpy on the per-run fields
        key = 'rad' if target == 'radiance' else 'flux'
        st = m.fused[key]
        assert st['nrun'] == 3 and st['runs'].shape == st['mean'].shape + (3,)
        assert np.allclose(st['mean'], st['runs'].astype(np.float64).mean(axis=-1), rtol=1e-6, atol=1e-12)
        assert np.allclose(st['std'], st['runs'].astype(np.float64).std(axis=-1), rtol=2e-3, atol=1e-6*st['mean'].max())
        # the same jobs again with the two solver handles taking turns (no per-job files: job i+1 is launched before job i is
        # folded into its run): the run fields are summed in job order all the same -- equal to float64-atomic order
        m1 = copy.copy(m); m1.keep_files = False
        assert get_runner().use_slots(2) == 2
        _quiet(m1.run_fused)
        for q in ('mean', 'std', 'runs'):
            assert m1.fused[key][q].shape == st[q].shape
            assert np.allclose(m1.fused[key][q], st[q], rtol=2e-5, atol=2e-6*st['mean'].max()), (target, q)
        # without files: nothing is written, the reader still works
        m2 = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], Ng=4, target=target, surface_albedo=0.05, solar_zenith_angle=40.0,
                    fdir=str(tmp_path/(target+'_nofiles')), Nrun=2, photons=1e5, weights=ab.coef['weight']['data'], solver='3D',
                    mp_mode='py', overwrite=True, date=gin.DATE, quiet=True, abs_obj=ab, keep_files=False)
        assert not any(os.path.exists(f) for row in m2.fnames_out for f in row)
        d = mca.mca_out_ng(mca_obj=m2, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data
        assert np.all(np.isfinite(d[keys[0]]['data'])) and d[keys[0]]['data'].mean() > 0.0
    # the C-ABI refuses statistics calls out of order
    sol = get_runner().sol
    sol.stats_begin()
    with pytest.raises(OSError):
        sol.stats_get(TARGET_FLUX)              # no run closed yet


def test_up_looking_sensor_through_the_driver(tmp_path):
    """er3t's sensor_zenith_angle > 90 ("looking up, 180 straight up", mcarats.py:499-502) with the sensor on the ground"""
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=2)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=2, target='radiance', surface_albedo=0.05, solar_zenith_angle=40.0, Nrun=2, photons=4e5,
              weights=ab.coef['weight']['data'], solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    up = _quiet(mca.mcarats_ng, fdir=str(tmp_path/'up'), sensor_zenith_angle=180.0, sensor_altitude=0.0, **kw)
    dn = _quiet(mca.mcarats_ng, fdir=str(tmp_path/'dn'), sensor_zenith_angle=0.0, **kw)
    assert up.nml[0]['Rad_the'] == 0.0 and dn.nml[0]['Rad_the'] == 180.0
    r_up = mca.mca_out_ng(mca_obj=up, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data['rad']['data']
    r_dn = mca.mca_out_ng(mca_obj=dn, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data['rad']['data']
    cot = cld.lay['extinction']['data'].sum(axis=2)                       # (nx, ny): proportional to the column optical depth
    assert r_up.shape == r_dn.shape == (12, 10) and np.all(np.isfinite(r_up)) and r_up.min() >= 0.0 and r_up.mean() > 0.0
    # both see the clouds (a clear-sky zenith is darker than a moderately thick cloud from below as well), as different images
    assert np.corrcoef(cot.ravel(), r_dn.ravel())[0, 1] > 0.3 and np.corrcoef(cot.ravel(), r_up.ravel())[0, 1] > 0.3
    assert not np.allclose(r_up, r_dn, rtol=0.05)


def test_the_binding_printed_in_integration_md_works():
    """INTEGRATION.md section 3 shows the ctypes stub a maintainer of the reference would add: run exactly that text against a
    job written by mcarats_ng and compare with the package's own route (tools/check_integration_stub.py)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_integration_stub.py')], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=root), timeout=300)
    assert r.returncode == 0 and 'INTEGRATION.md stub OK' in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_two_ranks_share_the_jobs(tmp_path):
    """row e through the drop-in layer: two ranks under torch.distributed.run ('gloo', both on this box's one GPU) --
    rank 0 writes the job files, every rank transports its share of every job's photon ids, tallies are all-reduced
    (per job on the file route, per run on the fused route).  tests/dist_gpu_worker.py holds the ranks' script."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'tests', 'dist_gpu_worker.py'), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(str(tmp_path/'result.npz'))
    for target in ('radiance', 'flux'):
        done, asked, done_f, asked_f = z[target+'_photons']
        assert done == asked and done_f == asked_f                    # the ranks' shares add up to every job's photon count
        a, b = z[target+'_job_dist'], z[target+'_job_solo']
        assert a.shape == b.shape and np.allclose(a, b, rtol=2e-3, atol=1e-6)   # same photon ids whoever transports them
        assert not z[target+'_fused_files'].any()
        m1, m2 = z[target+'_dist_mean'], z[target+'_fused_mean']      # different seeds: agreement of the domain means only
        assert m1.shape == m2.shape and abs(m1.mean()/m2.mean()-1.0) < 0.03
    # every flux variable, level by level, file route against fused route.  Above the 3-D region the direct beam is the
    # analytic term (no noise, no seed): there the two routes must agree to float32 rounding -- a fused route that adds the
    # term on every rank before the all-reduce would read world-size times too much.
    kdir = int(z['flux_kdir'][0])
    for v in ('f_down_direct', 'f_down', 'f_up'):
        a, b = z['flux_dist_'+v], z['flux_fused_'+v]
        assert a.shape == b.shape and a.shape[0] > kdir
        assert np.allclose(a, b, rtol=0.03, atol=2e-3), (v, a, b)
    a, b = z['flux_dist_f_down_direct'], z['flux_fused_f_down_direct']
    assert np.allclose(a[kdir:], b[kdir:], rtol=2e-5), (a[kdir:], b[kdir:])
    # heating rates through the batched exchange: the fourth tally and the total-down flux (its known direct part is the job's own)
    for v in ('heating_job', 'heating_flux'):
        a, b = z[v+'_dist'], z[v+'_solo']
        assert a.shape == b.shape and a.max() > 0.0 and np.allclose(a, b, rtol=2e-3, atol=1e-6*a.max()), v


def test_bench_rank_body_under_torchrun_with_rccl(tmp_path):
    """bench.py exactly as the driver starts it for N > 1 -- torch.distributed.run, backend nccl (= RCCL), device_id= --, at the one
    world size a one-GPU box can host: RCCL initialises, the in-place float64 all-reduce runs on the tally the kernel wrote,
    the barrier and the max-over-ranks timing are executed, and the line says so.  (No scaling figure comes out of this.)"""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--photons', '2e7',
           '--workload', 'les128', '--no-cpu-baseline', '--no-pmc']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['rccl_ranks'] == 1 and line['backend'] == 'nccl' and line['n_gpus'] == 1
    assert line['value'] > 1.0e8 and line['config']['mean_radiance'] > 0.0


@pytest.mark.parametrize('job_clock', [1759536000, 1759622400])
def test_func_ref_vs_cot_against_the_deterministic_answer(tmp_path, job_clock):
    """(Under two clocks -- `mcarats_ng` seeds its jobs from the clock, tests/conftest.py -- one of them the clock under which round 4's
    version of this test, at 2e6 photons and a bound that was 2.4 sigma of its own noise, read +0.62 %: at 2e7 photons per run a job's
    standard deviation is 0.18 %, the twelve jobs of an optical thickness give 0.05 %, and the 0.2 % bound is 3.8 sigma under ANY clock.)
    row a17: the reference's reflectance-vs-COT harness (er3t/rtm/mca/util.py:19-213) on the GPU, 1-D runs.  The reference
    compares this curve with libRadtran's DISORT (examples/00_er3t_bmk.py:470-579); here every job file the harness wrote goes
    through the deterministic plane-parallel solver K16 (tests/k16_adding_doubling.py: Rayleigh + gas absorption in 40 layers, the
    cloud slab with the Henyey-Greenstein TABLE the harness selects, per g), and the g-weighted reflectance must agree with the
    Monte-Carlo one to max(0.2 %, 4 standard errors of the mean of the three runs).  (Until round 2 the check was the
    two-stream curve, a band of +-0.12.)"""
    import glob
    from tests import k16_adding_doubling as k16
    atm = _atm(np.arange(0.0, 20.1, 0.5))
    ab = abs_synth(650.0, atm, Ng=4)
    pha = pha_hg_synth()
    cot = np.array([1.0, 4.0, 10.0, 30.0])
    f = mca.func_ref_vs_cot(cot, cer0=10.0, fdir=str(tmp_path/'lut'), wavelength=650.0, surface_albedo=0.03,
                            solar_zenith_angle=30.0, solar_azimuth_angle=0.0, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0,
                            cloud_top_height=2.0, cloud_geometrical_thickness=1.0, Nphoton=2e7, atm0=atm, abs0=ab, pha0=pha,
                            Ncpu=2, overwrite=True)
    from er3t_amd.rtm.mca.mca_exe import get_runner
    # (two 1-D constituents, the cloud slab's selector a table index: the lean loop's general-mixture build since round 5, not round 1's kernel)
    assert get_runner().sol.kernel_name().startswith('k_transport_lean<0,0,0,2>'), get_runner().sol.kernel_name()
    assert f.ref.shape == (4,) and np.all(np.diff(f.ref) > 0.0)           # brighter with optical thickness
    assert np.all(np.abs(f.ref-f.ref_2s) < 0.12)                          # (the two-stream curve of the reference's own plot)
    w, solar = ab.coef['weight']['data'], ab.coef['solar']['data']
    mu0 = np.cos(np.deg2rad(30.0))
    for ic, cot0 in enumerate(cot):
        files = sorted(glob.glob(str(tmp_path/'lut'/('*cot-%05.1f_cer-10.0' % cot0)/'rad'/'r00.g*.inp.txt')))
        assert len(files) == 4
        rad = []
        for fn in files:
            sc = Scene.from_nml(mca.mca_inp_read(fn), os.path.dirname(fn), solver=0)
            assert sc.nz3 == 0 and sc.np1d == 2 and sc.pha is not None
            rad.append(k16.solve_scene_1d(sc)['radiance'][0])
        want = np.pi*np.sum(np.array(rad)*solar*w)/(np.sum(solar*w)*mu0)
        se = f.ref_std[ic]/np.sqrt(3.0-1.0)                               # three runs: population std -> standard error of their mean
        assert abs(f.ref[ic]-want) < max(2.0e-3*want, 4.0*se), (job_clock, cot0, f.ref[ic], want, se)
    assert np.all(f.ref_std < 0.004*f.ref + 1e-4)
    assert abs(float(f.get_cot_from_ref(f.ref[2], method='linear'))-10.0) < 1e-6
    assert abs(float(f.get_ref_from_cot(10.0, method='linear'))-f.ref[2]) < 1e-9
    # overwrite=False re-loads the cached results without running
    g = mca.func_ref_vs_cot(cot, cer0=10.0, fdir=str(tmp_path/'lut'), surface_albedo=0.03, solar_zenith_angle=30.0,
                            atm0=atm, abs0=ab, pha0=pha, overwrite=False)
    assert np.array_equal(g.ref, f.ref)


@pytest.mark.parametrize('job_clock', [1759536000, 1759622400])
def test_func_ref_vs_cot_multi_pixel_against_the_deterministic_answer(tmp_path, job_clock):
    """`func_ref_vs_cot_multi_pixel` (er3t/rtm/mca/util.py:218-422): a homogeneous cloud on 3 x 2 columns under the independent-column
    solver, one `mcarats_ng` run of 3 runs x 4 g per optical thickness.  Every job file goes through the deterministic plane-parallel
    solver K16 (the 3-D constituent of the homogeneous grid put back as a 1-D one, layer for layer as the solver places it:
    Atm_iz3l honoured as handed over); the g-weighted reflectance of EVERY PIXEL must agree with it to max(0.4 %, 4 standard
    errors), the mean over the pixels to max(0.2 %, 4) -- at 2e7 photons per run, under two clocks (see the test above)."""
    import glob
    from er3t_amd.rtm.mca.mca_exe import get_runner
    from tests import k16_adding_doubling as k16
    atm = _atm(np.arange(0.0, 20.1, 0.5))
    ab = abs_synth(650.0, atm, Ng=4)
    cot = np.array([2.0, 10.0, 30.0])
    f = mca.func_ref_vs_cot_multi_pixel(cot, cer0=10.0, fdir=str(tmp_path/'lutmp'), wavelength=650.0, surface_albedo=0.03,
                                        solar_zenith_angle=30.0, solar_azimuth_angle=0.0, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0,
                                        cloud_top_height=2.0, cloud_geometrical_thickness=1.0, Nphoton=2e7, Nx=3, Ny=2, dx=0.1, dy=0.1,
                                        solver='ipa', atm0=atm, abs0=ab, pha0=None, Ncpu=2, overwrite=True)
    assert get_runner().sol.kernel_name().startswith('k_transport_lean<'), get_runner().sol.kernel_name()
    assert f.ref.shape == (3,) and np.all(np.diff(f.ref) > 0.0) and f.Nx == 3 and f.Ny == 2 and f.solver0 == 'ipa'
    assert np.all(np.abs(f.ref-f.ref_2s) < 0.12)
    w, solar = ab.coef['weight']['data'], ab.coef['solar']['data']
    mu0 = np.cos(np.deg2rad(30.0))
    for ic, cot0 in enumerate(cot):
        files = sorted(glob.glob(str(tmp_path/'lutmp'/('*cot-%05.1f_cer-10.0' % cot0)/'rad'/'r00.g*.inp.txt')))
        assert len(files) == 4
        rad = []
        for fn in files:
            sc = Scene.from_nml(mca.mca_inp_read(fn), os.path.dirname(fn), solver=2)
            assert (sc.nx, sc.ny, sc.np3d) == (3, 2, 1) and sc.nz3 == 2 and np.ptp(sc.extp) == 0.0
            assert abs(float(sc.extp[0, :, 0, 0].sum())*500.0-cot0) < 1e-4*cot0          # two layers of 500 m carry the optical thickness
            # the same atmosphere as a plane-parallel problem: the grid's constituent as a second 1-D one
            k0 = sc.iz3l-1
            col = lambda a3, fill: np.concatenate([np.full(k0, fill), np.asarray(a3[0, :, 0, 0], dtype=np.float64), np.full(sc.nz-k0-sc.nz3, fill)])
            sc1 = Scene(zgrd=sc.zgrd, ext1d=np.vstack([sc.ext1d, col(sc.extp, 0.0)]), omg1d=np.vstack([sc.omg1d, col(sc.omgp, 1.0)]),
                        apf1d=np.vstack([sc.apf1d, col(sc.apfp, 0.85)]), abs1d=sc.abs1d, nx=1, ny=1, sfc_mtype=1, sfc_param=sc.sfc_param,
                        src_the=sc.src_the, src_phi=sc.src_phi, src_qmax=0.0, view_the=sc.view_the, view_phi=sc.view_phi,
                        view_zloc=sc.view_zloc, nxr=1, nyr=1)
            rad.append(k16.solve_scene_1d(sc1)['radiance'][0])
        want = np.pi*np.sum(np.array(rad)*solar*w)/(np.sum(solar*w)*mu0)
        se = f.ref_std[ic]/np.sqrt(3.0-1.0)
        assert abs(f.ref[ic]-want) < max(2.0e-3*want, 4.0*se), (job_clock, cot0, f.ref[ic], want, se)
        pix = np.pi*np.asarray(f.rad_pixels[ic], dtype=np.float64)/(f.toa0*mu0)
        assert pix.shape == (3, 2)
        # (a pixel holds a sixth of the photons: its standard error is sqrt(6) x that of the mean of uncorrelated pixels -- 0.13 % here)
        assert np.all(np.abs(pix-want) < np.maximum(4.0e-3*want, 4.0*np.sqrt(6.0)*se)), (job_clock, cot0, pix, want, se)
    assert abs(float(f.get_cot_from_ref(f.ref[1], method='linear'))-10.0) < 1e-6
    g = mca.func_ref_vs_cot_multi_pixel(cot, cer0=10.0, fdir=str(tmp_path/'lutmp'), surface_albedo=0.03, solar_zenith_angle=30.0,
                                        Nx=3, Ny=2, atm0=atm, abs0=ab, overwrite=False)
    assert np.array_equal(g.ref, f.ref)


def test_heating_rate_target_through_the_dropin(tmp_path, oracle, nthreads):
    """target='heating rate' (er3t/rtm/mca/mcarats.py:279-283: Flx_mflx = 3, Flx_mhrt = 1) end to end: `mcarats_ng` writes the
    reference's job files, the jobs run on the GPU, every out.bin carries the three flux variables and a fourth on the layer grid,
    `mca_out_ng` returns the fluxes and `heating_rate`; one job against the oracle on the same file; the energy budget of the
    g-summed result"""
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=4)
    ab.coef['abso_coef']['data'] = ab.coef['abso_coef']['data']*40.0          # (an absorption band: heating that shows)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    nph = 400000
    m = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], Ng=4, target='heating rate', surface_albedo=0.2, solar_zenith_angle=40.0,
               solar_azimuth_angle=30.0, fdir=str(tmp_path/'hr'), Nrun=2, weights=ab.coef['weight']['data'], photons=nph,
               solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    assert m.target == 'heating rate' and int(mca.mca_inp_read(m.fnames_inp[0][0])['Flx_mhrt']) == 1
    raw = mca.mca_out_raw(m.fnames_out[1][2])
    nz = a1.nml[0]['Atm_nz']['data']
    assert [d['dims'][2] for d in raw.data] == [nz+1, nz+1, nz+1, nz] and raw.data[3]['name'].startswith('hrt')
    sc, o = _oracle_job(oracle, m.fnames_inp[1][2], int(m.photons[4+2]), 0, nthreads)
    got = raw.data[3]['data'][:, :, :, 0].mean(axis=(0, 1)); want = o['heat'].mean(axis=(1, 2))
    # (the two sides run the same job file, seed included, and follow the same histories until a float32 / float64 rounding parts them;
    #  absorption by the gas in a clear layer is a rare event -- a few hundred of this job's 1e5 photons per layer -- so where histories
    #  have parted the layer's value moves by its own Monte-Carlo noise, ~7 %: 10 % of a layer's value + 5 % of the largest.  The
    #  statistical comparison of heating rates is tests/test_gpu_parity.py::test_heating_rates_parity_and_energy_budget)
    assert want.max() > 0.0 and np.all(np.abs(got-want) < 0.05*want.max() + 0.10*want)
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data
    hr = out['heating_rate']['data']
    assert hr.shape == (12, 10, nz) and out['heating_rate_std']['data'].shape == hr.shape and out['f_up']['data'].shape == (12, 10, nz+1)
    # energy: net flux at the top - net flux at the surface = absorbed in between (domain means of the g-summed fields; roulette is
    # on, so to Monte-Carlo noise: 1.6e6 histories)
    dz = np.diff(atm.lev['altitude']['data'])*1000.0
    absorbed = (hr.mean(axis=(0, 1))*dz).sum()
    net = lambda lev: out['f_down']['data'][:, :, lev].mean()-out['f_up']['data'][:, :, lev].mean()
    toa = out['toa']['data']*np.cos(np.deg2rad(40.0))
    assert absorbed > 0.02*toa and abs(absorbed-(net(-1)-net(0))) < 4.0e-3*toa, (absorbed, net(-1)-net(0), toa)


def test_cox_munk_surface_through_the_dropin(tmp_path, oracle, nthreads):
    """the third surface branch of the reference's adapter (er3t/rtm/mca/mca_sfc.py:119-128, jsfc = 2): a wind-roughened sea
    under a cloud field, side file written by `mca_sfc_2d`, job run by `mcarats_ng`, the same job file through the oracle"""
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=2)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=4.0, seed=5)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    sfc = _quiet(mca.mca_sfc_2d, atm_obj=atm, sfc_obj=sfc_dsm_synth(12, 10), fname=str(tmp_path/'sfc.bin'), quiet=True)
    assert int(sfc.nml['Sfc_jsfc2d']['data'][0, 0]) == 2 and sfc.nml['Sfc_psfc2d']['data'].shape == (12, 10, 5)
    nph = 400000
    m = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], Ng=2, target='radiance', surface_albedo=sfc, solar_zenith_angle=35.0,
               solar_azimuth_angle=90.0, sensor_zenith_angle=35.0, sensor_azimuth_angle=270.0, fdir=str(tmp_path/'sim'), Nrun=1,
               photons=nph, solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    n = int(m.photons[1])
    raw = mca.mca_out_raw(m.fnames_out[0][1]).data[0]['data'][:, :, 0, 0]
    sc, o = _oracle_job(oracle, m.fnames_inp[0][1], n, 0, nthreads)
    assert sc.jsfc is not None and np.all(sc.jsfc == 2.0)
    orad = o['rad'][0].T
    assert abs(raw.mean()-orad.mean()) < 0.03*orad.mean()


def test_all_sky_camera_through_the_dropin(tmp_path, oracle, nthreads):
    """sensor_type='all-sky' (er3t/rtm/mca/mcarats.py:291-296, 369-372): `mcarats_ng` writes Rad_mrkind = 1 with its 500 x 500
    fish-eye image, the job runs on the GPU, `mca_out_ng` reads the image back; the same job file through the oracle"""
    atm = _atm(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=2)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=4.0, seed=5)
    a1 = _quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
    a3 = _quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, fname=str(tmp_path/'atm3d.bin'), quiet=True)
    nph = 300000
    m = _quiet(mca.mcarats_ng, atm_1ds=[a1], atm_3ds=[a3], Ng=2, target='radiance', surface_albedo=0.05, solar_zenith_angle=30.0,
               solar_azimuth_angle=45.0, sensor_zenith_angle=180.0, sensor_azimuth_angle=0.0, sensor_altitude=0.0, sensor_type='all-sky',
               sensor_xpos=0.4, sensor_ypos=0.6, fdir=str(tmp_path/'sim'), Nrun=1, photons=nph, solver='3D', mp_mode='py',
               overwrite=True, date=gin.DATE, quiet=True)
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True)
    rad = out.data['rad']['data']
    assert rad.shape == (500, 500) and np.isfinite(rad).all() and rad.max() > 0.0
    n = int(m.photons[1])
    raw = mca.mca_out_raw(m.fnames_out[0][1]).data[0]['data'][:, :, 0, 0]
    sc, o = _oracle_job(oracle, m.fnames_inp[0][1], n, 0, nthreads)
    assert sc.rad_kind == 1 and sc.cam_qmax == [178.0] and sc.cam_apsize == [0.05] and (sc.nxr, sc.nyr) == (500, 500)
    orad = o['rad'][0].T
    # the corners of the square image lie outside the 89-degree cone: dark in both
    assert raw[0, 0] == 0.0 and orad[0, 0] == 0.0
    # 1/r^2 spikes of scattering events next to the sensor (Rad_apsize = 0.05 m) make single pixels noisy: compare the medians of
    # 50 x 50 blocks over the lit disc
    gb = np.median(raw.reshape(10, 50, 10, 50).transpose(0, 2, 1, 3).reshape(10, 10, -1), axis=-1)
    ob = np.median(orad.reshape(10, 50, 10, 50).transpose(0, 2, 1, 3).reshape(10, 10, -1), axis=-1)
    lit = ob > 0.2*ob.max()
    assert lit.sum() >= 8 and np.all(np.abs(gb-ob)[lit] < 0.25*ob[lit]), (gb[lit]/ob[lit])


@pytest.mark.parametrize('case', ['c2_nadir', 'c2_slant', 'c3_flux', 'c4_absorb', 'c5_lsrt', 'c6_sea', 'c7_allsky'])
def test_ab_cases_run_and_follow_the_oracle(tmp_path, oracle, nthreads, case):
    """the committed input sets of the MCARaTS A/B run (tests/golden/ab/): each job through the solver's command-line route
    (`run_job`, what `python -m er3t_amd.rtm.mca.mca_exe` calls) and through the oracle, from the same files"""
    from er3t_amd.rtm.mca.mca_exe import JobRunner, run_job
    d = os.path.join(os.path.dirname(os.path.abspath(gin.__file__)), 'ab', case)
    n = 300000
    runner = JobRunner(device=0)
    res = run_job(os.path.join(d, 'r00.g000.inp.txt'), str(tmp_path/'out.bin'), n, 0, runner=runner)
    sc, o = _oracle_job(oracle, os.path.join(d, 'r00.g000.inp.txt'), n, 0, nthreads)
    raw = mca.mca_out_raw(str(tmp_path/'out.bin'))
    if case == 'c3_flux':
        assert len(raw.data) == 3
        for iv in range(3):
            a = raw.data[iv]['data'][:, :, :, 0].mean(axis=(0, 1)); b = o['flux'][iv].mean(axis=(1, 2))
            assert np.all(np.abs(a-b) < 6.0/np.sqrt(n) + 2e-3*np.abs(b)), (iv, np.abs(a-b).max())
    else:
        img = raw.data[0]['data'][:, :, 0, 0]
        oimg = o['rad'][0].T
        assert img.shape == oimg.shape
        if case == 'c7_allsky':
            # (single pixels carry 1/r^2 spikes: medians of 50 x 50 blocks over the lit part of the fish-eye image)
            gb = np.median(img.reshape(10, 50, 10, 50).transpose(0, 2, 1, 3).reshape(10, 10, -1), axis=-1)
            ob = np.median(oimg.reshape(10, 50, 10, 50).transpose(0, 2, 1, 3).reshape(10, 10, -1), axis=-1)
            lit = ob > 0.2*ob.max()
            assert lit.sum() >= 8 and np.all(np.abs(gb-ob)[lit] < 0.25*ob[lit])
        else:
            assert abs(img.mean()-oimg.mean()) < 0.02*oimg.mean(), (img.mean(), oimg.mean())
            assert np.corrcoef(img.ravel(), oimg.ravel())[0, 1] > 0.9


def test_two_handles_statistics_calls_out_of_order():
    """mi3d_stats_join / mi3d_stats_chain (two solver handles taking turns through the jobs of a run): the C-ABI refuses what
    would silently give wrong sums"""
    from er3t_amd.solver import Mi3dSolver
    from er3t_amd.synth import les_scene
    from er3t_amd.scene import TARGET_RADIANCE
    sc = les_scene(nx=8, ny=8, nz3=50)
    a, b = Mi3dSolver(0), Mi3dSolver(0)
    for s in (a, b):
        s.set_tuning(own_stream=1)
        s.load_scene(sc)
    with pytest.raises(OSError):
        b.stats_join(a)                          # the owner has not begun statistics
    a.stats_begin()
    b.stats_join(a)
    with pytest.raises(OSError):
        a.stats_join(b)                          # a joined handle owns no run fields
    with pytest.raises(OSError):
        b.stats_end_run()                        # the run is closed on the owner
    # two jobs, one on each handle, folded in job order; against the same two jobs on one handle
    n = 20000
    for s, seed in ((a, 3), (b, 4)):
        s.reset(); s.run(n, seed=seed)
    a.stats_chain(b); a.stats_add(n, factor_rad=1.0)
    b.stats_chain(a); b.stats_add(n, factor_rad=0.5)
    b.sync()
    two = a.stats_end_run(keep=True)['rad']
    c = Mi3dSolver(0); c.load_scene(sc); c.stats_begin()
    for seed, f in ((3, 1.0), (4, 0.5)):
        c.reset(); c.run(n, seed=seed); c.stats_add(n, factor_rad=f)
    one = c.stats_end_run(keep=True)['rad']
    assert two.shape == one.shape and np.allclose(two, one, rtol=1e-5, atol=1e-7*one.max()) and one.max() > 0.0
    mean, sdev, nrun = a.stats_get(TARGET_RADIANCE)
    assert nrun == 1 and np.allclose(mean, two, rtol=1e-6)
