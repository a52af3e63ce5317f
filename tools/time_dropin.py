"""end-to-end timing of the drop-in API: config-3 shape (128x128x50 + aerosol, 16 g, 3 runs, 1e8 photons per run);
   --grid 480: config-4 shape (480x480x100 cloud field, nadir radiance, 1e9 photons, one g): adapter + 460 MB side file, job files,
   scene upload, transport, output file, mca_out_ng -- and, for comparison, what the reference's way of writing ONE of the five
   fields of that side file costs (struct.pack('<%df', *array): er3t/rtm/mca/mca_atm.py:383-388)"""
import os, sys, time, io, contextlib, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import er3t_amd.rtm.mca as mca
from er3t_amd import synth
import datetime


def config4():
    import shutil, struct
    atm = synth.atm_synth(synth.z_levels_config4())
    atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
    atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
    ab = synth.abs_synth(650.0, atm, Ng=1)
    t0 = time.time()
    cld = synth.cld_synth(atm, nx=480, ny=480, nz=100, z_base=0.6, z_top=1.6, cot_mean=10.0, seed=20251004)
    tmp = tempfile.mkdtemp()
    t1 = time.time()
    a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
    a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, pha_obj=None, fname=tmp+'/atm3d.bin', quiet=True)
    t2 = time.time()
    size = os.path.getsize(tmp+'/atm3d.bin')
    print('synthetic cloud field %.1f s | mca_atm_1d + mca_atm_3d incl. side file of %.0f MB: %.2f s' % (t1-t0, size/1e6, t2-t1), flush=True)
    for nph in (1e7, 1e9):
        t3 = time.time()
        m = mca.mcarats_ng(atm_1ds=[a1], atm_3ds=[a3], Ng=1, weights=ab.coef['weight']['data'], target='radiance', surface_albedo=0.03,
                           solar_zenith_angle=30.0, solar_azimuth_angle=45.0, sensor_zenith_angle=0.0, fdir=tmp+'/rad', Nrun=1, photons=nph,
                           solver='3D', mp_mode='py', overwrite=True, date=datetime.datetime(2017, 8, 13), quiet=True)
        t4 = time.time()
        out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True)
        t5 = time.time()
        print('%-9s mcarats_ng %.2f s (input file, side file read + upload of %.0f MB, %.3g photons: kernels %.3f s, output file) | mca_out_ng %.3f s | mean radiance %.5f'
              % ('warm-up:' if nph < 1e8 else 'config 4:', t4-t3, size/1e6, nph, m.run0.kernel_ms*1e-3, t5-t4, out.data['rad']['data'].mean()), flush=True)
    # the reference's idiom on ONE field of this grid (23 040 000 values; the side file holds five)
    arr = a3.nml['Atm_extp3d']['data'][..., 0]
    t6 = time.time()
    with open(tmp+'/ref_style.bin', 'wb') as f:
        f.write(struct.pack('<%df' % arr.size, *arr.flatten(order='F')))
    t7 = time.time()
    with open(tmp+'/fast.bin', 'wb') as f:
        np.asarray(arr, dtype='<f4').ravel(order='F').tofile(f)
    t8 = time.time()
    same = open(tmp+'/ref_style.bin', 'rb').read() == open(tmp+'/fast.bin', 'rb').read()
    print('one field of the side file (%d values): struct.pack(*tuple) %.1f s (x 5 fields = %.0f s), ndarray.tofile %.2f s, bytes identical: %s'
          % (arr.size, t7-t6, 5*(t7-t6), t8-t7, same), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


if '--grid' in sys.argv and sys.argv[sys.argv.index('--grid')+1] == '480':
    config4()
    sys.exit(0)
atm = synth.atm_synth(synth.z_levels_config2())
atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
ab = synth.abs_synth(650.0, atm, Ng=16)
cld = synth.cld_synth(atm)
tmp = tempfile.mkdtemp()
t0 = time.time()
a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, pha_obj=None, fname=tmp+'/atm3d.bin', quiet=True)
t1 = time.time()
# (a first, untimed pass: library load, GPU initialisation and the first launch of every kernel are not the pipeline's cost)
for target, nph, fused in (('radiance', 1e6, False), ('radiance', 1e8, False), ('radiance', 1e8, True), ('flux', 1e8, False), ('flux', 1e8, True)):
    t2 = time.time()
    extra = dict(abs_obj=ab, keep_files=False) if fused else {}
    m = mca.mcarats_ng(**extra, atm_1ds=[a1], atm_3ds=[a3], Ng=16, weights=ab.coef['weight']['data'], target=target, surface_albedo=0.03,
                       solar_zenith_angle=30.0, solar_azimuth_angle=45.0, fdir=tmp+'/'+target+('_fused' if fused else ''), Nrun=3, photons=nph, solver='3D',
                       Ncpu=12, mp_mode='py', overwrite=True, date=datetime.datetime(2017, 8, 13), quiet=True)
    t3 = time.time()
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True)
    t4 = time.time()
    kms = m.kernel_ms if fused else m.run0.kernel_ms
    if nph < 1e7:
        continue
    print('%-8s %-5s: adapters %.2f s | mcarats_ng %.2f s (48 jobs, %.3g photons; kernels %.3f s) | mca_out_ng %.2f s' %
          (target, 'fused' if fused else 'files', t1-t0, t3-t2, 3*nph, kms*1e-3, t4-t3), flush=True)
