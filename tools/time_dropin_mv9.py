"""BASELINE config 5 through the drop-in classes: nine view zenith angles over an LSRT surface on the 480 x 480 x 100 grid -- ONE simulation with the
views as sequences (not in the reference) against the reference's way, a simulation per view: tools/time_dropin_mv9.py [photons]"""
import os, sys, time, tempfile, datetime, shutil, io, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import er3t_amd.rtm.mca as mca
from er3t_amd import synth
nph = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0e8
atm = synth.atm_synth(synth.z_levels_config4())
atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
ab = synth.abs_synth(650.0, atm, Ng=1)
cld = synth.cld_synth(atm, nx=480, ny=480, nz=100, z_base=0.6, z_top=1.6, cot_mean=10.0, seed=20251004)
tmp = tempfile.mkdtemp()
def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)
a1 = quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
a3 = quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, pha_obj=None, fname=tmp+'/atm3d.bin', quiet=True)
sfc = quiet(mca.mca_sfc_2d, atm_obj=atm, sfc_obj=synth.sfc_lsrt_synth(480, 480), fname=tmp+'/sfc.bin', quiet=True)
vza = [0.0, 26.1, 26.1, 45.6, 45.6, 60.0, 60.0, 70.5, 70.5]; vaa = [0.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0]
kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=1, weights=ab.coef['weight']['data'], target='radiance', surface_albedo=sfc, solar_zenith_angle=30.0,
          solar_azimuth_angle=45.0, Nrun=1, solver='3D', mp_mode='py', overwrite=True, date=datetime.datetime(2017, 8, 13), quiet=True)
quiet(mca.mcarats_ng, fdir=tmp+'/w9', sensor_zenith_angle=vza, sensor_azimuth_angle=vaa, photons=2.0e6, **kw)      # (event lists, first launches)
quiet(mca.mcarats_ng, fdir=tmp+'/w1', sensor_zenith_angle=vza[3], sensor_azimuth_angle=vaa[3], photons=2.0e6, **kw)
t0 = time.time(); m9 = quiet(mca.mcarats_ng, fdir=tmp+'/nine', sensor_zenith_angle=vza, sensor_azimuth_angle=vaa, photons=nph, **kw); t1 = time.time()
r9 = quiet(mca.mca_out_ng, mca_obj=m9, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data['rad']['data']
ts, means = [], []
for i in range(9):
    t2 = time.time(); m = quiet(mca.mcarats_ng, fdir=tmp+'/one%d' % i, sensor_zenith_angle=vza[i], sensor_azimuth_angle=vaa[i], photons=nph, **kw); ts.append(time.time()-t2)
    means.append(float(quiet(mca.mca_out_ng, mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data['rad']['data'].mean()))
print('nine views, %.3g photons: ONE simulation %.3f s (kernels %.3f s) | a simulation per view %.3f s in all (%s)'
      % (nph, t1-t0, m9.run0.kernel_ms*1e-3, sum(ts), ' '.join('%.3f' % t for t in ts)))
print('domain means, one simulation / nine: ' + ' '.join('%.4f/%.4f' % (r9[:, :, i].mean(), means[i]) for i in range(9)))
shutil.rmtree(tmp, ignore_errors=True)
