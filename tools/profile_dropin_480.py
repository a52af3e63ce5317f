"""where the host time of ONE radiance job on the 480 x 480 x 100 grid goes (config-4 shape through mcarats_ng): tools/profile_dropin_480.py"""
import os, sys, time, tempfile, cProfile, pstats, io, datetime, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import er3t_amd.rtm.mca as mca
from er3t_amd import synth
atm = synth.atm_synth(synth.z_levels_config4())
atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
ab = synth.abs_synth(650.0, atm, Ng=1)
cld = synth.cld_synth(atm, nx=480, ny=480, nz=100, z_base=0.6, z_top=1.6, cot_mean=10.0, seed=20251004)
tmp = tempfile.mkdtemp()
a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, pha_obj=None, fname=tmp+'/atm3d.bin', quiet=True)
def run(nph, tag):
    return mca.mcarats_ng(atm_1ds=[a1], atm_3ds=[a3], Ng=1, weights=ab.coef['weight']['data'], target='radiance', surface_albedo=0.03,
                          solar_zenith_angle=30.0, solar_azimuth_angle=45.0, sensor_zenith_angle=0.0, fdir=tmp+'/rad'+tag, Nrun=1, photons=nph,
                          solver='3D', mp_mode='py', overwrite=True, date=datetime.datetime(2017, 8, 13), quiet=True)
run(1e7, 'w')
t0 = time.time(); m = run(1e8, 'a'); print('mcarats_ng 1e8 photons: %.3f s (kernels %.3f s)' % (time.time()-t0, m.run0.kernel_ms*1e-3), flush=True)
pr = cProfile.Profile(); pr.enable(); m = run(1e8, 'p'); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(30); print(s.getvalue()[:6500])
shutil.rmtree(tmp, ignore_errors=True)
