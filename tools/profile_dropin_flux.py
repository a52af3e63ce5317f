"""where the host time of the file route of a flux simulation goes (config-3 shape: 16 g x 3 runs, 3e8 photons): tools/profile_dropin_flux.py"""
import os, sys, time, tempfile, cProfile, pstats, io, datetime
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import er3t_amd.rtm.mca as mca
from er3t_amd import synth
atm = synth.atm_synth(synth.z_levels_config2())
atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
ab = synth.abs_synth(650.0, atm, Ng=16)
cld = synth.cld_synth(atm)
tmp = tempfile.mkdtemp()
a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, pha_obj=None, fname=tmp+'/atm3d.bin', quiet=True)
def run(nph, tag):
    return mca.mcarats_ng(atm_1ds=[a1], atm_3ds=[a3], Ng=16, weights=ab.coef['weight']['data'], target='flux', surface_albedo=0.03,
                          solar_zenith_angle=30.0, solar_azimuth_angle=45.0, fdir=tmp+'/flux'+tag, Nrun=3, photons=nph, solver='3D',
                          Ncpu=12, mp_mode='py', overwrite=True, date=datetime.datetime(2017, 8, 13), quiet=True)
run(1e6, 'w1'); run(1e8, 'w2')
for rep in range(2):
    t0 = time.time(); m = run(1e8, 'r%d' % rep); t1 = time.time()
    out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True); t2 = time.time()
    print('flux files, warm: mcarats_ng %.3f s (kernels %.3f s) | mca_out_ng %.3f s' % (t1-t0, m.run0.kernel_ms*1e-3, t2-t1), flush=True)
pr = cProfile.Profile(); pr.enable(); m = run(1e8, 'p'); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
