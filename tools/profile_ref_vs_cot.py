"""func_ref_vs_cot as er3t's retrievals call it (a curve of a dozen optical thicknesses, 16 g x 3 runs each): where the time goes: tools/profile_ref_vs_cot.py [photons]"""
import os, sys, time, tempfile, cProfile, pstats, io, shutil, datetime
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.rtm.mca.util import func_ref_vs_cot
from er3t_amd import synth
nph = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0e7
atm = synth.atm_synth(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
ab = synth.abs_synth(650.0, atm, Ng=16)
tmp = tempfile.mkdtemp()
cot = np.concatenate([np.arange(0.0, 2.0, 0.5), np.arange(2.0, 30.0, 4.0), np.arange(30.0, 60.0, 10.0)])
kw = dict(cer0=10.0, date=datetime.datetime(2017, 8, 13), wavelength=650.0, surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=0.0,
          sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, Nphoton=nph, atm0=atm, abs0=ab, pha0=None, overwrite=True)
func_ref_vs_cot(cot[:2], fdir=tmp+'/w', **kw)
t0 = time.time(); f = func_ref_vs_cot(cot, fdir=tmp+'/a', **kw); t1 = time.time()
print('%d optical thicknesses x 16 g x 3 runs, %.3g photons per run: %.2f s (%.1f ms per job)' % (cot.size, nph, t1-t0, (t1-t0)/(cot.size*48)*1e3), flush=True)
print('reflectance', np.round(f.ref, 4))
pr = cProfile.Profile(); pr.enable(); func_ref_vs_cot(cot[:4], fdir=tmp+'/p', **kw); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(32); print(s.getvalue()[:7000])
shutil.rmtree(tmp, ignore_errors=True)
