"""end-to-end timing of the drop-in API: config-3 shape (128x128x50 + aerosol, 16 g, 3 runs, 1e8 photons per run)"""
import os, sys, time, io, contextlib, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import er3t_amd.rtm.mca as mca
from er3t_amd import synth
import datetime
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
