"""func_ref_vs_cot at one optical thickness under two frozen clocks (the seeds of the jobs): per-run reflectance from the GPU, the
same job files through the CPU oracle, and the deterministic answer (K16)."""
import glob, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import time as _time
import er3t_amd.rtm.mca as mca
import er3t_amd.rtm.mca.mcarats as _m
from er3t_amd.scene import Scene
from er3t_amd.synth import atm_synth, abs_synth, pha_hg_synth
from oracle import oracle
from tests import k16_adding_doubling as k16

class Still:
    def __init__(self, t): self.t = t
    def __getattr__(self, n): return getattr(_time, n)
    def time(self): return self.t

atm = atm_synth(np.arange(0.0, 20.1, 0.5))
atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
ab = abs_synth(650.0, atm, Ng=4); pha = pha_hg_synth()
w, solar = ab.coef['weight']['data'], ab.coef['solar']['data']; mu0 = np.cos(np.deg2rad(30.0))
for clock in (1759536000.0, 1759622400.0):
    _m.time = Still(clock)
    d = tempfile.mkdtemp(prefix='refcot_')
    f = mca.func_ref_vs_cot(np.array([4.0]), cer0=10.0, fdir=d, wavelength=650.0, surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=0.0,
                            sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, cloud_top_height=2.0, cloud_geometrical_thickness=1.0, Nphoton=2e6,
                            atm0=atm, abs0=ab, pha0=pha, Ncpu=2, overwrite=True)
    files = sorted(glob.glob(os.path.join(d, '*cot-004.0_cer-10.0', 'rad', 'r*.g*.inp.txt')))
    print('clock %d: %d job files; GPU reflectance %.6f +- %.6f (std of 3 runs)' % (clock, len(files), f.ref[0], f.ref_std[0]))
    rows = {}
    for fn in files:
        nml = mca.mca_inp_read(fn)
        sc = Scene.from_nml(nml, os.path.dirname(fn), solver=0)
        ir, ig = int(os.path.basename(fn)[1:3]), int(os.path.basename(fn)[5:8])
        nph = int(float(nml.get('Wld_nphoton', 0)) or 0)
        out = fn.replace('.inp.txt', '.out.bin')
        g = mca.mca_out_raw(out).data[0]['data'].mean()
        nph_job = 2000000 if not nph else nph
        rows[(ir, ig)] = (g, sc, int(nml['Wld_jseed']))
    for ir in range(3):
        gsum = sum(rows[(ir, ig)][0]*solar[ig]*w[ig] for ig in range(4))/np.sum(solar*w)
        print('   run %d: GPU pi I / mu0 = %.6f   seeds %s' % (ir, np.pi*gsum/mu0, [rows[(ir, ig)][2] for ig in range(4)]))
    want = np.pi*sum(k16.solve_scene_1d(rows[(0, ig)][1])['radiance'][0]*solar[ig]*w[ig] for ig in range(4))/(np.sum(solar*w)*mu0)
    print('   deterministic answer %.6f' % want)
    # the oracle on run 0's four jobs with the photons mcarats_ng gave them
    m_ph = {}
    for ig in range(4):
        sc, seed = rows[(0, ig)][1], rows[(0, ig)][2]
        nph = int(round(2e6*0.95*w[ig]/np.sum(w))) + int(2e6*0.05/4)
        o = oracle.run(sc, nph, seed=seed, nthreads=16)['rad'].mean()
        print('      run 0 g %d: GPU %.6f  oracle %.6f (%.2e photons, seed %d)' % (ig, rows[(0, ig)][0], o, nph, seed))
