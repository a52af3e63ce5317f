"""Are the streams of neighbouring seeds independent?  One 1-D job of func_ref_vs_cot (COT 4), 5e5 photons, under 64 consecutive seeds
and under 64 scattered ones: mean and spread of the 1-pixel radiance, and of means over groups of 12 consecutive seeds."""
import glob, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import er3t_amd.rtm.mca as mca
from er3t_amd.scene import Scene
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import atm_synth, abs_synth, pha_hg_synth
atm = atm_synth(np.arange(0.0, 20.1, 0.5))
atm.lay['co2'] = {'data': 4.0e-4*1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
atm.lay['air'] = {'data': 1.0e19*np.exp(-atm.lay['altitude']['data']/8.0)}
ab = abs_synth(650.0, atm, Ng=4); pha = pha_hg_synth()
d = tempfile.mkdtemp(prefix='seedcorr_')
f = mca.func_ref_vs_cot(np.array([4.0]), cer0=10.0, fdir=d, wavelength=650.0, surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=0.0,
                        sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, cloud_top_height=2.0, cloud_geometrical_thickness=1.0, Nphoton=2e5,
                        atm0=atm, abs0=ab, pha0=pha, Ncpu=2, overwrite=True)
fn = sorted(glob.glob(os.path.join(d, '*cot-004.0_cer-10.0', 'rad', 'r00.g000.inp.txt')))[0]
sc = Scene.from_nml(mca.mca_inp_read(fn), os.path.dirname(fn), solver=0)
sol = Mi3dSolver(0); sol.load_scene(sc); sol.set_counting(False)
nph = 500000
def run(seed):
    sol.reset(); sol.run(nph, seed=int(seed)); sol.sync(); return float(sol.radiance(nph).mean())
rng = np.random.default_rng(1)
for name, seeds in (('consecutive from 1759536000', 1759536000 + np.arange(96)), ('consecutive from 1759622400', 1759622400 + np.arange(96)),
                    ('scattered', rng.integers(1, 2**62, size=96))):
    v = np.array([run(s) for s in seeds])
    g = v.reshape(8, 12).mean(axis=1)
    print('%-30s mean %.6f  sd of a job %.2e (%.2f %%)  sd of the means of 12 jobs %.2e (independent jobs would give %.2e)  first three groups %s'
          % (name, v.mean(), v.std(ddof=1), 100*v.std(ddof=1)/v.mean(), g.std(ddof=1), v.std(ddof=1)/np.sqrt(12), np.round(g[:3], 6)))
