"""runs the ctypes stub printed in INTEGRATION.md section 3 against a job assembled by mcarats_ng and compares with the
package's own route (same input, same seed)"""
import os, re, sys, tempfile, contextlib, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import er3t_amd.rtm.mca as mca
from er3t_amd import synth
from tests.golden import inputs as gin

text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
code = re.search(r"## 3\..*?```python\n(.*?)```", text, re.S).group(1)
code = code.replace("C.CDLL('libmi3drt.so')", "C.CDLL(%r)" % os.path.join(ROOT, 'er3t_amd', 'libmi3drt.so'))
ns = {}
exec(code, ns)

tmp = tempfile.mkdtemp()
atm = synth.atm_synth(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
ab = synth.abs_synth(650.0, atm, Ng=2)
cld = synth.cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
with contextlib.redirect_stdout(io.StringIO()):
    a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
    a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, fname=tmp+'/atm3d.bin', quiet=True)
    m = mca.mcarats_ng(atm_1ds=[a1], atm_3ds=[a3], Ng=2, target='radiance', surface_albedo=0.05, solar_zenith_angle=40.0, fdir=tmp+'/sim',
                       Nrun=1, photons=4e5, weights=ab.coef['weight']['data'], solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
ig = 1
nml = mca.mca_inp_read(m.fnames_inp[0][ig])          # the job exactly as the solver executable would see it
n = int(m.photons[ig])
rad_stub = ns['run_job_on_gpu'](nml, m.fdir, n, 0)
rad_pkg = mca.mca_out_raw(m.fnames_out[0][ig]).data[0]['data'][:, :, 0, 0]
print('stub mean %.6f  package mean %.6f  max rel diff %.2e' % (rad_stub.mean(), rad_pkg.mean(), np.abs(rad_stub-rad_pkg).max()/rad_pkg.mean()))
assert rad_stub.shape == rad_pkg.shape and np.allclose(rad_stub, rad_pkg, rtol=3e-3, atol=1e-6)
print('INTEGRATION.md stub OK')
