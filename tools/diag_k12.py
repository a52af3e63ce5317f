import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from er3t_amd.solver import Mi3dSolver
from er3t_amd.scene import TARGET_RADIANCE
from oracle import oracle
from tests.util import slab_scene
from tests.test_oracle_kat import _chandrasekhar_h
from tests.test_gpu_parity import gpu_run
omega, sza = 0.9, 40.0
mu0 = np.cos(np.deg2rad(sza))
vza = np.array([0.0, 35.0, 65.0])
mu = np.cos(np.deg2rad(vza))
want = omega/(4.0*np.pi)*mu0/(mu+mu0)*_chandrasekhar_h(omega, mu)*_chandrasekhar_h(omega, mu0)
print('exact      ', want)
sol = Mi3dSolver(0)
for nz, tau in ((8, 40.0), (1, 40.0), (8, 8.0)):
    sc = slab_scene(tau=tau, omega=omega, apf=-2.0, albedo=0.0, sza=sza, nz=nz, vza=vza, vaa=(0.0, 90.0, 200.0), target=TARGET_RADIANCE)
    sc.le_tau1 = 0.0
    n = 4000000
    for col in (True, False):
        g = gpu_run(sol, sc, n, seed=31, column_le=col)
        print('nz %d tau %g GPU column_le=%s' % (nz, tau, col), g['rad'][:, 0, 0], {k: g['counters'][k]/n for k in ('le_rays', 'le_steps', 'le_column', 'scatter')})
    o = oracle.run(sc, n, seed=31, nthreads=16)
    print('nz %d tau %g oracle          ' % (nz, tau), o['rad'][:, 0, 0], {k: o['counters'][k]/n for k in ('le_rays', 'le_steps', 'scatter')})
