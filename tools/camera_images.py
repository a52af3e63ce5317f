"""What the periodic images of the domain add to an all-sky camera's image, ring by ring of the zenith angle, and what they cost:
   tools/camera_images.py [photons]     (bench workload les128_cam: er3t's camera on the ground, 178 degree cone, 500 x 500 pixels)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
sol = Mi3dSolver(0)
sc = make_scene('les128_cam')
du = np.deg2rad(178.0)/500
u = (np.arange(500)+0.5-250)*du
th = np.rad2deg(np.hypot(u[:, None], u[None, :]))
rings = [(0, 30), (30, 60), (60, 75), (75, 82), (82, 86), (86, 89)]
ref = None
for nimg in (0, 1, 2, 3):
    sc.cam_images = nimg
    sol.load_scene(sc); sol.set_counting(False)
    sol.reset(); sol.run(n//10, seed=5); sol.sync()
    sol.reset(); sol.run(n, seed=5); sol.sync()
    ms, _ = sol.timing()
    img = sol.radiance(n)[0].astype(np.float64)
    means = [img[(th >= a) & (th < b)].mean() for a, b in rings]
    if nimg == 0:
        ref = means
    print('cam_images %d (%2d images): %.3g photons/s; ring means %s; relative to the nearest image alone %s'
          % (nimg, (2*nimg+1)**2, n/(ms*1e-3), ' '.join('%.4f' % m for m in means), ' '.join('%+.1f%%' % (100*(m/r-1)) for m, r in zip(means, ref))), flush=True)
print('rings (degrees from the zenith):', rings)
