"""throughput of the all-sky camera path (Rad_mrkind = 1, er3t/rtm/mca/mcarats.py:291-296: a camera on the ground, 178 degree cone,
500 x 500 pixels) on the config-2 grid: tools/time_camera.py [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
sc = les_scene(surface_albedo=0.1)
sc.rad_kind = 1
sc.view_the = [0.0]; sc.view_phi = [0.0]; sc.view_zloc = [0.0]
sc.cam_psi = [0.0]; sc.cam_xpos = [0.5]; sc.cam_ypos = [0.5]
sc.cam_qmax = [178.0]; sc.cam_umax = [178.0]; sc.cam_vmax = [178.0]; sc.cam_apsize = [0.05]
sc.nxr = 500; sc.nyr = 500
sol = Mi3dSolver(0); sol.load_scene(sc); sol.set_counting(True)
sol.reset(); sol.run(1000000, seed=2); sol.sync(); c = sol.counters()
print(sol.kernel_name(), {k: round(v/1e6, 2) for k, v in c.items() if v and not k.startswith(('sched', 'ticks'))})
sol.set_counting(False)
for r in range(2):
    sol.reset(); sol.run(n, seed=3+r); sol.sync(); ms, _ = sol.timing(); print('%d photons %.1f ms %.4g photons/s' % (n, ms, n/(ms*1e-3)))
