"""Tally window: how many of the column view's tallies stay in the window (build with -DMI3D_WIN_DIAG: le_steps3d counts them, le_steps all)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
for wl in (sys.argv[1:] or ['les480', 'les128']):
    sol = Mi3dSolver(0); sc = make_scene(wl); sol.load_scene(sc); sol.set_counting(True)
    n = int(float(os.environ.get("WIN_DIAG_N", "2e8")))
    sol.reset(); sol.run(n, seed=5); sol.sync()
    c = sol.counters()
    print(wl, sol.kernel_name(), 'tallies per photon %.2f  in the window %.1f %%  (collisions per photon %.2f)' % (c['le_steps']/n, 100.0*c['le_steps3d']/max(c['le_steps'], 1), c['scatter']/n))
