import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
sol = Mi3dSolver(0)
for work, nph in (("les480_mv9", 2e7), ('les128_flux', 2e7), ('les128', 5e7)):
    sc = make_scene(work); sol.bind(None, None, None); sol.load_scene(sc)
    sol.set_counting(True); sol.reset(); sol.run(int(nph/10), seed=3); sol.sync(); c = sol.counters(); n = int(nph/10)
    print(work, {k: round(v/n, 2) for k, v in c.items() if v and not k.startswith('rsv')}, flush=True)
    sol.set_counting(False); sol.reset(); sol.run(int(nph), seed=4); sol.sync(); ms, _ = sol.timing()
    print('   %.4g photons/s (%.1f ms)' % (nph/(ms*1e-3), ms), flush=True)
