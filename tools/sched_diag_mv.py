import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
sol = Mi3dSolver(0); sc = make_scene('les480_mv9'); sol.load_scene(sc)
n = 20000000
sol.set_counting(True); sol.reset(); sol.run(n, seed=1234); sol.sync(); c = sol.counters()
print({k: round(v/n, 2) for k, v in c.items() if v})
print('phase A lane utilisation %.3f  phase B %.3f ; A slots/photon %.1f  B slots/photon %.1f' % (
    c['sched_a_lanes']/max(c['sched_a_slots'], 1), c['sched_b_lanes']/max(c['sched_b_slots'], 1), c['sched_a_slots']/n, c['sched_b_slots']/n))
tk = [c[k] for k in ('ticks_a', 'ticks_b0', 'ticks_b12', 'ticks_b34', 'ticks_b5', 'ticks_b6')]
print('share of wave time: A %.3f  B0 %.3f  B1+B2 %.3f  B3+B4 %.3f  B5 %.3f  B6 %.3f' % tuple(t/sum(tk) for t in tk))
sol.set_counting(False)
sol.reset(); sol.run(20000000, seed=77); sol.sync(); ms, _ = sol.timing(); print('%.4g photons/s' % (2e7/(ms*1e-3)))
