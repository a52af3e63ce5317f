"""k_tl_runs by phase (a -DMI3D_RUNS_DIAG build as MI3D_LIBRARY): tools/r06_runsdiag.py <workload> <photons>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1]; n = int(float(sys.argv[2]))
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc); sol.set_counting(False); sol.set_tuning(overlap_sort=0, overlap_pre=0)
sol.reset(); sol.run(n, seed=1); sol.sync(); sol.reset(); sol.run(n, seed=2); sol.sync()
c = sol.counters()
it, ln = c['ticks_b5'], c['ticks_b6']
print('%s %d photons: write pass wave clocks/64: load %d  classify+sort %d  levels %d;  level steps of waves %d (%.2f per photon), lanes at work %.3f, records %.2f per photon'
      % (work, n, c['ticks_a'], c['ticks_b0'], c['ticks_b12'], it, it/n, ln/max(64*it, 1), ln/n))
tiles = max(c['sched_b_slots'], 1)
print('  tiles %d (%.1f runs each): pieces per tile %.1f (%.1f of them full), classes per tile %.1f, the largest class %.1f runs, runs without a class of their own per tile %.1f'
      % (tiles, c['escaped']/tiles, c['ticks_b34']/tiles, c['sched_a_lanes']/tiles, c['sched_b_lanes']/tiles, c['absorbed']/tiles, c['sched_a_slots']/tiles))
