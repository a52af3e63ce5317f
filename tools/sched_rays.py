"""instrumented run of a marched-view workload: tools/sched_rays.py [workload] [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1] if len(sys.argv) > 1 else 'les480_mv9'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000000
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc)
sol.set_counting(True); sol.reset(); sol.run(n, seed=1234); sol.sync(); c = sol.counters()
print(sol.kernel_name(), {k: round(v/n, 3) for k, v in c.items() if v})
print('lane utilisation (photon loop and ray kernel together): phase A %.3f  phase B %.3f ; A wave-iterations/photon %.2f  B passes/photon %.2f' % (
    c['sched_a_lanes']/max(c['sched_a_slots'], 1), c['sched_b_lanes']/max(c['sched_b_slots'], 1), c['sched_a_slots']/n/64.0, c['sched_b_slots']/n/64.0))
sol.set_counting(False)
for r in range(2):
    sol.reset(); sol.run(4*n, seed=77+r); sol.sync(); ms, _ = sol.timing(); print('%.4g photons/s' % (4*n/(ms*1e-3)))
# (ticks_b12 ... ticks_b6 hold the photon loop's blocks AND, since round 5, the ray kernels' shares: run with MI3D_RAYS_ONLY_TICKS=1 in mind -- the
#  counters are sums over all kernels of the run; the ray kernels' are cyc[2..5] = ticks_b12, ticks_b34, ticks_b5, ticks_b6)
tot = sum(c[k] for k in ('ticks_b12', 'ticks_b34', 'ticks_b5', 'ticks_b6'))
print('wave ticks per photon (photon loop and ray kernels summed): walk-or-C %.1f  uniform+tally-or-B4 %.1f  batches-or-B5 %.1f  pop-or-B6/B7 %.1f' % tuple(c[k]/n for k in ('ticks_b12', 'ticks_b34', 'ticks_b5', 'ticks_b6')))
