"""throughput vs domain size (cache residency sweep); same cloud statistics, timing build"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene, z_levels_config4
sol = Mi3dSolver(0)
nph = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
for n in (16, 32, 64, 128, 256, 480):
    sc = les_scene(nx=n, ny=n, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
    sol.load_scene(sc); sol.set_counting(False); sol.reset()
    sol.run(nph//10, seed=1); sol.sync(); sol.reset()
    sol.run(nph, seed=1234); sol.sync()
    ms, nl = sol.timing()
    print('n=%4d  bext %.1f MB : %.1f ms -> %.3g photons/s' % (n, n*n*100*4/1e6, ms, nph/(ms*1e-3)), flush=True)
