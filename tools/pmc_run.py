"""single launch of the transport kernel for PMC passes: tools/pmc_run.py <photons> [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
nph = int(float(sys.argv[1])); work = sys.argv[2] if len(sys.argv) > 2 else 'les480'
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc); sol.set_counting(False)
sol.reset(); sol.run(nph, seed=1234); sol.sync()
ms, nl = sol.timing(); print('%d photons %.2f ms %.4g photons/s' % (nph, ms, nph/(ms*1e-3)))
