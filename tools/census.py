"""how many lanes of a tally instruction add to the same address (a wave-level combine could merge them)
   needs the census build:  make -C er3t_amd/csrc OUT=../../tools/ab_census.so EXTRA=-DMI3D_CENSUS
   MI3D_LIBRARY=tools/ab_census.so python tools/census.py [workload] [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1] if len(sys.argv) > 1 else 'les480'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000000
sol = Mi3dSolver(0); sc = make_scene(work); sol.load_scene(sc)
sol.set_counting(True); sol.reset(); sol.run(n, seed=1234); sol.sync(); c = sol.counters()
ninstr, lanes, distinct = c['ticks_b34'], c['ticks_b5'], c['ticks_b6']
print('%s  %s  %d photons: %.2f tally lanes per photon in %.3g tally instructions (%.1f lanes each); %.3g distinct addresses in all; lanes that share their address with an earlier lane of the same instruction: %.3f %%'
      % (work, sol.kernel_name(), n, lanes/n, ninstr, lanes/max(ninstr, 1), distinct, 100.0*(lanes-distinct)/max(lanes, 1)))
