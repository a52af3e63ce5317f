"""Do the pre-pass kernels of launch i + 1 run beside the photon loop of launch i?  Run under `rocprofv3 --kernel-trace`:
   MI3D_LIBRARY=... rocprofv3 --kernel-trace -d DIR -o trace -- python3 tools/overlap_trace.py [photons]
then `python tools/overlap_trace.py --read DIR/.../trace_kernel_trace.csv` prints the launches' intervals on one time axis."""
import csv, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
if len(sys.argv) > 2 and sys.argv[1] == '--read':
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_transport', 'k_entry', 'k_bin_', 'k_tl_', 'k_rays'))]
    t0 = min(int(r['Start_Timestamp']) for r in rows)
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    for r in rows[-int(os.environ.get('TRACE_ROWS', '24')):]:
        a, b = (int(r['Start_Timestamp']) - t0) * 1e-6, (int(r['End_Timestamp']) - t0) * 1e-6
        print('%-44s queue %-3s %10.3f -> %10.3f ms  (%8.3f)' % (r['Kernel_Name'][:44], r.get('Queue_Id', '?'), a, b, b - a))
    sys.exit(0)
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
nph = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
sol = Mi3dSolver(0); sol.load_scene(make_scene(os.environ.get('AB_WORKLOAD', 'les480'))); sol.set_counting(False)
sol.run(nph // 10, seed=1); sol.sync(); sol.reset()
for q in range(int(os.environ.get('AB_STEPS', '1'))): sol.run(nph, seed=7, offset=q*nph)
sol.sync()
print(sol.timing())
