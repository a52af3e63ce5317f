"""two solver handles of ONE process working at once on one GPU (a thread each, own non-blocking streams) against one handle
doing all the photons: tools/two_handles.py [workload] [photons in all]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from bench import make_scene
work = sys.argv[1] if len(sys.argv) > 1 else 'les480_mv9'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 80000000
sc = make_scene(work)
sols = [Mi3dSolver(0) for _ in range(2)]
for s in sols:
    s.set_tuning(own_stream=1); s.load_scene(sc); s.set_counting(False); s.reset(); s.run(200000, seed=1); s.sync()
def one(s, cnt, off):
    s.reset(); s.run(cnt, seed=7, offset=off); s.sync()
for rep in range(2):
    t0 = time.perf_counter(); one(sols[0], n, 0); t1 = time.perf_counter()
    th = [threading.Thread(target=one, args=(sols[i], n//2, i*(n//2))) for i in range(2)]
    t2 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t3 = time.perf_counter()
    print('%s, %d photons: one handle %.1f ms (%.4g photons/s); two handles at once %.1f ms (%.4g photons/s)' % (work, n, (t1-t0)*1e3, n/(t1-t0), (t3-t2)*1e3, n/(t3-t2)), flush=True)
