import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.solver import Mi3dSolver
from er3t_amd.synth import les_scene
dev = torch.device('cuda', 0)
rad = torch.zeros(16*16, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev)
print('stream handle', st.cuda_stream, 'ptr', hex(rad.data_ptr()))
sol = Mi3dSolver(0)
sc = les_scene(nx=16, ny=16, nz3=50)
for stream in (None, st.cuda_stream):
    try:
        sol.bind(rad_ptr=rad.data_ptr(), stream=stream)
        sol.load_scene(sc); sol.reset(); sol.run(10000, seed=1); sol.sync()
        print('stream', stream, 'ok', float(rad.sum()))
    except OSError as e:
        print('stream', stream, 'FAIL', e)
s2 = torch.cuda.Stream(dev)
with torch.cuda.stream(s2):
    try:
        sol.bind(rad_ptr=rad.data_ptr(), stream=s2.cuda_stream)
        sol.reset(); sol.run(10000, seed=1); sol.sync()
        print('side stream', s2.cuda_stream, 'ok', float(rad.sum()))
    except OSError as e:
        print('side stream FAIL', e)
