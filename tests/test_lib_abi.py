"""
The C-ABI boundary without a GPU: libmi3drt.so loads, exports every symbol include/mi3d.h declares, and
refuses to compute (loudly) when there is no device.  No compute calls here.
"""

import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mi3d.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mi3d_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ('mi3d_create', 'mi3d_destroy', 'mi3d_set_atm1d', 'mi3d_set_atm3d', 'mi3d_set_phase', 'mi3d_set_surface',
                 'mi3d_set_surface2d', 'mi3d_set_source', 'mi3d_set_views', 'mi3d_set_options', 'mi3d_run',
                 'mi3d_get_radiance', 'mi3d_get_flux', 'mi3d_get_counters', 'mi3d_last_error', 'mi3d_bind_device_buffers'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from er3t_amd import solver
    path = solver.library_path()
    assert os.path.exists(path), 'libmi3drt.so is not built: run __graft_entry__.build()'
    lib = C.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), 'missing export %s' % name
    # and the binding table of the Python wrapper covers the same set
    assert sorted(n for n, _, _ in solver._SIGNATURES) == declared_symbols()
    lib2 = solver.load_library()
    assert lib2.mi3d_version() == 100


def test_no_silent_cpu_fallback():
    """without a usable GPU mi3d_create must fail with an error message -- never compute on the host"""
    from er3t_amd import solver
    lib = solver.load_library()
    if lib.mi3d_device_count() > 0:
        pytest.skip('a GPU is visible: the failure path cannot be exercised here')
    with pytest.raises(OSError) as err:
        solver.Mi3dSolver(device=0)
    assert 'no HIP device' in str(err.value) or 'CPU fallback' in str(err.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'er3t_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'libmi3d_oracle' not in text, f


@pytest.mark.gpu
@pytest.mark.parametrize('job', ['flux', 'radiance'])
def test_mi3d_run_leaves_the_tallies_complete_in_stream_order(solver, job):
    """The ONE completion rule of the C-ABI (include/mi3d.h, mi3d_run): when mi3d_run returns, everything it has started is queued on, or
    joined to, the stream the caller has bound.  A plain asynchronous copy out of the bound tally buffer, queued on that stream straight
    behind three runs -- no mi3d call in between, the flux job's record sorts on a stream of the library's own under the default
    "overlap_sort" -- must read what mi3d_sync + a second copy read: byte for byte."""
    import numpy as np
    import torch
    from er3t_amd.synth import les_scene
    dev = torch.device('cuda:0')
    sc = les_scene(nx=48, ny=48, nz3=50, target='flux', aerosol=True) if job == 'flux' else les_scene(nx=48, ny=48, nz3=50)
    n = 2000000
    st = torch.cuda.Stream(device=dev)
    nel = 3*(sc.nz+1)*sc.ny*sc.nx if job == 'flux' else sc.nview*sc.nyr*sc.nxr
    buf = torch.zeros(nel, dtype=torch.float64, device=dev)
    host = torch.zeros(nel, dtype=torch.float64).pin_memory()
    torch.cuda.synchronize(dev)
    try:
        if job == 'flux':
            solver.bind(None, buf.data_ptr(), st.cuda_stream)
        else:
            solver.bind(buf.data_ptr(), None, st.cuda_stream)
        solver.load_scene(sc); solver.set_counting(False); solver.reset()
        for q in range(3):
            solver.run(n, seed=5, offset=q*n)
        with torch.cuda.stream(st):
            host.copy_(buf, non_blocking=True)          # (hipMemcpyAsync on the caller's stream, nothing of mi3d in between)
        st.synchronize()
        first = host.numpy().copy()
        solver.sync()
        again = buf.cpu().numpy()
        assert first.sum() > 0.0
        assert np.array_equal(first, again), 'a copy queued behind mi3d_run on the bound stream read tallies that were still being written'
    finally:
        solver.bind(None, None, None)


def test_the_tuning_table_of_the_header_names_the_keys_the_library_takes():
    """include/mi3d.h documents mi3d_set_tuning as a table (key, default, range, what, the log that set it): every key of the table is one
    the library compares against, and every key the library compares against stands in the table"""
    text = open(os.path.join(ROOT, 'include', 'mi3d.h')).read()
    block = text[text.index(' *   key            default'):text.index('int mi3d_set_tuning(')]
    keys = set()
    for ln in block.splitlines()[2:]:
        m = re.match(r' \*   ([a-z_0-9/]+)\s+(-?\d+)\s', ln)
        if m:
            for k in m.group(1).split('/'):
                keys.add(k if not k.startswith('_') else 'vpad' + k)
    src = open(os.path.join(ROOT, 'er3t_amd', 'csrc', 'mi3d_api.hip')).read()
    fn = src[src.index('int mi3d_set_tuning('):]
    fn = fn[:fn.index('\nint mi3d_', 10)]
    taken = set(re.findall(r'k == "([a-z_0-9]+)"', fn))
    assert keys == taken, (sorted(keys - taken), sorted(taken - keys))
