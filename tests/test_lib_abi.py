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
