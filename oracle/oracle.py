"""
ctypes wrapper of the CPU oracle (oracle/mi3d_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by er3t_amd.

    res = run(scene, nphoton, seed=1, offset=0, nthreads=8)
    res['rad']  (nview, nyr, nxr) float64   normalised radiance  (per unit Src_flx)
    res['flux'] (3, nz+1, ny, nx) float64   direct-down, total-down, up
    res['heat'] (nz, ny, nx) float64        absorbed power per unit volume (scene.target & 4: heating rates), else absent
    res['counters'] dict
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_VIEW = 16
NCOUNTER = 16
COUNTER_NAMES = ['photons', 'steps', 'steps3d', 'scatter', 'surface', 'le_rays', 'le_steps', 'le_steps3d',
                 'le_column', 'flux_tally', 'roulette', 'killed', 'escaped', 'absorbed', 'rsv14', 'rsv15']

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)


class _Config(C.Structure):
    _fields_ = [
        ('nz', C.c_int), ('zgrd', _dp), ('np1d', C.c_int),
        ('ext1d', _fp), ('omg1d', _fp), ('apf1d', _fp), ('abs1d', _fp),
        ('nx', C.c_int), ('ny', C.c_int), ('nz3', C.c_int), ('iz3l', C.c_int), ('np3d', C.c_int),
        ('dx', C.c_double), ('dy', C.c_double),
        ('abst', _fp), ('extp', _fp), ('omgp', _fp), ('apfp', _fp),
        ('nang', C.c_int), ('npf', C.c_int), ('ang', _fp), ('pha', _fp),
        ('sfc_mtype', C.c_int), ('sfc_param', C.c_float*5), ('nxb', C.c_int), ('nyb', C.c_int),
        ('jsfc', _fp), ('psfc', _fp),
        ('src_flx', C.c_double), ('src_qmax', C.c_double), ('src_the', C.c_double), ('src_phi', C.c_double),
        ('nview', C.c_int), ('view_the', C.c_double*MAX_VIEW), ('view_phi', C.c_double*MAX_VIEW),
        ('view_zloc', C.c_double*MAX_VIEW), ('zref', C.c_double), ('nxr', C.c_int), ('nyr', C.c_int),
        ('target', C.c_int), ('solver', C.c_int), ('wmin', C.c_double), ('wfac', C.c_double), ('nthreads', C.c_int),
        ('le_tau1', C.c_double),
        ('rad_kind', C.c_int), ('cam_xpos', C.c_double*MAX_VIEW), ('cam_ypos', C.c_double*MAX_VIEW), ('cam_psi', C.c_double*MAX_VIEW),
        ('cam_qmax', C.c_double*MAX_VIEW), ('cam_umax', C.c_double*MAX_VIEW), ('cam_vmax', C.c_double*MAX_VIEW),
        ('cam_apsize', C.c_double*MAX_VIEW),
        ('le_cmin', C.c_double),
        ('cam_images', C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, 'libmi3d_oracle.so')
    src = os.path.join(_HERE, 'mi3d_oracle.c')
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libmi3d_oracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_run.restype = C.c_int
        _LIB.orc_run.argtypes = [C.POINTER(_Config), C.c_uint64, C.c_uint64, C.c_uint64, _dp, _dp,
                                 C.POINTER(C.c_uint64)]
        _LIB.orc_run_heat.restype = C.c_int
        _LIB.orc_run_heat.argtypes = [C.POINTER(_Config), C.c_uint64, C.c_uint64, C.c_uint64, _dp, _dp, _dp,
                                      C.POINTER(C.c_uint64)]
        _LIB.orc_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]
        _LIB.orc_philox_raw.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _LIB.orc_lsrt.restype = C.c_double
        _LIB.orc_lsrt.argtypes = [C.c_double]*3 + [_dp, _dp]
        _LIB.orc_fresnel.restype = C.c_double
        _LIB.orc_fresnel.argtypes = [C.c_double]*3
        _LIB.orc_dsm.restype = None
        _LIB.orc_dsm.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        _LIB.orc_phase_eval.restype = C.c_double
        _LIB.orc_phase_eval.argtypes = [C.c_double, C.c_double]
        _LIB.orc_phase_sample.restype = C.c_double
        _LIB.orc_phase_sample.argtypes = [C.c_double, C.c_double]
        _LIB.orc_phase_table.argtypes = [C.POINTER(_Config), C.c_int, C.c_int, _dp, _dp, _dp, _dp]
    return _LIB


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _config(scene, nthreads=1):
    s = scene
    cfg = _Config()
    keep = []

    def f32(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float32)
        keep.append(a)
        return a

    zg = np.ascontiguousarray(s.zgrd, dtype=np.float64); keep.append(zg)
    cfg.nz = s.nz; cfg.zgrd = _ptr(zg, _dp); cfg.np1d = s.np1d
    cfg.ext1d = _ptr(f32(s.ext1d), _fp); cfg.omg1d = _ptr(f32(s.omg1d), _fp)
    cfg.apf1d = _ptr(f32(s.apf1d), _fp); cfg.abs1d = _ptr(f32(s.abs1d), _fp)
    cfg.nx = s.nx; cfg.ny = s.ny; cfg.nz3 = s.nz3; cfg.iz3l = s.iz3l; cfg.np3d = s.np3d
    cfg.dx = s.dx; cfg.dy = s.dy
    cfg.abst = _ptr(f32(s.abst), _fp); cfg.extp = _ptr(f32(s.extp), _fp)
    cfg.omgp = _ptr(f32(s.omgp), _fp); cfg.apfp = _ptr(f32(s.apfp), _fp)
    cfg.npf = s.npf; cfg.nang = 0 if s.ang is None else s.ang.size
    cfg.ang = _ptr(f32(s.ang), _fp); cfg.pha = _ptr(f32(s.pha), _fp)
    cfg.sfc_mtype = s.sfc_mtype
    for i in range(5):
        cfg.sfc_param[i] = float(s.sfc_param[i])
    if s.jsfc is not None:
        cfg.nyb, cfg.nxb = s.jsfc.shape
        cfg.jsfc = _ptr(f32(s.jsfc), _fp); cfg.psfc = _ptr(f32(s.psfc), _fp)
    cfg.src_flx = s.src_flx; cfg.src_qmax = s.src_qmax; cfg.src_the = s.src_the; cfg.src_phi = s.src_phi
    cfg.nview = s.nview
    for i in range(s.nview):
        cfg.view_the[i] = s.view_the[i]; cfg.view_phi[i] = s.view_phi[i]; cfg.view_zloc[i] = s.view_zloc[i]
    cfg.zref = s.zref; cfg.nxr = s.nxr; cfg.nyr = s.nyr
    cfg.target = s.target; cfg.solver = s.solver; cfg.wmin = s.wmin; cfg.wfac = s.wfac
    cfg.nthreads = nthreads
    cfg.le_tau1 = float(getattr(s, 'le_tau1', 0.0))
    cfg.le_cmin = float(getattr(s, 'le_cmin', 0.0))
    cfg.cam_images = int(getattr(s, 'cam_images', 0))
    if cfg.cam_images < 0:            # (Scene's default, -1: what the library's default route serves)
        cfg.cam_images = 2
    cfg.rad_kind = int(getattr(s, 'rad_kind', 2))
    if cfg.rad_kind == 1:
        for i in range(s.nview):
            cfg.cam_xpos[i] = s.cam_xpos[i]; cfg.cam_ypos[i] = s.cam_ypos[i]; cfg.cam_psi[i] = s.cam_psi[i]
            cfg.cam_qmax[i] = s.cam_qmax[i]; cfg.cam_umax[i] = s.cam_umax[i]; cfg.cam_vmax[i] = s.cam_vmax[i]
            cfg.cam_apsize[i] = s.cam_apsize[i]
    return cfg, keep


def run_raw(scene, nphoton, seed=1, offset=0, nthreads=1, heat=None):
    """raw (un-normalised) tallies: rad_sum, flux_sum, counters; heat (nz, ny, nx) float64 is added to when given"""
    cfg, keep = _config(scene, nthreads)
    rad = np.zeros((max(scene.nview, 1), scene.nyr, scene.nxr), dtype=np.float64)
    flux = np.zeros((3, scene.nz+1, scene.ny, scene.nx), dtype=np.float64)
    cnt = np.zeros(NCOUNTER, dtype=np.uint64)
    rc = lib().orc_run_heat(C.byref(cfg), int(nphoton), int(seed), int(offset), _ptr(rad, _dp), _ptr(flux, _dp), _ptr(heat, _dp),
                            cnt.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise OSError('Error [oracle]: orc_run returned %d.' % rc)
    return rad[:scene.nview], flux, cnt


def normalise(scene, rad_sum, flux_sum, nphoton):
    """radiance per unit Src_flx... times Src_flx; flux likewise (see include/mi3d.h: mi3d_get_radiance)"""
    mu0 = scene.mu0
    if getattr(scene, 'rad_kind', 2) == 1:     # camera: the tally already holds the 1 / (r^2 dOmega) of every contribution
        rad = rad_sum * (scene.src_flx * mu0 * scene.nx * scene.dx * scene.ny * scene.dy / float(nphoton))
    else:
        rad = rad_sum * (scene.src_flx * mu0 * scene.nxr * scene.nyr / float(nphoton))
    flux = flux_sum * (scene.src_flx * mu0 * scene.nx * scene.ny / float(nphoton))
    return rad, flux


def run(scene, nphoton, seed=1, offset=0, nthreads=1):
    heat = np.zeros((scene.nz, scene.ny, scene.nx), dtype=np.float64) if scene.target & 4 else None
    rad_sum, flux_sum, cnt = run_raw(scene, nphoton, seed, offset, nthreads, heat=heat)
    rad, flux = normalise(scene, rad_sum, flux_sum, nphoton)
    out = {'rad': rad, 'flux': flux, 'counters': dict(zip(COUNTER_NAMES, (int(v) for v in cnt))),
           'rad_sum': rad_sum, 'flux_sum': flux_sum}
    if heat is not None:
        # absorbed power per unit volume, per unit Src_flx: weight absorbed in the cell x (Src_flx mu0 nx ny / N) / layer thickness
        out['heat'] = heat*(scene.src_flx*scene.mu0*scene.nx*scene.ny/float(nphoton))/np.diff(scene.zgrd)[:, None, None]
    return out


def philox(seed, ident, draw):
    out = (C.c_uint32*4)()
    lib().orc_philox(int(seed), int(ident), int(draw), out)
    return np.array(out[:], dtype=np.uint32)


def philox_raw(ctr, key):
    c = (C.c_uint32*4)(*[int(x) for x in ctr]); k = (C.c_uint32*2)(*[int(x) for x in key]); out = (C.c_uint32*4)()
    lib().orc_philox_raw(c, k, out)
    return np.array(out[:], dtype=np.uint32)


def lsrt(fiso, fgeo, fvol, din, dout):
    a = np.ascontiguousarray(din, dtype=np.float64); b = np.ascontiguousarray(dout, dtype=np.float64)
    return lib().orc_lsrt(fiso, fgeo, fvol, _ptr(a, _dp), _ptr(b, _dp))


def fresnel(nr, ni, cos_inc):
    return lib().orc_fresnel(float(nr), float(ni), float(cos_inc))


def dsm(params, din, dout):
    """reflectance factor of the diffuse-specular mixture (jsfc = 2) for one incoming and n outgoing directions"""
    p = np.ascontiguousarray(params, dtype=np.float64); a = np.ascontiguousarray(din, dtype=np.float64)
    b = np.ascontiguousarray(np.atleast_2d(dout), dtype=np.float64)
    out = np.zeros(b.shape[0])
    lib().orc_dsm(_ptr(p, _dp), _ptr(a, _dp), b.shape[0], _ptr(b, _dp), _ptr(out, _dp))
    return out


def phase_eval(apf, mu):
    return lib().orc_phase_eval(float(apf), float(mu))


def phase_sample(apf, u):
    return lib().orc_phase_sample(float(apf), float(u))


def phase_table(scene, itable, mu, u):
    cfg, keep = _config(scene, 1)
    mu = np.ascontiguousarray(mu, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    p = np.zeros_like(mu); m = np.zeros_like(u)
    assert mu.size == u.size
    rc = lib().orc_phase_table(C.byref(cfg), int(itable), mu.size, _ptr(mu, _dp), _ptr(u, _dp), _ptr(p, _dp), _ptr(m, _dp))
    if rc != 0:
        raise OSError('Error [oracle]: orc_phase_table returned %d.' % rc)
    return p, m
