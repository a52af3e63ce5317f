"""
ctypes binding of libmi3drt.so (C-ABI: include/mi3d.h) and a small object wrapper.

This is the in-process replacement of the reference's process boundary
`os.system('<MCARATS_V010_EXE> <Nphoton> <solver> <inp> <out>')` (er3t/rtm/mca/mca_run.py:101-115,
179-181).  There is no CPU fallback: if the shared library is missing or no GPU is visible every
compute call raises OSError.
"""

import ctypes as C
import os

import numpy as np

from .scene import Scene, TARGET_FLUX, TARGET_RADIANCE, TARGET_HEAT

__all__ = ['Mi3dSolver', 'load_library', 'library_path', 'COUNTER_NAMES']

MAX_VIEW = 16
NCOUNTER = 24
COUNTER_NAMES = ['photons', 'steps', 'steps3d', 'scatter', 'surface', 'le_rays', 'le_steps', 'le_steps3d',
                 'le_column', 'flux_tally', 'roulette', 'killed', 'escaped', 'absorbed',
                 'sched_a_lanes', 'sched_a_slots', 'sched_b_lanes', 'sched_b_slots',
                 'ticks_a', 'ticks_b0', 'ticks_b12', 'ticks_b34', 'ticks_b5', 'ticks_b6']

_fp  = C.POINTER(C.c_float)
_dp  = C.POINTER(C.c_double)
_u64 = C.c_uint64
_LIB = None

# every symbol include/mi3d.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ('mi3d_version'            , C.c_int   , []),
    ('mi3d_last_error'         , C.c_char_p, []),
    ('mi3d_device_count'       , C.c_int   , []),
    ('mi3d_create'             , C.c_int   , [C.c_int, C.POINTER(C.c_void_p)]),
    ('mi3d_destroy'            , C.c_int   , [C.c_void_p]),
    ('mi3d_set_atm1d'          , C.c_int   , [C.c_void_p, C.c_int, _dp, C.c_int, _fp, _fp, _fp, _fp]),
    ('mi3d_set_atm3d'          , C.c_int   , [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _fp, _fp, _fp, _fp]),
    ('mi3d_set_phase'          , C.c_int   , [C.c_void_p, C.c_int, C.c_int, _fp, _fp]),
    ('mi3d_set_surface'        , C.c_int   , [C.c_void_p, C.c_int, _fp]),
    ('mi3d_set_surface2d'      , C.c_int   , [C.c_void_p, C.c_int, C.c_int, _fp, _fp, _fp]),
    ('mi3d_set_source'         , C.c_int   , [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    ('mi3d_set_views'          , C.c_int   , [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int]),
    ('mi3d_set_cameras'        , C.c_int   , [C.c_void_p, C.c_int] + [_dp]*10 + [C.c_int, C.c_int]),
    ('mi3d_set_options'        , C.c_int   , [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    ('mi3d_set_le_roulette'    , C.c_int   , [C.c_void_p, C.c_double]),
    ('mi3d_set_le_weight_roulette', C.c_int, [C.c_void_p, C.c_double]),
    ('mi3d_set_counting'       , C.c_int   , [C.c_void_p, C.c_int]),
    ('mi3d_bind_device_buffers', C.c_int   , [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('mi3d_bind_heating_buffer', C.c_int   , [C.c_void_p, C.c_void_p]),
    ('mi3d_prepare'            , C.c_int   , [C.c_void_p]),
    ('mi3d_reset'              , C.c_int   , [C.c_void_p]),
    ('mi3d_run'                , C.c_int   , [C.c_void_p, _u64, _u64, _u64]),
    ('mi3d_sync'               , C.c_int   , [C.c_void_p]),
    ('mi3d_last_kernel'        , C.c_char_p, [C.c_void_p]),
    ('mi3d_set_kernel'         , C.c_int   , [C.c_void_p, C.c_int]),
    ('mi3d_set_tuning'         , C.c_int   , [C.c_void_p, C.c_char_p, C.c_int]),
    ('mi3d_get_timing'         , C.c_int   , [C.c_void_p, _dp, C.POINTER(_u64)]),
    ('mi3d_get_radiance'       , C.c_int   , [C.c_void_p, _u64, _fp]),
    ('mi3d_get_flux'           , C.c_int   , [C.c_void_p, _u64, _fp]),
    ('mi3d_get_direct_levels'  , C.c_int   , [C.c_void_p, _dp]),
    ('mi3d_get_heating'        , C.c_int   , [C.c_void_p, _u64, _fp]),
    ('mi3d_get_counters'       , C.c_int   , [C.c_void_p, C.POINTER(_u64)]),
    ('mi3d_stats_begin'        , C.c_int   , [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('mi3d_stats_set_analytic_share', C.c_int, [C.c_void_p, C.c_double]),
    ('mi3d_stats_add'          , C.c_int   , [C.c_void_p, _u64, _fp, _fp]),
    ('mi3d_stats_join'         , C.c_int   , [C.c_void_p, C.c_void_p]),
    ('mi3d_stats_chain'        , C.c_int   , [C.c_void_p, C.c_void_p]),
    ('mi3d_stats_end_run'      , C.c_int   , [C.c_void_p, _fp, _fp]),
    ('mi3d_stats_get'          , C.c_int   , [C.c_void_p, C.c_int, _fp, _fp, C.POINTER(C.c_int)]),
    ('mi3d_debug_philox'       , C.c_int   , [C.c_void_p, _u64, _u64, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]),
    ('mi3d_debug_order'        , C.c_int   , [C.c_void_p, _u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]),
]


def library_path():
    # MI3D_LIBRARY selects another build of the SAME library (kernel A/B experiments); never a fallback
    return os.environ.get('MI3D_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmi3drt.so')


def load_library():

    """
    Load libmi3drt.so (built in-tree by `__graft_entry__.build()` / `make -C er3t_amd/csrc`).
    Raises OSError when it is missing -- the product path never substitutes anything for it.
    """

    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            msg = 'Error [Mi3dSolver]: Cannot find <%s>. Build it with `python -c "import __graft_entry__ as g; g.build()"`.' % path
            raise OSError(msg)
        lib = C.CDLL(path)
        for name, restype, argtypes in _SIGNATURES:
            try:
                fn = getattr(lib, name)     # AttributeError here means the library is stale
            except AttributeError:
                if os.environ.get('MI3D_LIBRARY'):   # an older build in a kernel A/B experiment (tools/ab.py): what it lacks fails when called
                    setattr(lib, name, _stale(name, path))
                    continue
                msg = 'Error [Mi3dSolver]: <%s> is stale: it lacks <%s>. Rebuild it with `make -C er3t_amd/csrc`.' % (path, name)
                raise OSError(msg)
            fn.restype  = restype
            fn.argtypes = argtypes
        _LIB = lib
    return _LIB


def _stale(name, path):
    def missing(*args):
        msg = 'Error [Mi3dSolver]: library is stale: <%s> missing from <%s>.' % (name, path)
        raise OSError(msg)
    return missing


def _ptr(a, typ=_fp):
    return None if a is None else a.ctypes.data_as(typ)


class Mi3dSolver:

    """
    One solver instance bound to one GPU.

        sol = Mi3dSolver(device=0)
        sol.load_scene(scene)                 # er3t_amd.scene.Scene
        sol.run(nphoton, seed=..., offset=0)  # asynchronous, accumulates
        rad  = sol.radiance(nphoton_total)    # (nview, nyr, nxr) float32
        flux = sol.flux(nphoton_total)        # (3, nz+1, ny, nx) float32
        heat = sol.heating(nphoton_total)     # (nz, ny, nx) float32, scene.target & TARGET_HEAT
    """

    def __init__(self, device=0):
        self.lib = load_library()
        self._h  = C.c_void_p()
        self._chk(self.lib.mi3d_create(int(device), C.byref(self._h)))
        self.device = device
        self.scene  = None

    def _chk(self, rc):
        if rc != 0:
            msg = 'Error [Mi3dSolver]: %s (code %d).' % (self.lib.mi3d_last_error().decode('utf-8', 'replace'), rc)
            raise OSError(msg)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.mi3d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ------------------------------------------------------------------------------
    def set_atm1d(self, zgrd, ext1d, omg1d, apf1d, abs1d):
        zgrd  = np.ascontiguousarray(zgrd, dtype=np.float64)
        ext1d = np.ascontiguousarray(np.atleast_2d(ext1d), dtype=np.float32)
        omg1d = np.ascontiguousarray(np.atleast_2d(omg1d), dtype=np.float32)
        apf1d = np.ascontiguousarray(np.atleast_2d(apf1d), dtype=np.float32)
        abs1d = np.ascontiguousarray(abs1d, dtype=np.float32)
        nz = zgrd.size - 1
        if ext1d.shape[1] != nz or omg1d.shape != ext1d.shape or apf1d.shape != ext1d.shape or abs1d.size != nz:
            raise ValueError('Error [Mi3dSolver]: 1-D profiles do not match <zgrd>.')
        self._chk(self.lib.mi3d_set_atm1d(self._h, nz, _ptr(zgrd, _dp), ext1d.shape[0], _ptr(ext1d), _ptr(omg1d), _ptr(apf1d), _ptr(abs1d)))

    def set_atm3d(self, nx, ny, dx, dy, nz3=0, iz3l=1, abst=None, extp=None, omgp=None, apfp=None):
        np3d = 0
        if nz3 > 0:
            extp = np.ascontiguousarray(extp, dtype=np.float32); omgp = np.ascontiguousarray(omgp, dtype=np.float32)
            apfp = np.ascontiguousarray(apfp, dtype=np.float32)
            if extp.ndim == 3:
                extp = extp[None]; omgp = omgp[None]; apfp = apfp[None]
            np3d = extp.shape[0]
            if extp.shape != (np3d, nz3, ny, nx) or omgp.shape != extp.shape or apfp.shape != extp.shape:
                raise ValueError('Error [Mi3dSolver]: 3-D arrays must be (np3d, nz3, ny, nx).')
            if abst is not None:
                abst = np.ascontiguousarray(abst, dtype=np.float32)
                if abst.shape != (nz3, ny, nx):
                    raise ValueError('Error [Mi3dSolver]: <abst> must be (nz3, ny, nx).')
        self._chk(self.lib.mi3d_set_atm3d(self._h, int(nx), int(ny), int(nz3), int(iz3l), int(np3d), float(dx), float(dy),
                                          _ptr(abst), _ptr(extp), _ptr(omgp), _ptr(apfp)))

    def set_phase(self, ang=None, pha=None):
        if pha is None:
            self._chk(self.lib.mi3d_set_phase(self._h, 0, 0, None, None))
            return
        ang = np.ascontiguousarray(ang, dtype=np.float32)
        pha = np.ascontiguousarray(np.atleast_2d(pha), dtype=np.float32)
        if pha.shape[1] != ang.size:
            raise ValueError('Error [Mi3dSolver]: <pha> must be (npf, nang).')
        self._chk(self.lib.mi3d_set_phase(self._h, ang.size, pha.shape[0], _ptr(ang), _ptr(pha)))

    def set_surface(self, mtype=1, param=(0.0, 0.0, 0.0, 0.0, 0.0)):
        p = np.zeros(5, dtype=np.float32)
        q = np.ravel(np.asarray(param, dtype=np.float32))
        p[:min(5, q.size)] = q[:5]
        self._chk(self.lib.mi3d_set_surface(self._h, int(mtype), _ptr(p)))

    def set_surface2d(self, jsfc, psfc):
        jsfc = np.ascontiguousarray(jsfc, dtype=np.float32)
        psfc = np.ascontiguousarray(psfc, dtype=np.float32)
        nyb, nxb = jsfc.shape
        if psfc.shape != (5, nyb, nxb):
            raise ValueError('Error [Mi3dSolver]: <psfc> must be (5, nyb, nxb).')
        self._chk(self.lib.mi3d_set_surface2d(self._h, nxb, nyb, None, _ptr(jsfc), _ptr(psfc)))

    def set_source(self, flx=1.0, qmax=0.533133, the=150.0, phi=270.0):
        self._chk(self.lib.mi3d_set_source(self._h, float(flx), float(qmax), float(the), float(phi)))

    def set_views(self, the, phi, zloc, zref=0.0, nxr=1, nyr=1):
        the = np.ascontiguousarray(np.atleast_1d(the), dtype=np.float64)
        phi = np.ascontiguousarray(np.atleast_1d(phi), dtype=np.float64)
        zloc = np.ascontiguousarray(np.atleast_1d(zloc), dtype=np.float64)
        if not (the.size == phi.size == zloc.size):
            raise ValueError('Error [Mi3dSolver]: view arrays differ in length.')
        self._chk(self.lib.mi3d_set_views(self._h, the.size, _ptr(the, _dp), _ptr(phi, _dp), _ptr(zloc, _dp), float(zref), int(nxr), int(nyr)))

    def set_cameras(self, the, phi, psi, xpos, ypos, zloc, qmax, umax, vmax, apsize, nxr, nyr):
        """all-sky cameras (Rad_mrkind = 1): point sensors at (xpos Lx, ypos Ly, zloc), Z-Y-Z rotations phi, the, psi"""
        arrs = [np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64) for a in (the, phi, psi, xpos, ypos, zloc, qmax, umax, vmax, apsize)]
        n = arrs[0].size
        if any(a.size != n for a in arrs):
            raise ValueError('Error [Mi3dSolver]: camera arrays differ in length.')
        self._chk(self.lib.mi3d_set_cameras(self._h, n, *[_ptr(a, _dp) for a in arrs], int(nxr), int(nyr)))

    def set_options(self, target=TARGET_FLUX, solver=0, wmin=0.2, wfac=1.0, column_le=True):
        self._chk(self.lib.mi3d_set_options(self._h, int(target), int(solver), float(wmin), float(wfac), 1 if column_le else 0))

    def set_le_roulette(self, tau1=0.0):
        self._chk(self.lib.mi3d_set_le_roulette(self._h, float(tau1)))

    def set_le_weight_roulette(self, cmin=0.0):
        self._chk(self.lib.mi3d_set_le_weight_roulette(self._h, float(cmin)))

    def set_counting(self, on=True):
        self._chk(self.lib.mi3d_set_counting(self._h, 1 if on else 0))

    def load_scene(self, scene, column_le=True):
        s = scene
        self.set_atm1d(s.zgrd, s.ext1d, s.omg1d, s.apf1d, s.abs1d)
        self.set_atm3d(s.nx, s.ny, s.dx, s.dy, nz3=s.nz3, iz3l=s.iz3l, abst=s.abst, extp=s.extp, omgp=s.omgp, apfp=s.apfp)
        self.set_phase(s.ang, s.pha)
        if s.jsfc is not None:
            self.set_surface2d(s.jsfc, s.psfc)
        else:
            self.set_surface(s.sfc_mtype, s.sfc_param)
        self.set_source(s.src_flx, s.src_qmax, s.src_the, s.src_phi)
        if getattr(s, 'rad_kind', 2) == 1 and s.nview > 0:
            self.set_cameras(s.view_the, s.view_phi, s.cam_psi, s.cam_xpos, s.cam_ypos, s.view_zloc, s.cam_qmax, s.cam_umax, s.cam_vmax,
                             s.cam_apsize, s.nxr, s.nyr)
            self.set_tuning(cam_images=int(getattr(s, 'cam_images', -1)))
        else:
            self.set_views(s.view_the, s.view_phi, s.view_zloc, zref=s.zref, nxr=s.nxr, nyr=s.nyr)
        self.set_options(s.target, s.solver, s.wmin, s.wfac, column_le)
        self.set_le_roulette(getattr(s, 'le_tau1', 0.0))
        self.set_le_weight_roulette(getattr(s, 'le_cmin', 0.0))
        self.scene = s
        self._shape_rad  = (s.nview, s.nyr, s.nxr)
        self._shape_flux = (3, s.nz+1, s.ny, s.nx)
        self.prepare()

    def update_atm1d(self, scene):
        """swap the 1-D profiles only (the per-g part of a correlated-k loop); 3-D arrays stay on the device"""
        self.set_atm1d(scene.zgrd, scene.ext1d, scene.omg1d, scene.apf1d, scene.abs1d)
        self.prepare()

    # ---- execution ---------------------------------------------------------------------------
    def bind(self, rad_ptr=None, flux_ptr=None, stream=None, heat_ptr=None):
        self._chk(self.lib.mi3d_bind_device_buffers(self._h, C.c_void_p(rad_ptr or 0), C.c_void_p(flux_ptr or 0), C.c_void_p(stream or 0)))
        self._chk(self.lib.mi3d_bind_heating_buffer(self._h, C.c_void_p(heat_ptr or 0)))

    def prepare(self):
        self._chk(self.lib.mi3d_prepare(self._h))

    def reset(self):
        self._chk(self.lib.mi3d_reset(self._h))

    def run(self, nphoton, seed=1, offset=0):
        self._chk(self.lib.mi3d_run(self._h, int(nphoton), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset)))

    def sync(self):
        self._chk(self.lib.mi3d_sync(self._h))

    def set_kernel(self, general=False):
        """general=True: always the general kernel build (k_transport), also where the lean ones apply (A/B and parity tests)"""
        self._chk(self.lib.mi3d_set_kernel(self._h, 1 if general else 0))

    def set_tuning(self, **knobs):
        """launch-machinery knobs (include/mi3d.h: mi3d_set_tuning), e.g. set_tuning(evcap_log2=12, own_stream=1);
        keys: tile_cols, batch_log2, evcap_log2, rad_spread, own_stream, tally_lists, tlcap_log2, entry_records, cam_images,
        tally_window, rad_row_pad, vpad_col, vpad_row, overlap_rays, overlap_sort, overlap_pre, tl_split, rays_wg, emit_wg"""
        for key, value in knobs.items():
            self._chk(self.lib.mi3d_set_tuning(self._h, key.encode(), int(value)))

    def kernel_name(self):
        """which build of the transport kernel served the last run (for logs; results do not depend on it)"""
        return self.lib.mi3d_last_kernel(self._h).decode()

    def timing(self):
        ms = C.c_double(0.0); n = _u64(0)
        self._chk(self.lib.mi3d_get_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    # ---- results -----------------------------------------------------------------------------
    def radiance(self, nphoton_total):
        out = np.zeros(self._shape_rad, dtype=np.float32)
        if out.size:
            self._chk(self.lib.mi3d_get_radiance(self._h, int(nphoton_total), _ptr(out)))
        return out

    def flux(self, nphoton_total):
        out = np.zeros(self._shape_flux, dtype=np.float32)
        self._chk(self.lib.mi3d_get_flux(self._h, int(nphoton_total), _ptr(out)))
        return out

    def direct_levels(self):
        """(nz+1,) the known part of the direct beam that mi3d_get_flux adds to the tallies, for the job that ran last"""
        out = np.zeros(self.scene.nz+1, dtype=np.float64)
        self._chk(self.lib.mi3d_get_direct_levels(self._h, _ptr(out, _dp)))
        return out

    def heating(self, nphoton_total):
        """absorbed power per unit volume and unit Src_flx, (nz, ny, nx): jobs whose target includes TARGET_HEAT (Flx_mhrt = 1)"""
        s = self.scene
        out = np.zeros((s.nz, s.ny, s.nx), dtype=np.float32)
        self._chk(self.lib.mi3d_get_heating(self._h, int(nphoton_total), _ptr(out)))
        return out

    # ---- run statistics on the device (sum over g per run, mean / std over runs) ---------------
    def stats_begin(self, rad_run_ptr=None, flux_run_ptr=None):
        self._chk(self.lib.mi3d_stats_begin(self._h, C.c_void_p(rad_run_ptr or 0), C.c_void_p(flux_run_ptr or 0)))

    def stats_join(self, owner):
        """this handle adds its jobs into the run fields of <owner> (another Mi3dSolver on the same device that has begun statistics)"""
        self._chk(self.lib.mi3d_stats_join(self._h, owner._h))

    def stats_chain(self, after):
        """this handle's next statistics kernel waits for the last one of <after> (another Mi3dSolver on the same device)"""
        self._chk(self.lib.mi3d_stats_chain(self._h, after._h))

    def stats_set_analytic_share(self, share):
        """photon-sharded jobs: 1 on the rank that adds the analytic direct beam to the run field, 0 on the others"""
        self._chk(self.lib.mi3d_stats_set_analytic_share(self._h, float(share)))

    def stats_add(self, nphoton_total, factor_rad=None, factor_flux=None):
        s = self.scene
        fr = ff = None
        if factor_rad is not None:
            fr = np.ascontiguousarray(np.broadcast_to(np.asarray(factor_rad, dtype=np.float32), (s.nview,)))
        if factor_flux is not None:
            ff = np.ascontiguousarray(np.broadcast_to(np.asarray(factor_flux, dtype=np.float32), (s.nz+1,)))
        self._chk(self.lib.mi3d_stats_add(self._h, int(nphoton_total), _ptr(fr), _ptr(ff)))

    def stats_end_run(self, keep=False):
        """close the run; with keep=True also return its fields {'rad': ..., 'flux': ...} (host copies)"""
        s = self.scene
        rad = np.zeros(self._shape_rad, dtype=np.float32) if keep and (s.target & TARGET_RADIANCE) else None
        flux = np.zeros(self._shape_flux, dtype=np.float32) if keep and (s.target & TARGET_FLUX) else None
        self._chk(self.lib.mi3d_stats_end_run(self._h, _ptr(rad), _ptr(flux)))
        out = {}
        if rad is not None:
            out['rad'] = rad
        if flux is not None:
            out['flux'] = flux
        return out

    def stats_get(self, which):
        """(mean, std, nrun) over the closed runs; which = TARGET_RADIANCE or TARGET_FLUX"""
        shape = self._shape_rad if which == TARGET_RADIANCE else self._shape_flux
        mean = np.zeros(shape, dtype=np.float32); sdev = np.zeros(shape, dtype=np.float32); n = C.c_int(0)
        self._chk(self.lib.mi3d_stats_get(self._h, int(which), _ptr(mean), _ptr(sdev), C.byref(n)))
        return mean, sdev, int(n.value)

    def counters(self):
        out = np.zeros(NCOUNTER, dtype=np.uint64)
        self._chk(self.lib.mi3d_get_counters(self._h, out.ctypes.data_as(C.POINTER(_u64))))
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))

    def debug_order(self, n, ntile_max=1024):
        """test hook: (photon order of the last launch: n indices sorted by start tile, where each tile's piece ends)"""
        order = np.zeros(n, dtype=np.uint32); tend = np.zeros(ntile_max, dtype=np.uint32)
        self._chk(self.lib.mi3d_debug_order(self._h, int(n), order.ctypes.data_as(C.POINTER(C.c_uint32)), tend.ctypes.data_as(C.POINTER(C.c_uint32)), int(ntile_max)))
        return order, tend

    def philox(self, seed, id0, draw, n):
        out = np.zeros((n, 4), dtype=np.uint32)
        self._chk(self.lib.mi3d_debug_philox(self._h, int(seed), int(id0), int(draw), int(n), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out
