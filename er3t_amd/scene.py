"""
Scene: the solver's complete input for one (run, g) job, as plain arrays.

It is the in-memory equivalent of one MCARaTS input namelist plus its three side files
(reference: er3t/rtm/mca/mca_inp.py:15-384 for the keys, er3t/rtm/mca/mca_atm.py:373-389,
mca_sca.py:82-92 and mca_sfc.py:136-146 for the files).  All multi-dimensional arrays are C-order
numpy arrays whose LAST index is x, i.e. they have exactly the byte image of the reference's
"x fastest, then y, then z" Fortran-order files.

`from_nml` builds a Scene from a namelist dictionary of the shape `mcarats_ng` assembles
(er3t/rtm/mca/mcarats.py:234-414) and the directory holding the side files.
"""

import os
import re
from dataclasses import dataclass, field

import numpy as np

__all__ = ['Scene', 'TARGET_FLUX', 'TARGET_RADIANCE', 'TARGET_HEAT', 'SOLVER_3D', 'SOLVER_P3D', 'SOLVER_IPA']

TARGET_FLUX     = 1
TARGET_RADIANCE = 2
TARGET_HEAT     = 4     # heating rates beside the fluxes (Flx_mhrt = 1, er3t/rtm/mca/mcarats.py:279-283): with TARGET_FLUX

SOLVER_3D  = 0
SOLVER_P3D = 1
SOLVER_IPA = 2


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


@dataclass
class Scene:

    # 1-D background (Atm_zgrd0, Atm_ext1d(1:,ip), Atm_omg1d, Atm_apf1d, Atm_abs1d)
    zgrd : np.ndarray                     # (nz+1,) float64 [m]
    ext1d: np.ndarray                     # (np1d, nz) float32 [1/m]
    omg1d: np.ndarray                     # (np1d, nz)
    apf1d: np.ndarray                     # (np1d, nz)
    abs1d: np.ndarray                     # (nz,)

    # horizontal grid (Atm_nx, Atm_ny, Atm_dx, Atm_dy)
    nx: int = 1
    ny: int = 1
    dx: float = 1.0e4
    dy: float = 1.0e4

    # 3-D region (Atm_nz3, Atm_iz3l, Atm_np3d, Atm_inpfile)
    nz3 : int = 0
    iz3l: int = 1
    abst: np.ndarray = None               # (nz3, ny, nx) or None
    extp: np.ndarray = None               # (np3d, nz3, ny, nx)
    omgp: np.ndarray = None
    apfp: np.ndarray = None

    # tabulated phase functions (Sca_npf, Sca_nangi, Sca_inpfile)
    ang: np.ndarray = None                # (nang,) degrees
    pha: np.ndarray = None                # (npf, nang)

    # surface: uniform (Sfc_mtype, Sfc_param) or 2-D (Sfc_nxb, Sfc_nyb, Sfc_inpfile)
    sfc_mtype: int = 1
    sfc_param: np.ndarray = field(default_factory=lambda: np.array([0.0, 0, 0, 0, 0], dtype=np.float32))
    jsfc: np.ndarray = None               # (nyb, nxb) float32
    psfc: np.ndarray = None               # (5, nyb, nxb) float32

    # source (Src_flx, Src_qmax, Src_the, Src_phi)
    src_flx : float = 1.0
    src_qmax: float = 0.533133
    src_the : float = 150.0
    src_phi : float = 270.0

    # radiance views (Rad_the, Rad_phi, Rad_zloc, Rad_zref, Rad_nxr, Rad_nyr)
    view_the : list = field(default_factory=list)
    view_phi : list = field(default_factory=list)
    view_zloc: list = field(default_factory=list)
    zref: float = 0.0
    nxr : int = 1
    nyr : int = 1
    # Rad_mrkind: 2 radiance averaged over the pixel's column cross-section (satellite); 1 local radiance at a point averaged
    # over the pixel's solid angle (all-sky camera, er3t/rtm/mca/mcarats.py:291-296): view i is then a camera at
    # (cam_xpos Lx, cam_ypos Ly, view_zloc) turned by the Z-Y-Z rotations view_phi, view_the, cam_psi, cone of view cam_qmax,
    # image cam_umax x cam_vmax degrees wide in the polar map (Rad_xpos, Rad_ypos, Rad_psi, Rad_qmax, Rad_umax, Rad_vmax),
    # nearest distance counted cam_apsize metres (Rad_apsize)
    rad_kind : int = 2
    cam_xpos : list = field(default_factory=list)
    cam_ypos : list = field(default_factory=list)
    cam_psi  : list = field(default_factory=list)
    cam_qmax : list = field(default_factory=list)
    cam_umax : list = field(default_factory=list)
    cam_vmax : list = field(default_factory=list)
    cam_apsize: list = field(default_factory=list)
    cam_images: int = -1                  # (-1: 2 where the ray kernel serves the job -- the default route --, else the nearest image alone, with a warning) the domain is cyclic: an event contributes to the periodic images of a camera within this many
                                          # domain lengths of the nearest one, the farther ones by Russian roulette on (r0 / r)^2 (unbiased).  er3t's
                                          # camera looks 89 degrees off its axis (mcarats.py:291-296); on the 12.8 km bench grid the nearest image
                                          # alone is complete to 75 degrees, lacks 43 % of the light at 82-86 and 96 % at 86-89; 1 / 2 / 3 images
                                          # recover 79 / 98 / 100 % of the last ring at 0.49 / 0.29 / 0.19 of the speed (profiles/r04/camera_images.log)

    # job
    target: int = TARGET_FLUX
    solver: int = SOLVER_3D
    wmin  : float = 0.2                   # Pho_wmin: Russian roulette below this weight ...
    wfac  : float = 1.0                   # Pho_wfac: ... survivors continue with this weight
    le_tau1: float = 2.0                  # Russian roulette on marched local-estimate rays beyond this optical depth (unbiased;
                                          # +64 % results per second on the nine-view configuration at 0.5 % more noise per photon); 0 = off

    le_cmin: float = 0.0796               # Russian roulette on the weight of marched local-estimate rays below this value, 1/(4 pi): the
                                          # estimates that look where the phase function is below its isotropic value (unbiased;
                                          # include/mi3d.h: mi3d_set_le_weight_roulette; nine-view configuration: 1.67 x the photons per
                                          # second at 0.2 % more per-pixel noise, profiles/r03/weight_roulette_sweep_mv9.log); 0 = off

    def __post_init__(self):
        self.zgrd  = np.ascontiguousarray(self.zgrd, dtype=np.float64)
        nz = self.zgrd.size - 1
        self.ext1d = _f32(np.atleast_2d(self.ext1d))
        self.omg1d = _f32(np.atleast_2d(self.omg1d))
        self.apf1d = _f32(np.atleast_2d(self.apf1d))
        self.abs1d = _f32(self.abs1d)
        if self.ext1d.shape[1] != nz or self.omg1d.shape != self.ext1d.shape or \
           self.apf1d.shape != self.ext1d.shape or self.abs1d.shape != (nz,):
            raise ValueError('Error [Scene]: 1-D profiles do not match the %d layers of <zgrd>.' % nz)
        if np.any(np.diff(self.zgrd) <= 0.0):
            raise ValueError('Error [Scene]: <zgrd> must be strictly ascending.')
        if self.nz3 > 0:
            shp = (self.nz3, self.ny, self.nx)
            self.extp = _f32(self.extp); self.omgp = _f32(self.omgp); self.apfp = _f32(self.apfp)
            if self.extp.ndim == 3:
                self.extp = self.extp[None]; self.omgp = self.omgp[None]; self.apfp = self.apfp[None]
            if self.extp.shape[1:] != shp or self.omgp.shape != self.extp.shape or self.apfp.shape != self.extp.shape:
                raise ValueError('Error [Scene]: 3-D arrays must be (np3d, nz3, ny, nx) = (*, %d, %d, %d).' % shp)
            if self.abst is not None:
                self.abst = _f32(self.abst)
                if self.abst.shape != shp:
                    raise ValueError('Error [Scene]: <abst> must be (nz3, ny, nx).')
            if self.iz3l < 1 or self.iz3l - 1 + self.nz3 > nz:
                raise ValueError('Error [Scene]: 3-D layers %d..%d do not fit the %d layers of the 1-D grid.' % (self.iz3l, self.iz3l+self.nz3-1, nz))
        if self.pha is not None:
            self.ang = _f32(self.ang)
            self.pha = _f32(np.atleast_2d(self.pha))
            if self.pha.shape[1] != self.ang.size:
                raise ValueError('Error [Scene]: <pha> must be (npf, nang).')
        self.sfc_param = _f32(np.resize(np.asarray(self.sfc_param, dtype=np.float32), 5)) if np.size(self.sfc_param) == 5 \
                         else _f32(np.concatenate([np.ravel(self.sfc_param), np.zeros(5)])[:5])
        if self.jsfc is not None:
            self.jsfc = _f32(self.jsfc)
            self.psfc = _f32(self.psfc)
            if self.psfc.shape != (5,) + self.jsfc.shape:
                raise ValueError('Error [Scene]: <psfc> must be (5, nyb, nxb).')
        if self.rad_kind == 1:
            n = len(self.view_the)
            for name, default in (('cam_xpos', 0.5), ('cam_ypos', 0.5), ('cam_psi', 0.0), ('cam_qmax', 180.0), ('cam_umax', 180.0),
                                  ('cam_vmax', 180.0), ('cam_apsize', 0.0)):
                v = list(np.resize(np.asarray(getattr(self, name) if len(getattr(self, name)) else [default], dtype=np.float64), n))
                setattr(self, name, v)
        elif self.rad_kind != 2:
            raise ValueError('Error [Scene]: <rad_kind=%s> (Rad_mrkind) must be 1 or 2.' % self.rad_kind)

    # convenient sizes
    @property
    def nz(self):
        return self.zgrd.size - 1

    @property
    def np1d(self):
        return self.ext1d.shape[0]

    @property
    def np3d(self):
        return 0 if self.nz3 == 0 else self.extp.shape[0]

    @property
    def npf(self):
        return 0 if self.pha is None else self.pha.shape[0]

    @property
    def nview(self):
        return len(self.view_the)

    @property
    def mu0(self):
        return abs(np.cos(np.deg2rad(self.src_the)))

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_nml(cls, nml, fdir='.', solver=SOLVER_3D):

        """
        Build a Scene from a flat namelist dictionary {key: value} as assembled by `mcarats_ng`
        (reference: er3t/rtm/mca/mcarats.py:234-414) -- or parsed back from an input file -- with
        side-file paths taken relative to <fdir> (mcarats.py:323,352,406).
        """

        def get(key, default=None):
            return nml[key] if key in nml and nml[key] is not None else default

        zgrd = np.asarray(get('Atm_zgrd0'), dtype=np.float64)
        nz   = int(get('Atm_nz', zgrd.size-1))
        zgrd = zgrd[:nz+1]
        np1d = int(get('Atm_np1d', 1))

        def col(base, ip, default):
            # indexed keys look like 'Atm_ext1d(1:, 2)' (mca_atm.py:91-102,135-137)
            for key in nml:
                m = re.match(r'%s\(\s*1:\s*,\s*(\d+)\s*\)' % base, key)
                if m and int(m.group(1)) == ip+1:
                    return np.resize(np.asarray(nml[key], dtype=np.float64), nz)
            if ip == 0 and base in nml and nml[base] is not None:
                return np.resize(np.asarray(nml[base], dtype=np.float64), nz)
            return np.full(nz, default)

        ext1d = np.stack([col('Atm_ext1d', ip, 0.0) for ip in range(np1d)])
        omg1d = np.stack([col('Atm_omg1d', ip, 1.0) for ip in range(np1d)])
        apf1d = np.stack([col('Atm_apf1d', ip, -1.0) for ip in range(np1d)])
        abs1d = col('Atm_abs1d', 0, 0.0)
        for base, arr in (('Atm_fext1d', ext1d),):
            fac = get(base)
            if fac is not None:
                arr *= np.resize(np.asarray(fac, dtype=np.float64), np1d)[:, None]
        abs1d = abs1d * float(get('Atm_fabs1d', 1.0))

        kw = dict(zgrd=zgrd, ext1d=ext1d, omg1d=omg1d, apf1d=apf1d, abs1d=abs1d)

        nx = int(get('Atm_nx', 1)); ny = int(get('Atm_ny', 1))
        kw.update(nx=nx, ny=ny, dx=float(get('Atm_dx', 1.0e4)), dy=float(get('Atm_dy', 1.0e4)))

        nz3 = int(get('Atm_nz3', 0))
        if nz3 > 0:
            np3d = int(get('Atm_np3d', 1))
            fname = os.path.join(fdir, get('Atm_inpfile'))
            nvox = nx*ny*nz3
            raw = np.fromfile(fname, dtype='<f4')
            if raw.size != nvox*(2+3*np3d):
                raise OSError('Error [Scene]: <%s> holds %d values, expected %d.' % (fname, raw.size, nvox*(2+3*np3d)))
            blocks = raw.reshape(2+3*np3d, nz3, ny, nx)
            fext3d = np.resize(np.asarray(get('Atm_fext3d', 1.0), dtype=np.float32), np3d)
            kw.update(nz3=nz3, iz3l=int(get('Atm_iz3l', 1)),
                      abst=blocks[1]*np.float32(get('Atm_fabs3d', 1.0)),
                      extp=blocks[2::3]*fext3d[:, None, None, None], omgp=blocks[3::3], apfp=blocks[4::3])

        npf = int(get('Sca_npf', 0))
        if npf > 0:
            nang  = int(get('Sca_nangi'))
            fname = os.path.join(fdir, get('Sca_inpfile'))
            raw   = np.fromfile(fname, dtype='<f4')
            if raw.size < nang*(1+npf):
                raise OSError('Error [Scene]: <%s> holds %d values, expected %d.' % (fname, raw.size, nang*(1+npf)))
            kw.update(ang=raw[:nang], pha=raw[nang:nang*(1+npf)].reshape(npf, nang))

        if get('Sfc_inpfile') is not None and int(get('Sfc_nxb', 0)) > 0:
            nxb = int(get('Sfc_nxb')); nyb = int(get('Sfc_nyb'))
            raw = np.fromfile(os.path.join(fdir, get('Sfc_inpfile')), dtype='<f4')
            if raw.size != 7*nxb*nyb:
                raise OSError('Error [Scene]: surface file holds %d values, expected %d.' % (raw.size, 7*nxb*nyb))
            blocks = raw.reshape(7, nyb, nxb)
            kw.update(jsfc=blocks[1], psfc=blocks[2:7])
        else:
            param = np.zeros(5, dtype=np.float32)
            if get('Sfc_param') is not None:
                p = np.ravel(np.asarray(get('Sfc_param'), dtype=np.float32))
                param[:p.size] = p[:5]
            for i in range(5):
                key = 'Sfc_param(%d)' % (i+1)
                if key in nml and nml[key] is not None:
                    param[i] = nml[key]
            kw.update(sfc_mtype=int(get('Sfc_mtype', 1)), sfc_param=param)

        kw.update(src_flx=float(get('Src_flx', 1.0)), src_qmax=float(get('Src_qmax', 0.0)),
                  src_the=float(get('Src_the', 120.0)), src_phi=float(get('Src_phi', 0.0)))

        mtarget = int(get('Wld_mtarget', 1))
        if mtarget == 2:
            nrad = int(get('Rad_nrad', 1))
            the  = np.resize(np.asarray(get('Rad_the', 180.0), dtype=np.float64), nrad)
            phi  = np.resize(np.asarray(get('Rad_phi', 0.0), dtype=np.float64), nrad)
            zloc = np.resize(np.asarray(get('Rad_zloc', 0.0), dtype=np.float64), nrad)
            kw.update(target=TARGET_RADIANCE, view_the=list(the), view_phi=list(phi), view_zloc=list(zloc),
                      zref=float(get('Rad_zref', 0.0)), nxr=int(get('Rad_nxr', 1)), nyr=int(get('Rad_nyr', 1)))
            mrkind = int(get('Rad_mrkind', 2))
            if mrkind == 1:
                if int(get('Rad_mpmap', 1)) != 1:
                    raise OSError('Error [Scene]: only the polar pixel map (<Rad_mpmap=1>) is supported for <Rad_mrkind=1>.')
                def per_view(key, default):
                    return list(np.resize(np.asarray(get(key, default), dtype=np.float64), nrad))
                kw.update(rad_kind=1, cam_xpos=per_view('Rad_xpos', 0.5), cam_ypos=per_view('Rad_ypos', 0.5),
                          cam_psi=per_view('Rad_psi', 0.0), cam_qmax=per_view('Rad_qmax', 180.0), cam_umax=per_view('Rad_umax', 180.0),
                          cam_vmax=per_view('Rad_vmax', 180.0), cam_apsize=per_view('Rad_apsize', 0.0))
            elif mrkind != 2:
                raise OSError('Error [Scene]: <Rad_mrkind=%d> is not supported (1: camera, 2: satellite).' % mrkind)
        elif mtarget == 1:
            kw.update(target=TARGET_FLUX | (TARGET_HEAT if int(get('Flx_mhrt', 0) or 0) == 1 else 0))
        else:
            raise OSError('Error [Scene]: <Wld_mtarget=%d> is not supported.' % mtarget)

        kw.update(solver=int(solver), wmin=float(get('Pho_wmin', 0.2)), wfac=float(get('Pho_wfac', 1.0)))

        return cls(**kw)
