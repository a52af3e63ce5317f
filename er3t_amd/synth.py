"""
Synthetic inputs for the BASELINE.json configurations.

The reference's own inputs need data files that are not available offline (AFGL profile, LES
netCDF, correlated-k / Mie databases: er3t/pre/atm/atm_atmmod.py:17, er3t/pre/cld/cld_les.py:16,
er3t/pre/abs/abs_crk.py:27, er3t/pre/pha/pha_mie.py:72).  These generators produce duck-typed
stand-ins with the same attributes (`.lay/.lev/.coef/.data` dictionaries of {'data': ...}) so they
can be fed to the adapters in er3t_amd.rtm.mca exactly like the reference's pre-processing objects,
plus `les_scene(...)` which assembles a ready Scene for tests and bench.py.
"""

import datetime

import numpy as np
from scipy import fft as _fft

from .scene import Scene, TARGET_FLUX, TARGET_RADIANCE, SOLVER_3D

__all__ = ['atm_synth', 'abs_synth', 'cld_synth', 'pha_hg_synth', 'pha_mie_synth', 'sfc_lsrt_synth', 'sfc_dsm_synth', 'les_scene',
           'z_levels_config2', 'z_levels_config4', 'weights_16g', 'rayleigh_tau']


# 16 g-point weights hard-coded by the reference's correlated-k module (er3t/pre/abs/abs_crk.py:693-701)
def weights_16g():
    return np.array([0.1527534276, 0.1491729617, 0.1420961469, 0.1316886544,
                     0.1181945205, 0.1019300893, 0.0832767040, 0.0626720116,
                     0.0424925000, 0.0046269894, 0.0038279891, 0.0030260086,
                     0.0022199750, 0.0014140010, 0.0005330000, 0.0000750000])


def rayleigh_tau(wvl_um, p_lower, p_upper):
    """Bodhaine-type Rayleigh optical thickness between two pressures (reference: er3t/util/util.py:1097-1099)"""
    num = 1.0455996 - 341.29061*wvl_um**(-2.0) - 0.90230850*wvl_um**2.0
    den = 1.0 + 0.0027059889*wvl_um**(-2.0) - 85.968563*wvl_um**2.0
    return 0.00210966*(num/den)*(p_lower-p_upper)/1013.25


def z_levels_config2():
    """50 layers of 40 m (0-2 km) + 18 layers of 1 km (2-20 km), in km"""
    return np.concatenate([np.arange(0, 51)*0.04, np.arange(3, 21)*1.0])


def z_levels_config4():
    """100 layers of 40 m (0-4 km) + 16 layers of 1 km (4-20 km), in km"""
    return np.concatenate([np.arange(0, 101)*0.04, np.arange(5, 21)*1.0])


class atm_synth:

    """
    Stand-in for `er3t.pre.atm.atm_atmmod` (attributes used downstream: er3t/rtm/mca/mca_atm.py:74-90,235-256):
    p(z) = 1013.25 exp(-z/8 km) hPa, T(z) = max(288 - 6.5 z, 216.65) K, air and CO2 number densities (the reference's
    Rayleigh routine reads their ratio: er3t/util/util.py:1030-1077).
    """

    def __init__(self, levels):
        lev = np.asarray(levels, dtype=np.float64)
        lay = 0.5*(lev[1:]+lev[:-1])
        self.lev = {'altitude': {'data': lev, 'units': 'km'},
                    'pressure': {'data': 1013.25*np.exp(-lev/8.0), 'units': 'mb'},
                    'temperature': {'data': np.maximum(288.0-6.5*lev, 216.65), 'units': 'K'}}
        air = 2.55e19*np.exp(-lay/8.0)                     # number density [cm^-3]; co2 at 400 ppm
        self.lay = {'altitude': {'data': lay, 'units': 'km'},
                    'thickness': {'data': lev[1:]-lev[:-1], 'units': 'km'},
                    'pressure': {'data': 1013.25*np.exp(-lay/8.0), 'units': 'mb'},
                    'temperature': {'data': np.maximum(288.0-6.5*lay, 216.65), 'units': 'K'},
                    'air': {'data': air, 'units': 'cm-3'}, 'co2': {'data': 4.0e-4*air, 'units': 'cm-3'}}


class abs_synth:

    """
    Stand-in for `er3t.pre.abs.abs_16g` (contract: er3t/pre/abs/abs_crk.py:622-628; consumers:
    er3t/rtm/mca/mca_atm.py:90, er3t/rtm/mca/mca_out.py:325-327):
    coef['abso_coef'] (nz, Ng) layer absorption optical thickness, 'weight', 'solar', 'slit_func' (nz, Ng).
    """

    def __init__(self, wavelength=650.0, atm_obj=None, Ng=16):
        self.wvl = wavelength
        self.nwl = 1
        self.Ng  = Ng
        self.wvl_info = '%.2f nm (synthetic %d g)' % (wavelength, Ng)
        z  = atm_obj.lay['altitude']['data']
        dz = atm_obj.lay['thickness']['data']
        ig = np.arange(Ng)
        weight = weights_16g() if Ng == 16 else np.repeat(1.0/Ng, Ng)
        self.coef = {
            'wavelength': {'data': wavelength},
            'abso_coef' : {'data': 1.0e-3*((ig[None, :]+1.0)/Ng)*np.exp(-z[:, None]/8.0)*dz[:, None]},
            'weight'    : {'data': weight},
            'solar'     : {'data': np.repeat(1.5, Ng)},
            'slit_func' : {'data': np.ones((z.size, Ng))},
            }


def _fractal_field(shape, rng, slope=-5.0/3.0):
    """Gaussian random field with an isotropic power spectrum whose 1-D spectrum falls as k^slope; unit variance"""
    ndim = len(shape)
    ks = np.meshgrid(*[np.fft.fftfreq(n)*n for n in shape], indexing='ij')
    k = np.sqrt(sum(kk**2 for kk in ks))
    k[(0,)*ndim] = 1.0
    amp = k**((slope-(ndim-1))/2.0)
    amp[(0,)*ndim] = 0.0
    f = _fft.ifftn(_fft.fftn(rng.standard_normal(shape), workers=4)*amp, workers=4).real
    return (f-f.mean())/f.std()


class cld_synth:

    """
    Stand-in for `er3t.pre.cld.cld_les` (attributes: er3t/pre/cld/cld_les.py:27-38; consumers:
    er3t/rtm/mca/mca_atm.py:235-257): a stratocumulus-like field,
    lay['extinction'] (nx, ny, nz) [1/m], lay['temperature'], lay['altitude'/'thickness'] [km], lay['nx','ny','dx','dy'].
    """

    def __init__(self, atm_obj, nx=128, ny=128, nz=50, dx=0.1, dy=0.1, z_base=0.6, z_top=1.4, cot_mean=10.0,
                 cloud_fraction=0.7, sigma_log=0.6, seed=20251003, cer=10.0):
        rng = np.random.default_rng(seed)
        z_lay = atm_obj.lay['altitude']['data'][:nz]
        dz    = atm_obj.lay['thickness']['data'][:nz]

        g2 = _fractal_field((nx, ny), rng)
        thresh = np.quantile(g2, 1.0-cloud_fraction)
        mask = g2 > thresh
        cot = np.exp(sigma_log*(g2-thresh))*mask
        cot *= cot_mean/cot.mean()

        # adiabatic shape: extinction grows as (height above base)^(2/3)
        f = np.clip((z_lay-z_base)/(z_top-z_base), 0.0, None)**(2.0/3.0)
        f[(z_lay < z_base) | (z_lay > z_top)] = 0.0
        f /= (f*dz*1000.0).sum()

        g3 = _fractal_field((nx, ny, nz), rng)
        ext = cot[:, :, None]*f[None, None, :]*np.exp(0.3*g3-0.045)
        tau = (ext*dz[None, None, :]*1000.0).sum(axis=-1)
        scale = np.where(tau > 0.0, cot/np.maximum(tau, 1e-30), 0.0)
        ext *= scale[:, :, None]

        self.lay = {
            'nx': {'data': nx}, 'ny': {'data': ny},
            'dx': {'data': dx, 'units': 'km'}, 'dy': {'data': dy, 'units': 'km'},
            'altitude'   : {'data': z_lay.copy(), 'units': 'km'},
            'thickness'  : {'data': dz.copy(), 'units': 'km'},
            'extinction' : {'data': ext.astype(np.float64), 'units': '/m'},
            'temperature': {'data': np.broadcast_to(atm_obj.lay['temperature']['data'][:nz], (nx, ny, nz)).copy(), 'units': 'K'},
            'cot'        : {'data': cot},
            'cer'        : {'data': np.where(ext > 0.0, cer, 0.0)},
            }
        self.lev = {'altitude': {'data': atm_obj.lev['altitude']['data'][:nz+1].copy(), 'units': 'km'}}


class cld_hom_synth:

    """
    Stand-in for `er3t.pre.cld.cld_gen_hom` (er3t/pre/cld/cld_gen.py:485-700; consumer: `func_ref_vs_cot_multi_pixel`,
    er3t/rtm/mca/util.py:330-333): a horizontally homogeneous cloud of optical thickness cot0 on Nx x Ny columns, its extinction
    cot0 / (Nz dz) the same in every layer whose centre is listed in `altitude` [km] (equidistant), temperatures interpolated from
    the atmosphere object.  Same attributes as the reference's object: lay['nx','ny','dx','dy','altitude','thickness',
    'temperature','extinction','cot','cer'], lev['altitude','cot_2d'].
    """

    def __init__(self, cot0=10.0, cer0=10.0, altitude=None, atm_obj=None, Nx=2, Ny=2, dx=0.1, dy=0.1, fname=None, overwrite=True,
                 verbose=False):
        if atm_obj is None:
            raise OSError('Error [cld_hom_synth]: Please provide an \'atm\' object for <atm_obj>.')
        altitude = np.arange(1.5, 2.5, 0.5) if altitude is None else np.asarray(altitude, dtype=np.float64)
        nz = altitude.size
        dz = float(altitude[1]-altitude[0]) if nz > 1 else float(atm_obj.lay['thickness']['data'][np.argmin(np.abs(atm_obj.lay['altitude']['data']-altitude[0]))])
        lev = np.append(altitude-0.5*dz, altitude[-1]+0.5*dz)
        t1d = np.interp(altitude, atm_obj.lay['altitude']['data'], atm_obj.lay['temperature']['data'])
        full = lambda v: np.full((Nx, Ny, nz), v, dtype=np.float64)
        self.lay = {
            'nx': {'data': Nx}, 'ny': {'data': Ny}, 'nz': {'data': nz},
            'dx': {'data': dx, 'units': 'km'}, 'dy': {'data': dy, 'units': 'km'}, 'dz': {'data': dz, 'units': 'km'},
            'altitude'   : {'data': altitude.copy(), 'units': 'km'},
            'thickness'  : {'data': np.diff(lev), 'units': 'km'},
            'temperature': {'data': np.broadcast_to(t1d, (Nx, Ny, nz)).copy(), 'units': 'K'},
            'extinction' : {'data': full(cot0/nz/dz/1000.0), 'units': '/m'},
            'cot'        : {'data': full(cot0/nz)},
            'cer'        : {'data': full(cer0), 'units': 'micron'},
            }
        self.lev = {'altitude': {'data': lev, 'units': 'km'}, 'cot_2d': {'data': np.full((Nx, Ny), float(cot0))}}


class pha_hg_synth:

    """
    Stand-in for `er3t.pre.pha.pha_hg` (er3t/pre/pha/pha_hg.py:31-95): Henyey-Greenstein tables on a regular
    angle grid; data['ang'] (nang,), data['pha'] (nang, n_g), data['asy'] (n_g,), data['id'] = 'HG'.
    """

    ID = 'HG (synthetic)'

    def __init__(self, asy=(0.80, 0.85, 0.90), nang=1801):
        ang = np.linspace(0.0, 180.0, nang)
        mu  = np.cos(np.deg2rad(ang))
        asy = np.asarray(asy, dtype=np.float64)
        pha = np.stack([(1.0-g*g)/(1.0+g*g-2.0*g*mu)**1.5 for g in asy], axis=1)
        self.data = {'id': {'data': 'HG'}, 'ang': {'data': ang}, 'pha': {'data': pha}, 'asy': {'data': asy},
                     'ssa': {'data': np.ones_like(asy)}}


class pha_mie_synth:

    """
    Stand-in for `er3t.pre.pha.pha_mie_wc` (er3t/pre/pha/pha_mie.py:72-113; the data base behind it, wc.sol.mie.cdf, is not part of the
    rtm.mca path): phase functions of water-cloud droplets on the reference's default grid of 498 scattering angles (0.01 degree steps
    in the diffraction peak, 1 degree beyond 15 degrees), one per effective radius -- data['ang'] (nang,), data['pha'] (nang, nref),
    data['ref'] (nref,) [micron], data['ssa'], data['asy'], data['id'] = 'Mie'.  Not a Mie calculation: a superposition with the features
    that matter to the sampling code -- a diffraction peak whose width goes as wavelength / radius and holds close to half of the
    scattered light, the broad refraction lobe, the rainbow near 140 degrees, the glory -- normalised so that the integral over the
    sphere is 4 pi; the asymmetry parameter is integrated from the table.
    """

    ID = 'Mie (synthetic)'

    def __init__(self, wavelength=650.0, ref=(6.0, 9.0, 12.0, 15.0)):
        ang = np.concatenate((np.arange(0.0, 2.0, 0.01), np.arange(2.0, 5.0, 0.05), np.arange(5.0, 10.0, 0.1), np.arange(10.0, 15.0, 0.5),
                              np.arange(15.0, 176.0, 1.0), np.arange(176.0, 180.1, 0.25)))
        th = np.deg2rad(ang); mu = np.cos(th)
        ref = np.asarray(ref, dtype=np.float64)
        hg = lambda g: (1.0-g*g)/(1.0+g*g-2.0*g*mu)**1.5
        pha, asy = [], []
        for r in ref:
            x = 2.0*np.pi*r/(wavelength*1.0e-3)                    # size parameter
            wd = 2.0/x                                             # angular width of the diffraction peak [rad]
            diff = np.exp(-0.5*(th/wd)**2)
            diff *= 2.0/np.trapz(diff*np.sin(th), th)              # (each part normalised to an integral of 4 pi over the sphere)
            rainbow = np.exp(-0.5*((ang-(138.0+12.0/np.sqrt(r)))/2.5)**2)
            rainbow *= 2.0/np.trapz(rainbow*np.sin(th), th)
            glory = np.exp(-(180.0-ang)/(60.0/np.sqrt(x)))
            glory *= 2.0/np.trapz(glory*np.sin(th), th)
            p = 0.48*diff + 0.47*hg(0.78+0.003*r) + 0.025*hg(-0.2) + 0.018*rainbow + 0.007*glory
            p *= 2.0/np.trapz(p*np.sin(th), th)
            pha.append(p); asy.append(0.5*np.trapz(p*mu*np.sin(th), th))
        self.data = {'id': {'data': 'Mie'}, 'ang': {'data': ang}, 'pha': {'data': np.stack(pha, axis=1)}, 'ref': {'data': ref},
                     'asy': {'data': np.array(asy)}, 'ssa': {'data': np.full(ref.size, 0.999995)}, 'wvl0': {'data': wavelength}}


class sfc_lsrt_synth:

    """
    Stand-in for `er3t.pre.sfc.sfc_2d_gen` with an LSRT dictionary (er3t/pre/sfc/sfc_gen.py:100-155; consumer
    er3t/rtm/mca/mca_sfc.py:104-117): data['sfc']['data'] (nx, ny, 3) = (fiso, fgeo, fvol).
    """

    def __init__(self, nx, ny, dx=0.1, dy=0.1, seed=7, fiso=0.25, fgeo=0.03, fvol=0.12):
        rng = np.random.default_rng(seed)
        par = np.zeros((nx, ny, 3))
        for i, m in enumerate((fiso, fgeo, fvol)):
            par[:, :, i] = np.clip(m*(1.0+0.2*_fractal_field((nx, ny), rng)), 0.0, None)
        self.Nx = nx; self.Ny = ny
        self.data = {'nx': {'data': nx}, 'ny': {'data': ny}, 'dx': {'data': dx}, 'dy': {'data': dy},
                     'sfc': {'data': par, 'name': 'BRDF-LSRT'}}


class sfc_dsm_synth:

    """
    Stand-in for `er3t.pre.sfc.sfc_2d_gen` with a Cox-Munk dictionary (er3t/pre/sfc/sfc_gen.py:131-145, parameters from
    er3t/pre/sfc/util.py:109-156; consumer er3t/rtm/mca/mca_sfc.py:119-128): data['sfc']['data'] (nx, ny, 5) =
    (diffuse albedo, diffuse fraction, Re m, Im m, slope variance) for a 10-m wind speed field around <u10> m/s.
    """

    def __init__(self, nx, ny, dx=0.1, dy=0.1, seed=9, u10=6.0, whitecaps=True):
        rng = np.random.default_rng(seed)
        u = np.clip(u10*(1.0+0.2*_fractal_field((nx, ny), rng)), 0.5, None)
        par = np.zeros((nx, ny, 5), dtype=np.float32)
        par[:, :, 0] = 0.22 if whitecaps else 0.0                 # effective reflectance of whitecaps in the visible (Koepke 1984)
        par[:, :, 1] = 2.95e-06*u**3.52 if whitecaps else 0.0     # whitecap coverage
        par[:, :, 2] = 1.34
        par[:, :, 3] = 1.0e-8
        par[:, :, 4] = 0.00512*u + 0.003                          # Cox and Munk 1954
        self.Nx = nx; self.Ny = ny
        self.data = {'nx': {'data': nx}, 'ny': {'data': ny}, 'dx': {'data': dx}, 'dy': {'data': dy},
                     'sfc': {'data': par, 'name': 'Cox-Munk'}}


# ----------------------------------------------------------------------------------------------
def les_scene(nx=128, ny=128, nz3=50, levels=None, wavelength=650.0, ig=7, Ng=16, sza=30.0, saa=45.0,
              vza=(0.0,), vaa=(0.0,), sensor_altitude=705000.0, surface_albedo=0.03, target='radiance',
              z_base=0.6, z_top=1.4, cot_mean=10.0, seed=20251003, aerosol=False, lsrt=False, solver=SOLVER_3D,
              asy=0.85, mie=False):

    """
    Assemble the Scene of BASELINE.json config 2 (defaults), 3 (aerosol=True), 4 (nx=ny=480, nz3=100,
    levels=z_levels_config4(), z_top=1.6, seed=20251004) or 5 (config 4 + nine vza + lsrt=True) the same
    way the adapters do (er3t/rtm/mca/mca_atm.py:68-102,231-337, mcarats.py:285-307,374-399), without
    touching the file system.  The index of the lowest 3-D layer reproduces the reference's value
    (`lay_index[0]+2`, mca_atm.py:242,330).  mie=True: the cloud scatters by TABULATED phase functions -- the Mie branch of
    mca_atm.py:275-303 with the table index the reference has commented out (`f_interp_ind(cer)`: a real 1-based index, its fraction
    mixing neighbouring tables) instead of the asymmetry parameter, as func_ref_vs_cot selects its table (rtm/mca/util.py:153); the
    droplets' effective radius grows with height above the cloud base as in an adiabatic cloud (6 to 14 micron), four tables of 498
    angles (pha_mie_synth).
    """

    if levels is None:
        levels = z_levels_config2()
    atm = atm_synth(levels)
    ab  = abs_synth(wavelength, atm, Ng=Ng)
    cld = cld_synth(atm, nx=nx, ny=ny, nz=nz3, z_base=z_base, z_top=z_top, cot_mean=cot_mean, seed=seed)

    dz_m  = atm.lay['thickness']['data']*1000.0
    p_lev = atm.lev['pressure']['data']
    ext1d = rayleigh_tau(wavelength*0.001, p_lev[:-1], p_lev[1:])/dz_m
    abs1d = ab.coef['abso_coef']['data'][:, ig]/dz_m
    nz = dz_m.size

    extp = np.transpose(cld.lay['extinction']['data'], (2, 1, 0))[None].astype(np.float32)
    omgp = np.ones_like(extp)
    apfp = np.full_like(extp, asy)
    ang = pha = None
    if mie:
        pm = pha_mie_synth(wavelength)
        z_lay = atm.lay['altitude']['data'][:nz3]
        cer = 6.0 + 8.0*np.clip((z_lay-z_base)/(z_top-z_base), 0.0, 1.0)**(1.0/3.0)
        ind = np.interp(cer, pm.data['ref']['data'], np.arange(pm.data['ref']['data'].size) + 1.0)
        apfp = np.where(extp > 0.0, ind[None, :, None, None], -1.0).astype(np.float32)
        omgp = np.where(extp > 0.0, np.float32(pm.data['ssa']['data'][0]), np.float32(1.0)).astype(np.float32)
        ang = pm.data['ang']['data'].astype(np.float32); pha = pm.data['pha']['data'].T.astype(np.float32)
    if aerosol:
        # (reference scenario: examples/00_er3t_mca.py:763-772)
        e2 = np.zeros_like(extp[0]); e2[0] = 0.00012; e2[1] = 0.00008
        extp = np.concatenate([extp, e2[None]]); omgp = np.concatenate([omgp, np.full_like(e2, 0.85)[None]])
        apfp = np.concatenate([apfp, np.full_like(e2, 0.6)[None]])

    kw = dict(zgrd=atm.lev['altitude']['data']*1000.0, ext1d=ext1d[None], omg1d=np.ones((1, nz)),
              apf1d=-np.ones((1, nz)), abs1d=abs1d,
              nx=nx, ny=ny, dx=cld.lay['dx']['data']*1000.0, dy=cld.lay['dy']['data']*1000.0,
              nz3=nz3, iz3l=2, abst=np.zeros_like(extp[0]), extp=extp, omgp=omgp, apfp=apfp,
              src_flx=1.0, src_qmax=0.533133, src_the=180.0-sza, src_phi=(270.0-saa) % 360.0, solver=solver, ang=ang, pha=pha)

    if lsrt:
        sfc = sfc_lsrt_synth(nx, ny)
        psfc = np.zeros((5, ny, nx), dtype=np.float32)
        psfc[:3] = np.transpose(sfc.data['sfc']['data'], (2, 1, 0))
        kw.update(jsfc=np.full((ny, nx), 4.0, dtype=np.float32), psfc=psfc)
    else:
        kw.update(sfc_mtype=1, sfc_param=[surface_albedo, 0, 0, 0, 0])

    if target == 'radiance':
        vza = np.atleast_1d(vza).astype(np.float64); vaa = np.resize(np.atleast_1d(vaa).astype(np.float64), vza.size)
        kw.update(target=TARGET_RADIANCE, view_the=list(180.0-vza), view_phi=list((270.0-vaa) % 360.0),
                  view_zloc=[sensor_altitude]*vza.size, nxr=nx, nyr=ny)
    else:
        kw.update(target=TARGET_FLUX)

    return Scene(**kw)
