"""
Worked examples of the drop-in layer on an MI355X, in the spirit of the reference's examples/00_er3t_mca.py (flux of a
clear sky, flux and radiance of a 3-D cloud field), with the synthetic atmosphere / absorption / cloud objects of
er3t_amd.synth standing in for er3t.pre.* (whose data bases are not part of this repository; er3t's own objects can be
passed instead, attribute for attribute).

    python examples/00_amd_mca.py [clear_sky_flux | cloud_flux | cloud_radiance | cloud_radiance_fused | cloud_radiance_multi_angle | cloud_heating_rate] [fdir]
"""

import datetime
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import er3t_amd.rtm.mca as mca                                  # noqa: E402   (same names as er3t.rtm.mca)
from er3t_amd import synth                                      # noqa: E402

DATE = datetime.datetime(2017, 8, 13)
WAVELENGTH = 650.0


def _atmosphere(levels):
    atm = synth.atm_synth(levels)
    ab = synth.abs_synth(WAVELENGTH, atm, Ng=16)               # 16 g of a correlated-k band
    return atm, ab


def clear_sky_flux(fdir):
    """1-D clear sky, 16 g x 3 runs, 1e5 photons per run (BASELINE config 1)"""
    atm, ab = _atmosphere(np.linspace(0.0, 20.0, 21))
    atm1d = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
    sim = mca.mcarats_ng(atm_1ds=[atm1d], atm_3ds=[], Ng=ab.Ng, weights=ab.coef['weight']['data'], target='flux', surface_albedo=0.03,
                         solar_zenith_angle=30.0, solar_azimuth_angle=0.0, fdir=fdir, Nrun=3, photons=1e5, solver='3D', date=DATE)
    out = mca.mca_out_ng(fname=os.path.join(fdir, 'flux.npz'), mca_obj=sim, abs_obj=ab, mode='mean', squeeze=True, overwrite=True)
    z = atm.lev['altitude']['data']
    for k in (0, len(z)//2, len(z)-1):
        print('z = %5.1f km: down %.4f (direct %.4f), up %.4f  +- %.4f W/m^2/nm' % (
            z[k], out.data['f_down']['data'][k], out.data['f_down_direct']['data'][k], out.data['f_up']['data'][k], out.data['f_up_std']['data'][k]))
    return out


def _cloud(fdir):
    atm, ab = _atmosphere(synth.z_levels_config2())
    cld = synth.cld_synth(atm)                                   # 128 x 128 x 50 stratocumulus-like field
    atm1d = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
    atm3d = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, fname=os.path.join(fdir, 'mca_atm_3d.bin'), overwrite=True)
    return ab, atm1d, atm3d


def cloud_flux(fdir):
    """3-D cloud field: fluxes on every level of every column, 3D against IPA at the surface"""
    ab, atm1d, atm3d = _cloud(fdir)
    res = {}
    for solver in ('3D', 'IPA'):
        sim = mca.mcarats_ng(atm_1ds=[atm1d], atm_3ds=[atm3d], Ng=ab.Ng, weights=ab.coef['weight']['data'], target='flux', surface_albedo=0.03,
                             solar_zenith_angle=30.0, solar_azimuth_angle=45.0, fdir=os.path.join(fdir, solver.lower()), Nrun=3, photons=1e8,
                             solver=solver, date=DATE, abs_obj=ab, keep_files=False)        # g-sum and run statistics on the GPU
        res[solver] = mca.mca_out_ng(mca_obj=sim, abs_obj=ab, mode='mean', squeeze=True).data
    for solver, d in res.items():
        f = d['f_down']['data'][:, :, 0]
        print('%-3s: surface downward flux mean %.4f, min %.4f, max %.4f W/m^2/nm' % (solver, f.mean(), f.min(), f.max()))
    return res


def cloud_radiance(fdir, fused=False):
    """3-D cloud field: nadir radiance image of a satellite sensor"""
    ab, atm1d, atm3d = _cloud(fdir)
    extra = dict(abs_obj=ab, keep_files=False) if fused else {}
    t0 = time.time()
    sim = mca.mcarats_ng(atm_1ds=[atm1d], atm_3ds=[atm3d], Ng=ab.Ng, weights=ab.coef['weight']['data'], target='radiance', surface_albedo=0.03,
                         solar_zenith_angle=30.0, solar_azimuth_angle=45.0, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0,
                         fdir=os.path.join(fdir, 'rad_fused' if fused else 'rad'), Nrun=3, photons=1e8, solver='3D', date=DATE, **extra)
    out = mca.mca_out_ng(mca_obj=sim, abs_obj=ab, mode='mean', squeeze=True)
    rad, std = out.data['rad']['data'], out.data['rad_std']['data']
    print('radiance image %s: mean %.5f, mean std over runs %.5f W/m^2/nm/sr (%.2f s for %d jobs, %.3g photons)' % (
        rad.shape, rad.mean(), std.mean(), time.time()-t0, sim.Nrun*sim.Ng, sim.photons.sum()))
    return out


def cloud_radiance_multi_angle(fdir):
    """3-D cloud field seen under nine view zenith angles (a MISR-like instrument) in ONE simulation: sequences of sensor angles (not in the
    reference, which runs a simulation per view); the views are the third axis of the radiance"""
    ab, atm1d, atm3d = _cloud(fdir)
    vza = [0.0, 26.1, 26.1, 45.6, 45.6, 60.0, 60.0, 70.5, 70.5]
    vaa = [0.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0]
    t0 = time.time()
    sim = mca.mcarats_ng(atm_1ds=[atm1d], atm_3ds=[atm3d], Ng=ab.Ng, weights=ab.coef['weight']['data'], target='radiance', surface_albedo=0.03,
                         solar_zenith_angle=30.0, solar_azimuth_angle=45.0, sensor_zenith_angle=vza, sensor_azimuth_angle=vaa,
                         fdir=os.path.join(fdir, 'rad9'), Nrun=3, photons=1e8, solver='3D', date=DATE)
    out = mca.mca_out_ng(mca_obj=sim, abs_obj=ab, mode='mean', squeeze=True)
    rad = out.data['rad']['data']                                # (Nx, Ny, 9)
    print('radiance images %s (%.2f s for %d jobs, %.3g photons)' % (rad.shape, time.time()-t0, sim.Nrun*sim.Ng, sim.photons.sum()))
    for i in range(len(vza)):
        print('  view zenith %5.1f azimuth %5.1f: mean %.5f W/m^2/nm/sr' % (vza[i], vaa[i], rad[:, :, i].mean()))
    return out


def cloud_heating_rate(fdir):
    """3-D cloud field: target='heating rate' (er3t/rtm/mca/mcarats.py:279-283) -- the fluxes and, per cell, the absorbed power"""
    ab, atm1d, atm3d = _cloud(fdir)
    sim = mca.mcarats_ng(atm_1ds=[atm1d], atm_3ds=[atm3d], Ng=ab.Ng, weights=ab.coef['weight']['data'], target='heating rate',
                         surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=45.0, fdir=os.path.join(fdir, 'hr'), Nrun=3,
                         photons=1e8, solver='3D', date=DATE)
    out = mca.mca_out_ng(mca_obj=sim, abs_obj=ab, mode='mean', squeeze=True)
    hr = out.data['heating_rate']['data']                        # (Nx, Ny, Nz layers), W/m^3/nm
    z = 0.5*(synth.z_levels_config2()[1:]+synth.z_levels_config2()[:-1])
    for k in (0, 20, 30, 49, 60):
        print('z = %5.2f km: absorbed %.3e W/m^3/nm (domain mean), %.3e (max column)' % (z[k], hr[:, :, k].mean(), hr[:, :, k].max()))
    net = lambda lev: out.data['f_down']['data'][:, :, lev].mean()-out.data['f_up']['data'][:, :, lev].mean()
    dz = np.diff(synth.z_levels_config2())*1000.0
    print('absorbed in the atmosphere %.5f W/m^2/nm; net flux at the top - at the surface %.5f' % ((hr.mean(axis=(0, 1))*dz).sum(), net(-1)-net(0)))
    return out


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'clear_sky_flux'
    fdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join('tmp-data', '00_amd_mca', what)
    os.makedirs(fdir, exist_ok=True)
    {'clear_sky_flux': clear_sky_flux, 'cloud_flux': cloud_flux, 'cloud_radiance': cloud_radiance,
     'cloud_radiance_fused': lambda d: cloud_radiance(d, fused=True), 'cloud_radiance_multi_angle': cloud_radiance_multi_angle,
     'cloud_heating_rate': cloud_heating_rate}[what](fdir)
