"""
`func_ref_vs_cot`: reflectance as a function of cloud optical thickness from 1-D (plane-parallel) radiance runs, the
reference's validation harness and retrieval look-up (er3t/rtm/mca/util.py:19-213); `func_ref_vs_cot_multi_pixel`: the same
curve from a homogeneous cloud on Nx x Ny columns under the independent-column solver, one `mcarats_ng` run per optical
thickness (er3t/rtm/mca/util.py:218-422).

The reference builds its atmosphere, absorption and Mie phase-function objects from data bases that are not part of the
rtm.mca path (`er3t.pre.*`); here they are passed in (`atm0`, `abs0`, `pha0`) -- er3t's own objects or the synthetic
stand-ins of er3t_amd.synth.  Everything downstream is the same chain the reference runs per optical thickness:
mca_sca -> mca_atm_1d.add_mca_1d_atm(ext, ssa, table index) -> mcarats_ng(target='radiance', Nrun=3) -> mca_out_ng ->
reflectance = pi * radiance / (toa * mu0)   (util.py:101-102).
"""

import os
import shutil

import numpy as np
from scipy.interpolate import interp1d

import er3t_amd.common
from er3t_amd.util import cal_r_twostream
from er3t_amd.rtm.mca.mca_atm import mca_atm_1d, mca_atm_3d
from er3t_amd.rtm.mca.mca_sca import mca_sca
from er3t_amd.rtm.mca.mcarats import mcarats_ng
from er3t_amd.rtm.mca.mca_out import mca_out_ng

__all__ = ['func_ref_vs_cot', 'func_ref_vs_cot_multi_pixel']


class func_ref_vs_cot:

    """
    Input:
        cot: array of cloud optical thicknesses
        cer0=: cloud effective radius [micron] (picks the phase-function table nearest to it)
        atm0=, abs0=, pha0=: atmosphere / absorption / phase-function objects (pha0 may be None: Henyey-Greenstein g=0.85)
        fdir=, date=, wavelength=, surface_albedo=, solar_*/sensor_* angles, sensor_altitude=,
        cloud_top_height=2.0 [km], cloud_geometrical_thickness=1.0 [km], solver='3d', Nphoton=, Ncpu=, output_tag=, overwrite=

    Output:
        self.rad, self.rad_std, self.ref, self.ref_std (one value per cot), self.ref_2s (two-stream), self.toa0
        get_cot_from_ref(ref), get_ref_from_cot(cot)
    """

    def __init__(self, cot, cer0=10.0, fdir=er3t_amd.common.params['fdir_tmp'], date=er3t_amd.common.params['date'],
                 wavelength=er3t_amd.common.params['wavelength'], surface_albedo=er3t_amd.common.params['surface_albedo'],
                 solar_zenith_angle=er3t_amd.common.params['solar_zenith_angle'],
                 solar_azimuth_angle=er3t_amd.common.params['solar_azimuth_angle'],
                 sensor_zenith_angle=er3t_amd.common.params['sensor_zenith_angle'],
                 sensor_azimuth_angle=er3t_amd.common.params['sensor_azimuth_angle'],
                 sensor_altitude=er3t_amd.common.params['sensor_altitude'],
                 cloud_top_height=2.0, cloud_geometrical_thickness=1.0, solver='3d',
                 Nphoton=er3t_amd.common.params['Nphoton'], atm0=None, abs0=None, pha0=None,
                 Ncpu=er3t_amd.common.params['Ncpu'], output_tag=er3t_amd.common.params['output_tag'],
                 overwrite=er3t_amd.common.params['overwrite'], quiet=True):

        if atm0 is None or abs0 is None:
            raise OSError('Error [func_ref_vs_cot]: Please provide <atm0> and <abs0> (the data bases behind er3t.pre are not part of this package).')

        self.cot  = np.atleast_1d(np.asarray(cot, dtype=np.float64))
        self.cer0 = cer0
        self.wvl0 = wavelength
        self.sza0 = solar_zenith_angle
        self.saa0 = solar_azimuth_angle
        self.vza0 = sensor_zenith_angle
        self.vaa0 = sensor_azimuth_angle
        self.alt0 = sensor_altitude
        self.cth0 = cloud_top_height
        self.cbh0 = cloud_top_height-cloud_geometrical_thickness
        self.alb0 = surface_albedo
        self.fdir = fdir
        self.output_tag = output_tag
        self.photon0 = Nphoton
        self.solver0 = solver
        self.cpu0 = Ncpu
        self.date0 = date
        self.atm0 = atm0
        self.abs0 = abs0
        self.pha0 = pha0
        self.quiet = quiet

        self.mu0 = np.cos(np.deg2rad(self.sza0))
        self.ref_2s = cal_r_twostream(self.cot, a=self.alb0, mu=self.mu0)

        if not overwrite:
            try:
                self.load_all()
            except (OSError, KeyError):
                self.run_all()
                self.load_all()
        else:
            self.run_all()
            self.load_all()

    def _fname(self, cot0):
        return '%s/%s_cot-%05.1f_cer-%04.1f.npz' % (self.fdir, self.output_tag, cot0, self.cer0)

    def load_all(self):
        rad, rad_std, toa0 = [], [], None
        for cot0 in self.cot:
            out = mca_out_ng(fname=self._fname(cot0), mode='mean', quiet=True)
            rad.append(np.mean(out.data['rad']['data']))
            rad_std.append(np.mean(out.data['rad_std']['data']))
            toa0 = out.data['toa']['data']
        self.rad = np.array(rad); self.rad_std = np.array(rad_std)
        self.toa0 = toa0
        self.ref     = np.pi*self.rad/(toa0*self.mu0)
        self.ref_std = np.pi*self.rad_std/(toa0*self.mu0)

    def run_all(self):
        shutil.rmtree(self.fdir, ignore_errors=True)
        os.makedirs(self.fdir)
        for cot0 in self.cot:
            self.run_one(cot0, self.cer0, cbh0=self.cbh0, cth0=self.cth0)

    def run_one(self, cot0, cer0, cbh0=1.0, cth0=2.0):

        name_tag = 'cot-%05.1f_cer-%04.1f' % (cot0, cer0)
        ext0 = cot0/(cth0-cbh0)/1000.0

        if self.pha0 is not None:
            sca0 = mca_sca(pha_obj=self.pha0, fname='%s/mca_sca-%06.1fnm.bin' % (self.fdir, self.wvl0), overwrite=True, quiet=True)
            # nearest table: by effective radius for Mie sets, by asymmetry parameter 0.85 for HG sets
            if 'ref' in self.pha0.data:
                iref = int(np.argmin(np.abs(self.pha0.data['ref']['data']-cer0)))
            else:
                iref = int(np.argmin(np.abs(self.pha0.data['asy']['data']-0.85)))
            ssa0 = float(np.ravel(self.pha0.data['ssa']['data'])[iref])
            apf0 = iref + 1
        else:
            sca0, ssa0, apf0 = None, 1.0, 0.85

        atm1d0 = mca_atm_1d(atm_obj=self.atm0, abs_obj=self.abs0)
        atm1d0.add_mca_1d_atm(ext1d=ext0, omg1d=ssa0, apf1d=apf0, z_bottom=cbh0, z_top=cth0)

        mca0 = mcarats_ng(date=self.date0, atm_1ds=[atm1d0], atm_3ds=[], sca=sca0, target='radiance',
                          surface_albedo=self.alb0, solar_zenith_angle=self.sza0, solar_azimuth_angle=self.saa0,
                          sensor_zenith_angle=self.vza0, sensor_azimuth_angle=self.vaa0, sensor_altitude=self.alt0,
                          fdir='%s/%s_%s/rad' % (self.fdir, self.output_tag, name_tag), Nrun=3, Ng=self.abs0.Ng,
                          weights=self.abs0.coef['weight']['data'], photons=self.photon0, solver=self.solver0,
                          Ncpu=self.cpu0, mp_mode='py', overwrite=True, quiet=self.quiet)
        mca_out_ng(fname=self._fname(cot0), mca_obj=mca0, abs_obj=self.abs0, mode='mean', squeeze=True, quiet=True, overwrite=True)

    def get_cot_from_ref(self, ref, method='cubic', mode='rt'):
        x = self.ref_2s if mode == '2s' else self.ref
        return interp1d(x, self.cot, kind=method, bounds_error=False, fill_value='extrapolate')(ref)

    def get_ref_from_cot(self, cot, method='cubic', mode='rt'):
        y = self.ref_2s if mode == '2s' else self.ref
        return interp1d(self.cot, y, kind=method, bounds_error=False)(cot)



class func_ref_vs_cot_multi_pixel:

    """
    The reflectance-vs-optical-thickness curve from 3-D job files: a horizontally homogeneous cloud on Nx x Ny columns, one
    `mcarats_ng(target='radiance', Nrun=3)` run per optical thickness under `solver` (default 'ipa': every column on its own, so
    the Nx x Ny pixels are Nx x Ny samples of one plane-parallel problem), `rad` = mean over the pixels
    (er3t/rtm/mca/util.py:218-422; same keyword arguments, same attributes).

    Input:
        cot: array of cloud optical thicknesses
        cer0=, fdir=, date=, wavelength=, surface_albedo=, solar_*/sensor_* angles, sensor_altitude=, Nphoton=,
        cloud_top_height=2.0 [km], cloud_geometrical_thickness=1.0 [km], solver='ipa', Nx=2, Ny=2, dx=0.1, dy=0.1 [km],
        Ncpu=, atm0=, output_tag=, overwrite=
        abs0=, pha0=: absorption / phase-function objects (the reference builds them from data bases outside rtm.mca; pha0 None:
                      Henyey-Greenstein g = 0.85 without a table)
        cld_gen=: callable(cot0=, cer0=, altitude=, atm_obj=, Nx=, Ny=, dx=, dy=) -> cloud object, default
                  er3t_amd.synth.cld_hom_synth (the reference calls er3t.pre.cld.cld_gen_hom, util.py:330-333)

    Output:
        self.rad, self.rad_std, self.ref, self.ref_std (one value per cot), self.ref_2s, self.toa0,
        self.rad_pixels[icot] (Nx, Ny): the pixels behind each mean
        get_cot_from_ref(ref), get_ref_from_cot(cot)
    """

    def __init__(self, cot, cer0=10.0, fdir=er3t_amd.common.params['fdir_tmp'], date=er3t_amd.common.params['date'],
                 wavelength=er3t_amd.common.params['wavelength'], surface_albedo=er3t_amd.common.params['surface_albedo'],
                 solar_zenith_angle=er3t_amd.common.params['solar_zenith_angle'],
                 solar_azimuth_angle=er3t_amd.common.params['solar_azimuth_angle'],
                 sensor_zenith_angle=er3t_amd.common.params['sensor_zenith_angle'],
                 sensor_azimuth_angle=er3t_amd.common.params['sensor_azimuth_angle'],
                 sensor_altitude=er3t_amd.common.params['sensor_altitude'],
                 Nphoton=er3t_amd.common.params['Nphoton'], cloud_top_height=2.0, cloud_geometrical_thickness=1.0, solver='ipa',
                 Nx=2, Ny=2, dx=0.1, dy=0.1, Ncpu=er3t_amd.common.params['Ncpu'], atm0=None, abs0=None, pha0=None, cld_gen=None,
                 output_tag=er3t_amd.common.params['output_tag'], overwrite=er3t_amd.common.params['overwrite'], quiet=True):

        if atm0 is None or abs0 is None:
            raise OSError('Error [func_ref_vs_cot_multi_pixel]: Please provide <atm0> and <abs0> (the data bases behind er3t.pre are not part of this package).')
        if cld_gen is None:
            from er3t_amd.synth import cld_hom_synth as cld_gen

        self.cot  = np.atleast_1d(np.asarray(cot, dtype=np.float64))
        self.cer0 = cer0
        self.wvl0 = wavelength
        self.sza0 = solar_zenith_angle
        self.saa0 = solar_azimuth_angle
        self.vza0 = sensor_zenith_angle
        self.vaa0 = sensor_azimuth_angle
        self.alt0 = sensor_altitude
        self.cth0 = cloud_top_height
        self.cbh0 = cloud_top_height-cloud_geometrical_thickness
        self.alb0 = surface_albedo
        self.fdir = fdir
        self.output_tag = output_tag
        self.photon0 = Nphoton
        self.solver0 = solver
        self.cpu0 = Ncpu
        self.date0 = date
        self.Nx = Nx
        self.Ny = Ny
        self.dx = dx
        self.dy = dy
        self.atm0 = atm0
        self.abs0 = abs0
        self.pha0 = pha0
        self.cld_gen = cld_gen
        self.quiet = quiet

        self.mu0 = np.cos(np.deg2rad(self.sza0))
        self.ref_2s = cal_r_twostream(self.cot, a=self.alb0, mu=self.mu0)

        if not overwrite:
            try:
                self.load_all()
            except (OSError, KeyError):
                self.run_all()
                self.load_all()
        else:
            self.run_all()
            self.load_all()

    def _fname(self, cot0):
        return '%s/%s_cot-%05.1f_cer-%04.1f.npz' % (self.fdir, self.output_tag, cot0, self.cer0)

    def load_all(self):
        rad, rad_std, pix, toa0 = [], [], [], None
        for cot0 in self.cot:
            out = mca_out_ng(fname=self._fname(cot0), mode='mean', quiet=True)
            pix.append(np.array(out.data['rad']['data']))
            rad.append(np.mean(out.data['rad']['data']))
            rad_std.append(np.mean(out.data['rad_std']['data']))
            toa0 = out.data['toa']['data']
        self.rad = np.array(rad); self.rad_std = np.array(rad_std)
        self.rad_pixels = pix
        self.toa0 = toa0
        self.ref     = np.pi*self.rad/(toa0*self.mu0)
        self.ref_std = np.pi*self.rad_std/(toa0*self.mu0)

    def run_all(self):
        shutil.rmtree(self.fdir, ignore_errors=True)
        os.makedirs(self.fdir)
        for cot0 in self.cot:
            self.run_one(cot0, self.cer0, Nx=self.Nx, Ny=self.Ny, dx=self.dx, dy=self.dy, cbh0=self.cbh0, cth0=self.cth0)

    def run_one(self, cot0, cer0, Nx=2, Ny=2, dx=0.1, dy=0.1, cbh0=1.0, cth0=2.0):

        name_tag = 'cot-%05.1f_cer-%04.1f' % (cot0, cer0)

        # the cloud: the atmosphere's layers whose centres lie between cloud base and top (util.py:331)
        z = self.atm0.lay['altitude']['data']
        altitude0 = z[(z >= cbh0) & (z <= cth0)]
        cld0 = self.cld_gen(cot0=cot0, cer0=cer0, altitude=altitude0, atm_obj=self.atm0, Nx=Nx, Ny=Ny, dx=dx, dy=dy)

        sca0 = None
        if self.pha0 is not None:
            sca0 = mca_sca(pha_obj=self.pha0, fname='%s/mca_sca-%06.1fnm.bin' % (self.fdir, self.wvl0), overwrite=True, quiet=True)

        atm1d0 = mca_atm_1d(atm_obj=self.atm0, abs_obj=self.abs0)
        atm3d0 = mca_atm_3d(cld_obj=cld0, atm_obj=self.atm0, pha_obj=self.pha0, fname='%s/mca_atm_3d_%s.bin' % (self.fdir, name_tag),
                            overwrite=True, quiet=True)

        mca0 = mcarats_ng(date=self.date0, atm_1ds=[atm1d0], atm_3ds=[atm3d0], sca=sca0, target='radiance',
                          surface_albedo=self.alb0, solar_zenith_angle=self.sza0, solar_azimuth_angle=self.saa0,
                          sensor_zenith_angle=self.vza0, sensor_azimuth_angle=self.vaa0, sensor_altitude=self.alt0,
                          fdir='%s/%s_%s/rad' % (self.fdir, self.output_tag, name_tag), Nrun=3, Ng=self.abs0.Ng,
                          weights=self.abs0.coef['weight']['data'], photons=self.photon0, solver=self.solver0,
                          Ncpu=self.cpu0, mp_mode='py', overwrite=True, quiet=self.quiet)
        mca_out_ng(fname=self._fname(cot0), mca_obj=mca0, abs_obj=self.abs0, mode='mean', squeeze=True, quiet=True, overwrite=True)

    def get_cot_from_ref(self, ref, method='cubic', mode='rt'):
        x = self.ref_2s if mode == '2s' else self.ref
        return interp1d(x, self.cot, kind=method, bounds_error=False, fill_value='extrapolate')(ref)

    def get_ref_from_cot(self, cot, method='cubic', mode='rt'):
        y = self.ref_2s if mode == '2s' else self.ref
        return interp1d(self.cot, y, kind=method, bounds_error=False)(cot)
