"""
`mcarats_ng`: set up and run a 3-D radiative-transfer simulation (flux or radiance) on MI355X GPUs.

Same constructor arguments, attributes, files and error conventions as the reference's `mcarats_ng`
(er3t/rtm/mca/mcarats.py:21-523), so it can be swapped in behind user scripts: it writes the reference's input files
(r%02d.g%03d.inp.txt), runs every (run, g) job through the HIP solver instead of launching MCARaTS processes, and
leaves r%02d.g%03d.out.bin + .ctl files that `mca_out_ng` (this package's or the reference's) reads.
"""

import datetime
import multiprocessing as mp
import os
import time

import numpy as np

from er3t_amd.rtm.mca.mca_inp import mca_inp_file
from er3t_amd.rtm.mca.mca_run import mca_run
from er3t_amd.rtm.mca.mca_sfc import mca_sfc_2d

__all__ = ['mcarats_ng', 'cal_mca_azimuth', 'distribute_photon']


_BIG_3D_KEYS = ('Atm_tmpa3d', 'Atm_abst3d', 'Atm_extp3d', 'Atm_omgp3d', 'Atm_apfp3d')
_SOLVER_IDS = {'3D': 0, 'Partial 3D': 1, 'IPA': 2}


class mcarats_ng:

    """
    Input (all keyword arguments, defaults as in the reference):
        atm_1ds=[], atm_3ds=[]: lists of mca_atm_1d / mca_atm_3d objects
        sca=None              : mca_sca object (tabulated phase functions)
        Ng=16, weights=None   : number of g and their weights (photons are split evenly when None)
        fdir=, Nrun=3, Ncpu='auto', mp_mode='py', overwrite=True
        date=, comment=False, tune=False, target='flux' | 'flux0' | 'radiance' | 'heating rate'
        surface_albedo=0.03   : float (Lambertian) or mca_sfc_2d object
        solar_zenith_angle=30, solar_azimuth_angle=0, sensor_zenith_angle=0, sensor_azimuth_angle=0,
        sensor_altitude=705000, sensor_type='satellite', sensor_xpos=0.5, sensor_ypos=0.5
        solver='3d' | 'p3d' | 'ipa', photons=1e7, base_ratio=0.05, verbose=False, quiet=False

    Output:
        input and output files under <fdir>; attributes Ng, Nrun, Nx, Ny, dx, dy, date, target, solver, photons,
        photons_per_set, fnames_inp, fnames_out, nml (list of Ng namelist dictionaries)
    """

    reference = '\nMCARaTS (Iwabuchi, 2006; Iwabuchi and Okamura, 2017):\n- Iwabuchi, H.: Efficient Monte Carlo methods for radiative transfer modeling, J. Atmos. Sci., 63, 2324-2339, https://doi.org/10.1175/JAS3755.1, 2006.\n- Iwabuchi, H., and Okamura, R.: Multispectral Monte Carlo radiative transfer simulation by using the maximum cross-section method, Journal of Quantitative Spectroscopy and Radiative Transfer, 193, 40-46, https://doi.org/10.1016/j.jqsrt.2017.01.025, 2017.'

    def __init__(self,
                 atm_1ds=[], atm_3ds=[], sca=None, Ng=16, weights=None,
                 fdir='tmp-data/sim', Nrun=3, Ncpu='auto', mp_mode='py', overwrite=True,
                 date=datetime.datetime.now(), comment=False, tune=False, target='flux',
                 surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=0.0,
                 sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, sensor_altitude=705000.0,
                 sensor_type='satellite', sensor_xpos=0.5, sensor_ypos=0.5,
                 solver='3d', photons=1e7, base_ratio=0.05, verbose=False, quiet=False):

        fdir = os.path.abspath(fdir)
        if not os.path.exists(fdir):
            os.makedirs(fdir)
            if not quiet:
                print('Message [mcarats_ng]: Directory <%s> is created.' % fdir)
        elif verbose:
            print('Message [mcarats_ng]: Directory <%s> already exists.' % fdir)

        self.Ng        = Ng
        self.date      = date
        self.fdir      = fdir
        self.verbose   = verbose
        self.quiet     = quiet
        self.overwrite = overwrite
        self.mp_mode   = mp_mode.lower()
        self.sca       = sca
        self.Nrun      = Nrun
        self.target    = target

        self.surface_albedo       = surface_albedo
        self.solar_zenith_angle   = solar_zenith_angle
        self.solar_azimuth_angle  = solar_azimuth_angle
        self.sensor_zenith_angle  = sensor_zenith_angle
        self.sensor_azimuth_angle = sensor_azimuth_angle
        self.sensor_altitude      = sensor_altitude
        self.sensor_type          = sensor_type
        self.sensor_xpos          = sensor_xpos
        self.sensor_ypos          = sensor_ypos

        key = solver.lower()
        if key in ['3d', '3 d', 'three d']:
            self.solver = '3D'
        elif key in ['p3d', 'p-3d', 'partial 3d', 'partial-3d']:
            self.solver = 'Partial 3D'
        elif key in ['ipa', 'independent pixel approximation']:
            self.solver = 'IPA'
        else:
            raise OSError('Error [mcarats_ng]: Cannot understand <solver=%s>.' % solver)

        if len(atm_3ds) > 0:
            self.Nx = atm_3ds[0].nml['Atm_nx']['data']
            self.Ny = atm_3ds[0].nml['Atm_ny']['data']
        else:
            self.Nx = 1
            self.Ny = 1

        # photons per g: by weight (plus an even share `base_ratio`) or evenly
        if weights is None:
            self.np_mode = 'evenly'
            weights = np.repeat(1.0/self.Ng, Ng)
        else:
            self.np_mode = 'weighted'
        photons_dist = distribute_photon(photons, weights, base_ratio=base_ratio)
        self.photons = np.tile(photons_dist, Nrun)
        self.photons_per_set = photons_dist.sum()

        # Ncpu only labels the banner and orders batch scripts: the jobs run on the GPU
        self.Ncpu_total = mp.cpu_count()
        if Ncpu == 'auto':
            self.Ncpu = self.Ncpu_total - 1
        elif Ncpu > 1:
            self.Ncpu = min(Ncpu, self.Ncpu_total)
        else:
            raise OSError('Error [mcarats_ng]: Cannot understand <Ncpu=%s>.' % Ncpu)

        # file names: r = run index, g = g index, both from 0; fnames_inp[ir][ig]
        self.fnames_inp = [['%s/r%2.2d.g%3.3d.inp.txt' % (self.fdir, ir, ig) for ig in range(self.Ng)] for ir in range(self.Nrun)]
        self.fnames_out = [['%s/r%2.2d.g%3.3d.out.bin' % (self.fdir, ir, ig) for ig in range(self.Ng)] for ir in range(self.Nrun)]

        if not self.quiet and not self.overwrite:
            print('Message [mcarats_ng]: Reading mode ...')

        if overwrite:
            self.nml = [{} for ig in range(self.Ng)]
            self.init_wld(verbose=verbose, tune=tune, sensor_zenith_angle=sensor_zenith_angle, sensor_azimuth_angle=sensor_azimuth_angle,
                          sensor_type=sensor_type, sensor_altitude=sensor_altitude, sensor_xpos=sensor_xpos, sensor_ypos=sensor_ypos)
            self.init_sca(sca=sca)
            self.init_atm(atm_1ds=atm_1ds, atm_3ds=atm_3ds)
            self.init_sfc(surface_albedo=surface_albedo)
            self.init_src(solar_zenith_angle=solar_zenith_angle, solar_azimuth_angle=solar_azimuth_angle)
            self.gen_mca_inp(comment=comment)
            self.gen_mca_out()

        if self.mp_mode not in ['batch', 'shell', 'bash', 'hpc', 'sh']:
            self.run_check()

    # ---- namelist assembly -------------------------------------------------------------------
    def init_wld(self, tune=False, verbose=False, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0,
                 sensor_type='satellite', sensor_altitude=705000.0, sensor_xpos=0.5, sensor_ypos=0.5):

        aliases = {'flux': ['f', 'flux', 'irradiance'], 'flux0': ['f0', 'flux0', 'irradiance0'],
                   'heating rate': ['heating rate', 'hr'], 'radiance': ['radiance', 'rad']}
        for name, alts in aliases.items():
            if self.target.lower() in alts:
                self.target = name
                break
        else:
            raise OSError('Error [mcarats_ng]: Cannot understand <target=%s>.' % self.target)

        for nml in self.nml:
            nml['Wld_mverb']  = 3 if verbose else 0
            nml['Wld_moptim'] = 2 if tune else 0
            nml['Wld_mbswap'] = 0
            nml['Wld_njob']   = 1

            if self.target == 'radiance':
                nml['Wld_mtarget'] = 2
                if 'satellite' in sensor_type.lower():
                    nml['Rad_mrkind'] = 2
                elif 'all-sky' in sensor_type.lower():
                    nml['Rad_mrkind'] = 1
                    nml['Rad_qmax']   = 178.0
                    nml['Rad_apsize'] = 0.05
                    nml['Rad_xpos']   = sensor_xpos
                    nml['Rad_ypos']   = sensor_ypos
                nml['Rad_mplen'] = 0
                nml['Rad_mpmap'] = 1
                nml['Rad_nrad']  = 1
                nml['Rad_difr0'] = 7.5
                nml['Rad_difr1'] = 0.0025
                nml['Rad_the']   = 180.0 - sensor_zenith_angle
                nml['Rad_phi']   = cal_mca_azimuth(sensor_azimuth_angle)
                nml['Rad_zloc']  = sensor_altitude
            else:
                nml['Wld_mtarget'] = 1
                nml['Flx_mflx']    = 1 if self.target == 'flux0' else 3
                nml['Flx_mhrt']    = 1 if self.target == 'heating rate' else 0

    def init_sca(self, sca=None):
        for nml in self.nml:
            if sca is None:
                nml['Sca_npf'] = 0            # must be given explicitly: the solver's default is not 0
            else:
                if os.path.exists(sca.nml['Sca_inpfile']['data']):
                    sca.nml['Sca_inpfile']['data'] = os.path.relpath(sca.nml['Sca_inpfile']['data'], start=self.fdir)
                for key in sca.nml.keys():
                    nml[key] = sca.nml[key]['data']

    def init_atm(self, atm_1ds=[], atm_3ds=[]):

        if len(atm_1ds) == 0:
            raise OSError('Error [mcarats_ng]: need <atm_1ds> to proceed.')

        for ig, nml in enumerate(self.nml):
            for atm_1d in atm_1ds:
                for key in atm_1d.nml[ig].keys():
                    nml[key] = atm_1d.nml[ig][key]['data']
            self.wvl_info = atm_1ds[-1].wvl_info

            for atm_3d in atm_3ds:
                if os.path.exists(atm_3d.nml['Atm_inpfile']['data']):
                    atm_3d.nml['Atm_inpfile']['data'] = os.path.relpath(atm_3d.nml['Atm_inpfile']['data'], start=self.fdir)
                for key in atm_3d.nml.keys():
                    if key not in _BIG_3D_KEYS:
                        nml[key] = atm_3d.nml[key]['data']

                self.Nx = atm_3d.nml['Atm_nx']['data']
                self.Ny = atm_3d.nml['Atm_ny']['data']
                self.dx = atm_3d.nml['Atm_dx']['data']
                self.dy = atm_3d.nml['Atm_dy']['data']

                if self.target == 'radiance':
                    if 'satellite' in self.sensor_type.lower():
                        nml['Rad_nxr'] = atm_3d.nml['Atm_nx']['data']
                        nml['Rad_nyr'] = atm_3d.nml['Atm_ny']['data']
                    elif 'all-sky' in self.sensor_type.lower():
                        nml['Rad_nxr'] = 500
                        nml['Rad_nyr'] = 500

    def init_src(self, solar_zenith_angle=0.0, solar_azimuth_angle=0.0):
        for nml in self.nml:
            nml['Src_flx']   = 1.0
            nml['Src_qmax']  = 0.533133
            nml['Src_dwlen'] = 0.0
            nml['Src_mtype'] = 1
            nml['Src_mphi']  = 0
            nml['Src_the']   = 180.0 - solar_zenith_angle
            nml['Src_phi']   = cal_mca_azimuth(solar_azimuth_angle)

    def init_sfc(self, surface_albedo=0.03):
        for nml in self.nml:
            if self.verbose:
                print('Message [mcarats_ng]: Assume Lambertian surface ...')
            if isinstance(surface_albedo, (float, np.float32, np.float64)):
                nml['Sfc_mbrdf']    = np.array([1, 0, 0, 0])
                nml['Sfc_mtype']    = 1
                nml['Sfc_param(1)'] = surface_albedo
                self.sfc_2d = False
            elif isinstance(surface_albedo, mca_sfc_2d) or (hasattr(surface_albedo, 'nml') and 'Sfc_inpfile' in surface_albedo.nml):
                if os.path.exists(surface_albedo.nml['Sfc_inpfile']['data']):
                    surface_albedo.nml['Sfc_inpfile']['data'] = os.path.relpath(surface_albedo.nml['Sfc_inpfile']['data'], start=self.fdir)
                for key in surface_albedo.nml.keys():
                    if '2d' not in key:
                        nml[key] = surface_albedo.nml[key]['data']
                self.sfc_2d = True
            else:
                raise ValueError('\nError [mcarats_ng]: Cannot ingest <surface_albedo>.')

    # ---- files and execution -----------------------------------------------------------------
    def gen_mca_inp(self, comment=False):

        """one input file per (run, g); every job gets its own random seed"""

        Nseed = int(time.time())
        rands = np.arange(self.Nrun*self.Ng).reshape((self.Nrun, self.Ng))
        np.random.shuffle(rands)
        for ir in range(self.Nrun):
            for ig in range(self.Ng):
                self.nml[ig]['Wld_jseed'] = Nseed + rands[ir, ig]
                mca_inp_file(self.fnames_inp[ir][ig], self.nml[ig], comment=comment)

        if not self.quiet:
            print('Message [mcarats_ng]: Created MCARaTS input files under <%s>.' % self.fdir)

    def gen_mca_out(self):

        """run every job (solver ids: 0 full 3-D, 1 partial 3-D, 2 independent columns)"""

        if self.target == 'heating rate':
            raise OSError('Error [mcarats_ng]: <target=heating rate> is not supported by the GPU solver.')
        if self.solver == 'Partial 3D':
            raise OSError('Error [mcarats_ng]: <solver=Partial 3D> is not supported by the GPU solver.')

        fnames_inp = [f for row in self.fnames_inp for f in row]
        fnames_out = [f for row in self.fnames_out for f in row]

        if not self.quiet:
            print('Message [mcarats_ng]: Running the GPU solver to get output files under <%s> ...' % self.fdir)
            self.print_info()

        self.run0 = mca_run(fnames_inp, fnames_out, photons=self.photons, solver=_SOLVER_IDS[self.solver], Ncpu=self.Ncpu,
                            verbose=self.verbose, quiet=self.quiet, mp_mode=self.mp_mode)

    def run_check(self):
        missing = [f for row in self.fnames_out for f in row if not os.path.exists(f)]
        if len(missing) > 0:
            raise OSError('Error [mcarats_ng]: Missing some output files.')

    def print_info(self):
        sfc = 'Surface Albedo : 2D domain' if self.sfc_2d else 'Surface Albedo : %.2f' % self.surface_albedo
        rows = [('Simulation', '%s %s' % (self.solver, self.target.title())),
                ('Wavelength', '%s' % self.wvl_info),
                ('Date (DOY)', '%s (%d)' % (self.date.strftime('%Y-%m-%d'), self.date.timetuple().tm_yday)),
                ('Solar Zenith Angle', '%.4f° (0 at local zenith)' % self.solar_zenith_angle),
                ('Solar Azimuth Angle', '%.4f° (0 at north; 90° at east)' % self.solar_azimuth_angle)]
        if self.target == 'radiance':
            looking = '(looking down, 0 straight down)' if self.sensor_zenith_angle < 90.0 else '(looking up, 180° straight up)'
            rows += [('Sensor Zenith Angle', '%.4f° %s' % (self.sensor_zenith_angle, looking)),
                     ('Sensor Azimuth Angle', '%.4f° (0 at north; 90° at east)' % self.sensor_azimuth_angle),
                     ('Sensor Altitude', '%.1f km' % (self.sensor_altitude/1000.0))]
        rows += [tuple(sfc.split(' : '))]
        rows += [('Phase Function', 'Henyey-Greenstein' if self.sca is None else '%s' % self.sca.pha.ID)]
        if (self.Nx > 1) | (self.Ny > 1):
            rows += [('Domain Size (Nx, Ny)', '(%d, %d)' % (self.Nx, self.Ny)),
                     ('Pixel Res. (dx, dy)', '(%.2f km, %.2f km)' % (self.dx/1000.0, self.dy/1000.0))]
        rows += [('Number of Photons / Set', '%.1e (%s over %d g)' % (self.photons_per_set, self.np_mode, self.Ng)),
                 ('Number of Runs', '%s (g) * %d (set)' % (self.Ng, self.Nrun)),
                 ('Solver', 'er3t_amd HIP kernels on MI355X (Ncpu=%d ignored)' % self.Ncpu)]
        print('╭────────────────────────────────────────────────────────╮')
        print('                 General Information                      ')
        for k, v in rows:
            print('%25s : %s' % (k, v))
        print('╰────────────────────────────────────────────────────────╯')


def cal_mca_azimuth(normal_azimuth_angle):

    """
    Azimuth measured clockwise from north (0 = from north, 90 = from east) -> the solver's azimuth of travel,
    counter-clockwise from +x (east): 270 - angle, folded into [0, 360)   (reference: er3t/rtm/mca/mcarats.py:527-549)
    """

    while normal_azimuth_angle < 0.0:
        normal_azimuth_angle += 360.0
    while normal_azimuth_angle > 360.0:
        normal_azimuth_angle -= 360.0
    mca_azimuth = 270.0 - normal_azimuth_angle
    if mca_azimuth < 0.0:
        mca_azimuth += 360.0
    return mca_azimuth


def distribute_photon(Nphoton, weights, base_ratio=0.05):

    """
    Photons per g: a fraction (1 - base_ratio) split by weight plus base_ratio split evenly, each truncated to an
    integer; the remainder goes to the lightest g (or is taken from the heaviest).   (reference: mcarats.py:553-565)
    """

    Ndist = weights.size
    photons_dist = np.int_(Nphoton*(1.0-base_ratio)*weights) + np.int_(Nphoton*base_ratio/Ndist)
    Ndiff = Nphoton - photons_dist.sum()
    if Ndiff >= 0:
        photons_dist[np.argmin(weights)] += Ndiff
    else:
        photons_dist[np.argmax(weights)] += Ndiff
    return photons_dist
