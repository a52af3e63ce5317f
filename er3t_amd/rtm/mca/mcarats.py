"""
`mcarats_ng`: set up and run a 3-D radiative-transfer simulation (flux or radiance) on MI355X GPUs.

Drop-in for the reference's `mcarats_ng` (er3t/rtm/mca/mcarats.py:21-523): same keyword arguments, attributes, file
names and error conventions.  It writes the reference's input files (r%02d.g%03d.inp.txt), runs every (run, g) job
through the HIP solver instead of launching MCARaTS processes, and leaves r%02d.g%03d.out.bin + .ctl files that
`mca_out_ng` (this package's or the reference's) reads.

The namelist content is assembled from the small tables below; which key gets which value is the reference's
choice (mcarats.py:234-414) and is pinned byte for byte by tests/test_golden_host.py.
"""

import datetime
import multiprocessing as mp
import os
import time

import numpy as np

from er3t_amd.rtm.mca.mca_inp import mca_inp_file
from er3t_amd.rtm.mca.mca_run import mca_run

__all__ = ['mcarats_ng', 'cal_mca_azimuth', 'distribute_photon']


# accepted spellings
_SOLVERS = {'3D': ('3d', '3 d', 'three d'),
            'Partial 3D': ('p3d', 'p-3d', 'partial 3d', 'partial-3d'),
            'IPA': ('ipa', 'independent pixel approximation')}
_SOLVER_IDS = {'3D': 0, 'Partial 3D': 1, 'IPA': 2}
_TARGETS = {'flux': ('f', 'flux', 'irradiance'),
            'flux0': ('f0', 'flux0', 'irradiance0'),
            'heating rate': ('heating rate', 'hr'),
            'radiance': ('radiance', 'rad')}

# fixed namelist content per group
_WLD_FIXED = {'Wld_mbswap': 0, 'Wld_njob': 1}
_FLX_FLAGS = {'flux': (3, 0), 'flux0': (1, 0), 'heating rate': (3, 1)}            # (Flx_mflx, Flx_mhrt)
_RAD_FIXED = {'Rad_mplen': 0, 'Rad_mpmap': 1, 'Rad_nrad': 1, 'Rad_difr0': 7.5, 'Rad_difr1': 0.0025}
_RAD_ALLSKY = {'Rad_mrkind': 1, 'Rad_qmax': 178.0, 'Rad_apsize': 0.05}
_SRC_FIXED = {'Src_flx': 1.0, 'Src_qmax': 0.533133, 'Src_dwlen': 0.0, 'Src_mtype': 1, 'Src_mphi': 0}
_VOXEL_ARRAYS = ('Atm_tmpa3d', 'Atm_abst3d', 'Atm_extp3d', 'Atm_omgp3d', 'Atm_apfp3d')   # travel in the side file


def _match(word, table, what):
    for name, spellings in table.items():
        if word.lower() in spellings:
            return name
    raise OSError('Error [mcarats_ng]: Cannot understand <%s=%s>.' % (what, word))


def _relative_side_file(obj_nml, key, fdir):
    """side-file paths are written relative to the directory of the input files"""
    if os.path.exists(obj_nml[key]['data']):
        obj_nml[key]['data'] = os.path.relpath(obj_nml[key]['data'], start=fdir)


class mcarats_ng:

    """
    Keyword arguments (defaults in brackets):
        atm_1ds [[]], atm_3ds [[]] : lists of mca_atm_1d / mca_atm_3d objects;  sca [None]: mca_sca object
        Ng [16], weights [None: photons split evenly over g]
        fdir ['tmp-data/sim'], Nrun [3], Ncpu ['auto'], mp_mode ['py'], overwrite [True]
        date [now], comment [False], tune [False], target ['flux'] : 'flux' | 'flux0' | 'radiance' | 'heating rate'
        surface_albedo [0.03] : float (Lambertian) or mca_sfc_2d object
        solar_zenith_angle [30], solar_azimuth_angle [0], sensor_zenith_angle [0], sensor_azimuth_angle [0],
        sensor_altitude [705000], sensor_type ['satellite'], sensor_xpos [0.5], sensor_ypos [0.5]
        solver ['3d'] : '3d' | 'p3d' | 'ipa',  photons [1e7],  base_ratio [0.05],  verbose [False],  quiet [False]

    Not in the reference:
        sensor_zenith_angle, sensor_azimuth_angle as SEQUENCES (satellite sensors): several views from one set of photon
                           histories -- Rad_nrad views in every job, the radiance of `mca_out_ng` gets the views as its third
                           axis (Nx, Ny, Nview) -- where the reference runs one simulation per view (nine for a nine-angle
                           instrument: the nine-view job costs 4.4 x LESS than nine single-view ones here)
        abs_obj [None]   : the absorption object `mca_out_ng` will be given.  With it the sum over g of every run and
                           the mean / standard deviation over runs are accumulated on the GPU while the jobs run
                           (attribute `fused`), and `mca_out_ng(mca_obj=..., abs_obj=...)` takes them from there
                           instead of reading Nrun*Ng files back.
        keep_files [True]: with abs_obj, False skips writing the r%02d.g%03d.out.bin files altogether
                           (a flux job on 480 x 480 x 100 is 0.3 GB per file).

    Afterwards: input and output files under <fdir>; attributes Ng, Nrun, Nx, Ny, dx, dy, date, target, solver,
    photons (Nrun*Ng,), photons_per_set, fnames_inp[ir][ig], fnames_out[ir][ig], nml (list of Ng dictionaries).
    """

    reference = '\nMCARaTS (Iwabuchi, 2006; Iwabuchi and Okamura, 2017):\n- Iwabuchi, H.: Efficient Monte Carlo methods for radiative transfer modeling, J. Atmos. Sci., 63, 2324-2339, https://doi.org/10.1175/JAS3755.1, 2006.\n- Iwabuchi, H., and Okamura, R.: Multispectral Monte Carlo radiative transfer simulation by using the maximum cross-section method, Journal of Quantitative Spectroscopy and Radiative Transfer, 193, 40-46, https://doi.org/10.1016/j.jqsrt.2017.01.025, 2017.'

    def __init__(self, atm_1ds=[], atm_3ds=[], sca=None, Ng=16, weights=None, fdir='tmp-data/sim', Nrun=3, Ncpu='auto',
                 mp_mode='py', overwrite=True, date=datetime.datetime.now(), comment=False, tune=False, target='flux',
                 surface_albedo=0.03, solar_zenith_angle=30.0, solar_azimuth_angle=0.0, sensor_zenith_angle=0.0,
                 sensor_azimuth_angle=0.0, sensor_altitude=705000.0, sensor_type='satellite', sensor_xpos=0.5,
                 sensor_ypos=0.5, solver='3d', photons=1e7, base_ratio=0.05, verbose=False, quiet=False,
                 abs_obj=None, keep_files=True):

        # one process per GPU (torch.distributed): rank 0 writes the job files, all ranks transport their share of every job
        from er3t_amd.dist import world_info, barrier
        self.rank, self.world = world_info()
        if self.rank != 0:
            quiet, verbose = True, False

        self.fdir = os.path.abspath(fdir)
        if not os.path.exists(self.fdir):
            os.makedirs(self.fdir, exist_ok=True)
            if not quiet:
                print('Message [mcarats_ng]: Directory <%s> is created.' % self.fdir)
        elif verbose:
            print('Message [mcarats_ng]: Directory <%s> already exists.' % self.fdir)

        # what was asked for, kept as attributes (mca_out_ng and user scripts read several of them)
        for name in ('Ng', 'date', 'verbose', 'quiet', 'overwrite', 'sca', 'Nrun', 'target', 'surface_albedo',
                     'solar_zenith_angle', 'solar_azimuth_angle', 'sensor_zenith_angle', 'sensor_azimuth_angle',
                     'sensor_altitude', 'sensor_type', 'sensor_xpos', 'sensor_ypos'):
            setattr(self, name, locals()[name])
        self.mp_mode = mp_mode.lower()
        self.Nview = max(int(np.size(sensor_zenith_angle)), int(np.size(sensor_azimuth_angle)))     # (several views: not in the reference, init_wld)
        self.abs_obj, self.keep_files, self.fused = abs_obj, keep_files, None
        self.solver  = _match(solver, _SOLVERS, 'solver')
        self.Nx, self.Ny = (atm_3ds[0].nml['Atm_nx']['data'], atm_3ds[0].nml['Atm_ny']['data']) if len(atm_3ds) > 0 else (1, 1)

        # photons of every (run, g) job
        self.np_mode = 'evenly' if weights is None else 'weighted'
        if weights is None:
            weights = np.repeat(1.0/Ng, Ng)
        per_g = distribute_photon(photons, weights, base_ratio=base_ratio)
        self.photons = np.tile(per_g, Nrun)
        self.photons_per_set = per_g.sum()

        # Ncpu only labels the banner and orders batch scripts: the jobs run on the GPU
        self.Ncpu_total = mp.cpu_count()
        if Ncpu == 'auto':
            self.Ncpu = self.Ncpu_total - 1
        elif Ncpu > 1:
            self.Ncpu = min(Ncpu, self.Ncpu_total)
        else:
            raise OSError('Error [mcarats_ng]: Cannot understand <Ncpu=%s>.' % Ncpu)

        # r = run index, g = g index, both counted from 0
        stem = lambda ir, ig: '%s/r%2.2d.g%3.3d' % (self.fdir, ir, ig)
        self.fnames_inp = [[stem(ir, ig)+'.inp.txt' for ig in range(Ng)] for ir in range(Nrun)]
        self.fnames_out = [[stem(ir, ig)+'.out.bin' for ig in range(Ng)] for ir in range(Nrun)]

        if overwrite:
            self.nml = [{} for ig in range(Ng)]
            self.init_wld(verbose=verbose, tune=tune, sensor_zenith_angle=sensor_zenith_angle, sensor_azimuth_angle=sensor_azimuth_angle,
                          sensor_type=sensor_type, sensor_altitude=sensor_altitude, sensor_xpos=sensor_xpos, sensor_ypos=sensor_ypos)
            self.init_sca(sca=sca)
            self.init_atm(atm_1ds=atm_1ds, atm_3ds=atm_3ds)
            self.init_sfc(surface_albedo=surface_albedo)
            self.init_src(solar_zenith_angle=solar_zenith_angle, solar_azimuth_angle=solar_azimuth_angle)
            if self.rank == 0:
                self.gen_mca_inp(comment=comment)
            barrier()               # the input files (and the seeds in them) are rank 0's; side files are complete on every rank
            self.gen_mca_out()
            barrier()               # rank 0 has written the outputs
        elif not quiet:
            print('Message [mcarats_ng]: Reading mode ...')

        if self.mp_mode not in ['batch', 'shell', 'bash', 'hpc', 'sh'] and (self.fused is None or self.keep_files):
            self.run_check()

    def _all(self, entries):
        for nml in self.nml:
            nml.update(entries)

    # ---- namelist assembly -------------------------------------------------------------------
    def init_wld(self, tune=False, verbose=False, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0,
                 sensor_type='satellite', sensor_altitude=705000.0, sensor_xpos=0.5, sensor_ypos=0.5):

        self.target = _match(self.target, _TARGETS, 'target')
        self._all(dict(_WLD_FIXED, Wld_mverb=3 if verbose else 0, Wld_moptim=2 if tune else 0))
        if tune and self.rank == 0:
            # (Wld_moptim=2 turns on MCARaTS' biasing optimisations -- collision forcing, truncation approximations,
            #  er3t/rtm/mca/mca_inp.py:27-33,52-54,193-199; the input files carry the flag as the reference writes it)
            print('Warning [mcarats_ng]: <tune=True> (Wld_moptim=2) is written to the input files, but the GPU solver runs its unbiased estimator (Wld_moptim=0) whatever the flag says.')

        if self.target != 'radiance':
            mflx, mhrt = _FLX_FLAGS[self.target]
            self._all({'Wld_mtarget': 1, 'Flx_mflx': mflx, 'Flx_mhrt': mhrt})
            return

        # Not in the reference: SEVERAL views in one simulation (sequences of sensor zenith / azimuth angles, a scalar stands for every
        # view): one set of photon histories serves them all, Rad_nrad views in the job files and the views as the third axis of the
        # radiance (the reference runs a simulation per view, mcarats.py:301: nine for a nine-angle instrument)
        # (arrays go into the job files as the reference writes arrays: '%12g', six significant digits -- 1e-4 degrees of a view angle)
        vza, vaa = np.atleast_1d(np.asarray(sensor_zenith_angle, dtype=np.float64)), np.atleast_1d(np.asarray(sensor_azimuth_angle, dtype=np.float64))
        self.Nview = max(vza.size, vaa.size)
        if self.Nview > 1:
            if 'satellite' not in sensor_type.lower() or vza.size not in (1, self.Nview) or vaa.size not in (1, self.Nview):
                raise OSError('Error [mcarats_ng]: several views need <sensor_type=\'satellite\'> and as many zenith as azimuth angles (or one of either).')
            vza, vaa = np.resize(vza, self.Nview), np.resize(vaa, self.Nview)
            rad = dict(_RAD_FIXED, Wld_mtarget=2, Rad_nrad=self.Nview, Rad_the=180.0-vza, Rad_phi=np.array([cal_mca_azimuth(a) for a in vaa]),
                       Rad_zloc=np.repeat(float(sensor_altitude), self.Nview))
        else:
            rad = dict(_RAD_FIXED, Wld_mtarget=2, Rad_the=180.0-sensor_zenith_angle, Rad_phi=cal_mca_azimuth(sensor_azimuth_angle),
                       Rad_zloc=sensor_altitude)
        if 'satellite' in sensor_type.lower():
            rad['Rad_mrkind'] = 2
        elif 'all-sky' in sensor_type.lower():
            rad.update(_RAD_ALLSKY, Rad_xpos=sensor_xpos, Rad_ypos=sensor_ypos)
        self._all(rad)

    def init_sca(self, sca=None):
        if sca is None:
            self._all({'Sca_npf': 0})                 # must be explicit: the solver's own default is not 0
        else:
            _relative_side_file(sca.nml, 'Sca_inpfile', self.fdir)
            self._all({key: item['data'] for key, item in sca.nml.items()})

    def init_atm(self, atm_1ds=[], atm_3ds=[]):
        if len(atm_1ds) == 0:
            raise OSError('Error [mcarats_ng]: need <atm_1ds> to proceed.')
        self.wvl_info = atm_1ds[-1].wvl_info

        for ig, nml in enumerate(self.nml):
            for atm_1d in atm_1ds:
                nml.update({key: item['data'] for key, item in atm_1d.nml[ig].items()})

        for atm_3d in atm_3ds:
            _relative_side_file(atm_3d.nml, 'Atm_inpfile', self.fdir)
            self._all({key: item['data'] for key, item in atm_3d.nml.items() if key not in _VOXEL_ARRAYS})
            self.Nx, self.Ny = atm_3d.nml['Atm_nx']['data'], atm_3d.nml['Atm_ny']['data']
            self.dx, self.dy = atm_3d.nml['Atm_dx']['data'], atm_3d.nml['Atm_dy']['data']
            if self.target == 'radiance':
                # one pixel per column for a satellite image; a 500 x 500 fish-eye image for an all-sky camera
                if 'satellite' in self.sensor_type.lower():
                    self._all({'Rad_nxr': self.Nx, 'Rad_nyr': self.Ny})
                elif 'all-sky' in self.sensor_type.lower():
                    self._all({'Rad_nxr': 500, 'Rad_nyr': 500})

    def init_src(self, solar_zenith_angle=0.0, solar_azimuth_angle=0.0):
        self._all(dict(_SRC_FIXED, Src_the=180.0-solar_zenith_angle, Src_phi=cal_mca_azimuth(solar_azimuth_angle)))

    def init_sfc(self, surface_albedo=0.03):
        if self.verbose:
            print('Message [mcarats_ng]: Assume Lambertian surface ...')
        if isinstance(surface_albedo, (float, np.float32, np.float64)):
            self.sfc_2d = False
            for nml in self.nml:
                nml.update({'Sfc_mbrdf': np.array([1, 0, 0, 0]), 'Sfc_mtype': 1, 'Sfc_param(1)': surface_albedo})
        elif hasattr(surface_albedo, 'nml') and 'Sfc_inpfile' in surface_albedo.nml:
            self.sfc_2d = True
            _relative_side_file(surface_albedo.nml, 'Sfc_inpfile', self.fdir)
            self._all({key: item['data'] for key, item in surface_albedo.nml.items() if '2d' not in key})
        else:
            raise ValueError('\nError [mcarats_ng]: Cannot ingest <surface_albedo>.')

    # ---- files and execution -----------------------------------------------------------------
    def gen_mca_inp(self, comment=False):

        """one input file per (run, g); the jobs get distinct seeds: the clock plus a shuffled job number"""

        base = int(time.time())
        order = np.arange(self.Nrun*self.Ng).reshape((self.Nrun, self.Ng))
        np.random.shuffle(order)
        for ir, row in enumerate(self.fnames_inp):
            for ig, fname in enumerate(row):
                self.nml[ig]['Wld_jseed'] = base + order[ir, ig]
                mca_inp_file(fname, self.nml[ig], comment=comment)
        if not self.quiet:
            print('Message [mcarats_ng]: Created MCARaTS input files under <%s>.' % self.fdir)

    def gen_mca_out(self):

        """run every job on the GPU (or write the batch script)"""

        if not self.quiet:
            print('Message [mcarats_ng]: Running the GPU solver to get output files under <%s> ...' % self.fdir)
            self.print_info()
        if self.abs_obj is not None and self.mp_mode not in ['batch', 'shell', 'bash', 'hpc', 'sh'] and self.target != 'heating rate':
            # (heating rates go through the job files: the run statistics on the device hold radiance and flux fields)
            self.run_fused()
            return
        self.run0 = mca_run(sum(self.fnames_inp, []), sum(self.fnames_out, []), photons=self.photons, solver=_SOLVER_IDS[self.solver],
                            Ncpu=self.Ncpu, verbose=self.verbose, quiet=self.quiet, mp_mode=self.mp_mode)

    def run_fused(self):

        """
        All (run, g) jobs back to back through one solver handle with the reduction of the reference's reader
        (er3t/rtm/mca/mca_out.py:313-352, 438-500) done on the device: per run the sum over g of factor[level, g] * result,
        over runs the mean and standard deviation.  self.fused = {'rad' | 'flux': {'mean', 'std', 'runs', 'nrun'}, 'toa'}
        with arrays in the solver's layout, rad (nview, ny, nx), flux (3: direct-down, total-down, up; nz+1; ny; nx).
        """

        from er3t_amd.rtm.mca.mca_exe import get_runner
        from er3t_amd.rtm.mca.mca_inp import mca_inp_read
        from er3t_amd.rtm.mca.mca_out import g_factors

        runner = get_runner()
        if self.keep_files and runner.world > 1:
            raise OSError('Error [mcarats_ng]: <keep_files=True> with <abs_obj> is a single-process option; use keep_files=False under torchrun.')
        solver = _SOLVER_IDS[self.solver]
        ms0, n0 = runner.kernel_ms, runner.photons_done
        photons = self.photons.reshape((self.Nrun, self.Ng))
        factors, runs = None, []
        # Two solver handles take turns (unless the per-job files are wanted: they are read back job by job): job i+1 is
        # launched before job i is folded into the run, so the tail of a launch -- as long as its longest history -- runs
        # beside the next launch.  The run field is still summed in job order (JobRunner.stats_add).
        nslot = runner.use_slots(1 if self.keep_files else 2)
        for ir in range(self.Nrun):
            waiting = None          # (photons, factors, slot) of the job launched last, not yet folded into the run
            for ig in range(self.Ng):
                slot = ig % nslot
                nml = mca_inp_read(self.fnames_inp[ir][ig])
                scene = runner.load(nml, self.fdir, solver, slot=slot)
                if factors is None:
                    if self.target == 'radiance':      # (every view scaled like the reference's one view: the slit function of the lowest layer)
                        factors, toa = g_factors(self, self.abs_obj, 1)
                        factors = np.repeat(factors, scene.nview, axis=0)
                    else:
                        factors, toa = g_factors(self, self.abs_obj, scene.nz+1)
                    runner.stats_begin()
                runner.launch(photons[ir, ig], int(nml['Wld_jseed']), slot=slot)
                if self.keep_files:
                    result = {'rad': runner.sol.radiance(photons[ir, ig])} if self.target == 'radiance' else {'flux': runner.sol.flux(photons[ir, ig])}
                    runner.write(self.fnames_out[ir][ig], result)
                # (a job is folded into the run before its slot is launched on again: with one slot at once, with two after the
                #  next job's launch)
                if waiting is not None:
                    runner.stats_add(*waiting)
                waiting = (photons[ir, ig], factors[:, ig], slot)
                if nslot == 1:
                    runner.stats_add(*waiting)
                    waiting = None
            if waiting is not None:
                runner.stats_add(*waiting)
            runs.append(runner.stats_end_run(keep=True))
        self.fused = runner.stats_result()
        for key in self.fused:
            self.fused[key]['runs'] = np.stack([r[key] for r in runs], axis=-1)
        self.fused['toa'] = toa
        self.kernel_ms = runner.kernel_ms - ms0
        self.photons_done = runner.photons_done - n0
        if not self.quiet and self.kernel_ms > 0.0:
            print('Message [mcarats_ng]: %d jobs fused on the device, %.3g photon histories per rank in %.1f ms of transport kernels (%.3g photons/s/GPU).'
                  % (self.Nrun*self.Ng, self.photons_done, self.kernel_ms, self.photons_done/(self.kernel_ms*1.0e-3)))

    def run_check(self):
        if not all(os.path.exists(f) for row in self.fnames_out for f in row):
            raise OSError('Error [mcarats_ng]: Missing some output files.')

    def print_info(self):
        rows = [('Simulation', '%s %s' % (self.solver, self.target.title())),
                ('Wavelength', '%s' % self.wvl_info),
                ('Date (DOY)', '%s (%d)' % (self.date.strftime('%Y-%m-%d'), self.date.timetuple().tm_yday)),
                ('Solar Zenith Angle', '%.4f° (0 at local zenith)' % self.solar_zenith_angle),
                ('Solar Azimuth Angle', '%.4f° (0 at north; 90° at east)' % self.solar_azimuth_angle)]
        if self.target == 'radiance':
            if getattr(self, 'Nview', 1) > 1:
                rows += [('Sensor Zenith Angles', ', '.join('%.1f°' % a for a in np.atleast_1d(self.sensor_zenith_angle)) + ' (%d views, one set of photons)' % self.Nview),
                         ('Sensor Azimuth Angles', ', '.join('%.1f°' % a for a in np.atleast_1d(self.sensor_azimuth_angle))),
                         ('Sensor Altitude', '%.1f km' % (self.sensor_altitude/1000.0))]
            else:
                looking = '(looking down, 0 straight down)' if self.sensor_zenith_angle < 90.0 else '(looking up, 180° straight up)'
                rows += [('Sensor Zenith Angle', '%.4f° %s' % (self.sensor_zenith_angle, looking)),
                         ('Sensor Azimuth Angle', '%.4f° (0 at north; 90° at east)' % self.sensor_azimuth_angle),
                         ('Sensor Altitude', '%.1f km' % (self.sensor_altitude/1000.0))]
        rows += [('Surface Albedo', '2D domain' if self.sfc_2d else '%.2f' % self.surface_albedo),
                 ('Phase Function', 'Henyey-Greenstein' if self.sca is None else '%s' % self.sca.pha.ID)]
        if (self.Nx > 1) | (self.Ny > 1):
            rows += [('Domain Size (Nx, Ny)', '(%d, %d)' % (self.Nx, self.Ny)),
                     ('Pixel Res. (dx, dy)', '(%.2f km, %.2f km)' % (self.dx/1000.0, self.dy/1000.0))]
        rows += [('Number of Photons / Set', '%.1e (%s over %d g)' % (self.photons_per_set, self.np_mode, self.Ng)),
                 ('Number of Runs', '%s (g) * %d (set)' % (self.Ng, self.Nrun)),
                 ('Solver', 'er3t_amd HIP kernels on MI355X (Ncpu=%d only labels batch scripts)' % self.Ncpu)]
        print('╭────────────────────────────────────────────────────────╮')
        print('                 General Information                      ')
        for k, v in rows:
            print('%25s : %s' % (k, v))
        print('╰────────────────────────────────────────────────────────╯')


def cal_mca_azimuth(normal_azimuth_angle):

    """
    Compass azimuth (0 = north, clockwise, 90 = east) -> the solver's azimuth (0 = travelling east, counter-clockwise):
    270 - angle, with the input first brought into [0, 360] and a negative result raised by 360.
    (reference: er3t/rtm/mca/mcarats.py:527-549)
    """

    a = normal_azimuth_angle
    while a < 0.0:
        a += 360.0
    while a > 360.0:
        a -= 360.0
    a = 270.0 - a
    return a + 360.0 if a < 0.0 else a


def distribute_photon(Nphoton, weights, base_ratio=0.05):

    """
    Photons per g: the share (1 - base_ratio) by weight plus the share base_ratio split evenly, each term truncated to
    an integer; what is then missing goes to the g of smallest weight, any excess is taken from the g of largest weight.
    (reference: er3t/rtm/mca/mcarats.py:553-565)
    """

    n = np.int_(Nphoton*(1.0-base_ratio)*weights) + np.int_(Nphoton*base_ratio/weights.size)
    rest = Nphoton - n.sum()
    n[np.argmin(weights) if rest >= 0 else np.argmax(weights)] += rest
    return n
