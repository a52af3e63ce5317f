"""
Atmosphere adapters: pre-processing objects -> solver inputs (namelist entries + the 3-D side file).

Counterparts of the reference's `mca_atm_1d` / `mca_atm_3d` (er3t/rtm/mca/mca_atm.py:18-139, 144-407).  Inputs are
duck-typed exactly like there: an atmosphere object with `.lay/.lev` dictionaries, an absorption object with
`.coef`, a cloud object with `.lay`, optionally a phase-function object with `.data`.  The 3-D side file is
written with one `ndarray.tofile` per field instead of expanding every value into a Python tuple (the
reference's largest host cost, mca_atm.py:383-388); the bytes are the same.
"""

import copy
import warnings

import numpy as np
from scipy import interpolate

from er3t_amd.util import cal_mol_ext, get_lay_index
from er3t_amd.rtm.mca._adapter import SideFileAdapter, fortran_f4

__all__ = ['mca_atm_1d', 'mca_atm_3d']


def _entry(data, units='N/A', name=''):
    return {'data': data, 'units': units, 'name': name}


class mca_atm_1d:

    """
    1-D background atmosphere, one namelist dictionary per g of the correlated-k set.

    Input:
        atm_obj=: atmosphere object (lev['altitude'|'pressure'], lay['altitude'|'thickness'|'temperature'|...])
        abs_obj=: absorption object (Ng, wvl, wvl_info, coef['abso_coef'] (nz, Ng))

    Output:
        self.nml[ig][key]['data'] for key in Atm_zgrd0 [m], Atm_wkd0, Atm_mtprof, Atm_tmp1d, Atm_nkd, Atm_np1d, Atm_nz,
        'Atm_abs1d(1:, 1)', 'Atm_ext1d(1:, 1)' (Rayleigh), 'Atm_omg1d(1:, 1)' (= 1), 'Atm_apf1d(1:, 1)' (= -1, Rayleigh)
    """

    ID = 'MCARaTS 1D Atmosphere'

    def __init__(self, atm_obj=None, abs_obj=None):

        if atm_obj is None:
            raise OSError('Error [mca_atm_1d]: please provide an \'atm\' object for <atm_obj>.')
        if abs_obj is None:
            raise OSError('Error [mca_atm_1d]: please provide an \'abs\' object for <abs_obj>.')

        self.atm = atm_obj
        self.abs = abs_obj
        self.Ng  = self.abs.Ng
        self.wvl_info = self.abs.wvl_info

        self.pre_mca_1d_atm()

    def pre_mca_1d_atm(self):

        lev = self.atm.lev
        lay = self.atm.lay
        dz_m = lay['thickness']['data']*1000.0
        nz = lay['altitude']['data'].size

        # molecular (Rayleigh) scattering does not depend on g
        ext_ray = cal_mol_ext(self.abs.wvl*0.001, lev['pressure']['data'][:-1], lev['pressure']['data'][1:], self.atm)/dz_m

        self.nml = {}
        for ig in range(self.Ng):
            nml = {}
            nml['Atm_zgrd0']  = _entry(lev['altitude']['data']*1000.0, 'm', 'Layer boundaries')
            nml['Atm_wkd0']   = _entry(1.0, name='Weight coefficients')
            nml['Atm_mtprof'] = _entry(0, name='Temperature profile flag')
            nml['Atm_tmp1d']  = _entry(lay['temperature']['data'], 'K', 'Temperature profile')
            nml['Atm_nkd']    = _entry(1, name='Number of K-distribution')
            nml['Atm_np1d']   = _entry(1, name='Number of 1D atmospheric constituents')
            nml['Atm_nz']     = _entry(nz, name='Number of z grid points')
            nml['Atm_abs1d(1:, 1)'] = _entry(self.abs.coef['abso_coef']['data'][:, ig]/dz_m, '/m', 'Absorption coefficients')
            nml['Atm_ext1d(1:, 1)'] = _entry(ext_ray, '/m', 'Extinction coefficients')
            nml['Atm_omg1d(1:, 1)'] = _entry(np.repeat(1.0, nz), name='Single scattering albedo')
            nml['Atm_apf1d(1:, 1)'] = _entry(np.repeat(-1, nz), name='Phase function')
            self.nml[ig] = nml

    def add_mca_1d_atm(self, ext1d=None, omg1d=None, apf1d=None, z_bottom=None, z_top=None):

        """append a horizontally uniform scattering component (e.g. a plane-parallel cloud between z_bottom and z_top [km])"""

        if (ext1d is None) or (omg1d is None) or (apf1d is None):
            raise OSError('Error [mca_atm_1d]: Please provide values of <ext1d>, <omg1d>, and <apf1d>.')

        z = self.atm.lay['altitude']['data']
        outside = np.zeros(z.size, dtype=bool)
        if z_bottom is not None:
            outside |= (z < z_bottom)
        if z_top is not None:
            outside |= (z > z_top)

        for ig in range(self.Ng):
            prof = {}
            for tag, val in (('ext', ext1d), ('omg', omg1d), ('apf', apf1d)):
                a = np.zeros(z.size)
                a[:] = val
                a[outside] = 0.0
                prof[tag] = a
            n = self.nml[ig]['Atm_np1d']['data'] + 1
            self.nml[ig]['Atm_ext1d(1:, %d)' % n] = _entry(prof['ext'], '/m', 'Extinction coefficients')
            self.nml[ig]['Atm_omg1d(1:, %d)' % n] = _entry(prof['omg'], name='Single scattering albedo')
            self.nml[ig]['Atm_apf1d(1:, %d)' % n] = _entry(prof['apf'], name='Phase function')
            self.nml[ig]['Atm_np1d']['data'] = n


class mca_atm_3d(SideFileAdapter):

    """
    3-D region of the atmosphere (clouds, aerosols) on the layers of the 1-D grid it coincides with.

    Input:
        atm_obj=, cld_obj=: atmosphere / cloud objects; cld.lay['extinction'|'temperature'] are (nx, ny, nz3),
                            cld.lay['dx'|'dy'] in km
        pha_obj=: None (Henyey-Greenstein g = 0.85 everywhere), an 'HG' table set (cloudy cells point at the table
                  nearest g = 0.85), or a 'Mie' set (cloudy cells get omega and the ASYMMETRY PARAMETER interpolated
                  at the cell's effective radius -- the reference stores g, not a table index: mca_atm.py:299-303)
        fname=  : side file to write (default 'mca_atm_3d.bin')

    Output:
        self.nml[key]['data'] for Atm_nx, Atm_ny, Atm_dx [m], Atm_dy [m], Atm_nz3, Atm_iz3l, Atm_np3d,
        Atm_tmpa3d (nx, ny, nz3), Atm_abst3d / Atm_extp3d / Atm_omgp3d / Atm_apfp3d (nx, ny, nz3, np3d), Atm_inpfile
    """

    ID = 'MCARaTS 3D Atmosphere'
    key_inpfile = 'Atm_inpfile'
    default_fname = 'mca_atm_3d.bin'
    tag = 'mca_atm_3d'

    def __init__(self, atm_obj=None, cld_obj=None, pha_obj=None, fname=None, overwrite=True, force=False,
                 verbose=False, quiet=False):

        self.overwrite = overwrite
        self.verbose   = verbose
        self.quiet     = quiet

        if atm_obj is None:
            raise OSError('Error [mca_atm_3d]: Please provide an \'atm\' object for <atm_obj>.')
        if cld_obj is None:
            raise OSError('Error [mca_atm_3d]: Please provide an \'cld\' object for <cld_obj>.')
        self.atm = atm_obj
        self.cld = cld_obj

        if pha_obj is None and self.verbose:
            warnings.warn('Warning [mca_atm_3d]: No phase function set specified - ignore thermodynamic phase/effective radius with g=0.85 (Henyey-Greenstein).')
        self.pha = pha_obj

        if self.cld.lay['altitude']['data'].size != self.cld.lay['thickness']['data'].size:
            msg = 'Error [mca_atm_3d]: Incorrect number of cloud layers (%d) vs layer thicknesses (%d).' % (self.cld.lay['altitude']['data'].size, self.cld.lay['thickness']['data'].size)
            raise ValueError(msg)

        self.pre_mca_3d_atm()
        self._settle_file(fname, overwrite, force, self.gen_mca_3d_atm_file)

    def pre_mca_3d_atm(self):

        cld = self.cld.lay
        lay_index = get_lay_index(cld['altitude']['data'], self.atm.lay['altitude']['data'])

        nx  = cld['nx']['data']
        ny  = cld['ny']['data']
        nz3 = int(lay_index.size)
        iz3l = int(lay_index[0]) + 1            # 1-based index of the first cloudy layer

        if (iz3l+nz3) > self.atm.lay['altitude']['data'].size:
            raise ValueError('Error [mca_atm_3d]: Non-homogeneous layer top exceeds atmosphere top.')

        ext_in = cld['extinction']['data']
        ext_np = ext_in.data if isinstance(ext_in, np.ma.MaskedArray) else ext_in

        atm_tmp = np.zeros((nx, ny, nz3), dtype=np.float32)
        atm_abs = np.zeros((nx, ny, nz3, 1), dtype=np.float32)
        atm_ext = np.zeros((nx, ny, nz3, 1), dtype=np.float32)
        atm_omg = np.ones((nx, ny, nz3, 1), dtype=np.float32)
        atm_apf = np.zeros((nx, ny, nz3, 1), dtype=np.float32)

        atm_tmp[...] = cld['temperature']['data'] - self.atm.lay['temperature']['data'][lay_index][None, None, :]
        atm_ext[..., 0] = ext_in

        if self.pha is None:
            atm_apf[...] = 0.85
        else:
            atm_apf[...] = -1.0                                   # Rayleigh unless the cell holds cloud
            cloudy = (ext_np > 0.0)
            kind = self.pha.data['id']['data'].lower()
            if kind == 'hg':
                atm_apf[cloudy, 0] = np.argmin(np.abs(self.pha.data['asy']['data']-0.85)) + 1.0
            elif kind == 'mie':
                cer = cld['cer']['data']
                cer = cer.data if isinstance(cer, np.ma.MaskedArray) else cer
                ref = self.pha.data['ref']['data']
                f_ssa = interpolate.interp1d(ref, self.pha.data['ssa']['data'], bounds_error=False, fill_value='extrapolate')
                f_asy = interpolate.interp1d(ref, self.pha.data['asy']['data'], bounds_error=False, fill_value='extrapolate')
                atm_omg[cloudy, 0] = f_ssa(cer[cloudy])
                atm_apf[cloudy, 0] = f_asy(cer[cloudy])

        self.nml = {}
        self.nml['Atm_nx'] = copy.deepcopy(cld['nx'])
        self.nml['Atm_ny'] = copy.deepcopy(cld['ny'])
        for key, src in (('Atm_dx', 'dx'), ('Atm_dy', 'dy')):
            self.nml[key] = copy.deepcopy(cld[src])
            self.nml[key]['data'] *= 1000.0
            self.nml[key]['units'] = 'm'

        self.nml['Atm_nz3']  = {'data': nz3, 'unit': 'N/A', 'name': 'number of 3D layer'}
        # NOTE the reference hands the solver iz3l+1, i.e. one more than the 1-based index of the first cloudy
        # layer (er3t/rtm/mca/mca_atm.py:242,330); reproduced so that both sides describe the same input
        self.nml['Atm_iz3l'] = {'data': iz3l+1, 'unit': 'N/A', 'name': 'layer index of first 3D layer'}

        self.nml['Atm_tmpa3d'] = _entry(atm_tmp, 'K', 'Temperature deviation')
        self.nml['Atm_abst3d'] = _entry(atm_abs, '/m', 'Absorption coefficients deviation')
        self.nml['Atm_extp3d'] = _entry(atm_ext, '/m', 'Extinction coefficients')
        self.nml['Atm_omgp3d'] = _entry(atm_omg, name='Single scattering Albedo')
        self.nml['Atm_apfp3d'] = _entry(atm_apf, name='Phase function')
        self.nml['Atm_np3d']   = _entry(1, name='Number of 3D atmospheric constituents')

    def add_mca_3d_atm(self, ext3d=None, omg3d=None, apf3d=None):

        """append one more 3-D scattering component given as (nx, ny, nz3) arrays"""

        if (ext3d is None) or (omg3d is None) or (apf3d is None):
            raise OSError('Error [mca_atm_3d]: Please provide an <ext3d>, <omg3d>, and <apf3d>.')
        for tag, a in (('ext3d', ext3d), ('omg3d', omg3d), ('apf3d', apf3d)):
            if isinstance(a, np.ndarray) and a.ndim != 3:
                raise ValueError('Error [mca_atm_3d]: <%s> should be in the dimension of (nx, ny, nz).' % tag)

        for key, a in (('Atm_extp3d', ext3d), ('Atm_omgp3d', omg3d), ('Atm_apfp3d', apf3d)):
            self.nml[key]['data'] = np.concatenate((self.nml[key]['data'], a[..., np.newaxis]), axis=-1)
        self.nml['Atm_np3d']['data'] += 1

    def gen_mca_3d_atm_file(self, fname):

        """side file: [tmpa3d][abst3d] then per component [ext][omg][apf]; float32 little-endian, x fastest, then y, then z"""

        if not self.quiet:
            print('Message [mca_atm_3d]: Creating 3D atm file <%s> for MCARaTS ...' % fname)
        fname = self._claim(fname)
        with open(fname, 'wb') as f:
            fortran_f4(self.nml['Atm_tmpa3d']['data']).tofile(f)
            fortran_f4(self.nml['Atm_abst3d']['data']).tofile(f)
            for i in range(self.nml['Atm_np3d']['data']):
                for key in ('Atm_extp3d', 'Atm_omgp3d', 'Atm_apfp3d'):
                    fortran_f4(self.nml[key]['data'][..., i]).tofile(f)
        self._done(fname)
