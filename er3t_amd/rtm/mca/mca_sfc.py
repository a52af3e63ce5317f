"""
Surface adapter: 2-D surface object -> namelist entries + surface side file.
Counterpart of the reference's `mca_sfc_2d` (er3t/rtm/mca/mca_sfc.py:16-164).
"""

import copy
import os

import numpy as np

import er3t_amd.common

__all__ = ['mca_sfc_2d']


class mca_sfc_2d:

    """
    Input:
        atm_obj=: atmosphere object (kept for interface parity)
        sfc_obj=: surface object; sfc.data['sfc']['data'] is (nx, ny) albedo ('lambertian' in its name),
                  (nx, ny, 3) = (fiso, fgeo, fvol) for the LSRT BRDF, or (nx, ny, 5) for Cox-Munk/DSM
        fname=  : side file to write (default 'mca_sfc_2d.bin')

    Output:
        self.nml: Sfc_nxb, Sfc_nyb, Sfc_tmps2d (nx, ny), Sfc_jsfc2d (nx, ny; 1 Lambert, 4 LSRT, 2 DSM), Sfc_psfc2d (nx, ny, 5),
                  Sfc_inpfile
        side file: [tmps2d][jsfc2d as float][psfc2d], float32 little-endian, x fastest
    """

    ID = 'MCARaTS 2D Surface'

    def __init__(self, atm_obj=None, sfc_obj=None, fname=None, overwrite=True, force=False, verbose=False, quiet=False):

        self.overwrite = overwrite
        self.verbose   = verbose
        self.quiet     = quiet

        if atm_obj is None:
            raise OSError('\nError [mca_sfc_2d]: Please provide an <atm> object for <atm_obj>.')
        if sfc_obj is None:
            raise OSError('\nError [mca_sfc_2d]: Please provide an <sfc> object for <sfc_obj>.')
        self.atm = atm_obj
        self.sfc = sfc_obj

        self.pre_mca_2d_sfc()

        if fname is None:
            fname = 'mca_sfc_2d.bin'

        if not self.overwrite:
            if (not os.path.exists(fname)) and (not force):
                self.gen_mca_2d_sfc_file(fname)
            self.nml['Sfc_inpfile'] = {'data': fname}
        else:
            self.gen_mca_2d_sfc_file(fname)

    def pre_mca_2d_sfc(self):

        f_dtype = er3t_amd.common.f_dtype
        nx, ny = self.sfc.Nx, self.sfc.Ny
        data = self.sfc.data['sfc']['data']
        name = self.sfc.data['sfc']['name'].lower()

        self.nml = {'Sfc_nxb': copy.deepcopy(self.sfc.data['nx']), 'Sfc_nyb': copy.deepcopy(self.sfc.data['ny'])}

        if ('lambertian' in name) and (np.squeeze(data).ndim == 2):
            model = 1
            psfc = np.zeros((nx, ny, 5), dtype=f_dtype)
            psfc[:, :, 0] = np.clip(np.squeeze(data), 0.0, 1.0)
        elif ('brdf-lsrt' in name) or (data.shape[-1] == 3):
            model = 4
            psfc = np.zeros((nx, ny, 5), dtype=f_dtype)
            psfc[:, :, :3] = data[:, :, :3]
        elif ('cox-munk' in name) or (data.shape[-1] == 5):
            model = 2
            psfc = data
        else:
            msg = '\nError [mca_sfc_2d]: Cannot determine surface type - currently only supports Lambertian surface and LSRT BRDF surface (e.g., MCD43A1).'
            raise OSError(msg)

        self.nml['Sfc_tmps2d'] = dict(data=np.zeros((nx, ny), dtype=f_dtype), name='Temperature anomalies', units='K')
        self.nml['Sfc_jsfc2d'] = dict(data=np.full((nx, ny), model, dtype=np.int16), name='Surface distribution type', units='N/A')
        self.nml['Sfc_psfc2d'] = dict(data=psfc, name='Surface distribution parameters', units='N/A')

    def gen_mca_2d_sfc_file(self, fname):
        fname = os.path.abspath(fname)
        self.nml['Sfc_inpfile'] = {'data': fname}
        with open(fname, 'wb') as f:
            for key in ('Sfc_tmps2d', 'Sfc_jsfc2d', 'Sfc_psfc2d'):
                np.asarray(self.nml[key]['data']).astype('<f4').ravel(order='F').tofile(f)
        if not self.quiet:
            print('Message [mca_sfc_2d]: File <%s> is created.' % fname)
