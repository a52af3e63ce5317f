"""
2-D surface for the solver (counterpart of the reference's `mca_sfc_2d`, er3t/rtm/mca/mca_sfc.py:16-164).
"""

import copy

import numpy as np

import er3t_amd.common
from er3t_amd.rtm.mca._adapter import SideFileAdapter, fortran_f4

__all__ = ['mca_sfc_2d']

# surface model ids of the solver and how a surface object selects them (mca_sfc.py:94-128)
LAMBERT, DSM, LSRT = 1, 2, 4


def _model_of(name, data):
    name = name.lower()
    if ('lambertian' in name) and (np.squeeze(data).ndim == 2):
        return LAMBERT
    if ('brdf-lsrt' in name) or (data.shape[-1] == 3):
        return LSRT
    if ('cox-munk' in name) or (data.shape[-1] == 5):
        return DSM
    msg = '\nError [mca_sfc_2d]: Cannot determine surface type - currently only supports Lambertian surface and LSRT BRDF surface (e.g., MCD43A1).'
    raise OSError(msg)


class mca_sfc_2d(SideFileAdapter):

    """
    mca_sfc_2d(atm_obj=, sfc_obj=, fname='mca_sfc_2d.bin', overwrite=True, force=False, verbose=False, quiet=False)

    sfc_obj.data['sfc']['data'] : (nx, ny) albedo with 'lambertian' in data['sfc']['name'];
                                  (nx, ny, 3) = (fiso, fgeo, fvol) for the LSRT BRDF; (nx, ny, 5) for Cox-Munk / DSM
    sfc_obj.Nx, .Ny, .data['nx'], .data['ny']

    -> self.nml : Sfc_nxb, Sfc_nyb, Sfc_tmps2d (nx, ny) zeros, Sfc_jsfc2d (nx, ny) int16 model id,
                  Sfc_psfc2d (nx, ny, 5) parameters (albedo clipped to [0, 1] for Lambert), Sfc_inpfile
    -> side file: the three arrays one after the other, float32 little-endian, x fastest (mca_sfc.py:136-146)
    """

    ID = 'MCARaTS 2D Surface'
    key_inpfile = 'Sfc_inpfile'
    default_fname = 'mca_sfc_2d.bin'
    tag = 'mca_sfc_2d'

    def __init__(self, atm_obj=None, sfc_obj=None, fname=None, overwrite=True, force=False, verbose=False, quiet=False):
        if atm_obj is None:
            raise OSError('\nError [mca_sfc_2d]: Please provide an <atm> object for <atm_obj>.')
        if sfc_obj is None:
            raise OSError('\nError [mca_sfc_2d]: Please provide an <sfc> object for <sfc_obj>.')
        self.atm, self.sfc, self.overwrite, self.verbose, self.quiet = atm_obj, sfc_obj, overwrite, verbose, quiet
        self.pre_mca_2d_sfc()
        self._settle_file(fname, overwrite, force, self.gen_mca_2d_sfc_file)

    def pre_mca_2d_sfc(self):
        f_dtype = er3t_amd.common.f_dtype
        shape = (self.sfc.Nx, self.sfc.Ny)
        data = self.sfc.data['sfc']['data']
        model = _model_of(self.sfc.data['sfc']['name'], data)

        if model == DSM:
            par = data
        else:
            par = np.zeros(shape+(5,), dtype=f_dtype)
            if model == LAMBERT:
                par[:, :, 0] = np.clip(np.squeeze(data), 0.0, 1.0)
            else:
                par[:, :, :3] = data[:, :, :3]

        self.nml = {'Sfc_nxb': copy.deepcopy(self.sfc.data['nx']), 'Sfc_nyb': copy.deepcopy(self.sfc.data['ny']),
                    'Sfc_tmps2d': dict(data=np.zeros(shape, dtype=f_dtype), name='Temperature anomalies', units='K'),
                    'Sfc_jsfc2d': dict(data=np.full(shape, model, dtype=np.int16), name='Surface distribution type', units='N/A'),
                    'Sfc_psfc2d': dict(data=par, name='Surface distribution parameters', units='N/A')}

    def gen_mca_2d_sfc_file(self, fname):
        fname = self._claim(fname)
        with open(fname, 'wb') as f:
            for key in ('Sfc_tmps2d', 'Sfc_jsfc2d', 'Sfc_psfc2d'):
                fortran_f4(self.nml[key]['data']).tofile(f)
        self._done(fname)
