"""
Scattering adapter: phase-function object -> namelist entries + table side file.
Counterpart of the reference's `mca_sca` (er3t/rtm/mca/mca_sca.py:15-114).
"""

import os

import numpy as np

__all__ = ['mca_sca']


class mca_sca:

    """
    Input:
        pha_obj=: phase-function object with data['ang']['data'] (nang,) [deg] and data['pha']['data'] (nang, npf)
        fname=  : side file to write (default 'mca_sca.bin')

    Output:
        self.nml: Sca_npf, Sca_nskip, Sca_nanci, Sca_nangi, Sca_inpfile
        side file: [ang(nang)] then one [pha(nang)] block per table, float32 little-endian
    """

    ID = 'MCARaTS Scattering'

    def __init__(self, pha_obj=None, fname=None, overwrite=True, force=False, verbose=False, quiet=False):

        self.overwrite = overwrite
        self.verbose   = verbose
        self.quiet     = quiet

        if pha_obj is None:
            raise OSError('Error [mca_sca]: Please provide an \'pha\' object for <pha_obj>.')
        self.pha = pha_obj

        self.pre_mca_sca()

        if fname is None:
            fname = 'mca_sca.bin'

        if not self.overwrite:
            if (not os.path.exists(fname)) and (not force):
                self.gen_mca_sca_file(fname)
            self.nml['Sca_inpfile'] = {'data': fname}
        else:
            self.gen_mca_sca_file(fname)

    def pre_mca_sca(self, nskip=0, nanci=0):
        pha = self.pha.data['pha']['data']
        self.nml = {
            'Sca_npf'  : dict(data=pha.shape[1], name='Number of tabulated phase functions', units='N/A'),
            'Sca_nskip': dict(data=nskip, name='Number of phase functions to be skipped', units='N/A'),
            'Sca_nanci': dict(data=nanci, name='Number of ancillary data', units='N/A'),
            'Sca_nangi': dict(data=self.pha.data['ang']['data'].size, name='Number of angles', units='N/A'),
            }

    def gen_mca_sca_file(self, fname):
        fname = os.path.abspath(fname)
        self.nml['Sca_inpfile'] = {'data': fname}
        with open(fname, 'wb') as f:
            np.asarray(self.pha.data['ang']['data']).astype('<f4').tofile(f)
            np.asarray(self.pha.data['pha']['data']).astype('<f4').T.copy().tofile(f)     # one table after the other
        if not self.quiet:
            print('Message [mca_sca]: File <%s> is created.' % fname)
