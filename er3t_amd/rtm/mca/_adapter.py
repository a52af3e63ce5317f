"""
Shared behaviour of the adapters that own a binary side file (3-D atmosphere, phase functions, 2-D surface):
when to (re)write the file and how the `*_inpfile` namelist entry is kept.

Semantics of the reference's adapters (er3t/rtm/mca/mca_atm.py:220-228, mca_sca.py:60-69, mca_sfc.py:69-78):
`overwrite=True` always writes; `overwrite=False` writes only when the file is missing (unless `force`), and in
both cases the namelist entry ends up pointing at the file.
"""

import os

import numpy as np

__all__ = ['SideFileAdapter', 'fortran_f4']


def fortran_f4(a):
    """little-endian float32 in Fortran element order (first index fastest): the byte order of every side file"""
    return np.asarray(a).astype('<f4').ravel(order='F')


class SideFileAdapter:

    key_inpfile = None       # namelist key naming the side file, e.g. 'Atm_inpfile'
    default_fname = None     # file name used when the caller gives none
    tag = 'adapter'          # name used in messages

    def _settle_file(self, fname, overwrite, force, writer):
        if fname is None:
            fname = self.default_fname
        if overwrite or ((not os.path.exists(fname)) and (not force)):
            writer(fname)
        else:
            self.nml[self.key_inpfile] = {'data': fname}

    def _claim(self, fname):
        fname = os.path.abspath(fname)
        self.nml[self.key_inpfile] = {'data': fname}
        return fname

    def _done(self, fname):
        if not self.quiet:
            print('Message [%s]: File <%s> is created.' % (self.tag, fname))
