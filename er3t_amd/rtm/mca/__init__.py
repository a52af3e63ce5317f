"""
MI355X-native counterpart of the reference package `er3t.rtm.mca` (er3t/rtm/mca/__init__.py:1-8): same public names.
"""

from .mca_inp import mca_inp_file, mca_inp_read
from .mca_run import mca_run, rearrange_jobs
from .mca_atm import mca_atm_1d, mca_atm_3d
from .mca_sca import mca_sca
from .mca_sfc import mca_sfc_2d
from .mca_out import mca_out_raw, mca_out_ng, mca_out_write, read_flux_mca_out, read_radiance_mca_out
from .mcarats import mcarats_ng, cal_mca_azimuth, distribute_photon
from .util import func_ref_vs_cot, func_ref_vs_cot_multi_pixel
