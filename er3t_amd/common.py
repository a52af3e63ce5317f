"""
Package-wide defaults (counterpart of the few lines of the reference's er3t/common.py that the rtm.mca path
uses: er3t/common.py:7-9,34-55).  The reference's data directories, satellite tables and external-solver
probes are outside the hot path and have no counterpart here.
"""

import datetime

import numpy as np

f_dtype = np.float32
i_dtype = np.int16

params = {
                 'wavelength': 650.0,
                       'date': datetime.datetime(2017, 8, 13),
         'solar_zenith_angle': 0.0,
        'solar_azimuth_angle': 0.0,
        'sensor_zenith_angle': 0.0,
       'sensor_azimuth_angle': 0.0,
            'sensor_altitude': 705000.0,
             'surface_albedo': 0.03,
                    'Nphoton': 1e8,
                       'Ncpu': 12,
                   'fdir_tmp': 'tmp-data/er3t_amd',
                 'output_tag': 'rtm-out_rad-3d',
                  'overwrite': True,
                    'verbose': False,
        }
