"""a few jobs of func_ref_vs_cot under rocprofv3 --kernel-trace: tools/trace_ref_vs_cot.py [photons]"""
import os, sys, tempfile, shutil, datetime
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.rtm.mca.util import func_ref_vs_cot
from er3t_amd import synth
nph = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0e7
atm = synth.atm_synth(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
ab = synth.abs_synth(650.0, atm, Ng=16)
tmp = tempfile.mkdtemp()
func_ref_vs_cot(np.array([10.0, 20.0]), fdir=tmp+'/a', cer0=10.0, date=datetime.datetime(2017, 8, 13), wavelength=650.0, surface_albedo=0.03, solar_zenith_angle=30.0,
                solar_azimuth_angle=0.0, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, Nphoton=nph, atm0=atm, abs0=ab, pha0=None, overwrite=True)
shutil.rmtree(tmp, ignore_errors=True)
