"""
Input sets for the A/B run against a real MCARaTS install (tools/ab_mcarats.sh): job files exactly as `mcarats_ng` hands them to
the solver -- namelist + side files, 32 x 32 x 20 voxels, <= 1 MB per case -- written by THIS build's host layer, whose
namelists and side files are byte-identical to the reference's (tests/test_golden_host.py).  Fixed seeds, three runs per case.

    python tests/golden/ab/make_ab_inputs.py          # rewrites tests/golden/ab/<case>/

Each case isolates decisions this build took without the solver's source:
  c2_nadir     3-D cloud, nadir radiance                      the baseline; Rad_difr0/1 smoothing shows as per-pixel z-scores at cloud edges
  c2_slant     same, view zenith 45                           pixel registration at Rad_zref (a shift of the whole image if it differs)
  c3_flux      cloud + aerosol (np3d = 2), flux               the flux grid (levels, columns), direct / diffuse split, Flx_mflx = 3
  c4_absorb    nadir radiance, strongly absorbing g, sza 60   gas absorption through the collision weight against path-length attenuation
  c5_lsrt      view zenith 60 + LSRT surface map              BRDF sampling and the surface local estimate
  c6_sea       nadir + Cox-Munk surface map (jsfc = 2)        the diffuse-specular mixture this build restates from the literature
  c7_allsky    all-sky camera on the ground                   Rad_mrkind = 1: camera frame, pixel map, 1/r^2 regularisation
"""
import contextlib
import datetime
import io
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

import er3t_amd.rtm.mca as mca                                        # noqa: E402
from er3t_amd.synth import atm_synth, abs_synth, cld_synth, sfc_lsrt_synth, sfc_dsm_synth      # noqa: E402
from er3t_amd.rtm.mca.mca_inp import mca_inp_file                     # noqa: E402

DATE = datetime.datetime(2017, 8, 13)
NX = NY = 32
NZC = 20


def main():
    sink = io.StringIO()
    levels = np.concatenate([np.arange(0, NZC+1)*0.1, np.arange(3, 21)*1.0])           # 20 layers of 100 m, then 1 km layers to 20 km
    atm = atm_synth(levels)
    ab = abs_synth(650.0, atm, Ng=4)
    cld = cld_synth(atm, nx=NX, ny=NY, nz=NZC, z_base=0.5, z_top=1.5, cot_mean=8.0, seed=11)
    cases = {
        'c2_nadir': dict(target='radiance', sza=30.0, saa=45.0, vza=0.0, vaa=0.0, ig=0),
        'c2_slant': dict(target='radiance', sza=30.0, saa=45.0, vza=45.0, vaa=90.0, ig=0),
        'c3_flux': dict(target='flux', sza=30.0, saa=45.0, aerosol=True, ig=1),
        'c4_absorb': dict(target='radiance', sza=60.0, saa=200.0, vza=0.0, vaa=0.0, ig=3),
        'c5_lsrt': dict(target='radiance', sza=40.0, saa=120.0, vza=60.0, vaa=300.0, ig=0, sfc='lsrt'),
        'c6_sea': dict(target='radiance', sza=35.0, saa=90.0, vza=0.0, vaa=0.0, ig=0, sfc='dsm'),
        'c7_allsky': dict(target='radiance', sza=30.0, saa=45.0, vza=180.0, vaa=0.0, ig=0, allsky=True),
    }
    for name, kw in cases.items():
        fdir = os.path.join(HERE, name)
        shutil.rmtree(fdir, ignore_errors=True)
        os.makedirs(fdir)
        with contextlib.redirect_stdout(sink):
            a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
            a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, fname=os.path.join(fdir, 'atm3d.bin'), quiet=True)
            if kw.get('aerosol'):
                aer = np.zeros((NX, NY, NZC)); aer[:, :, 0] = 1.2e-4; aer[:, :, 1] = 0.8e-4
                a3.add_mca_3d_atm(ext3d=aer, omg3d=np.full_like(aer, 0.85), apf3d=np.full_like(aer, 0.6))
                a3.gen_mca_3d_atm_file(os.path.join(fdir, 'atm3d.bin'))
            sfc = 0.05
            if kw.get('sfc') == 'lsrt':
                sfc = mca.mca_sfc_2d(atm_obj=atm, sfc_obj=sfc_lsrt_synth(NX, NY), fname=os.path.join(fdir, 'sfc.bin'), quiet=True)
            elif kw.get('sfc') == 'dsm':
                sfc = mca.mca_sfc_2d(atm_obj=atm, sfc_obj=sfc_dsm_synth(NX, NY), fname=os.path.join(fdir, 'sfc.bin'), quiet=True)
            extra = {}
            if kw['target'] == 'radiance':
                extra = dict(sensor_zenith_angle=kw['vza'], sensor_azimuth_angle=kw['vaa'],
                             sensor_altitude=0.0 if kw.get('allsky') else 705000.0)
                if kw.get('allsky'):
                    extra.update(sensor_type='all-sky', sensor_xpos=0.4, sensor_ypos=0.6)
            # one g (the ig-th of four) per case: the job is what a single solver process gets; mp_mode='sh' runs nothing
            a1g = _one_g(a1, kw['ig'])
            m = mca.mcarats_ng(atm_1ds=[a1g], atm_3ds=[a3], Ng=1, weights=np.array([1.0]), target=kw['target'], surface_albedo=sfc,
                               solar_zenith_angle=kw['sza'], solar_azimuth_angle=kw['saa'], fdir=fdir, Nrun=3, photons=1e6, solver='3D',
                               mp_mode='sh', overwrite=True, date=DATE, quiet=True, **extra)
        # fixed seeds instead of the clock
        for ir in range(3):
            m.nml[0]['Wld_jseed'] = 1000 + 17*ir
            mca_inp_file(m.fnames_inp[ir][0], m.nml[0])
        for f in os.listdir(fdir):
            if f.endswith('.sh') or f.endswith('.out.bin') or f.endswith('.ctl'):
                os.remove(os.path.join(fdir, f))
        size = sum(os.path.getsize(os.path.join(fdir, f)) for f in os.listdir(fdir))
        print('%-10s %d files, %.0f KB' % (name, len(os.listdir(fdir)), size/1024.0))


def _one_g(a1, ig):
    """an mca_atm_1d-like object holding only the ig-th g of <a1>"""
    import copy
    o = copy.copy(a1)
    o.nml = {0: a1.nml[ig]}
    o.Ng = 1
    return o


if __name__ == '__main__':
    main()
