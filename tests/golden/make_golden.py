"""
Generates the golden fixtures in this directory from the REFERENCE's own host layer.

Run in the build container only (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does: imports `er3t` from /root/reference (read-only; three optional packages that are not installed --
h5py, netCDF4, pyhdf -- are satisfied by empty stub modules so that `import er3t` succeeds; none of their functions
is called), feeds its adapters duck-typed inputs built by er3t_amd.synth, and records what the reference writes:

  golden_scalars.json   cal_mca_azimuth, distribute_photon, rearrange_jobs, cal_sol_fac, cal_r_twostream,
                        get_lay_index, cal_mol_ext, nice_array_str
  nml_*.txt             namelist files written by mcarats_ng(..., mp_mode='sh') (Wld_jseed line masked)
  side_*.bin            byte images of the 3-D atmosphere / phase-function / surface side files
  out_*.npz             mca_out_raw / read_flux_mca_out / read_radiance_mca_out results for synthetic out.bin+ctl
  adapters.npz          nml arrays produced by mca_atm_1d / mca_atm_3d / mca_sca / mca_sfc_2d

Only data (inputs and outputs) is stored here; no reference source text.
"""

import datetime
import importlib.abc
import importlib.machinery
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ('h5py', 'netCDF4', 'pyhdf')

    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference():
    os.environ.setdefault('MCARATS_V010_EXE', '/bin/true')      # mca_run only formats it into command strings
    sys.dont_write_bytecode = True
    sys.meta_path.append(_StubFinder())
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import er3t
    import er3t.rtm.mca
    return er3t


def mask_seed(text):
    return '\n'.join(' Wld_jseed       = <masked>' if l.startswith(' Wld_jseed') else l for l in text.split('\n'))


def relpaths(text, fdir):
    return text.replace(fdir, '<fdir>')


def write_ctl_bin(fname, arrays, names):
    """synthetic solver output in the shape mca_out_raw parses (er3t/rtm/mca/mca_out.py:48-103)"""
    nx, ny = arrays[0].shape[:2]
    with open(fname, 'wb') as f:
        for a in arrays:
            f.write(np.asarray(a, dtype='<f4').flatten(order='F').tobytes())
    with open(fname+'.ctl', 'w') as f:
        f.write('DSET ^%s\nTITLE synthetic\nUNDEF -9.99E33\n' % os.path.basename(fname))
        f.write('XDEF %d LINEAR 1 1\nYDEF %d LINEAR 1 1\nZDEF %d LINEAR 1 1\nTDEF 1 LINEAR 00:00Z01JAN2000 1mn\n' % (nx, ny, max(a.shape[2] for a in arrays)))
        f.write('VARS %d\n' % len(arrays))
        for a, n in zip(arrays, names):
            f.write('%s %d 99 %s description\n' % (n, a.shape[2], n))
        f.write('ENDVARS\n')


def main():
    er3t = import_reference()
    from er3t.rtm.mca import mcarats_ng, mca_out_ng, mca_out_raw, mca_atm_1d, mca_atm_3d, mca_sca, mca_sfc_2d
    from er3t.rtm.mca.mcarats import cal_mca_azimuth, distribute_photon
    from er3t.rtm.mca.mca_run import rearrange_jobs
    from er3t.rtm.mca.mca_out import read_flux_mca_out, read_radiance_mca_out
    import er3t.util as ru
    from er3t_amd import synth

    G = {}

    # ---------------------------------------------------------------- F1 scalars
    az = [0.0, 45.0, 90.0, 180.0, 270.0, 300.0, 360.0, -63.17, 400.0, -400.0, 270.5]
    G['cal_mca_azimuth'] = {'in': az, 'out': [float(cal_mca_azimuth(a)) for a in az]}
    w16 = synth.weights_16g()
    dp = []
    for n, w, br in ((1e8, w16, 0.05), (1e6, w16, 0.05), (1e5, w16, 0.05), (12345, w16, 0.1), (1e7, np.repeat(1.0/4, 4), 0.05),
                     (1000, np.array([0.7, 0.2, 0.1]), 0.0), (1e6, np.array([1.0]), 0.05)):
        dp.append({'N': n, 'w': list(map(float, w)), 'base_ratio': br,
                   'out': [int(v) for v in distribute_photon(n, np.asarray(w, dtype=np.float64), base_ratio=br)]})
    G['distribute_photon'] = dp
    rj = []
    for ncpu, wts in ((12, np.tile(distribute_photon(1e8, w16), 3)), (5, np.tile(distribute_photon(1e8, w16), 3)),
                      (3, np.array([5, 1, 9, 2, 2, 7, 3])), (4, np.array([10, 10, 10, 10, 10])), (8, np.tile(distribute_photon(1e6, w16), 2))):
        rj.append({'Ncpu': ncpu, 'w': [int(v) for v in wts], 'out': [int(v) for v in rearrange_jobs(ncpu, np.asarray(wts))]})
    G['rearrange_jobs'] = rj
    dates = ['2017-08-13', '2019-01-03', '2020-07-04', '2024-12-31', '2016-02-29']
    G['cal_sol_fac'] = {'in': dates, 'out': [float(ru.cal_sol_fac(datetime.datetime.strptime(d, '%Y-%m-%d'))) for d in dates]}
    tau = [0.0, 0.5, 1.0, 5.0, 10.0, 40.0, 100.0]
    G['cal_r_twostream'] = [{'tau': tau, 'a': a, 'g': g, 'mu': mu, 'out': [float(v) for v in ru.cal_r_twostream(np.array(tau), a=a, g=g, mu=mu)]}
                            for a, g, mu in ((0.0, 0.85, 1.0), (0.03, 0.85, 0.8660254), (0.3, 0.7, 0.5))]
    lay_ref = np.linspace(0.05, 19.95, 200)
    lay = np.array([0.65, 0.75, 0.85, 1.45])
    G['get_lay_index'] = {'lay': list(lay), 'lay_ref': list(lay_ref), 'out': [int(v) for v in ru.get_lay_index(lay, lay_ref)]}
    arrs = [np.arange(1, 14)*1.25, np.array([1.0e-5, 123456.789, -3.5]), np.linspace(0, 1, 6)]
    G['nice_array_str'] = [{'in': [float(v) for v in a], 'out': ru.nice_array_str(a)} for a in arrs]

    # ---------------------------------------------------------------- adapters
    from tests.golden import inputs as gin
    import er3t.rtm.mca as ref_mca
    import contextlib, io
    sink = io.StringIO()
    tmp = tempfile.mkdtemp(prefix='golden_')
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        inp = gin.make_inputs()
        rng = inp['rng']
        atm, ab = inp['atm'], inp['abs']
        ad = gin.build_adapters(ref_mca, inp, tmp)
        for src, dst in gin.SIDE_FILES.items():
            shutil.copy(os.path.join(tmp, src), os.path.join(HERE, dst))
        np.savez_compressed(os.path.join(HERE, 'adapters.npz'), **gin.adapter_arrays(ad))
        G['cal_mol_ext'] = {'wvl_um': 0.65, 'out': [float(v) for v in np.asarray(ad['a1'].nml[0]['Atm_ext1d(1:, 1)']['data'])*atm.lay['thickness']['data']*1000.0]}

        # ---------------------------------------------------------------- F2 namelists
        cases = gin.simulation_cases(ad['a1'], ad['a1b'], ad['a3'], ad['a3b'], ad['sca'], ad['s_l'], ad['s_b'], ab.coef['weight']['data'])
        objs = {}
        for name, kw in cases.items():
            fdir = '%s/%s' % (tmp, name)
            with contextlib.redirect_stdout(sink):
                m = mcarats_ng(fdir=fdir, Nrun=2, Ncpu=2, mp_mode='sh', overwrite=True, date=gin.DATE, quiet=True, **kw)
            objs[name] = m
            for ig in (0, 15):
                text = open(m.fnames_inp[1][ig]).read()
                with open(os.path.join(HERE, 'nml_%s_g%02d.txt' % (name, ig)), 'w') as f:
                    f.write(relpaths(mask_seed(text), tmp))
            G['mcarats_ng_%s' % name] = {'photons': [int(v) for v in m.photons], 'Nx': int(m.Nx), 'Ny': int(m.Ny), 'solver': m.solver,
                                         'target': m.target, 'photons_per_set': int(m.photons_per_set), 'np_mode': m.np_mode}

        # ---------------------------------------------------------------- F4 outputs
        for name, nz_out, nvar in (('rad_3d_hg', 1, 1), ('flux_1d', 21, 3), ('flux0_3d', 21, 3)):
            m = objs[name]
            nxo, nyo = int(m.Nx), int(m.Ny)
            for ir in range(m.Nrun):
                for ig in range(m.Ng):
                    arrs = [rng.uniform(0.0, 1.0, (nxo, nyo, nz_out, 1)).astype(np.float32) for _ in range(nvar)]
                    write_ctl_bin(m.fnames_out[ir][ig], arrs, ['v%d' % i for i in range(nvar)])
            raw = mca_out_raw(m.fnames_out[1][3])
            out = {'raw_%d' % i: d['data'] for i, d in enumerate(raw.data)}
            out['raw_names'] = np.array([d['name'] for d in raw.data])
            for mode in ('mean', 'all'):
                for squeeze in (True, False):
                    fn = read_radiance_mca_out if name.startswith('rad') else read_flux_mca_out
                    d = fn(m, ab, mode=mode, squeeze=squeeze)
                    for key in d:
                        out['%s_sq%d_%s' % (mode, int(squeeze), key)] = np.asarray(d[key]['data'])
                        if 'dims_info' in d[key]:
                            out['%s_sq%d_%s_dims' % (mode, int(squeeze), key)] = np.array(d[key]['dims_info'])
            # keep the synthetic solver outputs so the test can feed the same bytes to the build's reader
            for ir in range(m.Nrun):
                for ig in range(m.Ng):
                    out['bin_r%d_g%d' % (ir, ig)] = np.fromfile(m.fnames_out[ir][ig], dtype='<f4')
            out['ctl'] = np.array(open(m.fnames_out[0][0]+'.ctl').read())
            np.savez_compressed(os.path.join(HERE, 'out_%s.npz' % name), **out)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)

    with open(os.path.join(HERE, 'golden_scalars.json'), 'w') as f:
        json.dump(G, f, indent=1)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
