"""
Solver outputs: reader of one out.bin (+ .ctl), and the reduction over g and runs.

Counterparts of the reference's `mca_out_raw` and `mca_out_ng` / `read_flux_mca_out` / `read_radiance_mca_out`
(er3t/rtm/mca/mca_out.py:16-103, 107-505), plus `mca_out_write`, which produces the files those readers parse
(the reference never writes them: its solver executable does).
"""

import os

import numpy as np

from er3t_amd.util import cal_sol_fac

__all__ = ['mca_out_raw', 'mca_out_ng', 'mca_out_write', 'read_flux_mca_out', 'read_radiance_mca_out', 'read_heating_mca_out']


def mca_out_write(fname_bin, variables):

    """
    Write a solver output: <fname_bin> = float32 little-endian, every variable (nx, ny, nz, nt) in Fortran order
    one after the other; <fname_bin>.ctl = GrADS descriptor with the XDEF / YDEF / TDEF / VARS lines
    `mca_out_raw` reads (er3t/rtm/mca/mca_out.py:48-91).

    variables: list of (name, description, array) with array (nx, ny, nz) or (nx, ny, nz, nt)
    """

    arrays = []
    for name, desc, a in variables:
        a = np.asarray(a, dtype='<f4')
        if a.ndim == 3:
            a = a[..., np.newaxis]
        arrays.append((name, desc, a))
    nx, ny, _, nt = arrays[0][2].shape
    nzmax = max(a.shape[2] for _, _, a in arrays)

    with open(fname_bin, 'wb') as f:
        for _, _, a in arrays:
            a.ravel(order='F').tofile(f)

    lines = ['DSET ^%s' % os.path.basename(fname_bin),
             'TITLE er3t_amd solver output',
             'OPTIONS LITTLE_ENDIAN',
             'UNDEF -9.99E33',
             'XDEF %d LINEAR 1 1' % nx,
             'YDEF %d LINEAR 1 1' % ny,
             'ZDEF %d LINEAR 1 1' % nzmax,
             'TDEF %d LINEAR 00:00Z01JAN2000 1mn' % nt,
             'VARS %d' % len(arrays)]
    lines += ['%s %d 99 %s' % (name, a.shape[2], desc) for name, desc, a in arrays]
    lines += ['ENDVARS']
    with open(fname_bin+'.ctl', 'w') as f:
        f.write('\n'.join(lines)+'\n')


class mca_out_raw:

    """
    Read one solver output file using its .ctl descriptor.

    self.data: list (one entry per variable) of {'name', 'dims' [Nx, Ny, Nz, Nt], 'dims_info', 'data'}
    """

    def __init__(self, fname_bin):

        if not os.path.isfile(fname_bin):
            raise OSError('Error [mca_out_raw]: Cannot find <%s>.' % fname_bin)
        fname_ctl = fname_bin + '.ctl'
        if not os.path.isfile(fname_ctl):
            raise OSError('Error [mca_out_raw]: Cannot find <%s>.' % fname_ctl)

        self.fname_bin = fname_bin
        self.fname_ctl = fname_ctl
        self.data = []
        self.read_ctl()
        self.read_bin()

    def read_ctl(self):

        with open(self.fname_ctl, 'r') as f:
            lines = [l.strip() for l in f.readlines()]

        Nx = Ny = Nt = None
        start = 0
        for i, line in enumerate(lines):
            if 'XDEF' in line:
                Nx = int(line.replace('XDEF', '').replace('LINEAR', '').split()[0])
            elif 'YDEF' in line:
                Ny = int(line.replace('YDEF', '').replace('LINEAR', '').split()[0])
            elif 'TDEF' in line:
                Nt = int(line.replace('TDEF', '').split()[0])
            elif 'VARS' in line and 'ENDVARS' not in line:
                self.Nvar = int(line.replace('VARS', '').strip())
                for words in (l.split() for l in lines[i+1:i+1+self.Nvar]):
                    Nz = int(words[1])
                    size = Nx*Ny*Nz*Nt
                    self.data.append({'name': '%s (%s)' % (words[0], ' '.join(words[3:])), 'dims': [Nx, Ny, Nz, Nt],
                                      'dims_info': ['Nx', 'Ny', 'Nz', 'Nt'], 'Index_Start': start, 'Index_End': start+size})
                    start += size

    def read_bin(self, dtype='<f4'):
        raw = np.fromfile(self.fname_bin, dtype=dtype)
        for info in self.data:
            info['data'] = raw[info['Index_Start']:info['Index_End']].reshape(info['dims'], order='F')


# ----------------------------------------------------------------------------------------------
def g_factors(mca_obj, abs_obj, Nz):

    """
    factor[iz, ig] = sol_fac * solar[ig]*weight[ig]*slit[iz, ig] / sum_g(weight*slit[iz]) in float32, the slit
    function of the top level taken from the layer below it (reference: er3t/rtm/mca/mca_out.py:313-328)
    """

    zz = np.arange(Nz)
    if Nz > 1:
        zz[-1] = zz[-2]
    sol_fac = cal_sol_fac(mca_obj.date)
    weight = abs_obj.coef['weight']['data']
    solar  = abs_obj.coef['solar']['data']
    slit   = abs_obj.coef['slit_func']['data']
    factors = np.zeros((Nz, mca_obj.Ng), dtype=np.float32)
    for iz in range(Nz):
        norm = np.float32(sol_fac/(weight*slit[zz[iz], :]).sum())
        for ig in range(mca_obj.Ng):
            factors[iz, ig] = norm*solar[ig]*weight[ig]*slit[zz[iz], ig]
    toa = np.sum(sol_fac*solar*weight)
    return factors, toa


def _accumulate(mca_obj, abs_obj, nvar, squeeze):

    """sum over g of factor * variable for every run: list of nvar arrays (dims..., Nrun), plus dims_info and toa"""

    fused = getattr(mca_obj, 'fused', None)
    if fused is not None:
        return _from_fused(fused, nvar, squeeze)

    out0 = mca_out_raw(mca_obj.fnames_out[0][0])
    dims_info = list(out0.data[0]['dims_info'])
    dims = list(out0.data[0]['dims'])
    Nz = dims[dims_info.index('Nz')]
    if nvar == 1 and Nz > 1 and getattr(mca_obj, 'Nview', 1) > 1:
        # (a radiance file of several views -- mcarats_ng with sequences of sensor angles, not in the reference --: its third axis counts
        #  views, not levels; every view is scaled like the reference's one view)
        factors, toa = g_factors(mca_obj, abs_obj, 1)
        factors = np.repeat(factors, Nz, axis=0)
    else:
        factors, toa = g_factors(mca_obj, abs_obj, Nz)

    if squeeze:
        dims_info = [dims_info[i] for i in range(len(dims)) if dims[i] > 1]
        dims = [n for n in dims if n > 1]
    dims_info += ['Nr']
    dims += [mca_obj.Nrun]

    # The same float32 operations in the same order as the reference's loop (sum[..., ir] += raw*factor, g after g), but a run is
    # summed in an array of its own, laid out like the file (Fortran order): element after element in memory instead of every
    # Nrun-th float of a C-ordered array -- the reader of 48 flux files of 13.6 MB took longer than the simulation.
    sums = [np.zeros(dims, dtype=np.float32) for _ in range(nvar)]

    def one_run(ir):
        run = [np.zeros(dims[:-1], dtype=np.float32, order='F') for _ in range(nvar)]
        for ig in range(mca_obj.Ng):
            raw = mca_out_raw(mca_obj.fnames_out[ir][ig])
            fac = factors[:, ig][None, None, :, None]
            for iv in range(nvar):
                scaled = raw.data[iv]['data']*fac
                run[iv] += np.squeeze(scaled) if squeeze else scaled
        for iv in range(nvar):
            sums[iv][..., ir] = run[iv]

    # (the runs side by side, a thread each -- reading and the array operations release the interpreter lock --; inside a run g after g,
    #  as the float32 sum demands.  Reading further ahead or the variables of a file side by side bought nothing: tried)
    if mca_obj.Nrun > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(mca_obj.Nrun, 8)) as pool:
            list(pool.map(one_run, range(mca_obj.Nrun)))
    else:
        one_run(0)
    return sums, dims_info, toa


def _from_fused(fused, nvar, squeeze):

    """per-run fields gathered on the device by `mcarats_ng(abs_obj=...)`, brought to the file route's shapes"""

    runs = fused['flux']['runs'] if nvar == 3 else fused['rad']['runs'][None]      # (nvar, Nz, Ny, Nx, Nr)
    dims_info = ['Nx', 'Ny', 'Nz', 'Nt', 'Nr']
    sums = []
    for iv in range(nvar):
        a = np.transpose(runs[iv], (2, 1, 0, 3))[:, :, :, None, :]                   # (Nx, Ny, Nz, Nt=1, Nr)
        sums.append(a)
    if squeeze:
        keep = [i for i, n in enumerate(sums[0].shape[:-1]) if n > 1] + [4]
        dims_info = [dims_info[i] for i in keep]
        sums = [a.reshape([a.shape[i] for i in keep]) for a in sums]
    return [np.ascontiguousarray(a) for a in sums], dims_info, fused['toa']


def read_flux_mca_out(mca_obj, abs_obj, mode='mean', squeeze=True):

    """
    Fluxes summed over g, per run ('all') or mean and population standard deviation over runs ('mean').
    Output variables of the solver, in order: direct-down, total-down, up (reference: mca_out.py:350-352).
    keys: f_up, f_down, f_down_direct, f_down_diffuse (+ *_std for 'mean'), toa, N_photon, N_run
    """

    mode = mode.lower()
    (f_down_direct, f_down, f_up), dims_info, toa = _accumulate(mca_obj, abs_obj, 3, squeeze)
    fields = [('f_down', f_down, 'Global downwelling flux'), ('f_up', f_up, 'Global upwelling flux'),
              ('f_down_direct', f_down_direct, 'Direct downwelling flux'),
              ('f_down_diffuse', f_down-f_down_direct, 'Diffuse downwelling flux')]

    data = {'toa': {'data': toa, 'name': 'TOA without SZA', 'units': 'W/m^2/nm'}}
    if mode == 'all':
        for key, arr, name in fields:
            data[key] = {'data': arr, 'name': name, 'units': 'W/m^2/nm', 'dims_info': dims_info}
    elif mode == 'mean':
        # (numpy's own reductions, the reference's: eight of them over millions of cells, side by side -- they release the interpreter lock)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=8) as pool:
            means = list(pool.map(lambda f: np.mean(f[1], axis=-1), fields))
            stds = list(pool.map(lambda f: np.std(f[1], axis=-1), fields))
        for (key, arr, name), v in zip(fields, means):
            data[key] = {'data': v, 'name': name+' (mean)', 'units': 'W/m^2/nm', 'dims_info': dims_info[:-1]}
        for (key, arr, name), v in zip(fields, stds):
            data[key+'_std'] = {'data': v, 'name': name+' (standard deviation)', 'units': 'W/m^2/nm', 'dims_info': dims_info[:-1]}
    else:
        raise OSError('Error [read_flux_mca_out]: Do not support <mode=%s>.' % mode)
    data['N_photon'] = {'data': mca_obj.photons, 'name': 'Number of photons', 'units': 'N/A'}
    data['N_run']    = {'data': mca_obj.Nrun, 'name': 'Number of runs', 'units': 'N/A'}
    return data


def read_heating_mca_out(mca_obj, abs_obj, mode='mean', squeeze=True):

    """
    target='heating rate' (er3t/rtm/mca/mcarats.py:279-283: Flx_mflx = 3, Flx_mhrt = 1): the fluxes of `read_flux_mca_out` plus
    `heating_rate` (+ `heating_rate_std`): the fourth variable of every job's output, absorbed power per unit volume on the LAYER
    grid (Nx, Ny, Nz layers), scaled per g like the fluxes (the factor of a layer is that of its lower level: the slit function is
    given per layer, mca_out.py:313-328) and summed over g.  Units: W/m^3/nm; divided by air density x c_p: K/s.
    (The reference's reader has no such branch, er3t/rtm/mca/mca_out.py:202-205: its `mca_out_ng` ends without data for this target.)
    """

    data = read_flux_mca_out(mca_obj, abs_obj, mode=mode, squeeze=squeeze)
    out0 = mca_out_raw(mca_obj.fnames_out[0][0])
    if len(out0.data) < 4:
        raise OSError('Error [read_heating_mca_out]: <%s> holds no heating-rate variable.' % mca_obj.fnames_out[0][0])
    dims = list(out0.data[3]['dims']); dims_info = list(out0.data[3]['dims_info'])
    nlay = dims[dims_info.index('Nz')]
    factors, _ = g_factors(mca_obj, abs_obj, nlay+1)
    if squeeze:
        dims_info = [dims_info[i] for i in range(len(dims)) if dims[i] > 1]
        dims = [n for n in dims if n > 1]
    hr = np.zeros(dims+[mca_obj.Nrun], dtype=np.float32)
    for ir in range(mca_obj.Nrun):
        run = np.zeros(dims, dtype=np.float32, order='F')
        for ig in range(mca_obj.Ng):
            scaled = mca_out_raw(mca_obj.fnames_out[ir][ig]).data[3]['data']*factors[:nlay, ig][None, None, :, None]
            run += np.squeeze(scaled) if squeeze else scaled
        hr[..., ir] = run
    dims_info = dims_info+['Nr']
    if mode.lower() == 'all':
        data['heating_rate'] = {'data': hr, 'name': 'Absorbed power per unit volume', 'units': 'W/m^3/nm', 'dims_info': dims_info}
    else:
        data['heating_rate'] = {'data': np.mean(hr, axis=-1), 'name': 'Absorbed power per unit volume (mean)', 'units': 'W/m^3/nm', 'dims_info': dims_info[:-1]}
        data['heating_rate_std'] = {'data': np.std(hr, axis=-1), 'name': 'Absorbed power per unit volume (standard deviation)', 'units': 'W/m^3/nm', 'dims_info': dims_info[:-1]}
    return data


def read_radiance_mca_out(mca_obj, abs_obj, mode='mean', squeeze=True):

    """
    Radiance summed over g, per run ('all') or mean and population standard deviation over runs ('mean').
    keys: rad (+ rad_std for 'mean'), toa, N_photon, N_run      (reference: mca_out.py:412-505)
    """

    mode = mode.lower()
    (rad,), dims_info, toa = _accumulate(mca_obj, abs_obj, 1, squeeze)

    data = {'toa': {'data': toa, 'name': 'TOA without SZA', 'units': 'W/m^2/nm'}}
    if mode == 'all':
        data['rad'] = {'data': rad, 'name': 'Radiance', 'units': 'W/m^2/nm/sr', 'dims_info': dims_info}
    elif mode == 'mean':
        data['rad']     = {'data': np.mean(rad, axis=-1), 'name': 'Radiance (mean)', 'units': 'W/m^2/nm/sr', 'dims_info': dims_info[:-1]}
        data['rad_std'] = {'data': np.std(rad, axis=-1), 'name': 'Radiance (standard deviation)', 'units': 'W/m^2/nm/sr', 'dims_info': dims_info[:-1]}
    else:
        raise OSError('Error [read_radiance_mca_out]: Do not support <mode=%s>.' % mode)
    data['N_photon'] = {'data': mca_obj.photons, 'name': 'Number of photons', 'units': 'N/A'}
    data['N_run']    = {'data': mca_obj.Nrun, 'name': 'Number of runs', 'units': 'N/A'}
    return data


class mca_out_ng:

    """
    Collect the results of a `mcarats_ng` simulation.

    Input:
        fname=    : result cache to write/read (HDF5 when h5py is installed and the name ends in .h5/.hdf5, else .npz)
        mca_obj=  : the mcarats_ng object;   abs_obj=: the absorption object (coef['weight'|'solar'|'slit_func'])
        mode=     : 'mean' | 'all';   squeeze=: drop axes of length 1;   overwrite=: recompute even if <fname> exists

    Output:
        self.data[key]['data'|'name'|'units'|'dims_info'], keys as returned by read_flux_mca_out / read_radiance_mca_out

    Same three call shapes as the reference (er3t/rtm/mca/mca_out.py:160-177): cache only, objects + cache, objects only.
    """

    def __init__(self, fname=None, mca_obj=None, abs_obj=None, mode='mean', overwrite=False, squeeze=True,
                 quiet=False, verbose=False):

        self.mode      = mode
        self.quiet     = quiet
        self.verbose   = verbose
        self.overwrite = overwrite
        self.squeeze   = squeeze
        self.fname     = fname
        self.mca       = mca_obj
        self.abs       = abs_obj

        have_objs = (mca_obj is not None) and (abs_obj is not None)
        if (fname is not None) and os.path.exists(fname) and (not overwrite):
            self.load()
        elif have_objs and (fname is not None):
            self.run()
            self.dump()
        elif have_objs:
            self.run()
        else:
            raise OSError('Error [mca_out_ng]: Please provide both <mca_obj> and <abs_obj> to proceed.')

    def run(self):
        if self.verbose:
            print('Message [mca_out_ng]: Reading <%s> ...' % self.mca.target.lower())
        if self.mca.target in ['flux', 'flux0']:
            self.data = read_flux_mca_out(self.mca, self.abs, mode=self.mode, squeeze=self.squeeze)
        elif self.mca.target == 'radiance':
            self.data = read_radiance_mca_out(self.mca, self.abs, mode=self.mode, squeeze=self.squeeze)
        elif self.mca.target == 'heating rate':
            self.data = read_heating_mca_out(self.mca, self.abs, mode=self.mode, squeeze=self.squeeze)
        else:
            raise OSError('Error [mca_out_ng]: Cannot read results of <target=%s>.' % self.mca.target)

    # ---- result cache ------------------------------------------------------------------------
    @staticmethod
    def _is_hdf5(fname):
        return fname.lower().endswith(('.h5', '.hdf5', '.hdf'))

    def dump(self):
        if not self.quiet:
            print('Message [mca_out_ng]: Saving <%s> into <%s> ...' % (self.mca.target.lower(), self.fname))
        mode = self.mode.lower()
        if self._is_hdf5(self.fname):
            try:
                import h5py
            except ImportError:
                raise OSError('Error [mca_out_ng]: <%s> needs the h5py package; use a .npz file name instead.' % self.fname)
            with h5py.File(self.fname, 'w') as f:
                g = f.create_group(mode)
                for key, item in self.data.items():
                    if isinstance(item['data'], np.ndarray):
                        g.create_dataset(key, data=item['data'], compression='gzip', compression_opts=9, chunks=True)
                    else:
                        g[key] = item['data']
                    for k0, v0 in item.items():
                        if k0 != 'data':
                            g[key].attrs[k0] = np.bytes_(str(v0)) if k0 == 'dims_info' else v0
        else:
            flat = {}
            for key, item in self.data.items():
                for k0, v0 in item.items():
                    flat['%s/%s/%s' % (mode, key, k0)] = np.asarray(v0)
            with open(self.fname, 'wb') as f:
                np.savez_compressed(f, **flat)

    def load(self):
        if self.verbose:
            print('Message [mca_out_ng]: Reading from <%s> ...' % self.fname)
        self.data = {}
        if self._is_hdf5(self.fname):
            try:
                import h5py
            except ImportError:
                raise OSError('Error [mca_out_ng]: <%s> needs the h5py package.' % self.fname)
            with h5py.File(self.fname, 'r') as f:
                g = f[self.mode]
                for key in g.keys():
                    self.data[key] = {'data': g[key][...]}
                    for k0, v0 in g[key].attrs.items():
                        self.data[key][k0] = v0
        else:
            with np.load(self.fname, allow_pickle=False) as z:
                for full in z.files:
                    mode, key, k0 = full.split('/')
                    if mode != self.mode:
                        continue
                    v = z[full]
                    self.data.setdefault(key, {})[k0] = v if k0 == 'data' else (list(v) if v.ndim > 0 else v.item())
