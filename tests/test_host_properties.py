"""
Property tests of the host layer (no GPU, no oracle): the file formats round-trip for arbitrary shapes and values, the
photon bookkeeping conserves photons for arbitrary weights and rank counts.
"""

import os

import numpy as np
from hypothesis import given, settings, strategies as st

import er3t_amd.rtm.mca as mca
from er3t_amd.dist import photon_shard
from er3t_amd.rtm.mca.mca_out import mca_out_write


@settings(max_examples=60, deadline=None)
@given(n=st.integers(0, 10**12), world=st.integers(1, 64))
def test_photon_shard_partition(n, world):
    parts = [photon_shard(n, world, r) for r in range(world)]
    assert parts[0][0] == 0 and sum(c for _, c in parts) == n
    assert all(o0+c0 == o1 for (o0, c0), (o1, _) in zip(parts[:-1], parts[1:]))
    assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


@settings(max_examples=60, deadline=None)
@given(nphoton=st.integers(10**3, 10**10), ng=st.integers(1, 32), base=st.floats(0.0, 0.5), seed=st.integers(0, 2**31-1))
def test_distribute_photon_conserves_photons(nphoton, ng, base, seed):
    """every photon is handed to some g, whatever the weights (reference: er3t/rtm/mca/mcarats.py:553-565)"""
    w = np.random.default_rng(seed).uniform(1e-4, 1.0, ng); w /= w.sum()
    n = mca.distribute_photon(nphoton, w, base_ratio=base)
    assert n.sum() == nphoton and n.shape == (ng,)
    if ng > 1 and nphoton*base/ng >= 1.0:
        assert n.min() >= int(nphoton*base/ng) - 1       # the evenly split share reaches every g


@settings(max_examples=25, deadline=None)
@given(nx=st.integers(1, 7), ny=st.integers(1, 6), nz=st.integers(1, 9), nvar=st.integers(1, 3), seed=st.integers(0, 2**31-1))
def test_output_files_round_trip(tmp_path_factory, nx, ny, nz, nvar, seed):
    """what mca_out_write writes, mca_out_raw (the reference's reader restated, mca_out.py:16-103) reads back bit for bit"""
    rng = np.random.default_rng(seed)
    arrays = [rng.standard_normal((nx, ny, nz)).astype('<f4') for _ in range(nvar)]
    fname = os.path.join(str(tmp_path_factory.mktemp('out')), 'r00.g000.out.bin')
    mca_out_write(fname, [('v%d' % i, 'variable %d' % i, a) for i, a in enumerate(arrays)])
    raw = mca.mca_out_raw(fname)
    assert len(raw.data) == nvar
    for i, a in enumerate(arrays):
        got = raw.data[i]['data']
        assert got.shape == (nx, ny, nz, 1) and np.array_equal(got[..., 0], a) and raw.data[i]['name'].startswith('v%d' % i)


@settings(max_examples=25, deadline=None)
@given(nz=st.integers(1, 12), seed=st.integers(0, 2**31-1), sza=st.floats(0.0, 89.0), vza=st.floats(0.0, 180.0))
def test_namelist_round_trip(tmp_path_factory, nz, seed, sza, vza):
    """mca_inp_file -> mca_inp_read returns the values that went in (to the digits the file carries), arrays and scalars"""
    rng = np.random.default_rng(seed)
    nml = {'Wld_mverb': 0, 'Wld_jseed': int(rng.integers(1, 2**31-1)), 'Wld_mtarget': 2, 'Sca_npf': 0,
           'Atm_nz': nz, 'Atm_np1d': 1, 'Atm_zgrd0': np.concatenate([[0.0], np.cumsum(rng.uniform(10.0, 900.0, nz))]),
           'Atm_ext1d(1:, 1)': rng.uniform(0.0, 1e-3, nz), 'Atm_omg1d(1:, 1)': rng.uniform(0.0, 1.0, nz),
           'Atm_apf1d(1:, 1)': rng.uniform(-1.0, 0.9, nz), 'Atm_abs1d(1:, 1)': rng.uniform(0.0, 1e-5, nz),
           'Sfc_mtype': 1, 'Sfc_param(1)': float(rng.uniform(0.0, 1.0)), 'Src_the': 180.0-sza, 'Src_phi': float(rng.uniform(0.0, 360.0)),
           'Rad_mrkind': 2, 'Rad_the': 180.0-vza, 'Rad_phi': 12.5, 'Rad_zloc': 705000.0, 'Rad_nxr': 1, 'Rad_nyr': 1}
    fname = os.path.join(str(tmp_path_factory.mktemp('inp')), 'r00.g000.inp.txt')
    mca.mca_inp_file(fname, nml)
    back = mca.mca_inp_read(fname)
    for key, val in nml.items():
        got = back[key]
        if isinstance(val, np.ndarray):
            assert np.allclose(np.asarray(got, dtype=np.float64), val, rtol=1e-5, atol=1e-12), key
        elif isinstance(val, float):
            assert abs(float(got)-val) <= 1e-5*abs(val) + 1e-9, key
        else:
            assert int(got) == val, key


_NUM = r'[-+]?(?:\d+\.?\d*|\.\d+)(?:[eEdD][-+]?\d+)?'


def _parse_slow(text):
    """the namelist value parser as it was before its fast path (every token through the regular expressions)"""
    import re
    out = []
    for tok in re.split(r'[,\s]+', text.strip()):
        if tok == '':
            continue
        m = re.fullmatch(r'(\d+)\*(%s)' % _NUM, tok)
        if m:
            out += [float(m.group(2).replace('d', 'e').replace('D', 'e'))]*int(m.group(1))
        elif re.fullmatch(_NUM, tok):
            out.append(float(tok.replace('d', 'e').replace('D', 'e')))
        elif tok.upper() in ('.TRUE.', 'T'):
            out.append(1.0)
        elif tok.upper() in ('.FALSE.', 'F'):
            out.append(0.0)
        else:
            raise OSError(tok)
    return out


_token = st.one_of(
    st.floats(allow_nan=False, allow_infinity=False, width=64).map(lambda x: '%.16g' % x),
    st.floats(min_value=-1e30, max_value=1e30, allow_nan=False).map(lambda x: ('%.6e' % x).replace('e', 'd')),
    st.floats(min_value=-1e6, max_value=1e6, allow_nan=False).map(lambda x: '%12g' % x),
    st.integers(min_value=-10**9, max_value=10**9).map(str),
    st.integers(min_value=0, max_value=999).map(lambda i: '%d.' % i),
    st.integers(min_value=0, max_value=999).map(lambda i: '.%03d' % i),
    st.tuples(st.integers(min_value=1, max_value=5), st.floats(min_value=-10, max_value=10, allow_nan=False)).map(lambda t: '%d*%g' % t),
    st.sampled_from(['.TRUE.', '.false.', 'T', 'F', '+3', '-0', '1E5', '2e-3']))


@settings(max_examples=300, deadline=None)
@given(toks=st.lists(_token, min_size=2, max_size=12), sep=st.sampled_from([' ', ', ', ',', '  ']))
def test_namelist_values_fast_path_equals_the_regular_expressions(toks, sep):
    from er3t_amd.rtm.mca.mca_inp import _parse_values
    text = sep.join(t.strip() for t in toks)
    got = _parse_values(text)
    want = _parse_slow(text)
    assert isinstance(got, np.ndarray) and got.dtype == np.float64 and got.tolist() == want


def test_namelist_values_reject_what_fortran_would():
    from er3t_amd.rtm.mca.mca_inp import _parse_values
    import pytest
    for bad in ('nan', 'inf', '1_000', 'abc', '1e', '--2', '3*', '1.2.3'):
        with pytest.raises(OSError):
            _parse_values(bad + ' 1.0')
    assert _parse_values('12') == 12 and isinstance(_parse_values('12'), int)
    assert _parse_values("'name.bin'") == 'name.bin' and _parse_values('1.5d2') == 150.0


@settings(max_examples=15, deadline=None)
@given(nx=st.integers(1, 5), ny=st.integers(1, 4), nz=st.integers(1, 6), ng=st.integers(1, 4), nrun=st.integers(1, 3), nvar=st.sampled_from([1, 3]),
       squeeze=st.booleans(), seed=st.integers(0, 10**6))
def test_reader_sums_in_the_reference_order(tmp_path_factory, nx, ny, nz, ng, nrun, nvar, squeeze, seed):
    """the file-route reader sums a run in a file-ordered array of its own: the result must be BIT-identical to the reference's
    loop `sum[..., ir] += raw*factor[:, ig]` (er3t/rtm/mca/mca_out.py:340-352, 470-480), float32 operation for float32 operation"""
    import er3t_amd.rtm.mca.mca_out as mo
    tmp = str(tmp_path_factory.mktemp('acc'))
    rng = np.random.default_rng(seed)
    names = [('fdnd', 'a'), ('fdn', 'b'), ('fup', 'c')][:nvar] if nvar == 3 else [('rad', 'r')]
    fn = [[os.path.join(tmp, 'r%02d.g%03d.out.bin' % (r, g)) for g in range(ng)] for r in range(nrun)]
    for r in range(nrun):
        for g in range(ng):
            mca_out_write(fn[r][g], [(n, d, (rng.random((nx, ny, nz))*10.0**rng.integers(-3, 3)).astype(np.float32)) for n, d in names])
    fac = (rng.random((nz, ng))*3.0).astype(np.float32)

    class M:
        pass
    m = M(); m.fnames_out = fn; m.Nrun = nrun; m.Ng = ng; m.fused = None
    keep = mo.g_factors
    mo.g_factors = lambda mca_obj, abs_obj, Nz: (fac, 1.0)
    try:
        sums, dims_info, toa = mo._accumulate(m, None, nvar, squeeze)
    finally:
        mo.g_factors = keep
    for iv in range(nvar):
        ref = np.zeros(sums[iv].shape, dtype=np.float32)
        for ir in range(nrun):
            for ig in range(ng):
                scaled = mo.mca_out_raw(fn[ir][ig]).data[iv]['data']*fac[:, ig][None, None, :, None]
                ref[..., ir] += np.squeeze(scaled) if squeeze else scaled
        assert sums[iv].dtype == np.float32 and np.array_equal(sums[iv], ref)
    assert dims_info[-1] == 'Nr'


def test_two_instruction_uniform_equals_the_four_instruction_one_for_every_mantissa():
    """mi3d_device.h: u01(w) = ((w >> 9) + 0.5) 2^-23 and u01_fast(w) = float([0x7F | w >> 9]) - (1 - 2^-24) (v_alignbit_b32 + one add)
    must be the same float32 for every word; both depend on the upper 23 bits only: all 2^23 of them, in float32 arithmetic"""
    m = np.arange(1 << 23, dtype=np.uint32)
    slow = (m.astype(np.float32) + np.float32(0.5))*np.float32(1.0/8388608.0)
    fast = ((np.uint32(0x7F) << np.uint32(23)) | m).view(np.float32) + np.float32(-0.99999994)
    assert np.float32(-0.99999994) == -(np.float32(1.0) - np.float32(2.0**-24))
    assert np.array_equal(slow.view(np.uint32), fast.view(np.uint32))
    assert slow.min() > 0.0 and slow.max() < 1.0
