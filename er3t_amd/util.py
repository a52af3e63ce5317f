"""
Numerical helpers of the rtm.mca path (counterparts of the listed functions of er3t/util/util.py; the rest of
that module -- gridding, geodesy, satellite I/O -- is outside the hot path).
"""

import numpy as np

__all__ = ['nice_array_str', 'get_lay_index', 'cal_sol_fac', 'cal_mol_ext', 'cal_mol_ext_0', 'cal_ext',
           'cal_r_twostream', 'cal_t_twostream', 'g0_calc', 'g_alt_calc']


def nice_array_str(array1d, numPerLine=6):

    """
    1-D array -> text block for a namelist: `numPerLine` values per line, each '  %12g', every line (also the
    last) newline-terminated.   (reference: er3t/util/util.py:191-221)
    """

    array1d = np.asarray(array1d)
    if array1d.ndim > 1:
        raise ValueError('Error [nice_array_str]: Only support 1-D array.')

    lines = []
    for i0 in range(0, array1d.size, numPerLine):
        lines.append(''.join('  %12g' % v for v in array1d[i0:i0+numPerLine]) + '\n')
    return ''.join(lines)


def get_lay_index(lay, lay_ref):

    """
    Index of the nearest reference layer for every layer height in <lay>; raises ValueError when a height is
    further from every reference layer than half the largest reference spacing.
    (reference: er3t/util/util.py:804-831)
    """

    lay = np.asarray(lay); lay_ref = np.asarray(lay_ref)
    threshold = np.diff(lay_ref).max()/2.0
    index = np.abs(lay[:, None]-lay_ref[None, :]).argmin(axis=1)
    dd = np.abs(lay-lay_ref[index])
    if np.any(dd > threshold):
        bad = np.argmax(dd > threshold)
        raise ValueError('Error [get_layer_index]: Mismatch between layer and reference layer: '+str(dd[bad]))
    return index.astype(np.int32)


def cal_sol_fac(dtime):

    """Sun-Earth distance factor 1/r^2 for a datetime (reference: er3t/util/util.py:934-950)"""

    doy = dtime.timetuple().tm_yday
    rsun = 1.0 - 0.0167086*np.cos(0.017202124161707175*(doy-4.0))
    return 1.0/(rsun*rsun)


def g0_calc(lat):
    """surface gravity [m/s^2] at latitude <lat> (Bodhaine et al. 1999, Eq. 11; reference: er3t/util/util.py:1003-1010)"""
    c2 = np.cos(2.0*lat*np.pi/180.0)
    return 9.806160*(1.0 - 0.0026373*c2 + 0.0000059*c2**2)


def g_alt_calc(g0, lat, z):
    """gravity [m/s^2] at height z [m] (Bodhaine et al. 1999, Eq. 10; reference: er3t/util/util.py:1012-1028)"""
    c2 = np.cos(2.0*lat*np.pi/180.0)
    g = g0*100.0 - (3.085462e-4 + 2.27e-7*c2)*z + (7.254e-11 + 1.0e-13*c2)*z**2 - (1.517e-17 + 6.0e-20*c2)*z**3
    return g/100.0


def _rayleigh_cross_term(wv0):
    # Bodhaine et al. (1999) fit of the Rayleigh optical depth spectral dependence
    num = 1.0455996 - 341.29061*wv0**(-2.0) - 0.90230850*wv0**2.0
    den = 1.0 + 0.0027059889*wv0**(-2.0) - 85.968563*wv0**2.0
    return num/den


def cal_mol_ext(wv0, pz1, pz2, atm0):

    """
    Rayleigh optical thickness between pressures pz1 > pz2 [hPa] at wavelength wv0 [micron], scaled with the
    column air mass of the atmosphere object's surface (latitude-dependent gravity, CO2-dependent molar mass).
    (reference: er3t/util/util.py:1030-1077; its debug prints are not reproduced)
    """

    lat = getattr(atm0, 'lat', 30.0)
    avogadro = 6.02214179e23
    g0 = g0_calc(lat)*100.0                                                      # cm/s^2
    ma0 = 28.9595 + 15.0556*atm0.lay['co2']['data'][0]/atm0.lay['air']['data'][0]
    p_sfc = atm0.lev['pressure']['data'][0]*1000.0                               # dyne/cm^2
    const_sfc = p_sfc*avogadro/(g0*ma0)*1e-28
    return const_sfc*_rayleigh_cross_term(wv0)*(pz1-pz2)/1013.25


def cal_mol_ext_0(wv0, pz1, pz2, atm0=None):
    """Rayleigh optical thickness, classic Bodhaine form (reference: er3t/util/util.py:1080-1101)"""
    return 0.00210966*_rayleigh_cross_term(wv0)*(pz1-pz2)/1013.25


def cal_ext(cot, cer, dz=1.0, Qe=2.0):
    """extinction [1/m] of a layer dz [km] thick with optical thickness cot and effective radius cer [micron]
    (reference: er3t/util/util.py:1104-1131)"""
    lwp = 2.0/3000.0*cot*cer
    lwc = lwp/dz
    return 0.75*Qe*lwc/cer*1.0e3


def cal_r_twostream(tau, a=0.0, g=0.85, mu=1.0):
    """two-stream reflectance of a conservative layer over a surface of albedo a (reference: er3t/util/util.py:1135-1151)"""
    x = 2.0*mu/(1.0-g)/(1.0-a)
    return (tau + a*x)/(tau + x)


def cal_t_twostream(tau, a=0.0, g=0.85, mu=1.0):
    """two-stream transmittance of a conservative layer over a surface of albedo a (reference: er3t/util/util.py:1155-1170)"""
    x = 2.0*mu/(1.0-g)/(1.0-a)
    return x*(1.0-a)/(tau + x)
