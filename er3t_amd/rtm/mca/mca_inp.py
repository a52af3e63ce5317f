"""
MCARaTS-format input namelists: catalogue of the variables, writer and reader.

Wire format (what the reference writes, er3t/rtm/mca/mca_inp.py:636-697): fourteen `&group ... /` blocks in a
fixed order; inside a block the variables appear in catalogue order; only variables that were given a value are
written; scalars as ' %-15s = %-.16g', strings single-quoted, arrays through `nice_array_str` (on the same line
when the text is at most 80 characters, otherwise on the following lines); indexed variables such as
'Atm_ext1d(1:, 2)' follow the last variable of the same family.

`mca_inp_read` parses such a file back into a flat {key: value} dictionary -- the input side of a drop-in
solver executable.
"""

import os
import re
from collections import OrderedDict

import numpy as np

from er3t_amd.util import nice_array_str

__all__ = ['mca_inp_file', 'mca_inp_read', 'mca_inp_groups']


# group -> variables, in file order (the namelist definition of MCARaTS 0.10 as catalogued by the reference,
# er3t/rtm/mca/mca_inp.py:36-382); a short meaning for each variable is kept for `comment=True`
_CATALOGUE = [
    ('mcarWld_nml_init', [
        ('Wld_mverb', 'verbosity (0 quiet .. 3 most)'), ('Wld_jseed', 'random seed (0 = automatic)'),
        ('Wld_mbswap', 'byte swapping of binary inputs (0/1)'),
        ('Wld_mtarget', 'target: 1 fluxes and heating rates, 2 radiances, 3 quasi-radiances by volume rendering'),
        ('Wld_moptim', 'optimisation level: -2 default, -1 none, 0 unbiased only, 1 conservative, 2 standard, 3 quick'),
        ('Wld_njob', 'number of jobs per experiment')]),
    ('mcarSca_nml_init', [
        ('Sca_inpfile', 'file of tabulated phase functions'), ('Sca_npf', 'number of tabulated phase functions'),
        ('Sca_nanci', 'number of ancillary data'), ('Sca_nangi', 'number of angles'), ('Sca_nskip', 'records to skip'),
        ('Sca_ndfl', 'number of phase-function files'), ('Sca_nchi', 'orders of the truncation approximation'),
        ('Sca_ntg', 'table grid size for angles and probabilities'), ('Sca_qtfmax', 'geometrical truncation angle [deg]')]),
    ('mcarAtm_nml_init', [
        ('Atm_inpfile', 'file of 3-D optical properties'), ('Atm_np1d', 'scattering components of the 1-D medium'),
        ('Atm_np3d', 'scattering components of the 3-D medium'), ('Atm_nx', 'X grid points'), ('Atm_ny', 'Y grid points'),
        ('Atm_nz', 'Z grid points'), ('Atm_iz3l', 'index of the lowest 3-D layer'), ('Atm_nz3', 'number of 3-D layers'),
        ('Atm_nkd', 'k-distribution terms'), ('Atm_mtprof', 'temperature profile given per layer (0) or level (1)'),
        ('Atm_nwl', 'number of wavelengths'), ('Atm_nqlay', 'Gaussian quadrature points per layer'),
        ('Atm_iipfd1d', 'phase-function file index per 1-D component'), ('Atm_iipfd3d', 'phase-function file index per 3-D component')]),
    ('mcarSfc_nml_init', [
        ('Sfc_inpfile', 'file of 2-D surface properties'), ('Sfc_mbrdf', 'on/off flags of the four BRDF models'),
        ('Sfc_nxb', 'X grid points'), ('Sfc_nyb', 'Y grid points'), ('Sfc_nsco', 'coefficient-table size'),
        ('Sfc_nsuz', 'albedo look-up table size')]),
    ('mcarSrc_nml_init', [('Src_nsrc', 'number of sources')]),
    ('mcarFlx_nml_init', [
        ('Flx_mflx', 'flux density calculation flag'), ('Flx_mhrt', 'heating rate calculation flag'),
        ('Flx_nxf', 'flux cells along X'), ('Flx_nyf', 'flux cells along Y'), ('Flx_diff0', 'numerical diffusion parameter'),
        ('Flx_diff1', 'numerical diffusion parameter'), ('Flx_cf_dtau', 'layer optical thickness for collision forcing')]),
    ('mcarRad_nml_init', [
        ('Rad_mrkind', 'radiance kind: 0 none, 1 local (solid-angle averaged), 2 pixel (column cross-section) averaged'),
        ('Rad_mpmap', 'pixel mapping: 1 polar, 2 rectangular'), ('Rad_mplen', 'path-length statistics method'),
        ('Rad_nrad', 'number of radiances'), ('Rad_nxr', 'X pixels'), ('Rad_nyr', 'Y pixels'),
        ('Rad_nwf', 'number of weighting functions'), ('Rad_ntp', 'total path-length bins'),
        ('Rad_tpmin', 'minimum total path length'), ('Rad_tpmax', 'maximum total path length')]),
    ('mcarVis_nml_init', [
        ('Vis_mrend', 'rendering method'), ('Vis_epserr', 'convergence criterion'), ('Vis_fpsmth', 'phase-function smoothing fraction'),
        ('Vis_fatten', 'attenuation factor'), ('Vis_nqhem', 'quadrature points per hemisphere')]),
    ('mcarPho_nml_init', [
        ('Pho_iso_SS', 'scattering order at which 1-D transfer begins'), ('Pho_iso_tru', 'scattering order after which truncation is used'),
        ('Pho_iso_max', 'maximum scattering order sampled'), ('Pho_wmin', 'minimum photon weight'), ('Pho_wmax', 'maximum photon weight'),
        ('Pho_wfac', 'factor for the ideal photon weight'), ('Pho_pfpeak', 'phase-function peak threshold')]),
    ('mcarWld_nml_job', [('Wld_nplcf', 'dummy variable')]),
    ('mcarAtm_nml_job', [
        ('Atm_idread', 'location of the data to read'), ('Atm_wkd0', 'k-distribution weights'), ('Atm_dx', 'X cell size [m]'),
        ('Atm_dy', 'Y cell size [m]'), ('Atm_zgrd0', 'layer interface heights [m]'), ('Atm_tmp1d', 'temperatures [K]'),
        ('Atm_ext1d', 'extinction coefficients [1/m]'), ('Atm_omg1d', 'single-scattering albedos'),
        ('Atm_apf1d', 'phase-function selectors'), ('Atm_abs1d', 'absorption coefficients [1/m]'),
        ('Atm_fext1d', 'scaling of Atm_ext1d'), ('Atm_fext3d', 'scaling of the 3-D extinction'), ('Atm_fabs1d', 'scaling of Atm_abs1d'),
        ('Atm_fabs3d', 'scaling of the 3-D absorption'), ('Atm_mcs_rat', 'max/mean extinction ratio threshold'),
        ('Atm_mcs_frc', 'super-voxel fraction threshold'), ('Atm_mcs_dtauz', 'super-voxel vertical optical thickness threshold'),
        ('Atm_mcs_dtauxy', 'super-voxel horizontal optical thickness threshold')]),
    ('mcarSfc_nml_job', [
        ('Sfc_idread', 'index of the data to read'), ('Sfc_mtype', 'surface BRDF type'), ('Sfc_param', 'BRDF parameters (5)'),
        ('Sfc_nudsm', 'table size, DSM model'), ('Sfc_nurpv', 'table size, RPV model'), ('Sfc_nulsrt', 'table size, LSRT model'),
        ('Sfc_nqpot', 'quadrature points for preprocessing'), ('Sfc_rrmax', 'max relative-BRDF factor for random directions'),
        ('Sfc_rrexp', 'scaling exponent of the relative BRDF')]),
    ('mcarSrc_nml_job', [
        ('Src_mtype', 'source type: 0 local, 1 solar, 2 solar+thermal, 3 thermal'), ('Src_dwlen', 'band width [micron]'),
        ('Src_mphi', 'random azimuth flag'), ('Src_flx', 'source flux density'), ('Src_qmax', 'full cone angle [deg]'),
        ('Src_the', 'zenith angle of photon travel [deg]'), ('Src_phi', 'azimuth angle of photon travel [deg]')]),
    ('mcarRad_nml_job', [
        ('Rad_mrproj', 'angular weighting flag'), ('Rad_difr0', 'numerical diffusion parameter'), ('Rad_difr1', 'numerical diffusion parameter'),
        ('Rad_zetamin', 'threshold of the radiance contribution function'), ('Rad_npwrn', 'near-field scaling exponent'),
        ('Rad_npwrf', 'far-field scaling exponent'), ('Rad_cf_dmax', 'max layer optical thickness for collision forcing'),
        ('Rad_cf_taus', 'scattering optical thickness for collision forcing'), ('Rad_wfunc0', 'weighting functions'),
        ('Rad_rmin0', 'min distance from the camera'), ('Rad_rmid0', 'moderate distance from the camera'),
        ('Rad_rmax0', 'max distance from the camera'), ('Rad_phi', 'camera rotation about Z0 [deg]'),
        ('Rad_the', 'camera rotation about Y1 [deg]'), ('Rad_psi', 'camera rotation about Z2 [deg]'), ('Rad_umax', 'max angle along U'),
        ('Rad_vmax', 'max angle along V'), ('Rad_qmax', 'max angle of the field-of-view cone'), ('Rad_xpos', 'relative X position'),
        ('Rad_ypos', 'relative Y position'), ('Rad_zloc', 'Z location [m]'), ('Rad_apsize', 'aperture size'),
        ('Rad_zref', 'Z of the reference level')]),
]


def mca_inp_groups():
    """OrderedDict group -> list of variable names (file order)"""
    return OrderedDict((g, [k for k, _ in v]) for g, v in _CATALOGUE)


def _layout(input_dict):

    """
    Place every given key into its group.  Returns OrderedDict group -> list of (key, value) in file order.
    Indexed keys ('Name(...)') go right after the last already-placed key of the family 'Name(', or after 'Name'.
    """

    order = []          # (key, group) for the whole catalogue, extended by indexed keys
    for group, items in _CATALOGUE:
        order += [(k, group) for k, _ in items]
    keys = [k for k, _ in order]
    values = {}
    for key, val in input_dict.items():
        if key in keys:
            values[key] = val
            continue
        if '(' in key and ')' in key:
            base = key[:key.index('(')]
            if base not in keys:
                raise OSError('Error [mca_inp_nml]: please check input variable <%s>.' % key)
            family = [i for i, k in enumerate(keys) if base in k and '(' in k]
            at = family[-1] if len(family) > 0 else keys.index(base)
            keys.insert(at+1, key)
            order.insert(at+1, (key, order[at][1]))
            values[key] = val
        else:
            raise OSError('Error [mca_inp_nml]: please check input variable <%s>.' % key)

    out = OrderedDict((g, []) for g, _ in _CATALOGUE)
    for key, group in order:
        if key in values and values[key] is not None:
            out[group].append((key, values[key]))
    return out


def _describe(key):
    base = key[:key.index('(')] if '(' in key else key
    for _, items in _CATALOGUE:
        for k, text in items:
            if k == base:
                return text
    return ''


def mca_inp_file(input_fname, input_dict, verbose=True, comment=True):

    """
    Write one MCARaTS input file (reference: er3t/rtm/mca/mca_inp.py:636-697).
    With comment=False the text is byte-identical to the reference's for the same dictionary.
    """

    layout = _layout(input_dict)

    input_fname = os.path.abspath(input_fname)
    os.makedirs(os.path.dirname(input_fname), exist_ok=True)

    chunks = []
    for group, items in layout.items():
        chunks.append('&%s\n' % group)
        for key, var in items:
            if isinstance(var, (bool, np.bool_)):
                var = int(var)
            if isinstance(var, (int, float, np.integer, np.floating)):
                chunks.append(' %-15s = %-.16g\n' % (key, var))
            elif isinstance(var, str):
                chunks.append((' %-15s = %s\n' if '*' in var else ' %-15s = \'%s\'\n') % (key, var))
            elif isinstance(var, np.ndarray):
                if var.size > 1:
                    text = nice_array_str(var)
                    chunks.append((' %-15s = %s\n' if len(text) <= 80 else ' %-15s =\n%s\n') % (key, text))
                elif var.size == 1:
                    chunks.append(' %-15s = %-g\n' % (key, var.reshape(-1)[0]))
            else:
                msg = 'Error [mca_inp_file]: only types of int, float, str, ndarray are supported (do not support <%s> as %s).' % (key, type(var))
                raise ValueError(msg)
            if comment:
                chunks.append(' !----> %s\n\n' % _describe(key))
        chunks.append('/\n')

    with open(input_fname, 'w') as f:
        f.write(''.join(chunks))


# ----------------------------------------------------------------------------------------------
_num = r'[-+]?(?:\d+\.?\d*|\.\d+)(?:[eEdD][-+]?\d+)?'


_re_split = re.compile(r'[,\s]+')
_re_repeat = re.compile(r'(\d+)\*(%s)' % _num)
_re_num = re.compile(_num)
_re_int = re.compile(r'[-+]?\d+')
_re_assign = re.compile(r'^([A-Za-z_][A-Za-z_0-9]*(?:\([^)]*\))?)\s*=\s*(.*)$')


def _parse_values(text):
    text = text.strip()
    if text.startswith("'") or text.startswith('"'):
        return text.strip('\'"')
    out = []
    for tok in _re_split.split(text):
        if tok == '':
            continue
        # the common case first: a plain number as Python reads it too (digits at both ends rule out nan, inf and 1_000)
        if (tok[-1].isdigit() or tok[-1] == '.') and '_' not in tok and '*' not in tok:
            try:
                out.append(float(tok))
                continue
            except ValueError:
                pass
        m = _re_repeat.fullmatch(tok)                         # Fortran repeat count, e.g. 3*0.5
        if m:
            out += [float(m.group(2).replace('d', 'e').replace('D', 'e'))]*int(m.group(1))
        elif _re_num.fullmatch(tok):
            out.append(float(tok.replace('d', 'e').replace('D', 'e')))
        elif tok.upper() in ('.TRUE.', 'T'):
            out.append(1.0)
        elif tok.upper() in ('.FALSE.', 'F'):
            out.append(0.0)
        else:
            raise OSError('Error [mca_inp_read]: cannot parse <%s>.' % tok)
    if len(out) == 1:
        return int(text) if _re_int.fullmatch(text) else out[0]
    return np.array(out, dtype=np.float64)


def mca_inp_read(fname):

    """
    Parse an MCARaTS input file into a flat dictionary {key: int | float | str | ndarray}.
    Comment lines (starting with '!') are ignored; continuation lines extend the previous assignment.
    """

    if not os.path.isfile(fname):
        raise OSError('Error [mca_inp_read]: Cannot find <%s>.' % fname)

    nml = OrderedDict()
    key, buf = None, []

    def flush():
        if key is not None:
            nml[key] = _parse_values(' '.join(buf))

    with open(fname, 'r') as f:
        for line in f:
            s = line.strip()
            if s == '' or s.startswith('!'):
                continue
            if s.startswith('&') or s == '/':
                flush(); key, buf = None, []
                continue
            m = _re_assign.match(s)
            if m:
                flush()
                key, buf = m.group(1), [m.group(2)]
            else:
                buf.append(s)
    flush()
    return nml
