"""
K16 -- a deterministic plane-parallel radiative-transfer solver (adding-doubling, azimuthal Fourier decomposition): the
independent known answer for ANISOTROPIC MULTIPLE scattering that the Monte-Carlo oracle (oracle/mi3d_oracle.c) and the HIP
path (er3t_amd/csrc) are both held against.  Test infrastructure; shares no code, formula source or random numbers with either.

In the reference the role is played by libRadtran's DISORT (examples/00_er3t_bmk.py:470-579 compares the reflectance-vs-COT
curve of er3t/rtm/mca/util.py:19-213 with it); libRadtran is an external binary that is not available here, so the discrete-
ordinate answer is computed by this file instead.

Method (Hansen & Travis 1974, Space Sci. Rev. 16, 527; de Haan, Bosma & Hovenier 1987, A&A 183, 371):
  * directions: n Gauss-Legendre nodes on (0, 1) per hemisphere ("double Gauss") + the view cosines as extra nodes of weight 0
  * phase function P(cos Theta) = sum_l chi_l P_l(cos Theta), chi_0 = 1, expanded to l <= 2n-1 (then the quadrature keeps
    the normalisation exact); its m-th azimuthal component between two directions is
        P^m(mu, mu') = sum_{l>=m} chi_l  Pbar_l^m(mu) Pbar_l^m(mu'),   Pbar = sqrt((l-m)!/(l+m)!) P_l^m
    with P = sum_m (2 - delta_m0) P^m cos m(phi - phi')
  * every homogeneous layer starts as a slice of optical thickness tau / 2^K whose reflection and transmission are the exact
    single-scattering expressions for a finite slice, and is doubled K times; layers are then added from the top down and the
    Lambertian surface last.  Operators act on vectors of intensity samples: Rop = R(mu_i, mu_j) 2 mu_j w_j, Top = T(...) 2 mu_j
    w_j + diag(exp(-tau/mu_i)); the collimated solar beam (flux 1 through a plane normal to it) is carried as a pair of source
    vectors (diffuse field leaving the top / the bottom) and its own attenuation.
  * the radiance is summed as exact single scattering (closed form, every azimuthal order) + sum_m (2 - delta_m0) cos m dphi
    [I^m - I^m_single]: the Fourier series is only asked for the smooth multiply-scattered part.

Units as the solver's: radiance per unit solar irradiance normal to the beam; fluxes per unit horizontal area.
"""

import numpy as np

__all__ = ['hg_moments', 'rayleigh_moments', 'isotropic_moments', 'table_moments', 'solve', 'solve_scene_1d']


def hg_moments(g, nmom):
    """Henyey-Greenstein: chi_l = (2l+1) g^l"""
    l = np.arange(nmom+1)
    return (2.0*l+1.0)*np.power(float(g), l)


def rayleigh_moments(nmom):
    """scalar Rayleigh 3/4 (1 + cos^2): 1 + P_2/2"""
    chi = np.zeros(nmom+1)
    chi[0] = 1.0
    if nmom >= 2:
        chi[2] = 0.5
    return chi


def isotropic_moments(nmom):
    chi = np.zeros(nmom+1)
    chi[0] = 1.0
    return chi


def table_moments(ang_deg, pha, nmom):
    """Legendre moments of a tabulated phase function taken as piecewise linear in mu = cos(angle) and renormalised to
    (1/2) int P dmu = 1 (what include/mi3d.h: mi3d_set_phase specifies): chi_l = (2l+1)/2 int P P_l dmu, four Gauss points per
    interval of the table"""
    mu = np.cos(np.deg2rad(np.asarray(ang_deg, dtype=np.float64)))[::-1]
    p = np.asarray(pha, dtype=np.float64)[::-1]
    mu[0], mu[-1] = -1.0, 1.0
    xg, wg = np.polynomial.legendre.leggauss(4)
    a, b = mu[:-1], mu[1:]
    x = 0.5*(a+b)[:, None] + 0.5*(b-a)[:, None]*xg[None, :]
    w = 0.5*(b-a)[:, None]*wg[None, :]
    f = p[:-1, None] + (p[1:]-p[:-1])[:, None]*(x-a[:, None])/(b-a)[:, None]
    f = f/(0.5*np.sum(w*f))
    chi = np.zeros(nmom+1)
    p0, p1 = np.ones_like(x), x.copy()
    chi[0] = 0.5*np.sum(w*f)
    if nmom >= 1:
        chi[1] = 1.5*np.sum(w*f*p1)
    for l in range(2, nmom+1):
        p0, p1 = p1, ((2*l-1)*x*p1 - (l-1)*p0)/l
        chi[l] = (l+0.5)*np.sum(w*f*p1)
    return chi


def _phase_from_moments(chi, cos_theta):
    """P(cos Theta) = sum chi_l P_l: Legendre recurrence"""
    x = np.asarray(cos_theta, dtype=np.float64)
    p0, p1 = np.ones_like(x), x.copy()
    out = chi[0]*p0
    if len(chi) > 1:
        out = out + chi[1]*p1
    for l in range(2, len(chi)):
        p0, p1 = p1, ((2*l-1)*x*p1 - (l-1)*p0)/l
        if chi[l] != 0.0:
            out = out + chi[l]*p1
    return out


def _pbar(m, lmax, mu):
    """normalised associated Legendre functions Pbar_l^m(mu), l = m..lmax, as rows [l-m, point]"""
    mu = np.asarray(mu, dtype=np.float64)
    out = np.zeros((lmax-m+1, mu.size))
    s = np.sqrt(np.maximum(1.0-mu*mu, 0.0))
    pmm = np.ones_like(mu)
    for k in range(1, m+1):
        pmm = pmm*np.sqrt((2.0*k-1.0)/(2.0*k))*s
    out[0] = pmm
    if lmax > m:
        out[1] = mu*np.sqrt(2.0*m+1.0)*pmm
    for l in range(m+2, lmax+1):
        out[l-m] = ((2.0*l-1.0)*mu*out[l-m-1] - np.sqrt((l-1.0)**2-m*m)*out[l-m-2])/np.sqrt(l*l-m*m)
    return out


class _Layer:
    """operators and solar source vectors of a (stack of) layer(s) for one azimuthal order"""
    __slots__ = ('R', 'T', 'Rs', 'Ts', 'r', 't', 'e')


def _slice(m, chi, omega, dtau, mu, cw, mu0):
    """a homogeneous slice thin enough for single scattering: exact first-order expressions for a slice of finite thickness"""
    lmax = len(chi)-1
    n = mu.size
    L = _Layer()
    if m > lmax or omega <= 0.0:
        Z = np.zeros((n, n))
        E = np.diag(np.exp(-dtau/mu))
        L.R, L.T, L.Rs, L.Ts = Z, E, Z, E
        L.r = np.zeros(n); L.t = np.zeros(n)
        L.e = np.exp(-dtau/mu0)
        return L
    pb = _pbar(m, lmax, mu)                           # [l-m, i]
    pb0 = _pbar(m, lmax, np.array([mu0]))[:, 0]
    c = chi[m:]
    sign = (-1.0)**(np.arange(m, lmax+1)+m)           # Pbar_l^m(-mu) = (-1)^(l+m) Pbar_l^m(mu)
    Ptt = (pb.T*c) @ pb                               # same hemisphere  (transmission)
    Prr = (pb.T*(c*sign)) @ pb                        # opposite hemispheres (reflection)
    ptt0 = (pb.T*c) @ pb0
    prr0 = (pb.T*(c*sign)) @ pb0

    def refl(mi, mj):
        s = 1.0/mi + 1.0/mj
        return -np.expm1(-dtau*s)/(mi+mj)

    def tran(mi, mj):
        d = mi-mj
        same = np.abs(d) < 1e-12*np.maximum(mi, mj)
        dd = np.where(same, 1.0, d)
        a = np.exp(-dtau/mj)*np.expm1(-dtau*(1.0/mi-1.0/mj))/dd      # (exp(-dtau/mi) - exp(-dtau/mj)) / (mi - mj)
        return np.where(same, dtau*np.exp(-dtau/mi)/(mi*mi), a)

    mi, mj = mu[:, None], mu[None, :]
    Rf = 0.25*omega*Prr*refl(mi, mj)                  # R(mu_i, mu_j) in I_r = 2 int R mu' I dmu'
    Tf = 0.25*omega*Ptt*tran(mi, mj)
    E = np.diag(np.exp(-dtau/mu))
    L.R = Rf*cw[None, :]
    L.T = Tf*cw[None, :] + E
    L.Rs, L.Ts = L.R, L.T
    # unit solar beam on the top: I^m = R^m(mu, mu0) mu0 / pi
    L.r = 0.25*omega*prr0*refl(mu, mu0)*mu0/np.pi
    L.t = 0.25*omega*ptt0*tran(mu, mu0)*mu0/np.pi
    L.e = np.exp(-dtau/mu0)
    return L


def _add(a, b):
    """layer (stack) a on top of b"""
    n = a.R.shape[0]
    I = np.eye(n)
    c = _Layer()
    G = np.linalg.solve(I - a.Rs @ b.R, np.column_stack([a.T, (a.t + a.Rs @ (a.e*b.r))[:, None]]))
    GT, D = G[:, :n], G[:, n]                          # D: diffuse field going down through the interface (solar illumination)
    U = b.R @ D + a.e*b.r                              # ... and going up
    c.R = a.R + a.Ts @ (b.R @ GT)
    c.T = b.T @ GT
    H = np.linalg.solve(I - b.R @ a.Rs, b.Ts)
    c.Rs = b.Rs + b.T @ (a.Rs @ H)
    c.Ts = a.Ts @ H
    c.r = a.r + a.Ts @ U
    c.t = b.T @ D + a.e*b.t
    c.e = a.e*b.e
    return c


def _double_layer(m, chi, omega, tau, mu, cw, mu0, dtau_max):
    K = max(0, int(np.ceil(np.log2(max(tau/dtau_max, 1.0)))))
    L = _slice(m, chi, omega, tau/2.0**K, mu, cw, mu0)
    for _ in range(K):
        L = _add(L, L)
    L.e = np.exp(-tau/mu0)      # (K squarings of 1 - 1e-9 carry its rounding: the direct beam is known exactly)
    return L


def _single_up(layers, mu, mu0, pfun):
    """singly scattered radiance leaving the top towards mu (array), closed form; pfun(layer index) -> phase value(s)"""
    out = np.zeros_like(mu)
    t = 0.0
    s = 1.0/mu + 1.0/mu0
    for il, (tau, omega, chi) in enumerate(layers):
        if omega > 0.0 and tau > 0.0:
            out = out + omega*pfun(il)/(4.0*np.pi)*(mu0/(mu0+mu))*(np.exp(-t*s) - np.exp(-(t+tau)*s))
        t += tau
    return out


def solve(layers, mu0, albedo=0.0, view_mu=(), view_dphi=(), nstream=48, dtau_max=2.0e-9, mmax=None, tol=2.0e-8):
    """
    layers    : [(tau, omega, chi)] from the top down; chi = Legendre moments of the phase function (chi[0] = 1)
    mu0       : cosine of the solar zenith angle;  albedo: Lambertian surface
    view_mu   : cosines of the view zenith angles (radiance leaving the top, sensor above the atmosphere)
    view_dphi : azimuth of the direction the light travels to the sensor minus that of the direct beam's travel, radians
                (0: forward scattering side)
    returns {'albedo', 'transmittance' (total), 'transmittance_direct', 'radiance' [nview], 'nmode'}
    """
    view_mu = np.atleast_1d(np.asarray(view_mu, dtype=np.float64))
    view_dphi = np.atleast_1d(np.asarray(view_dphi, dtype=np.float64))
    x, w = np.polynomial.legendre.leggauss(nstream)
    gmu, gw = 0.5*(x+1.0), 0.5*w
    uniq = np.unique(view_mu)
    mu = np.concatenate([gmu, uniq])
    wt = np.concatenate([gw, np.zeros(uniq.size)])
    cw = 2.0*mu*wt
    vidx = np.array([nstream + int(np.searchsorted(uniq, v)) for v in view_mu], dtype=int)
    nmom = 2*nstream-1
    lay = [(float(t), float(o), np.asarray(c, dtype=np.float64)[:nmom+1]) for (t, o, c) in layers]
    if mmax is None:
        mmax = nmom
    n = mu.size

    rad_ms = np.zeros(view_mu.size)       # sum over m of (2 - delta) cos(m dphi) [I^m - I^m_single]
    res = {}
    quiet = 0
    nmode = 0
    for m in range(0, mmax+1):
        stack = None
        for (tau, omega, chi) in lay:
            L = _double_layer(m, chi, omega, tau, mu, cw, mu0, dtau_max)
            stack = L if stack is None else _add(stack, L)
        # the surface below: Lambertian, m = 0 only
        if m == 0:
            D = np.linalg.solve(np.eye(n) - stack.Rs @ (albedo*np.tile(cw, (n, 1))), stack.t + stack.Rs @ (stack.e*np.full(n, albedo*mu0/np.pi)))
            f_dn_diffuse = np.pi*np.dot(cw, D)                         # 2 pi sum w mu D
            f_dn = f_dn_diffuse + mu0*stack.e
            U = np.full(n, albedo*f_dn/np.pi)
            top = stack.r + stack.Ts @ U
            res['albedo'] = np.pi*np.dot(cw, top)/mu0
            res['transmittance'] = f_dn/mu0
            res['transmittance_direct'] = stack.e
        else:
            top = stack.r
        nmode = m+1
        if view_mu.size == 0:
            break
        # singly scattered part of this order (atmosphere only), closed form with P^m
        def pm(il, m=m):
            chi = lay[il][2]
            if m > len(chi)-1:
                return np.zeros(view_mu.size)
            pbv = _pbar(m, len(chi)-1, view_mu)
            pb0 = _pbar(m, len(chi)-1, np.array([mu0]))[:, 0]
            sign = (-1.0)**(np.arange(m, len(chi))+m)
            return (pbv.T*(chi[m:]*sign)) @ pb0
        i1 = _single_up(lay, view_mu, mu0, pm)
        term = (1.0 if m == 0 else 2.0)*np.cos(m*view_dphi)*(top[vidx]-i1)
        rad_ms += term
        scale = np.maximum(np.abs(rad_ms), 1e-300)
        if m > 0 and np.all(np.abs(top[vidx]-i1) <= tol*scale):
            quiet += 1
            if quiet >= 2:
                break
        else:
            quiet = 0
    if view_mu.size:
        cos_theta = -mu0*view_mu + np.sqrt(max(1.0-mu0*mu0, 0.0))*np.sqrt(np.maximum(1.0-view_mu**2, 0.0))*np.cos(view_dphi)
        def pexact(il):
            return _phase_from_moments(lay[il][2], cos_theta)
        res['radiance'] = rad_ms + _single_up(lay, view_mu, mu0, pexact)
        res['radiance_single'] = _single_up(lay, view_mu, mu0, pexact)
    else:
        res['radiance'] = np.zeros(0)
    res['nmode'] = nmode
    return res


def solve_scene_1d(sc, nstream=48, **kw):
    """the deterministic answer for a horizontally homogeneous er3t_amd.scene.Scene (no 3-D region, Lambertian surface, sensors
    above the atmosphere): its 1-D constituents layer by layer -- extinction, single-scattering albedo and phase-function selector
    (apf <= -1.5 isotropic, <= -1 Rayleigh, (-1, 1) Henyey-Greenstein, >= 1 the tables of the scene) -- and the gas absorption
    become one mixture per layer; returns solve()'s dictionary, 'radiance' in the order of the scene's views"""
    if sc.nz3 > 0 or int(sc.sfc_mtype) != 1 or sc.jsfc is not None:
        raise ValueError('solve_scene_1d: plane-parallel scenes over a Lambertian surface only')
    nmom = 2*nstream-1
    tabs = {}

    def moments(apf):
        apf = float(apf)
        if apf <= -1.5:
            return isotropic_moments(nmom)
        if apf <= -1.0:
            return rayleigh_moments(nmom)
        if apf < 1.0:
            return hg_moments(apf, nmom)
        t = apf-1.0
        i0 = min(max(int(np.floor(t)), 0), sc.pha.shape[0]-1)
        fr = t-i0 if i0 < sc.pha.shape[0]-1 else 0.0
        for i in (i0, i0+1):
            if i not in tabs and i < sc.pha.shape[0]:
                tabs[i] = table_moments(sc.ang, sc.pha[i], nmom)
        return tabs[i0] if fr <= 0.0 else (1.0-fr)*tabs[i0] + fr*tabs[i0+1]

    layers = []
    dz = np.diff(sc.zgrd)
    for k in range(sc.nz-1, -1, -1):                 # from the top down
        ext = np.asarray(sc.ext1d[:, k], dtype=np.float64)
        ks = ext*np.asarray(sc.omg1d[:, k], dtype=np.float64)
        bt = ext.sum() + float(sc.abs1d[k])
        if bt <= 0.0:
            continue
        chi = np.zeros(nmom+1)
        if ks.sum() > 0.0:
            for p in range(ext.size):
                if ks[p] > 0.0:
                    chi += ks[p]*moments(sc.apf1d[p, k])
            chi /= ks.sum()
        else:
            chi[0] = 1.0
        layers.append((bt*dz[k], ks.sum()/bt, chi))
    mu0 = abs(np.cos(np.deg2rad(sc.src_the)))
    the, phi = np.asarray(sc.view_the, dtype=np.float64), np.asarray(sc.view_phi, dtype=np.float64)
    vmu = -np.cos(np.deg2rad(the))                   # Rad_the = 180 - view zenith angle
    dphi = np.deg2rad(phi + 180.0 - sc.src_phi)      # azimuth of travel towards the sensor - azimuth of the beam's travel
    return solve(layers, mu0, float(sc.sfc_param[0]), view_mu=vmu, view_dphi=dphi, nstream=nstream, **kw)
