/*
 * mi3d_oracle.c — CPU ORACLE (test infrastructure, NOT part of the product).
 *
 * Plain-C, double-precision restatement of the photon-transport algorithm that the reference
 * toolbox delegates to the external MCARaTS 0.10.4 program (hong-chen/er3t launches it at
 * er3t/rtm/mca/mca_run.py:101-115; the program itself is an un-vendored third-party Fortran
 * dependency named only in docs/source/tutorial/install.rst:39-48 and er3t/common.py:10, so its
 * source is NOT available: this file restates the PUBLISHED algorithm — forward Monte Carlo with
 * local-estimate radiance, Iwabuchi 2006 (J. Atmos. Sci. 63, 2324) as cited at
 * er3t/rtm/mca/mcarats.py:59 — on the input contract er3t defines:
 *     namelist catalogue            er3t/rtm/mca/mca_inp.py:15-384, 388-571
 *     values er3t actually sets     er3t/rtm/mca/mcarats.py:234-414
 *     3-D / phase / surface files   er3t/rtm/mca/mca_atm.py:373-389, mca_sca.py:82-92, mca_sfc.py:136-146
 *     output variables & order      er3t/rtm/mca/mca_out.py:350-352, 473
 *     radiance normalisation        er3t/rtm/mca/util.py:101-102  (reflectance = pi*I/(F*mu0))
 *
 * PARITY STATUS: "parity unpinned" against MCARaTS itself — the reference holds no golden output,
 * tolerance or assertion for solver results.  This oracle is pinned instead by
 * analytic known-answer tests (tests/test_oracle_kat.py: Beer's law, Lambert surface, single
 * scattering, energy conservation, 3D==1D on homogeneous grids) and by Random123's published
 * Philox4x32-10 vectors.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Deliberately written differently from the HIP kernels (absolute double coordinates, every
 * local-estimate ray marched cell by cell, no tables in fast memory, no shortcuts) while
 * consuming the SAME random-number protocol, so that a photon id follows the same history in
 * both up to floating-point rounding:
 *
 *   Philox4x32-10, key = (seed lo, seed hi), counter = (id lo, id hi, draw, 0)
 *   u_j = ((word_j >> 9) + 0.5) * 2^-23                      (exact in float32 and float64)
 *   draw 0           : u0,u1 -> launch x,y ; u2,u3 -> direction inside the solar cone
 *   draw c>=1 (cycle): u0 -> optical path -ln(u0); u1 -> component / table choice;
 *                      u2 -> cos(scattering angle) or surface cos^2(zenith); u3 -> azimuth
 *   roulette         : consumes one extra draw (its u0) only when weight < wmin; survival probability
 *                      weight/wfac, survivors continue with weight wfac (Pho_wmin = 0.2, Pho_wfac = 1)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_VIEW 16
#define ORC_NCOUNTER 16
#define PI 3.14159265358979323846

typedef struct {
    /* 1-D background (mca_atm.py:68-139) */
    int nz;
    const double *zgrd; /* [nz+1] */
    int np1d;
    const float *ext1d, *omg1d, *apf1d; /* [np1d][nz] */
    const float *abs1d;                 /* [nz] */
    /* 3-D region, file layout x fastest (mca_atm.py:373-389) */
    int nx, ny, nz3, iz3l, np3d;
    double dx, dy;
    const float *abst;                /* [nz3][ny][nx] or NULL */
    const float *extp, *omgp, *apfp;  /* [np3d][nz3][ny][nx] */
    /* phase tables (mca_sca.py:82-92) */
    int nang, npf;
    const float *ang; /* [nang] */
    const float *pha; /* [npf][nang] */
    /* surface (mcarats.py:393-399, mca_sfc.py:136-146) */
    int sfc_mtype;
    float sfc_param[5];
    int nxb, nyb;
    const float *jsfc; /* [nyb][nxb] or NULL */
    const float *psfc; /* [5][nyb][nxb] */
    /* source (mcarats.py:374-383) */
    double src_flx, src_qmax, src_the, src_phi;
    /* views (mcarats.py:285-307) */
    int nview;
    double view_the[ORC_MAX_VIEW], view_phi[ORC_MAX_VIEW], view_zloc[ORC_MAX_VIEW];
    double zref;
    int nxr, nyr;
    /* options */
    int target; /* bits: 1 flux, 2 radiance, 4 heating rates (Flx_mhrt = 1, mcarats.py:279-283; with bit 1) */
    int solver; /* 0 3D, 1 partial 3D (3-D direct beam, independent columns for everything scattered), 2 IPA */
    double wmin, wfac; /* Russian roulette below wmin; survivors restart with weight wfac (Pho_wmin, Pho_wfac) */
    int nthreads;
    double le_tau1;    /* > 0: Russian roulette on local-estimate rays beyond this optical depth (see le_roulette) */
    /* Rad_mrkind (mca_inp.py:141-144): 2 = radiance averaged over the pixel's column cross-section (satellite, views above),
     * 1 = local radiance at a point, averaged over the pixel's solid angle (all-sky camera, mcarats.py:291-296, 369-372):
     * view i is then a camera at (cam_xpos Lx, cam_ypos Ly, view_zloc) whose axes are the world axes turned by the Z-Y-Z
     * rotations view_phi, view_the, cam_psi (mca_inp.py:324-330); it looks along its z axis; polar pixel map (Rad_mpmap = 1). */
    int rad_kind;
    double cam_xpos[ORC_MAX_VIEW], cam_ypos[ORC_MAX_VIEW], cam_psi[ORC_MAX_VIEW];
    double cam_qmax[ORC_MAX_VIEW], cam_umax[ORC_MAX_VIEW], cam_vmax[ORC_MAX_VIEW], cam_apsize[ORC_MAX_VIEW];
    double le_cmin;    /* > 0: Russian roulette on the WEIGHT of local-estimate rays of marched satellite views: a ray that would carry
                        * c = w P / 4 pi (surface: w R cos / pi) below le_cmin is marched with probability c / le_cmin, and then
                        * carries le_cmin (unbiased; the decision is a hash of seed, photon id, Philox block index and view) */
    int cam_images;    /* cameras in the cyclic domain: an event contributes to the periodic images of the camera within this many
                        * domain lengths of the nearest one, (2 cam_images + 1)^2 images in all (0: the nearest image only) */
} orc_config;

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; constants from the Random123 distribution)            */
/* ------------------------------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                          uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(uint64_t seed, uint64_t id, uint32_t draw, uint32_t out[4]) {
    philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), draw, 0u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), out);
}

/* raw Philox block for arbitrary counter/key: used to check Random123's known answers */
void orc_philox_raw(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

static void draw4(uint64_t seed, uint64_t id, uint32_t draw, double u[4]) {
    uint32_t w[4];
    orc_philox(seed, id, draw, w);
    for (int j = 0; j < 4; ++j) u[j] = ((double)(w[j] >> 9) + 0.5) * (1.0 / 8388608.0);
}

/* ------------------------------------------------------------------------------------------ */
/* scene: everything converted to double once                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const orc_config *c;
    int nz, k3lo, k3hi; /* 3-D layers are k3lo <= k < k3hi (0-based) */
    double Lx, Ly;
    double *bt1d;   /* [nz] total extinction of the 1-D part: sum ext1d + abs1d */
    double *ks1d;   /* [np1d][nz] scattering coefficient omg*ext */
    /* tables, ascending mu */
    int nang, npf;
    double *tmu;  /* [nang] */
    double *tp;   /* [npf][nang] normalised P */
    double *tcdf; /* [npf][nang] */
    /* source / views */
    double sdir[3], cos_cone;
    double vdir[ORC_MAX_VIEW][3];
    double cam_pos[ORC_MAX_VIEW][3], cam_ax[ORC_MAX_VIEW][3][3]; /* cameras: position; x, y, z axis in world coordinates */
    /* tallies (shared, updated atomically) */
    double *rad;  /* [nview][nyr][nxr] */
    double *flux; /* [3][nz+1][ny][nx] */
    double *heat; /* [nz][ny][nx] weight absorbed in every cell (target & 4: heating rates, Flx_mhrt = 1), or NULL */
} scene_t;

static inline long vox(const scene_t *s, int ix, int iy, int k3) {
    return ((long)k3 * s->c->ny + iy) * s->c->nx + ix;
}

/* total extinction of the cell (ix,iy,k) */
static double cell_bt(const scene_t *s, int ix, int iy, int k) {
    double b = s->bt1d[k];
    if (k >= s->k3lo && k < s->k3hi) {
        const orc_config *c = s->c;
        long v = vox(s, ix, iy, k - s->k3lo);
        long nvox = (long)c->nx * c->ny * c->nz3;
        if (c->abst) b += (double)c->abst[v];
        for (int ip = 0; ip < c->np3d; ++ip) b += (double)c->extp[ip * nvox + v];
    }
    return b > 0.0 ? b : 0.0;
}

/* phase function value with ∫P dΩ = 4π */
static double table_eval(const scene_t *s, int it, double mu) {
    const double *m = s->tmu, *p = s->tp + (long)it * s->nang;
    int lo = 0, hi = s->nang - 1;
    if (mu <= m[0]) return p[0];
    if (mu >= m[hi]) return p[hi];
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (m[mid] <= mu) lo = mid; else hi = mid;
    }
    double f = (mu - m[lo]) / (m[hi] - m[lo]);
    return p[lo] + f * (p[hi] - p[lo]);
}

static double table_sample(const scene_t *s, int it, double u) {
    const double *m = s->tmu, *p = s->tp + (long)it * s->nang, *cdf = s->tcdf + (long)it * s->nang;
    int lo = 0, hi = s->nang - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid; else hi = mid;
    }
    double r = 2.0 * (u - cdf[lo]);
    double sl = (p[hi] - p[lo]) / (m[hi] - m[lo]);
    double disc = p[lo] * p[lo] + 2.0 * sl * r;
    if (disc < 0.0) disc = 0.0;
    double den = p[lo] + sqrt(disc);
    double t = den > 0.0 ? 2.0 * r / den : 0.0;
    double mu = m[lo] + t;
    if (mu > m[hi]) mu = m[hi];
    return mu;
}

/* decode the table selection of apf >= 1: tables i0 and i0+1 mixed with fraction f */
static void table_pick(const scene_t *s, double apf, int *i0, double *f) {
    double t = apf - 1.0;
    int i = (int)floor(t);
    double fr = t - i;
    if (i < 0) { i = 0; fr = 0.0; }
    if (i >= s->npf - 1) { i = s->npf - 1; fr = 0.0; }
    *i0 = i; *f = fr;
}

static double phase_eval(const scene_t *s, double apf, double mu) {
    if (apf <= -1.5) return 1.0;
    if (apf <= -1.0) return 0.75 * (1.0 + mu * mu);
    if (apf < 1.0) {
        double g = apf, d = 1.0 + g * g - 2.0 * g * mu;
        return (1.0 - g * g) / (d * sqrt(d));
    }
    if (s->npf <= 0) return 1.0;
    int i0; double f;
    table_pick(s, apf, &i0, &f);
    double p = table_eval(s, i0, mu);
    if (f > 0.0) p = (1.0 - f) * p + f * table_eval(s, i0 + 1, mu);
    return p;
}

/* sample cos(scattering angle); `usel` is a spare uniform for the table mix */
static double phase_sample(const scene_t *s, double apf, double u, double usel) {
    if (apf <= -1.5) return 2.0 * u - 1.0;
    if (apf <= -1.0) {
        double q = 8.0 * u - 4.0;
        double a = cbrt(0.5 * q + sqrt(0.25 * q * q + 1.0));
        return a - 1.0 / a;
    }
    if (apf < 1.0) {
        double g = apf;
        if (fabs(g) < 1e-3) return 2.0 * u - 1.0;
        double t = (1.0 - g * g) / (1.0 - g + 2.0 * g * u);
        double mu = (1.0 + g * g - t * t) / (2.0 * g);
        return mu < -1.0 ? -1.0 : (mu > 1.0 ? 1.0 : mu);
    }
    if (s->npf <= 0) return 2.0 * u - 1.0;
    int i0; double f;
    table_pick(s, apf, &i0, &f);
    if (f > 0.0 && usel < f) i0 += 1;
    return table_sample(s, i0, u);
}

/* rotate direction d by polar cosine mu and azimuth phi */
static void rotate_dir(double d[3], double mu, double phi) {
    double st = sqrt(fmax(0.0, 1.0 - mu * mu)), cp = cos(phi), sp = sin(phi);
    double ux = d[0], uy = d[1], uz = d[2];
    double den2 = 1.0 - uz * uz;
    double nx_, ny_, nz_;
    if (den2 < 1e-10) {
        double sg = uz >= 0.0 ? 1.0 : -1.0;
        nx_ = st * cp; ny_ = st * sp; nz_ = mu * sg;
    } else {
        double den = sqrt(den2);
        nx_ = st * (ux * uz * cp - uy * sp) / den + ux * mu;
        ny_ = st * (uy * uz * cp + ux * sp) / den + uy * mu;
        nz_ = -st * cp * den + uz * mu;
    }
    double n = 1.0 / sqrt(nx_ * nx_ + ny_ * ny_ + nz_ * nz_);
    d[0] = nx_ * n; d[1] = ny_ * n; d[2] = nz_ * n;
}

/* ------------------------------------------------------------------------------------------ */
/* surface models                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* Ross-Thick / Li-Sparse-Reciprocal reflectance factor R (BRDF = R/pi), MODIS MCD43 kernels:
 * packing of (fiso, fgeo, fvol) follows er3t/rtm/mca/mca_sfc.py:104-117 */
static double lsrt_R(double fiso, double fgeo, double fvol, const double din[3], const double dout[3]) {
    double ci = -din[2], cv = dout[2]; /* cos of zenith angles (to sun, to viewer) */
    if (ci < 1e-6) ci = 1e-6;
    if (cv < 1e-6) cv = 1e-6;
    double si = sqrt(fmax(0.0, 1.0 - ci * ci)), sv = sqrt(fmax(0.0, 1.0 - cv * cv));
    double cphi = 1.0;
    double hi = sqrt(din[0] * din[0] + din[1] * din[1]), hv = sqrt(dout[0] * dout[0] + dout[1] * dout[1]);
    if (hi > 1e-12 && hv > 1e-12) cphi = (-din[0] * dout[0] - din[1] * dout[1]) / (hi * hv);
    if (cphi > 1.0) cphi = 1.0;
    if (cphi < -1.0) cphi = -1.0;
    double sphi2 = 1.0 - cphi * cphi;
    double cxi = ci * cv + si * sv * cphi;
    if (cxi > 1.0) cxi = 1.0;
    if (cxi < -1.0) cxi = -1.0;
    double xi = acos(cxi), sxi = sin(xi);
    double kvol = ((0.5 * PI - xi) * cxi + sxi) / (ci + cv) - 0.25 * PI;
    double ti = si / ci, tv = sv / cv; /* b/r = 1: primed angles equal the angles */
    double seci = 1.0 / ci, secv = 1.0 / cv;
    double D2 = ti * ti + tv * tv - 2.0 * ti * tv * cphi;
    if (D2 < 0.0) D2 = 0.0;
    double cost = 2.0 * sqrt(D2 + ti * ti * tv * tv * sphi2) / (seci + secv); /* h/b = 2 */
    if (cost > 1.0) cost = 1.0;
    double t = acos(cost);
    double O = (t - sin(t) * cost) * (seci + secv) / PI;
    double kgeo = O - seci - secv + 0.5 * (1.0 + cxi) * seci * secv;
    double R = fiso + fgeo * kgeo + fvol * kvol;
    return R > 0.0 ? R : 0.0;
}

/* Diffuse-specular mixture ("DSM", jsfc = 2; parameters packed (diffuse albedo, diffuse fraction, Re m, Im m, slope variance)
 * as er3t/rtm/mca/mca_sfc.py:119-128 hands them over from er3t/pre/sfc/sfc_gen.py:131-145 and util.py:109-112,150-156:
 * whitecaps as a Lambertian part, slope variance sigma^2 = 0.003 + 0.00512 u10 of Cox and Munk 1954).  The solver's own
 * formulation is not in the reference tree; this is the standard rough-ocean reflectance those parameters define
 * (Cox and Munk 1954 isotropic Gaussian facet slopes; Fresnel reflection of unpolarised light by a facet of complex index m;
 * bidirectional shadowing after Smith 1967 / Sancer 1969 -- e.g. Mishchenko and Travis 1997, JGR 102, eq. 11-15):
 *     R = f_d a_d + (1 - f_d) pi F(cos chi) P(mu_n) S(mu_i, mu_v) / (4 mu_i mu_v mu_n^4),   BRDF = R / pi
 *     P = exp(-tan^2(theta_n) / sigma^2) / (pi sigma^2),  S = 1 / (1 + L(mu_i) + L(mu_v)),
 *     L(mu) = [exp(-nu^2) / (sqrt(pi) nu) - erfc(nu)] / 2,  nu = mu / (sigma sqrt(1 - mu^2))
 * with the facet normal n bisecting the directions to the source and to the viewer, chi the angle of incidence on the facet. */
static double fresnel_unpolarised(double nr, double ni, double c) {
    if (c > 1.0) c = 1.0;
    if (c < 1e-9) c = 1e-9;
    double s2 = 1.0 - c * c;
    double u = nr * nr - ni * ni - s2, v = sqrt(u * u + 4.0 * nr * nr * ni * ni);
    double a2 = 0.5 * (v + u), b2 = 0.5 * (v - u);
    if (a2 < 0.0) a2 = 0.0;
    if (b2 < 0.0) b2 = 0.0;
    double a = sqrt(a2);
    double rs = ((a - c) * (a - c) + b2) / ((a + c) * (a + c) + b2);
    double q = s2 / c;
    double rp = rs * ((a - q) * (a - q) + b2) / ((a + q) * (a + q) + b2);
    return 0.5 * (rs + rp);
}

static double dsm_shadow_lambda(double mu, double sigma) {
    if (mu >= 1.0) return 0.0;
    double nu = mu / (sigma * sqrt(1.0 - mu * mu));
    return 0.5 * (exp(-nu * nu) / (sqrt(PI) * nu) - erfc(nu));
}

static double dsm_R(const double p[5], const double din[3], const double dout[3]) {
    double ad = p[0], fd = p[1], nr = p[2], ni = p[3], s2 = p[4];
    if (fd < 0.0) fd = 0.0;
    if (fd > 1.0) fd = 1.0;
    if (ad < 0.0) ad = 0.0;
    if (ad > 1.0) ad = 1.0;
    double R = fd * ad;
    double mi = -din[2], mv = dout[2];
    if (s2 > 0.0 && fd < 1.0 && mi > 1e-6 && mv > 1e-6) {
        double hx = dout[0] - din[0], hy = dout[1] - din[1], hz = dout[2] - din[2];   /* towards the viewer + towards the source */
        double hn = sqrt(hx * hx + hy * hy + hz * hz);
        if (hn > 1e-12) {
            double mun = hz / hn;
            double cchi = (-din[0] * hx - din[1] * hy - din[2] * hz) / hn;
            if (mun > 1e-6) {
                double t2 = (1.0 - mun * mun) / (mun * mun);
                double P = exp(-t2 / s2) / (PI * s2);
                double sg = sqrt(s2);
                double S = 1.0 / (1.0 + dsm_shadow_lambda(mi, sg) + dsm_shadow_lambda(mv, sg));
                double mun2 = mun * mun;
                R += (1.0 - fd) * PI * fresnel_unpolarised(nr, ni, cchi) * P * S / (4.0 * mi * mv * mun2 * mun2);
            }
        }
    }
    return R > 0.0 ? R : 0.0;
}

typedef struct { int type; double p[5]; } sfc_t;

static void surface_at(const scene_t *s, double x, double y, sfc_t *o) {
    const orc_config *c = s->c;
    if (c->jsfc && c->nxb > 0 && c->nyb > 0) {
        int ib = (int)floor(x / s->Lx * c->nxb), jb = (int)floor(y / s->Ly * c->nyb);
        if (ib < 0) ib = 0; if (ib >= c->nxb) ib = c->nxb - 1;
        if (jb < 0) jb = 0; if (jb >= c->nyb) jb = c->nyb - 1;
        long i = (long)jb * c->nxb + ib, n = (long)c->nxb * c->nyb;
        o->type = (int)lrintf(c->jsfc[i]);
        for (int q = 0; q < 5; ++q) o->p[q] = (double)c->psfc[q * n + i];
    } else {
        o->type = c->sfc_mtype;
        for (int q = 0; q < 5; ++q) o->p[q] = (double)c->sfc_param[q];
    }
}

static double surface_R(const sfc_t *sf, const double din[3], const double dout[3]) {
    if (sf->type == 4) return lsrt_R(sf->p[0], sf->p[1], sf->p[2], din, dout);
    if (sf->type == 2) return dsm_R(sf->p, din, dout);
    double a = sf->p[0];
    return a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
}

/* ------------------------------------------------------------------------------------------ */
/* photon state and marching                                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    double x, y, z;
    double d[3];
    int ix, iy, k;
    double w;
    int nscat;
    int ipa; /* horizontal transport switched off: solver 2 always, solver 1 (partial 3-D) after the first event */
    uint64_t seed, id; uint32_t draw; /* seed, photon id and index of its next Philox block (the roulette of local-estimate rays hashes them) */
} photon_t;

static inline double wrap(double x, double L) {
    x = fmod(x, L);
    if (x < 0.0) x += L;
    if (x >= L) x = 0.0;
    return x;
}

static inline void add_atomic(double *p, double v) {
#pragma omp atomic
    *p += v;
}

static void flux_tally(const scene_t *s, const photon_t *ph, int level, int going_up, uint64_t *cnt) {
    const orc_config *c = s->c;
    if (!(c->target & 1)) return;
    int ix = (int)floor(ph->x / c->dx), iy = (int)floor(ph->y / c->dy);
    if (ix < 0) ix = 0; if (ix >= c->nx) ix = c->nx - 1;
    if (iy < 0) iy = 0; if (iy >= c->ny) iy = c->ny - 1;
    long plane = (long)c->nx * c->ny, nlev = s->nz + 1;
    long i = ((long)level * c->ny + iy) * c->nx + ix;
    if (going_up) {
        add_atomic(&s->flux[2 * nlev * plane + i], ph->w);
    } else {
        add_atomic(&s->flux[1 * nlev * plane + i], ph->w);
        if (ph->nscat == 0) add_atomic(&s->flux[0 * nlev * plane + i], ph->w);
    }
    cnt[9]++;
}

enum { EV_COLLISION = 0, EV_SURFACE = 1, EV_ESCAPE = 2 };

/* One cell step.  Returns the geometric length to the nearest face of the current cell and
 * which face it is (0 x, 1 y, 2 z); in 1-D layers only z faces exist. */
static double face_distance(const scene_t *s, const photon_t *ph, int *axis) {
    const orc_config *c = s->c;
    double best = INFINITY; int ax = 2;
    double uz = ph->d[2];
    if (uz > 0.0) best = (s->c->zgrd[ph->k + 1] - ph->z) / uz;
    else if (uz < 0.0) best = (ph->z - s->c->zgrd[ph->k]) / (-uz);
    if (best < 0.0) best = 0.0;
    if (ph->k >= s->k3lo && ph->k < s->k3hi) {
        double ux = ph->d[0], uy = ph->d[1];
        if (ux != 0.0) {
            double sx = ux > 0.0 ? ((ph->ix + 1) * c->dx - ph->x) / ux : (ph->x - ph->ix * c->dx) / (-ux);
            if (sx < 0.0) sx = 0.0;
            if (sx < best) { best = sx; ax = 0; }
        }
        if (uy != 0.0) {
            double sy = uy > 0.0 ? ((ph->iy + 1) * c->dy - ph->y) / uy : (ph->y - ph->iy * c->dy) / (-uy);
            if (sy < 0.0) sy = 0.0;
            if (sy < best) { best = sy; ax = 1; }
        }
    }
    *axis = ax;
    return best;
}

/* move the photon a distance sgeo inside its cell (no face reached) */
static void advance_inside(const scene_t *s, photon_t *ph, double sgeo) {
    const orc_config *c = s->c;
    ph->x += ph->d[0] * sgeo; ph->y += ph->d[1] * sgeo; ph->z += ph->d[2] * sgeo;
    if (ph->k >= s->k3lo && ph->k < s->k3hi) {
        /* stay inside the cell against rounding */
        double x0 = ph->ix * c->dx, y0 = ph->iy * c->dy;
        if (ph->x < x0) ph->x = x0; if (ph->x > x0 + c->dx) ph->x = x0 + c->dx;
        if (ph->y < y0) ph->y = y0; if (ph->y > y0 + c->dy) ph->y = y0 + c->dy;
    } else if (ph->ipa) {
        /* independent columns: a 1-D layer is this column repeated for ever, so the event belongs to the column the
         * photon is in (its tallies go to that column's pixel) */
        double x0 = ph->ix * c->dx, y0 = ph->iy * c->dy;
        ph->x = x0 + wrap(ph->x - x0, c->dx); ph->y = y0 + wrap(ph->y - y0, c->dy);
    } else {
        ph->x = wrap(ph->x, s->Lx); ph->y = wrap(ph->y, s->Ly);
    }
    if (ph->z < c->zgrd[ph->k]) ph->z = c->zgrd[ph->k];
    if (ph->z > c->zgrd[ph->k + 1]) ph->z = c->zgrd[ph->k + 1];
}

/* move the photon onto the face `axis` of its cell and into the neighbouring cell.
 * returns 0 normally, EV_SURFACE / EV_ESCAPE when it leaves the atmosphere */
static int cross_face(const scene_t *s, photon_t *ph, double sgeo, int axis, int ipa) {
    const orc_config *c = s->c;
    int in3d = (ph->k >= s->k3lo && ph->k < s->k3hi);
    ph->x += ph->d[0] * sgeo; ph->y += ph->d[1] * sgeo; ph->z += ph->d[2] * sgeo;
    if (axis == 0) {
        if (ph->d[0] > 0.0) {
            if (ipa) { ph->x = ph->ix * c->dx; }
            else { ph->ix += 1; if (ph->ix >= c->nx) ph->ix = 0; ph->x = ph->ix * c->dx; }
        } else {
            if (ipa) { ph->x = (ph->ix + 1) * c->dx; }
            else { ph->ix -= 1; if (ph->ix < 0) ph->ix = c->nx - 1; ph->x = (ph->ix + 1) * c->dx; }
        }
    } else if (in3d) {
        double x0 = ph->ix * c->dx;
        if (ph->x < x0) ph->x = x0; if (ph->x > x0 + c->dx) ph->x = x0 + c->dx;
    }
    if (axis == 1) {
        if (ph->d[1] > 0.0) {
            if (ipa) { ph->y = ph->iy * c->dy; }
            else { ph->iy += 1; if (ph->iy >= c->ny) ph->iy = 0; ph->y = ph->iy * c->dy; }
        } else {
            if (ipa) { ph->y = (ph->iy + 1) * c->dy; }
            else { ph->iy -= 1; if (ph->iy < 0) ph->iy = c->ny - 1; ph->y = (ph->iy + 1) * c->dy; }
        }
    } else if (in3d) {
        double y0 = ph->iy * c->dy;
        if (ph->y < y0) ph->y = y0; if (ph->y > y0 + c->dy) ph->y = y0 + c->dy;
    }
    if (!in3d) {
        if (ipa) {
            /* independent columns: stay in the column you are in */
            double x0 = ph->ix * c->dx, y0 = ph->iy * c->dy;
            ph->x = x0 + wrap(ph->x - x0, c->dx); ph->y = y0 + wrap(ph->y - y0, c->dy);
        } else {
            ph->x = wrap(ph->x, s->Lx); ph->y = wrap(ph->y, s->Ly);
            ph->ix = (int)floor(ph->x / c->dx); if (ph->ix >= c->nx) ph->ix = c->nx - 1;
            ph->iy = (int)floor(ph->y / c->dy); if (ph->iy >= c->ny) ph->iy = c->ny - 1;
        }
    }
    if (axis == 2) {
        if (ph->d[2] > 0.0) {
            ph->k += 1; ph->z = c->zgrd[ph->k];
            if (ph->k >= s->nz) return EV_ESCAPE;
        } else {
            ph->z = c->zgrd[ph->k]; ph->k -= 1;
            if (ph->k < 0) return EV_SURFACE;
        }
    } else {
        if (ph->z < c->zgrd[ph->k]) ph->z = c->zgrd[ph->k];
        if (ph->z > c->zgrd[ph->k + 1]) ph->z = c->zgrd[ph->k + 1];
    }
    return -1;
}

/* transport flight: advance until optical path tau is used up */
static int flight(const scene_t *s, photon_t *ph, double tau, double *bt_hit, uint64_t *cnt) {
    int ipa = ph->ipa;
    for (;;) {
        int axis;
        double sgeo = face_distance(s, ph, &axis);
        double bt = cell_bt(s, ph->ix, ph->iy, ph->k);
        cnt[1]++;
        if (ph->k >= s->k3lo && ph->k < s->k3hi) cnt[2]++;
        if (bt * sgeo >= tau) {
            advance_inside(s, ph, tau / bt);
            *bt_hit = bt;
            return EV_COLLISION;
        }
        tau -= bt * sgeo;
        int up = ph->d[2] > 0.0;
        int ev = cross_face(s, ph, sgeo, axis, ipa);
        if (axis == 2) {
            int level = up ? ph->k : ph->k + 1; /* level index just crossed */
            flux_tally(s, ph, level, up, cnt);
        }
        if (ev >= 0) return ev;
    }
}

/* optical depth from the photon position along direction v to the height zstop of the sensor plane (above the start for
 * a down-looking sensor, v[2] > 0; below it for an up-looking one) or to the boundary of the atmosphere */
static double le_tau(const scene_t *s, const photon_t *from, const double v[3], double zstop, uint64_t *cnt) {
    photon_t r = *from;
    r.d[0] = v[0]; r.d[1] = v[1]; r.d[2] = v[2];
    int ipa = (s->c->solver != 0); /* scattered light of the partial 3-D solver stays in its column as well */
    double tau = 0.0;
    if (r.k < 0) r.k = 0; /* ray starts on the surface */
    for (;;) {
        if (r.k >= s->nz || r.k < 0) break;
        int axis;
        double sgeo = face_distance(s, &r, &axis);
        double bt = cell_bt(s, r.ix, r.iy, r.k);
        cnt[6]++;
        if (r.k >= s->k3lo && r.k < s->k3hi) cnt[7]++;
        double zend = r.z + r.d[2] * sgeo;
        if (r.d[2] > 0.0 ? zend >= zstop : zend <= zstop) { /* sensor plane inside this cell */
            tau += bt * (zstop - r.z) / r.d[2];
            break;
        }
        tau += bt * sgeo;
        int ev = cross_face(s, &r, sgeo, axis, ipa);
        if (ev == EV_ESCAPE || ev == EV_SURFACE) break;
    }
    return tau;
}

/* Direction of the light that reaches sensor `iv` from the photon's position: the fixed view direction of a satellite image,
 * or -- camera -- the unit vector towards the camera (nearest periodic image of it) and its distance r.  Returns 0 when the
 * sensor cannot see the point: outside the camera's cone of view, or a line of sight within 0.06 degrees of the horizontal
 * (its optical depth is found by marching to the camera's height). */
static int view_dir_img(const scene_t *s, const photon_t *ph, int iv, int ii, int jj, double v[3], double *r, double *r0sq) {
    const orc_config *c = s->c;
    if (r0sq) *r0sq = 0.0;
    if (c->rad_kind != 1) { v[0] = s->vdir[iv][0]; v[1] = s->vdir[iv][1]; v[2] = s->vdir[iv][2]; *r = 0.0; return 1; }
    double rx = s->cam_pos[iv][0] - ph->x, ry = s->cam_pos[iv][1] - ph->y, rz = s->cam_pos[iv][2] - ph->z;
    rx -= s->Lx * floor(rx / s->Lx + 0.5); ry -= s->Ly * floor(ry / s->Ly + 0.5);
    if (r0sq) *r0sq = rx * rx + ry * ry + rz * rz;   /* (squared distance of the nearest image: the roulette of the farther ones) */
    rx += ii * s->Lx; ry += jj * s->Ly;      /* the image (ii, jj) domain lengths beyond the nearest one */
    double rr = sqrt(rx * rx + ry * ry + rz * rz);
    if (!(rr > 0.0)) return 0;
    v[0] = rx / rr; v[1] = ry / rr; v[2] = rz / rr; *r = rr;
    if (fabs(v[2]) < 1e-3) return 0;
    double cz = -(v[0] * s->cam_ax[iv][2][0] + v[1] * s->cam_ax[iv][2][1] + v[2] * s->cam_ax[iv][2][2]);
    return cz >= cos(0.5 * c->cam_qmax[iv] * PI / 180.0);
}

/* Russian roulette on the images of a camera beyond the nearest one: an image at distance r is served with probability
 * (r0 / r)^2, r0 the distance of the nearest image, and then carries (r / r0)^2 times its contribution -- unbiased, and the rays per
 * event grow with the logarithm of the number of images instead of with the number.  One hashed uniform number per (event, view,
 * image), as in the camera build of k_rays.  Returns the factor on the contribution (0: not served). */
static double cam_image_roulette(const photon_t *ph, int iv, int img, double r0sq, double r) {
    if (img == 0) return 1.0;
    uint32_t h = (uint32_t)ph->id ^ ((uint32_t)(ph->id >> 32) * 0x9E3779B9u) ^ (ph->draw * 0x85EBCA6Bu)
                 ^ ((uint32_t)(iv + 64 * img + 16 + 1) * 0xC2B2AE35u) ^ (uint32_t)ph->seed;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    double u = ((double)(h >> 9) + 0.5) * (1.0 / 8388608.0);
    double rsq = r * r;
    if (!(u * rsq < r0sq)) return 0.0;
    return rsq / r0sq;
}

/* The images of a camera an event contributes to: index 0 is the nearest one, 1 ... (2N+1)^2 - 1 the others row by row
 * (N = cam_images); the same enumeration as the camera build of k_rays (er3t_amd/csrc/mi3d_kernel_rays.hip). */
static int cam_nimg(const orc_config *c) { return c->rad_kind == 1 ? (2 * c->cam_images + 1) * (2 * c->cam_images + 1) : 1; }
static void cam_image(const orc_config *c, int img, int *ii, int *jj) {
    const int n = 2 * c->cam_images + 1;
    if (img == 0) { *ii = 0; *jj = 0; return; }
    int t = img - 1;
    if (t >= (n * n - 1) / 2) t += 1;          /* (skip the centre: it is index 0) */
    *ii = t % n - c->cam_images; *jj = t / n - c->cam_images;
}

/* camera: the ray from the event reaches the camera at distance r; tally into the pixel its direction falls in */
static void camera_tally(const scene_t *s, const photon_t *ph, double contrib_no_T, int iv, int img, const double v[3], double r, uint64_t *cnt) {
    const orc_config *c = s->c;
    double ztoa = c->zgrd[s->nz];
    double zc = s->cam_pos[iv][2];
    cnt[5]++;
    double tau = le_tau(s, ph, v, zc < ztoa ? zc : INFINITY, cnt);
    double T = exp(-tau);
    if (c->le_tau1 > 0.0) {
        uint32_t h = (uint32_t)ph->id ^ ((uint32_t)(ph->id >> 32) * 0x9E3779B9u) ^ (ph->draw * 0x85EBCA6Bu)
                     ^ ((uint32_t)(iv + 64 * img + 1) * 0xC2B2AE35u) ^ (uint32_t)ph->seed;   /* (a ray of its own per image) */
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        double u = ((double)(h >> 9) + 0.5) * (1.0 / 8388608.0);
        if (tau > c->le_tau1 - log(u)) return;
        T = exp(-(tau < c->le_tau1 ? tau : c->le_tau1));
    }
    /* direction the camera looks in to see the event, in camera coordinates; polar map U = theta cos(phi), V = theta sin(phi) */
    double dxc = -(v[0] * s->cam_ax[iv][0][0] + v[1] * s->cam_ax[iv][0][1] + v[2] * s->cam_ax[iv][0][2]);
    double dyc = -(v[0] * s->cam_ax[iv][1][0] + v[1] * s->cam_ax[iv][1][1] + v[2] * s->cam_ax[iv][1][2]);
    double dzc = -(v[0] * s->cam_ax[iv][2][0] + v[1] * s->cam_ax[iv][2][1] + v[2] * s->cam_ax[iv][2][2]);
    if (dzc > 1.0) dzc = 1.0;
    double theta = acos(dzc), rho = sqrt(dxc * dxc + dyc * dyc);
    double U = rho > 1e-12 ? theta * dxc / rho : 0.0, V = rho > 1e-12 ? theta * dyc / rho : 0.0;
    double du = c->cam_umax[iv] * PI / 180.0 / c->nxr, dv = c->cam_vmax[iv] * PI / 180.0 / c->nyr;
    int ir = (int)floor(U / du + 0.5 * c->nxr), jr = (int)floor(V / dv + 0.5 * c->nyr);
    if (ir < 0 || ir >= c->nxr || jr < 0 || jr >= c->nyr) return;
    /* radiance = power per unit area normal to the ray and per unit solid angle: the point source of the local estimate gives
     * contrib T / r^2 per unit area at the camera (r not below the aperture size, Rad_apsize); the solid angle of the patch
     * dU dV of the polar map is (sin theta / theta) dU dV */
    double a = c->cam_apsize[iv], r2 = r * r > a * a ? r * r : a * a;
    double sinc = theta > 1e-6 ? sin(theta) / theta : 1.0;
    add_atomic(&s->rad[((long)iv * c->nyr + jr) * c->nxr + ir], contrib_no_T * T / (r2 * sinc * du * dv));
}

static void radiance_tally(const scene_t *s, const photon_t *ph, double contrib_no_T, int iv, uint64_t *cnt) {
    const orc_config *c = s->c;
    const double *v = s->vdir[iv];
    double ztoa = c->zgrd[s->nz];
    double zs = c->view_zloc[iv] < ztoa ? c->view_zloc[iv] : ztoa;
    if (v[2] > 0.0) {
        if (ph->z >= zs && !(ph->z == ztoa && zs == ztoa)) return; /* down-looking sensor: event above it */
    } else {
        if (zs < c->zgrd[0]) zs = c->zgrd[0];
        if (ph->z <= zs) return;                                     /* up-looking sensor: event below (or level with) it */
    }
    cnt[5]++;
    if (c->le_cmin > 0.0 && !(v[2] >= 1.0 && c->view_zloc[iv] >= ztoa) && contrib_no_T < c->le_cmin) {
        /* Russian roulette on the weight the ray would carry (not for views answered from the column table, which cost nothing):
         * most local estimates look away from the forward peak of the phase function and carry a few per cent of what the few
         * near it carry; marching only a share c / cmin of them, at weight cmin, leaves the mean alone and the noise where the
         * large contributions put it.  Its own uniform number: the hash of the optical-depth roulette with the view moved on by 16. */
        uint32_t h = (uint32_t)ph->id ^ ((uint32_t)(ph->id >> 32) * 0x9E3779B9u) ^ (ph->draw * 0x85EBCA6Bu)
                     ^ ((uint32_t)(iv + 16 + 1) * 0xC2B2AE35u) ^ (uint32_t)ph->seed;
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        double u = ((double)(h >> 9) + 0.5) * (1.0 / 8388608.0);
        if (!(u * c->le_cmin < contrib_no_T)) return;
        contrib_no_T = c->le_cmin;
    }
    double tau = le_tau(s, ph, v, zs, cnt);
    double T = exp(-tau);
    if (c->le_tau1 > 0.0 && !(v[2] >= 1.0 && c->view_zloc[iv] >= ztoa)) {
        /* Russian roulette on the ray (not for exactly vertical views of a sensor above the atmosphere, whose optical depth
         * comes from a table at no cost): the ray survives to optical depth tau with probability min(1, exp(-(tau - tau1)))
         * and then carries exp(-tau)/that = exp(-min(tau, tau1)).  One uniform number per ray, a hash of (seed, photon id,
         * index of the photon's next Philox block, view), decides: the ray ends at tau_kill = tau1 - ln u. */
        uint32_t h = (uint32_t)ph->id ^ ((uint32_t)(ph->id >> 32) * 0x9E3779B9u) ^ (ph->draw * 0x85EBCA6Bu)
                     ^ ((uint32_t)(iv + 1) * 0xC2B2AE35u) ^ (uint32_t)ph->seed;
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        double u = ((double)(h >> 9) + 0.5) * (1.0 / 8388608.0);
        if (tau > c->le_tau1 - log(u)) return;
        T = exp(-(tau < c->le_tau1 ? tau : c->le_tau1));
    }
    double xr = ph->x, yr = ph->y;
    if (c->solver == 0) {
        /* the pixel is where the line of sight meets the reference level (down-looking: Rad_zref, the level the image is
         * geolocated on) or the sensor's own level (up-looking: where the instrument stands) */
        double zreg = v[2] > 0.0 ? c->zref : zs;
        xr = wrap(ph->x - v[0] / v[2] * (ph->z - zreg), s->Lx);
        yr = wrap(ph->y - v[1] / v[2] * (ph->z - zreg), s->Ly);
    }
    int ir = (int)floor(xr / s->Lx * c->nxr), jr = (int)floor(yr / s->Ly * c->nyr);
    if (ir < 0) ir = 0; if (ir >= c->nxr) ir = c->nxr - 1;
    if (jr < 0) jr = 0; if (jr >= c->nyr) jr = c->nyr - 1;
    add_atomic(&s->rad[((long)iv * c->nyr + jr) * c->nxr + ir], contrib_no_T * T / fabs(v[2]));
}

/* ------------------------------------------------------------------------------------------ */
static void run_photon(const scene_t *s, uint64_t seed, uint64_t id, uint64_t *cnt) {
    const orc_config *c = s->c;
    uint32_t draw = 0;
    double u[4];
    photon_t ph;
    draw4(seed, id, draw++, u);
    ph.x = u[0] * s->Lx; ph.y = u[1] * s->Ly;
    if (ph.x >= s->Lx) ph.x = 0.0;
    if (ph.y >= s->Ly) ph.y = 0.0;
    ph.ix = (int)floor(ph.x / c->dx); if (ph.ix >= c->nx) ph.ix = c->nx - 1;
    ph.iy = (int)floor(ph.y / c->dy); if (ph.iy >= c->ny) ph.iy = c->ny - 1;
    ph.k = s->nz - 1; ph.z = c->zgrd[s->nz];
    ph.d[0] = s->sdir[0]; ph.d[1] = s->sdir[1]; ph.d[2] = s->sdir[2];
    if (s->cos_cone < 1.0) {
        double mu = 1.0 - u[2] * (1.0 - s->cos_cone);
        rotate_dir(ph.d, mu, 2.0 * PI * u[3]);
    }
    ph.w = 1.0; ph.nscat = 0; ph.ipa = (c->solver == 2);
    flux_tally(s, &ph, s->nz, 0, cnt);

    ph.seed = seed; ph.id = id;
    for (;;) {
        draw4(seed, id, draw++, u);
        ph.draw = draw;
        double tau = -log(u[0]);
        double bt = 0.0;
        int ev = flight(s, &ph, tau, &bt, cnt);
        if (ev == EV_ESCAPE) { cnt[12]++; break; }
        if (ev == EV_SURFACE) {
            cnt[4]++;
            ph.k = 0; ph.z = c->zgrd[0];
            sfc_t sf;
            surface_at(s, ph.x, ph.y, &sf);
            if (c->target & 2) {
                photon_t q = ph; q.k = 0;
                for (int iv = 0; iv < c->nview; ++iv)
                for (int img = 0; img < cam_nimg(c); ++img) {
                    double v[3], r;
                    int ii, jj;
                    cam_image(c, img, &ii, &jj);
                    double r0sq;
                    if (!view_dir_img(s, &q, iv, ii, jj, v, &r, &r0sq)) continue;
                    if (v[2] <= 0.0) continue; /* an up-looking sensor does not see the surface */
                    double R = surface_R(&sf, ph.d, v);
                    if (!(R > 0.0)) continue;
                    double fimg = c->rad_kind == 1 ? cam_image_roulette(&q, iv, img, r0sq, r) : 1.0;
                    if (!(fimg > 0.0)) continue;
                    if (c->rad_kind == 1) camera_tally(s, &q, fimg * ph.w * R * v[2] / PI, iv, img, v, r, cnt);
                    else radiance_tally(s, &q, ph.w * R * v[2] / PI, iv, cnt);
                }
            }
            double nd[3] = {0.0, 0.0, 1.0};
            rotate_dir(nd, sqrt(u[2]), 2.0 * PI * u[3]);
            if (nd[2] < 1e-9) nd[2] = 1e-9;
            ph.w *= surface_R(&sf, ph.d, nd);
            ph.d[0] = nd[0]; ph.d[1] = nd[1]; ph.d[2] = nd[2];
            ph.nscat++;
            if (ph.w <= 0.0) { cnt[13]++; break; }
            flux_tally(s, &ph, 0, 1, cnt);
        } else {
            /* collision: gather the scattering coefficients of all components of this cell */
            double ks[8], apf[8]; int ncomp = 0; double kstot = 0.0;
            for (int ip = 0; ip < c->np1d; ++ip) {
                ks[ncomp] = s->ks1d[(long)ip * s->nz + ph.k];
                apf[ncomp] = (double)c->apf1d[(long)ip * s->nz + ph.k];
                kstot += ks[ncomp++];
            }
            if (ph.k >= s->k3lo && ph.k < s->k3hi) {
                long v = vox(s, ph.ix, ph.iy, ph.k - s->k3lo), nvox = (long)c->nx * c->ny * c->nz3;
                for (int ip = 0; ip < c->np3d; ++ip) {
                    ks[ncomp] = (double)c->omgp[ip * nvox + v] * (double)c->extp[ip * nvox + v];
                    apf[ncomp] = (double)c->apfp[ip * nvox + v];
                    kstot += ks[ncomp++];
                }
            }
            cnt[3]++;
            if (s->heat) {
                /* heating rates (Flx_mhrt = 1, mca_inp.py:124): what the collision takes from the photon's weight -- gas absorption
                 * and the absorbing part of every constituent -- stays in this cell.  The column is where the photon IS (inside a
                 * 1-D layer the cell index is only brought up to date at level crossings), as for the flux tallies. */
                int ix = (int)floor(ph.x / c->dx), iy = (int)floor(ph.y / c->dy);
                if (ix < 0) ix = 0; if (ix >= c->nx) ix = c->nx - 1;
                if (iy < 0) iy = 0; if (iy >= c->ny) iy = c->ny - 1;
                if (ph.k >= s->k3lo && ph.k < s->k3hi) { ix = ph.ix; iy = ph.iy; }
                double dw = ph.w * (1.0 - kstot / bt);
                if (dw > 0.0) add_atomic(&s->heat[((long)ph.k * c->ny + iy) * c->nx + ix], dw);
            }
            ph.w *= kstot / bt;
            if (!(ph.w > 0.0)) { cnt[13]++; break; }
            if (c->target & 2) {
                for (int iv = 0; iv < c->nview; ++iv)
                for (int img = 0; img < cam_nimg(c); ++img) {
                    double v[3], r;
                    int ii, jj;
                    cam_image(c, img, &ii, &jj);
                    double r0sq;
                    if (!view_dir_img(s, &ph, iv, ii, jj, v, &r, &r0sq)) continue;
                    double fimg = c->rad_kind == 1 ? cam_image_roulette(&ph, iv, img, r0sq, r) : 1.0;
                    if (!(fimg > 0.0)) continue;
                    double mu = ph.d[0] * v[0] + ph.d[1] * v[1] + ph.d[2] * v[2];
                    double P = 0.0;
                    for (int q = 0; q < ncomp; ++q)
                        if (ks[q] > 0.0) P += ks[q] * phase_eval(s, apf[q], mu);
                    P /= kstot;
                    if (c->rad_kind == 1) camera_tally(s, &ph, fimg * ph.w * P / (4.0 * PI), iv, img, v, r, cnt);
                    else radiance_tally(s, &ph, ph.w * P / (4.0 * PI), iv, cnt);
                }
            }
            /* choose the scattering component */
            double target = u[1] * kstot, acc = 0.0, usel = 0.0; int sel = ncomp - 1;
            for (int q = 0; q < ncomp; ++q) {
                if (target < acc + ks[q] || q == ncomp - 1) {
                    sel = q;
                    usel = ks[q] > 0.0 ? (target - acc) / ks[q] : 0.0;
                    break;
                }
                acc += ks[q];
            }
            if (usel < 0.0) usel = 0.0; if (usel > 1.0) usel = 1.0;
            double mu = phase_sample(s, apf[sel], u[2], usel);
            rotate_dir(ph.d, mu, 2.0 * PI * u[3]);
            ph.nscat++;
        }
        if (c->solver == 1 && !ph.ipa) {
            /* first event behind us: from here on independent columns, namely the column of that event
             * (inside a 1-D layer the column index is only brought up to date at level crossings) */
            ph.ipa = 1;
            if (!(ph.k >= s->k3lo && ph.k < s->k3hi)) {
                ph.ix = (int)floor(ph.x / c->dx); if (ph.ix >= c->nx) ph.ix = c->nx - 1; if (ph.ix < 0) ph.ix = 0;
                ph.iy = (int)floor(ph.y / c->dy); if (ph.iy >= c->ny) ph.iy = c->ny - 1; if (ph.iy < 0) ph.iy = 0;
            }
        }
        if (ph.w < c->wmin) {
            cnt[10]++;
            draw4(seed, id, draw++, u);
            if (u[0] * c->wfac < ph.w) ph.w = c->wfac;
            else { cnt[11]++; break; }
        }
    }
    cnt[0]++;
}

/* ------------------------------------------------------------------------------------------ */
static int build_scene(scene_t *s, const orc_config *c, double *rad, double *flux, double *heat) {
    memset(s, 0, sizeof(*s));
    s->c = c; s->nz = c->nz;
    s->k3lo = c->nz3 > 0 ? c->iz3l - 1 : 0;
    s->k3hi = c->nz3 > 0 ? s->k3lo + c->nz3 : 0;
    if (c->nz3 > 0 && (s->k3lo < 0 || s->k3hi > c->nz)) return -1;
    s->Lx = c->nx * c->dx; s->Ly = c->ny * c->dy;
    s->bt1d = (double *)calloc(c->nz, sizeof(double));
    s->ks1d = (double *)calloc((size_t)c->nz * (c->np1d > 0 ? c->np1d : 1), sizeof(double));
    for (int k = 0; k < c->nz; ++k) {
        double b = c->abs1d ? (double)c->abs1d[k] : 0.0;
        for (int ip = 0; ip < c->np1d; ++ip) {
            double e = (double)c->ext1d[(long)ip * c->nz + k];
            b += e;
            s->ks1d[(long)ip * c->nz + k] = e * (double)c->omg1d[(long)ip * c->nz + k];
        }
        s->bt1d[k] = b;
    }
    s->nang = c->nang; s->npf = c->npf;
    if (c->npf > 0) {
        int n = c->nang;
        s->tmu = (double *)malloc(sizeof(double) * n);
        s->tp = (double *)malloc(sizeof(double) * n * c->npf);
        s->tcdf = (double *)malloc(sizeof(double) * n * c->npf);
        for (int j = 0; j < n; ++j) s->tmu[j] = cos((double)c->ang[n - 1 - j] * PI / 180.0);
        s->tmu[0] = -1.0; s->tmu[n - 1] = 1.0;
        for (int t = 0; t < c->npf; ++t) {
            double *p = s->tp + (long)t * n, *cdf = s->tcdf + (long)t * n;
            for (int j = 0; j < n; ++j) p[j] = (double)c->pha[(long)t * n + (n - 1 - j)];
            double tot = 0.0;
            for (int j = 1; j < n; ++j) tot += 0.25 * (p[j] + p[j - 1]) * (s->tmu[j] - s->tmu[j - 1]);
            for (int j = 0; j < n; ++j) p[j] /= tot;
            cdf[0] = 0.0;
            for (int j = 1; j < n; ++j) cdf[j] = cdf[j - 1] + 0.25 * (p[j] + p[j - 1]) * (s->tmu[j] - s->tmu[j - 1]);
            cdf[n - 1] = 1.0;
        }
    }
    double th = c->src_the * PI / 180.0, phi = c->src_phi * PI / 180.0;
    s->sdir[0] = sin(th) * cos(phi); s->sdir[1] = sin(th) * sin(phi); s->sdir[2] = cos(th);
    s->cos_cone = cos(0.5 * c->src_qmax * PI / 180.0);
    for (int iv = 0; iv < c->nview; ++iv) {
        double t = c->view_the[iv] * PI / 180.0, p = c->view_phi[iv] * PI / 180.0;
        s->vdir[iv][0] = -sin(t) * cos(p); s->vdir[iv][1] = -sin(t) * sin(p); s->vdir[iv][2] = -cos(t);
        if ((c->target & 2) && c->rad_kind != 1 && fabs(s->vdir[iv][2]) <= 1e-6) return -2; /* no horizontal lines of sight */
        if (c->rad_kind == 1) {
            double q = c->cam_psi[iv] * PI / 180.0;
            double ct = cos(t), st = sin(t), cp = cos(p), sp = sin(p), cq = cos(q), sq = sin(q);
            double *X = s->cam_ax[iv][0], *Y = s->cam_ax[iv][1], *Z = s->cam_ax[iv][2];
            Z[0] = st * cp; Z[1] = st * sp; Z[2] = ct;                                  /* Rz(phi) Ry(the) Rz(psi) applied to z */
            X[0] = cp * ct * cq - sp * sq; X[1] = sp * ct * cq + cp * sq; X[2] = -st * cq; /* ... to x */
            Y[0] = Z[1] * X[2] - Z[2] * X[1]; Y[1] = Z[2] * X[0] - Z[0] * X[2]; Y[2] = Z[0] * X[1] - Z[1] * X[0];
            s->cam_pos[iv][0] = c->cam_xpos[iv] * s->Lx; s->cam_pos[iv][1] = c->cam_ypos[iv] * s->Ly;
            s->cam_pos[iv][2] = c->view_zloc[iv];
            if (c->solver != 0) return -3;  /* a point sensor needs the 3-D solver */
        }
    }
    s->rad = rad; s->flux = flux; s->heat = (c->target & 4) ? heat : NULL;
    return 0;
}

static void free_scene(scene_t *s) {
    free(s->bt1d); free(s->ks1d); free(s->tmu); free(s->tp); free(s->tcdf);
}

/* Raw tallies (sums of weights / local-estimate contributions), double precision:
 *   rad_sum [nview][nyr][nxr], flux_sum [3][nz+1][ny][nx], counters[ORC_NCOUNTER] — all ADDED to. */
int orc_run_heat(const orc_config *c, uint64_t nphoton, uint64_t seed, uint64_t offset, double *rad_sum,
                 double *flux_sum, double *heat_sum, uint64_t *counters);

int orc_run(const orc_config *c, uint64_t nphoton, uint64_t seed, uint64_t offset, double *rad_sum,
            double *flux_sum, uint64_t *counters) {
    return orc_run_heat(c, nphoton, seed, offset, rad_sum, flux_sum, NULL, counters);
}

/* ... and heat_sum [nz][ny][nx]: weight absorbed per cell (tallied when c->target & 4), or NULL */
int orc_run_heat(const orc_config *c, uint64_t nphoton, uint64_t seed, uint64_t offset, double *rad_sum,
                 double *flux_sum, double *heat_sum, uint64_t *counters) {
    scene_t s;
    int rc = build_scene(&s, c, rad_sum, flux_sum, heat_sum);
    if (rc) return rc;
    int nt = c->nthreads > 0 ? c->nthreads : 1;
#ifdef _OPENMP
    omp_set_num_threads(nt);
#endif
    uint64_t total[ORC_NCOUNTER] = {0};
#pragma omp parallel
    {
        uint64_t cnt[ORC_NCOUNTER] = {0};
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)nphoton; ++i) run_photon(&s, seed, offset + (uint64_t)i, cnt);
#pragma omp critical
        for (int q = 0; q < ORC_NCOUNTER; ++q) total[q] += cnt[q];
    }
    for (int q = 0; q < ORC_NCOUNTER; ++q) counters[q] += total[q];
    free_scene(&s);
    return 0;
}

/* helpers exported for unit tests of the pieces */
double orc_lsrt(double fiso, double fgeo, double fvol, const double din[3], const double dout[3]) {
    return lsrt_R(fiso, fgeo, fvol, din, dout);
}

double orc_fresnel(double nr, double ni, double cos_inc) { return fresnel_unpolarised(nr, ni, cos_inc); }

/* reflectance factor of the diffuse-specular mixture for n outgoing directions */
void orc_dsm(const double p[5], const double din[3], int n, const double *dout, double *out) {
    for (int i = 0; i < n; ++i) out[i] = dsm_R(p, din, dout + 3 * i);
}

int orc_phase_table(const orc_config *c, int itable, int n, const double *mu, const double *u, double *p_out,
                    double *mu_out) {
    scene_t s;
    orc_config cc = *c;
    cc.nview = 0;
    if (build_scene(&s, &cc, NULL, NULL, NULL)) return -1;
    for (int i = 0; i < n; ++i) {
        p_out[i] = table_eval(&s, itable, mu[i]);
        mu_out[i] = table_sample(&s, itable, u[i]);
    }
    free_scene(&s);
    return 0;
}

double orc_phase_eval(double apf, double mu) {
    scene_t s; memset(&s, 0, sizeof(s));
    return phase_eval(&s, apf, mu);
}
double orc_phase_sample(double apf, double u) {
    scene_t s; memset(&s, 0, sizeof(s));
    return phase_sample(&s, apf, u, 0.0);
}
