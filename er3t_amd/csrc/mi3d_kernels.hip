// mi3d_kernels.hip — CDNA4 (gfx950) kernels of the photon-transport hot path.
//
// Replaces the main loop of the external solver the reference launches as
// "<exe> <Nphoton> <solver> <inp> <out>" (er3t/rtm/mca/mca_run.py:113).  No reference kernels
// exist to mirror; the algorithm is the forward Monte Carlo / local-estimate method the
// reference cites (er3t/rtm/mca/mcarats.py:59) on er3t's input contract (see mi3d_device.h).
//
// Kernels
//   k_build_grid    file-layout 3-D arrays -> z-fastest total extinction + collision records
//   k_build_column  per-column optical depth from every 3-D level up to the top of atmosphere
//   k_transport     persistent photon loop: Philox draws, cell march, collisions, local-estimate
//                   radiance tally, flux tally
//   k_philox        test hook
//
// Random-number protocol, geometry and estimator are specified in DESIGN.md §3 and restated
// independently (double precision) in oracle/mi3d_oracle.c.
#include "mi3d_device.h"

namespace mi3d {

// ---------------------------------------------------------------------------------------------
// scene builders
// ---------------------------------------------------------------------------------------------
// One thread per voxel, x fastest on the read side (coalesced reads of the file-layout arrays).
__global__ void __launch_bounds__(256)
k_build_grid(int nx, int ny, int nz3, int k3lo, int np3d, const LayerRec *lay, const float *abst,
             const float *extp, const float *omgp, const float *apfp, float *bext, float2 *csca) {
    const long nvox = (long)nx * ny * nz3;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x; // file index: (k3*ny + iy)*nx + ix
    if (v >= nvox) return;
    const int ix = (int)(v % nx);
    const int iy = (int)((v / nx) % ny);
    const int k3 = (int)(v / ((long)nx * ny));
    float bt = lay[k3lo + k3].bt1d;
    if (abst) bt += abst[v];
    const long o = ((long)iy * nx + ix) * nz3 + k3;
    for (int ip = 0; ip < np3d; ++ip) {
        const float e = extp[ip * nvox + v];
        bt += e;
        csca[o * np3d + ip] = make_float2(omgp[ip * nvox + v] * e, apfp[ip * nvox + v]);
    }
    bext[o] = fmaxf(bt, 0.0f);
}

// One thread per column: tcol[c][j] = vertical optical depth from level k3lo+j to TOA.
__global__ void __launch_bounds__(256)
k_build_column(int ncol, int nz3, int k3lo, int nz, const LayerRec *lay, const float *bext, float *tcol) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int k3hi = k3lo + nz3;
    float tau = 0.0f;
    for (int k = nz - 1; k >= k3hi; --k) tau += lay[k].bt1d * lay[k].dz;
    float *t = tcol + (long)c * (nz3 + 1);
    t[nz3] = tau;
    for (int k3 = nz3 - 1; k3 >= 0; --k3) {
        tau += bext[(long)c * nz3 + k3] * lay[k3lo + k3].dz;
        t[k3] = tau;
    }
}

__global__ void k_philox(uint64_t seed, uint64_t id0, uint32_t draw, int n, uint32_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t id = id0 + (uint64_t)i;
    uint32_t w[4];
    philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), draw, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    out[4 * i + 0] = w[0]; out[4 * i + 1] = w[1]; out[4 * i + 2] = w[2]; out[4 * i + 3] = w[3];
}

// ---------------------------------------------------------------------------------------------
// transport
// ---------------------------------------------------------------------------------------------
struct Ray {
    float px, py, pz;  // position inside the current cell: [0,dx] x [0,dy] x [0,dz_k]
    float ux, uy, uz;  // direction of travel
    int ix, iy, k;     // column and layer
};

struct Counters {
    uint32_t steps, steps3d, scatter, surface, le_rays, le_steps, le_steps3d, le_column, flux_tally,
        roulette, killed, escaped, absorbed, photons;
};

__device__ inline int wrapi(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

// Horizontal bookkeeping inside horizontally homogeneous (1-D) layers: fold the local position
// back into [0,dx) and move the column index with it (3-D / IPA aware).
__device__ inline void fold_xy(const DevScene &S, Ray &r, bool ipa) {
    const float fx = floorf(r.px / S.dx), fy = floorf(r.py / S.dy);
    r.px = fminf(fmaxf(r.px - fx * S.dx, 0.0f), S.dx);
    r.py = fminf(fmaxf(r.py - fy * S.dy, 0.0f), S.dy);
    if (!ipa) {
        if (fx != 0.0f) r.ix = wrapi(r.ix + (int)fx, S.nx);
        if (fy != 0.0f) r.iy = wrapi(r.iy + (int)fy, S.ny);
    }
}

// Geometric distance to the nearest face of the current cell; axis 0/1/2 = x/y/z.
__device__ inline float face_dist(const DevScene &S, const Ray &r, float dz, bool in3d, float iux,
                                  float iuy, float iuz, int &axis) {
    float s = (r.uz > 0.0f ? dz - r.pz : r.pz) * iuz;
    axis = 2;
    if (in3d) {
        const float sx = (r.ux > 0.0f ? S.dx - r.px : r.px) * iux;
        const float sy = (r.uy > 0.0f ? S.dy - r.py : r.py) * iuy;
        if (sx < s) { s = sx; axis = 0; }
        if (sy < s) { s = sy; axis = 1; }
    }
    return fmaxf(s, 0.0f);
}

// Move onto face `axis` and into the neighbour cell.  Returns +1 left through the top,
// -1 reached the surface, 0 otherwise.  `lay` is the LDS layer table.
__device__ inline int cross_face(const DevScene &S, const LayerRec *lay, Ray &r, float s, int axis,
                                 bool in3d, bool ipa, float dz) {
    r.px += r.ux * s; r.py += r.uy * s; r.pz += r.uz * s;
    if (in3d) {
        if (axis == 0) {
            if (r.ux > 0.0f) { r.px = 0.0f; if (!ipa) { r.ix += 1; if (r.ix >= S.nx) r.ix = 0; } }
            else { r.px = S.dx; if (!ipa) { r.ix -= 1; if (r.ix < 0) r.ix = S.nx - 1; } }
        } else {
            r.px = fminf(fmaxf(r.px, 0.0f), S.dx);
        }
        if (axis == 1) {
            if (r.uy > 0.0f) { r.py = 0.0f; if (!ipa) { r.iy += 1; if (r.iy >= S.ny) r.iy = 0; } }
            else { r.py = S.dy; if (!ipa) { r.iy -= 1; if (r.iy < 0) r.iy = S.ny - 1; } }
        } else {
            r.py = fminf(fmaxf(r.py, 0.0f), S.dy);
        }
    } else {
        fold_xy(S, r, ipa);
    }
    if (axis == 2) {
        if (r.uz > 0.0f) {
            r.k += 1; r.pz = 0.0f;
            if (r.k >= S.nz) return 1;
        } else {
            r.k -= 1;
            if (r.k < 0) { r.pz = 0.0f; return -1; }
            r.pz = lay[r.k].dz;
        }
    } else {
        r.pz = fminf(fmaxf(r.pz, 0.0f), dz);
    }
    return 0;
}

template <bool COUNT>
__device__ inline void flux_add(const DevScene &S, const Ray &r, float w, bool direct, int level, bool up,
                                Counters &cnt) {
    const long plane = (long)S.nx * S.ny, nlev = S.nz + 1;
    const long i = ((long)level * S.ny + r.iy) * S.nx + r.ix;
    if (up) {
        atomicAdd(&S.flux[2 * nlev * plane + i], w);
    } else {
        atomicAdd(&S.flux[1 * nlev * plane + i], w);
        if (direct) atomicAdd(&S.flux[i], w);
    }
    if (COUNT) cnt.flux_tally++;
}

// Local estimate: contribution of an event at ray position `e` (direction fields unused) towards
// view iv.  `contrib` = w*P/(4π) for a scattering event or w*R*μv/π for a surface reflection;
// `bt_here` is the total extinction of the event's cell.
template <bool COUNT>
__device__ inline void le_tally(const DevScene &S, const LayerRec *lay, const Ray &e, float bt_here,
                                float contrib, int iv, bool ipa, Counters &cnt) {
    const float vx = S.vdir[iv][0], vy = S.vdir[iv][1], vz = S.vdir[iv][2];
    const float zs = S.vzs[iv];
    const LayerRec L0 = lay[e.k];
    const float z = L0.zlo + e.pz;
    if (z >= zs) return;
    if (COUNT) cnt.le_rays++;
    float tau;
    const bool in3d0 = (e.k >= S.k3lo && e.k < S.k3hi);
    if (S.vcol[iv]) {
        // exactly vertical line of sight to a sensor above the atmosphere: the optical depth is
        // the rest of this cell plus a per-column table entry
        const long c = (long)e.iy * S.nx + e.ix;
        tau = bt_here * (L0.dz - e.pz);
        if (in3d0) {
            tau += S.tcol[c * (S.nz3 + 1) + (e.k - S.k3lo) + 1];
        } else {
            tau += L0.tabove;
            if (e.k < S.k3lo && S.nz3 > 0) tau += S.tcol[c * (S.nz3 + 1)];
        }
        if (COUNT) cnt.le_column++;
    } else {
        Ray r = e;
        r.ux = vx; r.uy = vy; r.uz = vz;
        const float iux = 1.0f / fmaxf(fabsf(vx), 1e-20f), iuy = 1.0f / fmaxf(fabsf(vy), 1e-20f),
                    iuz = 1.0f / vz;
        tau = 0.0f;
        float bt = bt_here;
        int guard = 0;
        for (;;) {
            const LayerRec L = lay[r.k];
            const bool in3d = (r.k >= S.k3lo && r.k < S.k3hi);
            int axis;
            const float s = face_dist(S, r, L.dz, in3d, iux, iuy, iuz, axis);
            if (COUNT) { cnt.le_steps++; if (in3d) cnt.le_steps3d++; }
            const float zend = L.zlo + r.pz + vz * s;
            if (zend >= zs && zs < S.ztoa) { // sensor inside this cell
                tau += bt * (zs - (L.zlo + r.pz)) * iuz;
                break;
            }
            tau += bt * s;
            if (tau > kTauCut) break;
            if (cross_face(S, lay, r, s, axis, in3d, ipa, L.dz) != 0) break;
            if (++guard > (1 << 22)) break;
            bt = (r.k >= S.k3lo && r.k < S.k3hi)
                     ? S.bext[((long)r.iy * S.nx + r.ix) * S.nz3 + (r.k - S.k3lo)]
                     : lay[r.k].bt1d;
        }
    }
    if (tau > kTauCut) return;
    const float T = __expf(-tau);
    // pixel registration: where the line of sight meets z = zref
    float xr = (float)e.ix * S.dx + e.px, yr = (float)e.iy * S.dy + e.py;
    if (!ipa) {
        const float t = (z - S.zref) / vz;
        xr -= vx * t; yr -= vy * t;
        xr -= floorf(xr / S.Lx) * S.Lx; yr -= floorf(yr / S.Ly) * S.Ly;
    }
    int ir = (int)(xr / S.Lx * (float)S.nxr), jr = (int)(yr / S.Ly * (float)S.nyr);
    ir = min(max(ir, 0), S.nxr - 1); jr = min(max(jr, 0), S.nyr - 1);
    atomicAdd(&S.rad[((long)iv * S.nyr + jr) * S.nxr + ir], contrib * T / vz);
}

template <bool COUNT>
__global__ void __launch_bounds__(256)
k_transport(const DevScene S, const uint64_t nphoton, const uint64_t seed, const uint64_t offset) {
    extern __shared__ float4 smem[];
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(smem);
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.lay);
        for (int i = threadIdx.x; i < S.nz * (kLayStride / 4); i += blockDim.x) smem[i] = src[i];
    }
    __syncthreads();

    const bool ipa = (S.solver == MI3D_SOLVER_IPA);
    const bool do_flux = (S.target & MI3D_TARGET_FLUX) != 0;
    const bool do_rad = (S.target & MI3D_TARGET_RADIANCE) != 0 && S.nview > 0;
    Counters cnt = {};

    for (;;) {
        const unsigned long long n = atomicAdd(S.next_photon, 1ull);
        if (n >= nphoton) break;
        const uint64_t id = offset + n;
        uint32_t draw = 0;
        float u[4];
        draw4(seed, id, draw++, u);

        // ---- launch at the top of the atmosphere
        Ray r;
        {
            float x = u[0] * S.Lx, y = u[1] * S.Ly;
            if (x >= S.Lx) x = 0.0f;
            if (y >= S.Ly) y = 0.0f;
            r.ix = min((int)(x / S.dx), S.nx - 1);
            r.iy = min((int)(y / S.dy), S.ny - 1);
            r.px = fminf(fmaxf(x - (float)r.ix * S.dx, 0.0f), S.dx);
            r.py = fminf(fmaxf(y - (float)r.iy * S.dy, 0.0f), S.dy);
        }
        r.k = S.nz - 1;
        r.pz = lay[r.k].dz;
        r.ux = S.sdir[0]; r.uy = S.sdir[1]; r.uz = S.sdir[2];
        if (S.cos_cone < 1.0f) rotate_dir(r.ux, r.uy, r.uz, 1.0f - u[2] * (1.0f - S.cos_cone), u[3]);
        float w = 1.0f;
        bool direct = true;
        if (do_flux) flux_add<COUNT>(S, r, w, true, S.nz, false, cnt);

        // ---- history
        bool alive = true;
        while (alive) {
            draw4(seed, id, draw++, u);
            float tau = -__logf(u[0]);
            const float iux = 1.0f / fmaxf(fabsf(r.ux), 1e-20f), iuy = 1.0f / fmaxf(fabsf(r.uy), 1e-20f),
                        iuz = 1.0f / fmaxf(fabsf(r.uz), 1e-20f);
            int ev = 0; // 0 collision, -1 surface, +1 escape
            float bt;
            int guard = 0;
            for (;;) {
                const LayerRec L = lay[r.k];
                const bool in3d = (r.k >= S.k3lo && r.k < S.k3hi);
                bt = in3d ? S.bext[((long)r.iy * S.nx + r.ix) * S.nz3 + (r.k - S.k3lo)] : L.bt1d;
                int axis;
                const float s = face_dist(S, r, L.dz, in3d, iux, iuy, iuz, axis);
                if (COUNT) { cnt.steps++; if (in3d) cnt.steps3d++; }
                if (bt * s >= tau) {
                    const float sc = tau / bt;
                    r.px += r.ux * sc; r.py += r.uy * sc;
                    r.pz = fminf(fmaxf(r.pz + r.uz * sc, 0.0f), L.dz);
                    if (in3d) { // stay inside the cell against rounding (1-D layers: folded below)
                        r.px = fminf(fmaxf(r.px, 0.0f), S.dx);
                        r.py = fminf(fmaxf(r.py, 0.0f), S.dy);
                    }
                    ev = 0;
                    break;
                }
                tau -= bt * s;
                const bool up = r.uz > 0.0f;
                ev = cross_face(S, lay, r, s, axis, in3d, ipa, L.dz);
                if (axis == 2 && do_flux) {
                    const int level = up ? r.k : r.k + 1;
                    flux_add<COUNT>(S, r, w, direct, level, up, cnt);
                }
                if (ev != 0) break;
                if (++guard > (1 << 22)) { ev = 2; break; }
            }
            if (ev == 2) { alive = false; break; }
            if (ev > 0) { if (COUNT) cnt.escaped++; alive = false; break; }

            if (ev < 0) {
                // ---- surface reflection
                if (COUNT) cnt.surface++;
                r.k = 0; r.pz = 0.0f;
                Sfc sf;
                if (S.sfc2d) {
                    const float xa = (float)r.ix * S.dx + r.px, ya = (float)r.iy * S.dy + r.py;
                    const int ib = min(max((int)(xa / S.Lx * (float)S.nxb), 0), S.nxb - 1);
                    const int jb = min(max((int)(ya / S.Ly * (float)S.nyb), 0), S.nyb - 1);
                    const float4 q = *reinterpret_cast<const float4 *>(S.sfc2d + ((long)jb * S.nxb + ib) * 8);
                    sf.type = (int)(q.x + 0.5f); sf.p0 = q.y; sf.p1 = q.z; sf.p2 = q.w;
                } else {
                    sf.type = S.sfc_mtype; sf.p0 = S.sfc_param[0]; sf.p1 = S.sfc_param[1]; sf.p2 = S.sfc_param[2];
                }
                if (do_rad) {
                    const bool in3d = (0 >= S.k3lo && 0 < S.k3hi);
                    const float bt0 = in3d ? S.bext[((long)r.iy * S.nx + r.ix) * S.nz3 - S.k3lo] : lay[0].bt1d;
                    for (int iv = 0; iv < S.nview; ++iv) {
                        const float R = surface_R(sf, r.ux, r.uy, r.uz, S.vdir[iv][0], S.vdir[iv][1], S.vdir[iv][2]);
                        if (R > 0.0f)
                            le_tally<COUNT>(S, lay, r, bt0, w * R * S.vdir[iv][2] * (1.0f / kPi), iv, ipa, cnt);
                    }
                }
                float nx = 0.0f, ny = 0.0f, nz = 1.0f;
                rotate_dir(nx, ny, nz, sqrtf(u[2]), u[3]);
                nz = fmaxf(nz, 1e-9f);
                w *= surface_R(sf, r.ux, r.uy, r.uz, nx, ny, nz);
                r.ux = nx; r.uy = ny; r.uz = nz;
                direct = false;
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; alive = false; break; }
                if (do_flux) flux_add<COUNT>(S, r, w, false, 0, true, cnt);
            } else {
                // ---- collision
                const LayerRec L = lay[r.k];
                const bool in3d = (r.k >= S.k3lo && r.k < S.k3hi);
                if (!in3d) fold_xy(S, r, ipa);
                float ks[MI3D_MAX_NP1D + MI3D_MAX_NP3D], apf[MI3D_MAX_NP1D + MI3D_MAX_NP3D];
                int ncomp = 0;
                float kstot = 0.0f;
#pragma unroll
                for (int ip = 0; ip < MI3D_MAX_NP1D; ++ip)
                    if (ip < S.np1d) { ks[ncomp] = L.ks1d[ip]; apf[ncomp] = L.apf1d[ip]; kstot += ks[ncomp]; ++ncomp; }
                if (in3d) {
                    const long v = ((long)r.iy * S.nx + r.ix) * S.nz3 + (r.k - S.k3lo);
#pragma unroll
                    for (int ip = 0; ip < MI3D_MAX_NP3D; ++ip)
                        if (ip < S.np3d) {
                            const float2 c = S.csca[v * S.np3d + ip];
                            ks[ncomp] = c.x; apf[ncomp] = c.y; kstot += c.x; ++ncomp;
                        }
                }
                if (COUNT) cnt.scatter++;
                w *= kstot / bt;
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; alive = false; break; }
                if (do_rad) {
                    for (int iv = 0; iv < S.nview; ++iv) {
                        const float mu = r.ux * S.vdir[iv][0] + r.uy * S.vdir[iv][1] + r.uz * S.vdir[iv][2];
                        float P = 0.0f;
                        for (int q = 0; q < ncomp; ++q)
                            if (ks[q] > 0.0f) P += ks[q] * phase_eval(S, apf[q], mu);
                        P /= kstot;
                        le_tally<COUNT>(S, lay, r, bt, w * P * (0.25f / kPi), iv, ipa, cnt);
                    }
                }
                // choose the component that scatters
                const float target = u[1] * kstot;
                float acc = 0.0f, usel = 0.0f, apf_sel = apf[0];
                bool found = false;
                for (int q = 0; q < ncomp; ++q) {
                    if (!found && (target < acc + ks[q] || q == ncomp - 1)) {
                        found = true;
                        apf_sel = apf[q];
                        usel = ks[q] > 0.0f ? (target - acc) / ks[q] : 0.0f;
                    }
                    acc += ks[q];
                }
                usel = fminf(fmaxf(usel, 0.0f), 1.0f);
                const float mu = phase_sample(S, apf_sel, u[2], usel);
                rotate_dir(r.ux, r.uy, r.uz, mu, u[3]);
                direct = false;
            }
            if (w < S.wmin) {
                if (COUNT) cnt.roulette++;
                draw4(seed, id, draw++, u);
                if (u[0] * S.wmin < w) w = S.wmin;
                else { if (COUNT) cnt.killed++; alive = false; }
            }
        }
        cnt.photons++;
    }

    // ---- counters: wave reduction, one atomic per wave and counter
    {
        uint32_t vals[14] = {cnt.photons, cnt.steps, cnt.steps3d, cnt.scatter, cnt.surface, cnt.le_rays,
                             cnt.le_steps, cnt.le_steps3d, cnt.le_column, cnt.flux_tally, cnt.roulette,
                             cnt.killed, cnt.escaped, cnt.absorbed};
        const int ncnt = COUNT ? 14 : 1;
        for (int q = 0; q < ncnt; ++q) {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&S.counters[q], v);
        }
    }
}

template __global__ void k_transport<false>(const DevScene, const uint64_t, const uint64_t, const uint64_t);
template __global__ void k_transport<true>(const DevScene, const uint64_t, const uint64_t, const uint64_t);

} // namespace mi3d
