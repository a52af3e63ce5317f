// mi3d_kernel_rays.hip — the local-estimate rays of marched satellite views as a kernel of their own.
//
// A local-estimate ray feeds nothing back into the photon it comes from, so the two need not share a lane.  Inside one loop
// (k_transport, k_transport_lean<.,.,1>) a lane walks the eight rays of an event one after the other while its photon waits,
// and the blocks that serve the photons' own events run for the few lanes that happen to need them: on the nine-view
// configuration 25 % of the lanes are active per vector instruction (profiles/r02/pmc_les480_mv9_lean.txt).  Here the photon
// loop (k_transport_lean<.,.,2>, as lean as the nadir build) only WRITES an event record per collision and reflection -- 64
// bytes, into the list of the XCD it runs on -- and carries on; this kernel then takes (event, view) pairs off those lists,
// one ray per lane at a time, every lane doing nothing but rays: start (phase function of the event towards the view, roulette
// budget, DDA set-up), walk, tally, next.  The lists are per XCD and in the order the photon loop met the events, which is the
// order of the photons' start tiles: an XCD marches rays through the voxels its L2 still holds.
//
// Same estimator, same numbers: which ray carries what is a function of (photon id, index of the photon's Philox block, view)
// as before; only the order of the sums changes.  Serves what k_transport_lean serves.
#include "mi3d_device.h"

namespace mi3d {

#ifndef MI3D_RAYS_THRESH
#define MI3D_RAYS_THRESH 24   // phase A keeps stepping while at least this many lanes of the wave are walking
#endif
#ifndef MI3D_RAYS_WAVES
#define MI3D_RAYS_WAVES(COUNT) ((COUNT) ? 4 : 5)
#endif
constexpr unsigned kEvChunk = 64;   // events a wave takes from a list at a time

template <bool COUNT, bool P3D>
__global__ void __launch_bounds__(256, MI3D_RAYS_WAVES(COUNT))
k_rays(const DevScene S, const uint64_t seed) {
    extern __shared__ float4 smem[];
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(smem);
    const float4 *lay4 = smem;
    const ViewRec *views = reinterpret_cast<const ViewRec *>(smem + S.nz * (kLayStride / 4));
    const DevCold *cold = reinterpret_cast<const DevCold *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2);
    int *mview = reinterpret_cast<int *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + kColdF4);   // [nmarch] marched views
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.cold->lay);
        for (int i = threadIdx.x; i < S.nz * (kLayStride / 4); i += blockDim.x) smem[i] = src[i];
        const float4 *vsrc = reinterpret_cast<const float4 *>(S.cold->views);
        for (int i = threadIdx.x; i < S.nview * 2; i += blockDim.x) smem[S.nz * (kLayStride / 4) + i] = vsrc[i];
        const float4 *csrc = reinterpret_cast<const float4 *>(S.cold);
        if (threadIdx.x < kColdF4) smem[S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + threadIdx.x] = csrc[threadIdx.x];
        if (threadIdx.x == 0) {
            int n = 0;
            for (int v = 0; v < S.nview; ++v)
                if (!S.cold->views[v].column) mview[n++] = v;
        }
    }
    __syncthreads();

    const bool ipa = (S.solver == MI3D_SOLVER_IPA) || P3D;   // everything scattered stays in its column under both
    Counters cnt = {};
    const unsigned sx_b = (unsigned)S.nz3 * 16u, sy_b = (unsigned)S.nx * sx_b;
    const char *vbase = reinterpret_cast<const char *>(S.vrec) - (long)S.k3lo * 16;
    const unsigned nm = (unsigned)S.nmarch;
    const float inv_nm = 1.0f / (float)nm;

    // ---- lane state: one ray
    float epx = 0, epy = 0, zev = 0;            // the event: position inside its voxel, height
    int eix = 0, eiy = 0;
    float ux = 0, uy = 0, uz = 1, iux = 1, iuy = 1, iuz = 1;
    float t = 0, tx = 0, ty = 0, tz = 0;
    int ix = 0, iy = 0, k = 0, ncx = 0, ncy = 0, stepx = 0, stepy = 0;
    float rem = 0.0f, tkill = kTauCut, contrib = 0.0f, zstop = 0.0f;
    float rox = 0, roy = 0, roz = 0, rpz = 0;
    int iv = 0, mode = M_NEED;
    bool setup = false;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    unsigned victim = 0, list = xcc;
    // wave-uniform: events [ev_next, ev_end) of list `list` are this wave's; `sub` rays of them have been handed out
    unsigned long long ev_next = 0, ev_end = 0;
    unsigned sub = 0;

    for (;;) {
        // =================================== phase A: voxel steps ===================================
        for (;;) {
            const bool flying = (mode == M_LE);
            const int nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (nfly < MI3D_RAYS_THRESH && __ballot(mode != M_LE && mode != M_DONE) != 0ull) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (flying) {
                const float4 r4 = *reinterpret_cast<const float4 *>(vbase + ((unsigned)iy * sy_b + (unsigned)ix * sx_b + (unsigned)k * 16u));
                const float tn = fminf(fminf(tx, ty), tz);
                float dtau = r4.x * (tn - t);
                if (COUNT) { cnt.le_steps++; cnt.le_steps3d++; }
                bool plane = false;
                if (zstop < INFINITY) {
                    const float zn = fmaf(uz, tn, roz);
                    plane = uz > 0.0f ? zn >= zstop : zn <= zstop;
                    if (plane) dtau = r4.x * fabsf(zstop - fmaf(uz, t, roz)) * iuz;
                }
                if (dtau >= rem) { rem = -1.0f; mode = M_LEEND; }   // the ray's budget is used up: given up
                else if (plane) { rem -= dtau; mode = M_LEEND; }
                else {
                    rem -= dtau;
                    t = tn;
                    if (tz == tn) {
                        const bool up = uz > 0.0f;
                        const int knew = up ? k + 1 : k - 1;
                        if (knew >= S.nz || knew < 0) mode = M_LEEND;
                        else {
                            const float4 Ln = lay4[knew * (kLayStride / 4)];
                            tz = fmaf(Ln.x, iuz, tz);
                            if (!(__float_as_int(Ln.w) & kLayStep3d)) { mode = M_LEUNIF; rpz = up ? 0.0f : Ln.x; }
                            k = knew;
                        }
                    } else if (tx == tn) {
                        tx = fmaf(S.dx, iux, tx);
                        ncx++;
                        int c = ix + stepx;
                        c = c >= S.nx ? 0 : (c < 0 ? S.nx - 1 : c);
                        ix = c;
                    } else {
                        ty = fmaf(S.dy, iuy, ty);
                        ncy++;
                        int c = iy + stepy;
                        c = c >= S.ny ? 0 : (c < 0 ? S.ny - 1 : c);
                        iy = c;
                    }
                }
            }
        }

        // =================================== phase B ===================================
        if (COUNT) { cnt.b_slots++; if (mode != M_LE && mode != M_DONE) cnt.b_lanes++; }
        // ---- rays inside runs of uniform layers (as k_transport_lean, block B0')
        if (mode == M_LEUNIF) {
            const bool up = uz > 0.0f;
            bool reenter = false;
            if (!(zstop < INFINITY)) {
                const LayerRec &Lk = lay[k];
                const int kend = up ? Lk.run_hi : Lk.run_lo;
                const LayerRec &Le = lay[kend];
                const float tv = up ? (Le.tauz + Le.bt * Le.dz - Lk.tauz) - Lk.bt * rpz
                                    : (Lk.tauz - Le.tauz) + Lk.bt * rpz;
                const float hv = up ? (Le.zlo + Le.dz) - (Lk.zlo + rpz) : (Lk.zlo + rpz) - Le.zlo;
                const float tpath = tv * iuz;
                if (COUNT) cnt.le_steps++;
                if (tpath >= rem) { rem = -1.0f; mode = M_LEEND; }
                else {
                    rem -= tpath;
                    t += hv * iuz;
                    if (up) { k = kend + 1; rpz = 0.0f; if (k >= S.nz) mode = M_LEEND; else reenter = true; }
                    else { k = kend - 1; if (k < 0) { k = 0; mode = M_LEEND; } else { rpz = lay[k].dz; reenter = true; } }
                }
            } else {
                for (int guard = 0; guard < kMaxLayers + 2; ++guard) {
                    const float4 L = lay4[k * (kLayStride / 4)];
                    if (__float_as_int(L.w) & kLayStep3d) { reenter = true; break; }
                    const float s = fmaxf((up ? L.x - rpz : rpz) * iuz, 0.0f);
                    if (COUNT) cnt.le_steps++;
                    const float zn = L.z + rpz + uz * s;
                    if (up ? zn >= zstop : zn <= zstop) {
                        rem -= L.y * fabsf(zstop - (L.z + rpz)) * iuz;
                        mode = M_LEEND;
                        break;
                    }
                    rem -= L.y * s;
                    t += s;
                    const int knew = up ? k + 1 : k - 1;
                    if (knew >= S.nz || knew < 0) { mode = M_LEEND; break; }
                    k = knew;
                    rpz = up ? 0.0f : lay4[k * (kLayStride / 4)].x;
                    if (rem < 0.0f) { mode = M_LEEND; break; }
                }
            }
            if (reenter) {
                float xo = fmaf(ux, t, rox) - (ux > 0.0f ? S.dx : -S.dx) * (float)ncx;
                float yo = fmaf(uy, t, roy) - (uy > 0.0f ? S.dy : -S.dy) * (float)ncy;
                fold_xy(S, cold, xo, yo, ix, iy, ipa);
                rox = xo; roy = yo; roz = lay[k].zlo + rpz;
                t = 0.0f; ncx = 0; ncy = 0;
                mode = M_LE; setup = true;
            }
        }

        // ---- a ray has arrived (or has been given up): tally it, the lane is free
        if (mode == M_LEEND) {
            if (rem >= 0.0f) {
                const ViewRec V = views[iv];
                const float acc = tkill - rem;
                float xr = (float)eix * S.dx + epx, yr = (float)eiy * S.dy + epy;
                if (!ipa) {
                    const float tt = (zev - V.zreg) * frcp(V.vz);
                    xr -= V.vx * tt; yr -= V.vy * tt;
                    xr -= floorf(xr * cold->inv_Lx) * cold->Lx; yr -= floorf(yr * cold->inv_Ly) * cold->Ly;
                }
                const int ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                const int jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                RAD_ADD(&S.rad[(unsigned)((iv * S.nyr + jr) * S.nxr + ir) * (unsigned)S.rad_stride],
                        contrib * fexp_neg(V.roulette ? fminf(acc, cold->le_tau1) : acc) * frcp(fabsf(V.vz)));
            }
            mode = M_NEED;
        }

        // ---- free lanes take the next (event, view) pairs; twice per pass: a pair whose view does not see the event costs nothing more
        for (int round = 0; round < 2; ++round) {
            const unsigned long long need = __ballot(mode == M_NEED);
            if (need == 0ull) break;
            if (ev_next >= ev_end || sub >= (unsigned)(ev_end - ev_next) * nm) {
                // this wave's events are all handed out: the next chunk of its XCD's list, then of the others'
                const int leader = __ffsll((long long)need) - 1;
                bool got = false;
                while (victim < 8u) {
                    list = (xcc + victim) & 7u;
                    const unsigned long long have = cold->ev_ctr[list * kCtrStride] < (unsigned long long)cold->ev_cap
                                                        ? cold->ev_ctr[list * kCtrStride] : (unsigned long long)cold->ev_cap;
                    unsigned long long b = 0;
                    if ((int)(threadIdx.x & 63) == leader) b = atomicAdd(cold->ev_ctr + (9 + list) * kCtrStride, (unsigned long long)kEvChunk);
                    b = __shfl(b, leader, 64);
                    if (b < have) { ev_next = b; ev_end = b + kEvChunk < have ? b + kEvChunk : have; sub = 0; got = true; break; }
                    victim++;
                }
                if (!got) {
                    if (mode == M_NEED) mode = M_DONE;
                    break;
                }
            }
            const unsigned total = (unsigned)(ev_end - ev_next) * nm, avail = total - sub;
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            const unsigned nn = (unsigned)__popcll(need);
            if (mode == M_NEED && rank < avail) {
                const unsigned r = sub + rank;
                const unsigned el = (unsigned)(((float)r + 0.5f) * inv_nm);       // r / nm, exact: r < 2^13
                iv = mview[r - el * nm];
                const float4 *e = cold->ev_list + ((size_t)list * cold->ev_cap + (ev_next + el)) * kEventF4;
                const float4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
                const bool filled = e0.w > 0.0f;      // (a record a wave of the photon loop reserved and did not use has weight 0)
                const int cell = __float_as_int(e2.z), kk = filled ? __float_as_int(e2.w) : 0;
                const int ek = kk & 0xffff, kind = kk >> 16;
                const LayerRec &Lk = lay[ek];
                const ViewRec V = views[iv];
                const float zz = Lk.zlo + e0.z;
                // the sensor on the wrong side of the event, an up-looking one for a surface event: no ray
                const bool sees = filled && (V.vz > 0.0f ? zz < V.zs : (zz > V.zs && (kind & 15) != E_SURFACE));
                float c = 0.0f;
                if (sees) {
                    if ((kind & 15) == E_SURFACE) {
                        const int six = cell & 0xffff, siy = cell >> 16;
                        const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, six, siy, e0.x, e0.y) : Sfc{kind >> 4, e1.w, e2.x, e2.y, 0.0f, 0.0f};
                        c = e0.w * surface_R(sf, e1.x, e1.y, e1.z, V.vx, V.vy, V.vz) * V.vz * (1.0f / kPi);
                    } else {
                        const float mu = e1.x * V.vx + e1.y * V.vy + e1.z * V.vz;
                        const float ks1 = Lk.ks1d[0], ks3 = (Lk.flags & kLayIn3d) ? e1.w : 0.0f;
                        float P = 0.0f;
                        if (ks1 > 0.0f) P = ks1 * phase_eval_analytic(Lk.apf1d[0], mu);
                        if (ks3 > 0.0f) P += ks3 * phase_eval_analytic(e2.x, mu);
                        c = e0.w * P * frcp(ks1 + ks3) * (0.25f / kPi);
                    }
                    if (COUNT) cnt.le_rays++;
                }
                if (c > 0.0f) {
                    contrib = c;
                    epx = e0.x; epy = e0.y; zev = zz; eix = cell & 0xffff; eiy = cell >> 16;
                    ux = V.vx; uy = V.vy; uz = V.vz;
                    zstop = (uz < 0.0f || V.zs < cold->ztoa) ? V.zs : INFINITY;
                    const uint64_t pid = (uint64_t)(unsigned)__float_as_int(e3.x) | ((uint64_t)(unsigned)__float_as_int(e3.y) << 32);
                    tkill = V.roulette ? cold->le_tau1 - 0.69314718f * __builtin_amdgcn_logf(le_roulette_u(seed, pid, (uint32_t)__float_as_int(e3.z), iv)) : kTauCut;
                    rem = tkill;
                    ix = eix; iy = eiy; k = ek;
                    rox = epx; roy = epy; roz = zev; rpz = e0.z;
                    t = 0.0f; ncx = 0; ncy = 0;
                    if (Lk.flags & kLayStep3d) { mode = M_LE; setup = true; }
                    else { mode = M_LEUNIF; iuz = frcp(fmaxf(fabsf(uz), 1e-20f)); }
                }
            }
            sub += nn < avail ? nn : avail;
        }

        // ---- a ray about to walk: the parameters of its first three faces
        if (setup && mode == M_LE) {
            setup = false;
            const float4 L = lay4[k * (kLayStride / 4)];
            iux = frcp(fmaxf(fabsf(ux), 1e-20f)); iuy = frcp(fmaxf(fabsf(uy), 1e-20f)); iuz = frcp(fmaxf(fabsf(uz), 1e-20f));
            tx = (ux > 0.0f ? S.dx - rox : rox) * iux;
            ty = (uy > 0.0f ? S.dy - roy : roy) * iuy;
            tz = (uz > 0.0f ? L.x - rpz : rpz) * iuz;
            stepx = ipa ? 0 : (ux > 0.0f ? 1 : -1);
            stepy = ipa ? 0 : (uy > 0.0f ? 1 : -1);
        }

        if (__ballot(mode != M_DONE) == 0ull) break;
    }

    if (COUNT) {
        uint32_t vals[24] = {0, 0, 0, 0, 0, cnt.le_rays, cnt.le_steps, cnt.le_steps3d, 0, 0, 0, 0, 0, 0,
                             cnt.a_lanes, cnt.a_slots, cnt.b_lanes, cnt.b_slots, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < 24; ++q) {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&S.cold->counters[q], v);
        }
    }
}

template __global__ void k_rays<false, false>(const DevScene, const uint64_t);
template __global__ void k_rays<false, true>(const DevScene, const uint64_t);
template __global__ void k_rays<true, false>(const DevScene, const uint64_t);
template __global__ void k_rays<true, true>(const DevScene, const uint64_t);

} // namespace mi3d
