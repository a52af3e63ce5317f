// mi3d_kernel_pool.hip — the photon loop of k_transport_lean with the photons' EVENTS served in full-width batches.
//
// In k_transport_lean a lane owns one photon: it walks voxels (phase A), and when the walk ends it waits until enough other
// lanes of the wave have ended theirs; then phase B serves the events of those lanes while the ones still walking wait.
// A flight is 3.4 voxel steps long on the bench scene and an event costs ten times a step, so the lanes spend half the
// issue slots waiting for each other (profiles/r02/pmc_les480_lean.txt: 47 % of the lanes active per vector instruction).
//
// Here a wave owns 128 photons: 64 in its lanes, walking, and 64 parked in LDS (7 KiB per wave, private to it: no atomics,
// no barriers).  A lane whose walk has ended SWAPS its photon for a parked one that is ready to walk -- seven 16-byte LDS
// writes, seven reads -- and walks on; the events of the parked photons are served by a BATCH: lane i loads the photon of
// slot i, all 64 at once, runs one pass of the event blocks of k_transport_lean on it (the walkers' own state stays in their
// registers, untouched), and stores it back, ready to walk or waiting for another pass.  A batch runs when no parked photon
// is ready any more, i.e. when all 64 slots hold events: the event blocks run with every lane active, and the walk runs with
// nearly every lane active because a finished lane finds a ready photon at once.
//
// Same random-number protocol, same estimator, same sampling formulas, same numbers: photon id -> history is the function
// DESIGN.md §3 specifies; only which lane computes what, and the order of the sums, differ.  Serves what
// k_transport_lean<.,.,0> and <.,.,2> serve (EMIT: event records for k_rays).
#include "mi3d_device.h"

namespace mi3d {

#ifndef MI3D_POOL_SWAP
#define MI3D_POOL_SWAP 20     // lanes without a walking photon before the walk pauses for a swap (while parked photons are ready)
#endif
#ifndef MI3D_POOL_WAVES
#define MI3D_POOL_WAVES 3     // waves per SIMD the register budget allows: walker + parked photon + the event blocks' temporaries need ~160 registers
#endif
#ifndef MI3D_POOL_PASS
#define MI3D_POOL_PASS 3      // every third batch is a full one (launches, surface, roulette: see k_transport), the others serve collisions
#endif
constexpr int kPoolSlotF4 = 7;      // float4 per parked photon
constexpr int kPoolWaveBytes = 64 * kPoolSlotF4 * 16 + 64 * 2 * 4;   // slots + two index lists

__host__ __device__ inline size_t pool_lds_extra() { return (size_t)4 * kPoolWaveBytes; }

// Parked photon, field-major ([field][slot] float4):
//   0  px, py, pz, w          position inside the voxel (walk origin when ready: pz absolute)
//   1  ux, uy, uz, rem
//   2  u1, u2, u3, pend_val
//   3  id lo, id hi, draw, pend_pix
//   4  ix | iy << 16, k | mode << 16, kind | dkind << 8 | direct << 16 | walked << 17, t
//   5  ready: tx, ty, tz, -     after a walk: ncx, ncy (integers), -, -
//   6  ready: 1/|ux|, 1/|uy|, 1/|uz|, -       after a walk: the voxel record the walk ended in (bt, tab, ks0, apf0)
template <bool COUNT, bool P3D, bool EMIT>
__global__ void __launch_bounds__(256, MI3D_POOL_WAVES)
k_transport_pool(const DevScene S, const uint64_t nphoton, const uint64_t seed, const uint64_t offset) {
    constexpr bool MIXED = EMIT;
    extern __shared__ float4 smem[];
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(smem);
    const float4 *lay4 = smem;
    const ViewRec *views = reinterpret_cast<const ViewRec *>(smem + S.nz * (kLayStride / 4));
    const DevCold *cold = reinterpret_cast<const DevCold *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2);
    char *wave_lds = reinterpret_cast<char *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + kColdF4) + (threadIdx.x >> 6) * kPoolWaveBytes;
    float4 *slots = reinterpret_cast<float4 *>(wave_lds);                     // [kPoolSlotF4][64]
    int *rdy = reinterpret_cast<int *>(wave_lds + 64 * kPoolSlotF4 * 16);     // [64] ready slots, [64] empty slots, in rank order
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.cold->lay);
        for (int i = threadIdx.x; i < S.nz * (kLayStride / 4); i += blockDim.x) smem[i] = src[i];
        const float4 *vsrc = reinterpret_cast<const float4 *>(S.cold->views);
        for (int i = threadIdx.x; i < S.nview * 2; i += blockDim.x) smem[S.nz * (kLayStride / 4) + i] = vsrc[i];
        const float4 *csrc = reinterpret_cast<const float4 *>(S.cold);
        if (threadIdx.x < kColdF4) smem[S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + threadIdx.x] = csrc[threadIdx.x];
    }
    __syncthreads();

    const bool ipa_all = (S.solver == MI3D_SOLVER_IPA);
#define IPA_NOW(is_le_) (ipa_all || (P3D && ((is_le_) || !direct)))
    Counters cnt = {};
    const unsigned sx_b = (unsigned)S.nz3 * 16u, sy_b = (unsigned)S.nx * sx_b;
    const char *vbase = reinterpret_cast<const char *>(S.vrec) - (long)S.k3lo * 16;
    const unsigned lane = threadIdx.x & 63u;

    // ---- the walker: the photon this lane walks.  c0..c3: what it carries along untouched (fields 0-3 of a slot)
    bool has = false;
    int wmode = M_NEED;                         // M_FLY while walking; what the walk ended in afterwards
    float4 c0 = make_float4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0;
    int wflags = 0;
    float wt = 0, wtx = 0, wty = 0, wtz = 0, wdtx = 1, wdty = 1, wiuz = 1, wrem = 0;   // (wdtx, wdty: 1/|ux|, 1/|uy|)
    int wix = 0, wiy = 0, wk = 0, wncx = 0, wncy = 0, wsx = 0, wsy = 0;
    float4 wev = make_float4(0, 0, 0, 0);       // the voxel record the walk ended in

    // ---- wave-uniform: the state of the 64 slots, the photon ids this wave has taken from the launch
    unsigned long long m_ready = 0ull, m_work = ~0ull, m_done = 0ull;
    unsigned long long pool_next = 0, pool_end = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID (speed only)
    unsigned victim = 0;
    unsigned long long ev_lo = 0, ev_hi = 0;   // EMIT: slots of this XCD's event list reserved by this wave and not yet used
    unsigned pass_ctr = 0;

    // every slot starts empty: a photon that wants its id
    slots[0 * 64 + lane] = make_float4(0, 0, 0, 0);
    slots[1 * 64 + lane] = make_float4(0, 0, 1, 0);
    slots[2 * 64 + lane] = make_float4(0, 0, 0, 0);
    slots[3 * 64 + lane] = make_float4(0, 0, 0, __int_as_float(-1));
    slots[4 * 64 + lane] = make_float4(0, __int_as_float(M_NEED << 16), __int_as_float(E_LAUNCH | (D_LAUNCH << 8)), 0);
    slots[5 * 64 + lane] = make_float4(0, 0, 0, 0);
    slots[6 * 64 + lane] = make_float4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (;;) {
        // =================================== the walk (phase A of k_transport_lean) ===================================
        int nfly = 0;
        for (;;) {
            const bool flying = has && wmode == M_FLY;
            nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (m_ready != 0ull ? (64 - nfly >= MI3D_POOL_SWAP) : (m_work != 0ull && 64 - nfly >= MI3D_POOL_SWAP)) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (flying) {
                const float4 r4 = *reinterpret_cast<const float4 *>(vbase + ((unsigned)wiy * sy_b + (unsigned)wix * sx_b + (unsigned)wk * 16u));
                const float tn = fminf(fminf(wtx, wty), wtz);
                const float dtau = r4.x * (tn - wt);
                if (COUNT) { cnt.steps++; cnt.steps3d++; }
                if (dtau >= wrem) {
                    wev = r4;                   // the collision lies inside this voxel: at t + rem / bt (worked out by the batch)
                    wmode = M_COLL;
                } else {
                    wrem -= dtau;
                    wt = tn;
                    if (wtz == tn) {
                        const bool up = c1.z > 0.0f;
                        const int knew = up ? wk + 1 : wk - 1;
                        if (knew >= S.nz) { if (COUNT) cnt.escaped++; wmode = M_NEED; }
                        else if (knew < 0) { wmode = M_SURF; wev = r4; }
                        else {
                            const float4 Ln = lay4[knew * (kLayStride / 4)];
                            wtz = fmaf(Ln.x, wiuz, wtz);
                            if (!(__float_as_int(Ln.w) & kLayStep3d)) wmode = M_UNIF;
                            wk = knew;
                        }
                    } else if (wtx == tn) {
                        wtx = fmaf(S.dx, wdtx, wtx);
                        wncx++;
                        int c = wix + wsx;
                        c = c >= S.nx ? 0 : (c < 0 ? S.nx - 1 : c);
                        wix = c;
                    } else {
                        wty = fmaf(S.dy, wdty, wty);
                        wncy++;
                        int c = wiy + wsy;
                        c = c >= S.ny ? 0 : (c < 0 ? S.ny - 1 : c);
                        wiy = c;
                    }
                }
            }
        }

        // =================================== swap: finished walks against ready photons ===================================
        {
            const bool fin = has && wmode != M_FLY;
            const unsigned long long wm = __ballot(fin || !has), fm = __ballot(fin);
            if ((wm != 0ull && m_ready != 0ull) || (fm != 0ull && m_done != 0ull)) {
                if (COUNT) { cnt.cyc[0]++; if (fin || !has) cnt.cyc[1]++; }
                // the ready slots and the empty ones, each in rank order
                const unsigned rr = __builtin_amdgcn_mbcnt_hi((unsigned)(m_ready >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_ready, 0u));
                const unsigned dr = __builtin_amdgcn_mbcnt_hi((unsigned)(m_done >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_done, 0u));
                if ((m_ready >> lane) & 1ull) rdy[rr] = (int)lane;
                if ((m_done >> lane) & 1ull) rdy[64 + dr] = (int)lane;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const unsigned nr = (unsigned)__popcll(m_ready), nd = (unsigned)__popcll(m_done);
                const unsigned wrank = __builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u));
                const bool take = (fin || !has) && wrank < nr;
                // finished walks no ready photon is left for go to empty slots (the end of the launch)
                const unsigned long long lm = __ballot(fin && !take);
                const unsigned lrank = __builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
                const bool leave = fin && !take && lrank < nd;
                if (take || leave) {
                    const int s = take ? rdy[wrank] : rdy[64 + lrank];
                    float4 a0, a1, a2, a3, a4, a5, a6;
                    if (take) {
                        a0 = slots[0 * 64 + s]; a1 = slots[1 * 64 + s]; a2 = slots[2 * 64 + s]; a3 = slots[3 * 64 + s];
                        a4 = slots[4 * 64 + s]; a5 = slots[5 * 64 + s]; a6 = slots[6 * 64 + s];
                    }
                    if (fin) {
                        slots[0 * 64 + s] = c0;
                        slots[1 * 64 + s] = make_float4(c1.x, c1.y, c1.z, wrem);
                        slots[2 * 64 + s] = c2;
                        slots[3 * 64 + s] = c3;
                        slots[4 * 64 + s] = make_float4(__int_as_float(wix | (wiy << 16)), __int_as_float(wk | (wmode << 16)), __int_as_float(wflags | (1 << 17)), wt);
                        slots[5 * 64 + s] = make_float4(__int_as_float(wncx), __int_as_float(wncy), 0.0f, 0.0f);
                        slots[6 * 64 + s] = wev;
                    } else {
                        // the lane had no photon: the slot is left empty (wants an id)
                        slots[3 * 64 + s] = make_float4(0, 0, 0, __int_as_float(-1));
                        slots[4 * 64 + s] = make_float4(0, __int_as_float(M_NEED << 16), __int_as_float(E_LAUNCH | (D_LAUNCH << 8)), 0);
                    }
                    if (take) {
                        c0 = a0; c1 = a1; c2 = a2; c3 = a3;
                        const int cell = __float_as_int(a4.x);
                        wix = cell & 0xffff; wiy = cell >> 16;
                        wk = __float_as_int(a4.y) & 0xffff;
                        wflags = __float_as_int(a4.z) & 0x1ffff;
                        wt = 0.0f; wncx = 0; wncy = 0;
                        wtx = a5.x; wty = a5.y; wtz = a5.z;
                        wdtx = a6.x; wdty = a6.y; wiuz = a6.z;
                        wrem = a1.w;
                        const bool direct = (wflags >> 16) & 1;
                        const bool ipa = IPA_NOW(false);
                        wsx = ipa ? 0 : (c1.x > 0.0f ? 1 : -1);
                        wsy = ipa ? 0 : (c1.y > 0.0f ? 1 : -1);
                        wmode = M_FLY; has = true;
                    } else has = false;
                }
                // slots taken (the first ntake ready ones) and slots filled (the first nleave empty ones) now hold work
                const unsigned ntake = (unsigned)__popcll(wm) < nr ? (unsigned)__popcll(wm) : nr;
                const unsigned nleave = (unsigned)__popcll(lm) < nd ? (unsigned)__popcll(lm) : nd;
                const unsigned long long tk = __ballot(((m_ready >> lane) & 1ull) && rr < ntake);
                const unsigned long long lv = __ballot(((m_done >> lane) & 1ull) && dr < nleave);
                m_ready &= ~tk; m_done &= ~lv; m_work |= tk | lv;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }

        // =================================== batch: one pass of the event blocks over the parked photons ===================================
        // (when no parked photon is ready any more, or nothing walks)
        if (m_work != 0ull && (m_ready == 0ull || __ballot(has && wmode == M_FLY) == 0ull)) {
            const bool wk_ = (m_work >> lane) & 1ull;
            float px = 0, py = 0, pz = 0, w = 0, ux = 0, uy = 0, uz = 1, rem = 0, u1 = 0, u2 = 0, u3 = 0, pend_val = 0;
            uint64_t id = 0;
            uint32_t draw = 0;
            int pend_pix = -1, ix = 0, iy = 0, k = 0, mode = M_DONE, kind = E_LAUNCH, dkind = D_LAUNCH, ncx = 0, ncy = 0;
            bool direct = false, walked = false;
            float t = 0, bt_ev = 0, ev_tab = 0, ev_ks0 = 0, ev_apf0 = 0;
            float &ev_sfc = ev_tab;
            if (wk_) {
                const float4 a0 = slots[0 * 64 + lane], a1 = slots[1 * 64 + lane], a2 = slots[2 * 64 + lane], a3 = slots[3 * 64 + lane];
                const float4 a4 = slots[4 * 64 + lane], a5 = slots[5 * 64 + lane], a6 = slots[6 * 64 + lane];
                px = a0.x; py = a0.y; pz = a0.z; w = a0.w;
                ux = a1.x; uy = a1.y; uz = a1.z; rem = a1.w;
                u1 = a2.x; u2 = a2.y; u3 = a2.z; pend_val = a2.w;
                id = (uint64_t)(unsigned)__float_as_int(a3.x) | ((uint64_t)(unsigned)__float_as_int(a3.y) << 32);
                draw = (uint32_t)__float_as_int(a3.z); pend_pix = __float_as_int(a3.w);
                const int cell = __float_as_int(a4.x), km = __float_as_int(a4.y), fl = __float_as_int(a4.z);
                ix = cell & 0xffff; iy = cell >> 16; k = km & 0xffff; mode = km >> 16;
                kind = fl & 0xff; dkind = (fl >> 8) & 0xff; direct = (fl >> 16) & 1; walked = (fl >> 17) & 1;
                t = a4.w;
                ncx = __float_as_int(a5.x); ncy = __float_as_int(a5.y);
                bt_ev = a6.x; ev_tab = a6.y; ev_ks0 = a6.z; ev_apf0 = a6.w;
            }
            if (COUNT) { cnt.b_slots++; if (wk_) cnt.b_lanes++; }
            const bool full = MI3D_POOL_PASS <= 1 || ((pass_ctr++ % (unsigned)(MI3D_POOL_PASS)) == 0u) ||
                              __ballot(mode == M_COLL || (mode == M_FINISH && (kind & 15) != E_SURFACE) || (mode == M_DRAW && dkind == D_FLIGHT)) == 0ull;
            bool emit = false;
            float iux = 1, iuy = 1, iuz = 1, tx = 0, ty = 0, tz = 0;

            // ---- where the walk has ended
            if (walked && (mode == M_COLL || mode == M_SURF || mode == M_UNIF)) {
                walked = false;
                const float tc = (mode == M_COLL) ? fmaf(rem, frcp(bt_ev), t) : t;
                const float4 L = lay4[k * (kLayStride / 4)];
                const float xo = fmaf(ux, tc, px) - (ux > 0.0f ? S.dx : -S.dx) * (float)ncx;
                const float yo = fmaf(uy, tc, py) - (uy > 0.0f ? S.dy : -S.dy) * (float)ncy;
                px = fminf(fmaxf(xo, 0.0f), S.dx);
                py = fminf(fmaxf(yo, 0.0f), S.dy);
                if (mode == M_COLL) pz = fminf(fmaxf(fmaf(uz, tc, pz) - L.z, 0.0f), L.x);
                else pz = (mode == M_SURF || uz > 0.0f) ? 0.0f : L.x;
            }
            if (mode == M_NEED) walked = false;   // (escaped: nothing to reconstruct)

            // ---- B0: photons inside runs of horizontally uniform layers: the whole rest of the run at once
            if (full && mode == M_UNIF) {
                const bool up = uz > 0.0f;
                const LayerRec &Lk = lay[k];
                const int kend = up ? Lk.run_hi : Lk.run_lo;
                const LayerRec &Le = lay[kend];
                const float tv = up ? (Le.tauz + Le.bt * Le.dz - Lk.tauz) - Lk.bt * pz
                                    : (Lk.tauz - Le.tauz) + Lk.bt * pz;          // vertical optical depth
                const float hv = up ? (Le.zlo + Le.dz) - (Lk.zlo + pz) : (Lk.zlo + pz) - Le.zlo;
                const float iuzl = frcp(fmaxf(fabsf(uz), 1e-20f));
                const float tpath = tv * iuzl;
                if (tpath < rem) {
                    rem -= tpath;
                    const float s = hv * iuzl;
                    px += ux * s; py += uy * s;
                    if (COUNT) cnt.steps++;
                    if (up) {
                        k = kend + 1; pz = 0.0f;
                        if (k >= S.nz) { if (COUNT) cnt.escaped++; mode = M_NEED; }
                        else { fold_xy(S, cold, px, py, ix, iy, IPA_NOW(false)); mode = M_FLY; walked = true; }   // (walked: the walk is set up in B7)
                    } else {
                        k = kend - 1;
                        if (k < 0) { k = 0; pz = 0.0f; mode = M_SURF; }
                        else { pz = lay[k].dz; fold_xy(S, cold, px, py, ix, iy, IPA_NOW(false)); mode = M_FLY; walked = true; }
                    }
                } else {
                    // the collision lies inside the run: bisection on the vertical optical depth below every layer
                    const float T = Lk.tauz + Lk.bt * pz + (up ? rem : -rem) * fabsf(uz);
                    int lo = up ? k : kend, hi = up ? kend : k;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (lay[mid].tauz <= T) lo = mid; else hi = mid - 1;
                    }
                    const float4 Lj = lay4[lo * (kLayStride / 4)];     // {dz, bt, zlo, flags}
                    const float pzn = fminf(fmaxf((T - lay[lo].tauz) * frcp(fmaxf(Lj.y, 1e-30f)), 0.0f), Lj.x);
                    const float s = fabsf((Lj.z + pzn) - (Lk.zlo + pz)) * iuzl;
                    px += ux * s; py += uy * s;
                    k = lo; pz = pzn;
                    bt_ev = Lj.y;
                    if (COUNT) cnt.steps++;
                    mode = M_COLL;
                }
            }

            // ---- B2: a new event: weight, local estimates answered from the column table
            if (mode == M_COLL || (full && mode == M_SURF)) {
                const float4 L = lay4[k * (kLayStride / 4)];              // {dz, bt, zlo, flags}
                const int flags = __float_as_int(L.w);
                const bool in3d = (flags & kLayIn3d) != 0;
                const LayerRec &Lk = lay[k];
                if (!(flags & kLayStep3d)) fold_xy(S, cold, px, py, ix, iy, IPA_NOW(false));
                const unsigned col = (unsigned)(iy * S.nx + ix);
                if (!(flags & kLayStep3d)) {
                    // the event was found by the uniform-layer code: no voxel step has brought the record
                    float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (in3d) rec = *reinterpret_cast<const float4 *>(vbase + ((unsigned)iy * sy_b + (unsigned)ix * sx_b + (unsigned)k * 16u));
                    ev_tab = rec.y; ev_ks0 = rec.z; ev_apf0 = rec.w;
                }
                const float tcol_here = in3d ? ev_tab : Lk.tabove + ((k < S.k3lo && S.nz3 > 0) ? cold->tcol0[col] : 0.0f);
                const float ks1 = Lk.ks1d[0];
                const bool any_col = !MIXED || S.nmarch < S.nview;
                float c = 0.0f;
                bool dead = false;
                if (mode == M_SURF) {
                    if (COUNT) cnt.surface++;
                    const Sfc sf = load_sfc(S, cold, ix, iy, px, py);
                    if (!(flags & kLayStep3d)) bt_ev = L.y;
                    if (any_col) c = w * surface_R(sf, ux, uy, uz, 0.0f, 0.0f, 1.0f) * (1.0f / kPi);
                    ev_ks0 = sf.p0; ev_apf0 = sf.p1; ev_sfc = sf.p2; kind = E_SURFACE | (sf.type << 4);
                } else {
                    if (COUNT) cnt.scatter++;
                    const float ks3 = in3d ? ev_ks0 : 0.0f;
                    const float kstot = ks1 + ks3;
                    // (exactly 1 for conservative scattering: the approximate reciprocal must not nudge a weight that sits on
                    //  the roulette threshold below it)
                    w *= (kstot >= bt_ev) ? 1.0f : kstot * frcp(bt_ev);
                    if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; dead = true; }
                    if (any_col) {
                        // mixture phase function towards the zenith (a column view looks straight down): cos(angle) = uz
                        float P = 0.0f;
                        if (ks1 > 0.0f) P = ks1 * phase_eval_analytic(Lk.apf1d[0], uz);
                        if (ks3 > 0.0f) P += ks3 * phase_eval_analytic(ev_apf0, uz);
                        c = w * P * frcp(kstot) * (0.25f / kPi);
                    }
                    kind = E_SCATTER;
                }
                if (dead) {
                    mode = M_NEED;
                } else {
                    if (any_col) {
                        const float tau = bt_ev * (L.x - pz) + tcol_here;
                        const float xr = (float)ix * S.dx + px, yr = (float)iy * S.dy + py;
                        const int ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                        const int jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                        const float val = c * fexp_neg(tau);
                        // consecutive tallies of one history into the same pixel are summed in a register (first column view: S.col0)
                        const int jv0 = MIXED ? S.col0 : 0;
                        if (COUNT) { const int nc = MIXED ? S.nview - S.nmarch : S.nview; cnt.le_rays += nc; cnt.le_column += nc; }
                        if (c > 0.0f) {
                            const int pix = (jv0 * S.nyr + jr) * S.nxr + ir;
                            if (pix == pend_pix) pend_val += val;
                            else {
                                if (pend_pix >= 0) RAD_ADD(&S.rad[(unsigned)pend_pix * (unsigned)S.rad_stride], pend_val);
                                pend_pix = pix; pend_val = val;
                            }
                            for (int jv = jv0 + 1; jv < S.nview; ++jv)
                                if (!MIXED || views[jv].column) RAD_ADD(&S.rad[(unsigned)((jv * S.nyr + jr) * S.nxr + ir) * (unsigned)S.rad_stride], val);
                        }
                    }
                    mode = M_FINISH;
                    if (EMIT) emit = true;
                }
            }

            if (EMIT) {
                // ---- the event goes to this XCD's list for k_rays (as k_transport_lean<.,.,2>)
                const unsigned long long em = __ballot(emit);
                if (em != 0ull) {
                    const unsigned n = (unsigned)__popcll(em);
                    if (ev_lo + n > ev_hi) {
                        for (unsigned long long q = ev_lo + lane; q < ev_hi; q += 64)
                            if (q < (unsigned long long)cold->ev_cap) cold->ev_list[ev_list_f4(cold->ev_cap) * xcc + ev_index((unsigned)q)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        const int leader = __ffsll((long long)em) - 1;
                        unsigned long long base = 0;
                        if ((int)lane == leader) base = atomicAdd(cold->ev_ctr + xcc * kCtrStride, (unsigned long long)kEvBlock);
                        base = __shfl(base, leader, 64);
                        ev_lo = base; ev_hi = base + kEvBlock;
                    }
                    if (emit) {
                        const unsigned long long slot = ev_lo + __builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, 0u));
                        if (slot < (unsigned long long)cold->ev_cap) {
                            float4 *lbase = cold->ev_list + ev_list_f4(cold->ev_cap) * xcc; float4 *e = lbase + ev_index((unsigned)slot);
                            e[0] = make_float4(px, py, pz, w);
                            e[kEvStride] = make_float4(ux, uy, uz, ev_ks0);
                            e[2 * kEvStride] = make_float4(ev_apf0, ev_sfc, __int_as_float(ix | (iy << 16)), __int_as_float(k | (kind << 16)));
                            reinterpret_cast<uint32_t *>(lbase)[ev_word((unsigned)slot)] = le_hash_base(seed, id, draw);
                        } else cold->ev_ctr[8 * kCtrStride] = 1ull;   // list full: the launch is reported as failed (mi3d_run), never silently short
                    }
                    ev_lo += n;
                }
            }

            // ---- B4: next photon
            if (full && mode == M_NEED && (id != 0 || draw != 0)) { // a history just ended
                cnt.photons++; id = 0; draw = 0;
                if (pend_pix >= 0) { RAD_ADD(&S.rad[(unsigned)pend_pix * (unsigned)S.rad_stride], pend_val); pend_pix = -1; }
            }
            for (;;) {
                const unsigned long long need = __ballot(full && mode == M_NEED);
                if (need == 0ull) break;
                if (pool_next >= pool_end) {
                    const int leader = __ffsll((long long)need) - 1;
                    bool got = false;
                    while (victim < 8u) {
                        const unsigned x = (xcc + victim) & 7u;
                        const unsigned long long lo = (nphoton * x) >> 3, hi = (nphoton * (x + 1u)) >> 3;
                        unsigned long long b = 0;
                        if ((int)lane == leader) b = atomicAdd(cold->next_photon + x * kCtrStride, (unsigned long long)kChunk);
                        b = __shfl(b, leader, 64);
                        if (lo + b < hi) {
                            pool_next = lo + b;
                            pool_end = lo + b + kChunk < hi ? lo + b + kChunk : hi;
                            got = true;
                            break;
                        }
                        victim++;
                    }
                    if (!got) {
                        if (mode == M_NEED) mode = M_DONE;
                        break;
                    }
                }
                const unsigned long long avail = pool_end - pool_next;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
                const unsigned long long nn = (unsigned long long)__popcll(need);
                if (mode == M_NEED && rank < avail) {
                    const uint32_t *order = cold->order;
                    id = offset + (order ? (unsigned long long)order[pool_next + rank] : pool_next + rank);
                    draw = 0;
                    dkind = D_LAUNCH;
                    mode = M_DRAW;
                }
                pool_next += nn < avail ? nn : avail;
            }

            // ---- B5: finish the event (scattering, surface reflection or launch): new direction and weight
            if (mode == M_FINISH && (full || (kind & 15) != E_SURFACE)) {
                float bx = ux, by = uy, bz = uz, mu_rot = u2;
                if ((kind & 15) == E_SURFACE) {
                    bx = 0.0f; by = 0.0f; bz = 1.0f;
                    mu_rot = fsqrt(u2);
                } else if ((kind & 15) == E_SCATTER) {
                    const LayerRec &Lk = lay[k];
                    const bool in3d = (Lk.flags & kLayIn3d) != 0;
                    const float ks1 = Lk.ks1d[0], ks3 = in3d ? ev_ks0 : 0.0f;
                    // choose the constituent that scatters: the 1-D one first, then the 3-D one
                    const float target = u1 * (ks1 + ks3);
                    const bool first = (target < ks1) || !in3d;
                    const float apf_sel = first ? Lk.apf1d[0] : ev_apf0;
                    mu_rot = phase_sample_analytic(apf_sel, u2);
                }
                if (!(kind == E_LAUNCH && cold->cos_cone >= 1.0f)) rotate_dir(bx, by, bz, mu_rot, u3);
                if ((kind & 15) == E_SURFACE) {
                    const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ix, iy, px, py) : Sfc{kind >> 4, ev_ks0, ev_apf0, ev_sfc, 0.0f, 0.0f};
                    bz = fmaxf(bz, 1e-9f);
                    w *= surface_R(sf, ux, uy, uz, bx, by, bz);
                }
                ux = bx; uy = by; uz = bz;
                if (kind != E_LAUNCH) direct = false;
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; mode = M_NEED; }
                else {
                    mode = M_DRAW;
                    dkind = D_FLIGHT;
                    if (w < S.wmin) { if (COUNT) cnt.roulette++; dkind = D_ROULETTE; }
                }
            }

            // ---- B6: the one Philox block
            if (mode == M_DRAW && (full || dkind == D_FLIGHT)) {
                float r0, r1, r2, r3;
                draw4(seed, id, draw++, r0, r1, r2, r3);
                if (dkind == D_FLIGHT) {
                    rem = -0.69314718f * __builtin_amdgcn_logf(r0);
                    u1 = r1; u2 = r2; u3 = r3;
                    if (lay[k].flags & kLayStep3d) { mode = M_FLY; walked = true; } else mode = M_UNIF;
                } else if (dkind == D_ROULETTE) {
                    if (r0 * S.wfac < w) { w = S.wfac; dkind = D_FLIGHT; }
                    else { if (COUNT) cnt.killed++; mode = M_NEED; }
                } else { // D_LAUNCH
                    float x = r0 * cold->Lx, y = r1 * cold->Ly;
                    if (x >= cold->Lx) x = 0.0f;
                    if (y >= cold->Ly) y = 0.0f;
                    ix = min((int)(x * cold->inv_dx), S.nx - 1);
                    iy = min((int)(y * cold->inv_dy), S.ny - 1);
                    px = fminf(fmaxf(x - (float)ix * S.dx, 0.0f), S.dx);
                    py = fminf(fmaxf(y - (float)iy * S.dy, 0.0f), S.dy);
                    k = S.nz - 1;
                    pz = lay[k].dz;
                    ux = cold->sdx; uy = cold->sdy; uz = cold->sdz;
                    u2 = 1.0f - r2 * (1.0f - cold->cos_cone);
                    u3 = r3;
                    asm volatile("" : "+v"(u3));
                    w = 1.0f;
                    direct = true;
                    kind = E_LAUNCH;
                    mode = M_FINISH;
                }
            }

            // ---- B7: a photon about to walk: the parameters of the walk's first three faces, seen from its origin
            if (walked && mode == M_FLY) {
                walked = false;
                const float4 L = lay4[k * (kLayStride / 4)];
                iux = frcp(fmaxf(fabsf(ux), 1e-20f)); iuy = frcp(fmaxf(fabsf(uy), 1e-20f)); iuz = frcp(fmaxf(fabsf(uz), 1e-20f));
                tx = (ux > 0.0f ? S.dx - px : px) * iux;
                ty = (uy > 0.0f ? S.dy - py : py) * iuy;
                tz = (uz > 0.0f ? L.x - pz : pz) * iuz;
                pz += L.z;
            }

            // ---- back to the slot; what it holds now
            const bool now_ready = wk_ && mode == M_FLY, now_done = wk_ && mode == M_DONE;
            if (wk_) {
                slots[0 * 64 + lane] = make_float4(px, py, pz, w);
                slots[1 * 64 + lane] = make_float4(ux, uy, uz, rem);
                slots[2 * 64 + lane] = make_float4(u1, u2, u3, pend_val);
                slots[3 * 64 + lane] = make_float4(__int_as_float((int)(unsigned)id), __int_as_float((int)(unsigned)(id >> 32)), __int_as_float((int)draw), __int_as_float(pend_pix));
                slots[4 * 64 + lane] = make_float4(__int_as_float(ix | (iy << 16)), __int_as_float(k | (mode << 16)),
                                                   __int_as_float(kind | (dkind << 8) | ((direct ? 1 : 0) << 16) | ((walked ? 1 : 0) << 17)), t);
                if (now_ready) {
                    slots[5 * 64 + lane] = make_float4(tx, ty, tz, 0.0f);
                    slots[6 * 64 + lane] = make_float4(iux, iuy, iuz, 0.0f);
                } else {
                    slots[5 * 64 + lane] = make_float4(__int_as_float(ncx), __int_as_float(ncy), 0.0f, 0.0f);
                    slots[6 * 64 + lane] = make_float4(bt_ev, ev_tab, ev_ks0, ev_apf0);
                }
            }
            const unsigned long long nrdy = __ballot(now_ready), ndn = __ballot(now_done);
            m_ready |= nrdy; m_done |= ndn; m_work &= ~(nrdy | ndn);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        if (m_work == 0ull && m_ready == 0ull && __ballot(has) == 0ull) break;
    }

    if (EMIT) {
        for (unsigned long long q = ev_lo + lane; q < ev_hi; q += 64)
            if (q < (unsigned long long)S.cold->ev_cap) S.cold->ev_list[ev_list_f4(S.cold->ev_cap) * xcc + ev_index((unsigned)q)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // ---- counters: wave reduction, one atomic per wave and counter (cyc[0], cyc[1]: swaps and the lanes that wanted one)
    {
        uint32_t vals[24] = {cnt.photons, cnt.steps, cnt.steps3d, cnt.scatter, cnt.surface, cnt.le_rays,
                             cnt.le_steps, cnt.le_steps3d, cnt.le_column, cnt.flux_tally, cnt.roulette,
                             cnt.killed, cnt.escaped, cnt.absorbed, cnt.a_lanes, cnt.a_slots, cnt.b_lanes, cnt.b_slots,
                             cnt.cyc[0], cnt.cyc[1], cnt.cyc[2], cnt.cyc[3], cnt.cyc[4], cnt.cyc[5]};
        const int ncnt = COUNT ? 24 : 1;
        for (int q = 0; q < ncnt; ++q) {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&S.cold->counters[q], v);
        }
    }
#undef IPA_NOW
}

#define MI3D_POOL_INST(C, P) template __global__ void k_transport_pool<C, P, false>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_pool<C, P, true>(const DevScene, const uint64_t, const uint64_t, const uint64_t);
MI3D_POOL_INST(false, false) MI3D_POOL_INST(false, true) MI3D_POOL_INST(true, false) MI3D_POOL_INST(true, true)
#undef MI3D_POOL_INST

} // namespace mi3d
