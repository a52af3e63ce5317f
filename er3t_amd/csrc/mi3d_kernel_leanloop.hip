// mi3d_kernel_leanloop.hip — the lean photon loop with the local-estimate rays of the marched views walked INSIDE the loop
// (k_transport_leanloop, mi3d_set_kernel choice 2 / MI3D_KERNEL=loop).  Round 2's kernel as it stood at the end of round 3, kept
// for two reasons: it is what serves marched views when the event lists of k_transport_lean<.,.,2> + k_rays find no device memory
// (mi3d_run's fall-back), and it is a third, independently scheduled implementation of the marched views that the parity tests hold
// against the oracle next to the ray kernel and the general loop (tests/test_gpu_fullsize.py::test_config5_*[loop]).  The builds
// without rays in the loop -- column views only, event records for k_rays -- live in mi3d_kernel_lean.hip and have moved on (block C,
// entry records); this file is not where speed is looked for.  One 1-D and one 3-D constituent, analytic phase functions,
// satellite views, any surface model, any solver; same random-number protocol, estimator and formulas as every other build.
#include "mi3d_device.h"

namespace mi3d {

#ifndef MI3D_LEAN_THRESH
#define MI3D_LEAN_THRESH 16   // phase A keeps stepping while at least this many lanes of the wave are walking
#endif
#ifndef MI3D_LEAN_PASS
#define MI3D_LEAN_PASS 2      // column views only: every second pass of phase B is a full one (see k_transport); 3 until the voxel step
                              // was halved: 1 / 2 / 3 / 4 / 6 now give 1.84 / 1.96 / 1.94 / 1.90 / 1.78e9 photons/s (profiles/r03/ab_thresh_pass_les480.log)
#endif
#ifndef MI3D_LEAN_PASS_MARCH
#define MI3D_LEAN_PASS_MARCH 2 // marched views: every second pass serves the photons' events, every pass the rays
#endif
#ifndef MI3D_LEAN_PREFETCH
#define MI3D_LEAN_PREFETCH 1  // 1: the voxel walk asks for the next cell's record before it loops back (0 / 1: 1.943 / 1.988e9 photons/s; a hand-pipelined
                              // walk on two register sets, the read a whole step ahead: 1.836e9 -- profiles/r03/ab_walk_prefetch.log)
#endif
#ifndef MI3D_LEAN_EMIT4
#define MI3D_LEAN_EMIT4 1     // 1: the build that writes event records gets the register budget of four waves per SIMD like the marching one
#endif
#ifndef MI3D_LEAN_WAVES
#define MI3D_LEAN_WAVES(COUNT, MARCH) (((COUNT) || (MARCH)) ? 4 : 5)   // waves per SIMD the register budget must allow
#endif

// MARCH: 0 every view is answered from the column table; 1 the rays of the other views are marched inside this loop;
//        2 they are marched by k_rays: this kernel only writes an event record for every collision and reflection (k_rays' header)
// TWO: the voxels carry a second 3-D constituent (er3t's cloud + aerosol scenes); a build of its own because even wave-uniform
//      branches around it cost the one-constituent bench 0.8 % (profiles/r02/ab_second_constituent_cost.log)
template <bool COUNT, bool P3D, int MARCH, bool TWO>
__global__ void __launch_bounds__(256, MI3D_LEAN_WAVES(COUNT, MARCH == 1 || (MARCH == 2 && MI3D_LEAN_EMIT4)))
k_transport_leanloop(const DevScene S, const uint64_t nphoton, const uint64_t seed, const uint64_t offset) {
    constexpr bool MLOOP = (MARCH == 1), MIXED = (MARCH != 0), EMIT = (MARCH == 2);
    extern __shared__ float4 smem[];
    // layer table in LDS with one record more at either end: layer -1 (below the surface) and layer nz (above the top) read as
    // horizontally uniform layers of no thickness, so that the voxel walk needs no bounds check when it crosses a level: a photon
    // that leaves the atmosphere either way is found by the block that serves uniform layers
    constexpr int kL4 = kLayStride / 4;
    const float4 *lay4 = smem + kL4;
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(lay4);
    const int o_view = (S.nz + 2) * kL4;
    const ViewRec *views = reinterpret_cast<const ViewRec *>(smem + o_view);
    const DevCold *cold = reinterpret_cast<const DevCold *>(smem + o_view + MI3D_MAX_VIEW * 2);
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.cold->lay);
        for (int i = threadIdx.x; i < S.nz * kL4; i += blockDim.x) smem[kL4 + i] = src[i];
        if (threadIdx.x < 2 * kL4) smem[threadIdx.x < kL4 ? threadIdx.x : (S.nz + 1) * kL4 + (threadIdx.x - kL4)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 *vsrc = reinterpret_cast<const float4 *>(S.cold->views);
        for (int i = threadIdx.x; i < S.nview * 2; i += blockDim.x) smem[o_view + i] = vsrc[i];
        const float4 *csrc = reinterpret_cast<const float4 *>(S.cold);
        if (threadIdx.x < kColdF4) smem[o_view + MI3D_MAX_VIEW * 2 + threadIdx.x] = csrc[threadIdx.x];
    }
    __syncthreads();

    const bool ipa_all = (S.solver == MI3D_SOLVER_IPA);
    const bool same_grid = (S.nxr == S.nx) && (S.nyr == S.ny);
    const bool plain = (S.target & kTargetPlainPhase) != 0;   // Rayleigh + Henyey-Greenstein: no selector is looked at
#define IPA_NOW(is_le_) (ipa_all || (P3D && ((is_le_) || !direct)))
    Counters cnt = {};
    // byte offsets into the voxel records: record of (ix, iy, k) at vbase + iy*sy_b + ix*sx_b + k*16
    const unsigned sx_b = S.vcol_f4 * 16u, sy_b = S.vrow_f4 * 16u;
    const char *vbase = reinterpret_cast<const char *>(S.vrec) - (long)S.k3lo * 16;

    // ---- lane state
    // photon: outside its walk (px, py, pz) is the position inside the voxel (ix, iy, k).  During a walk no position is carried at
    // all: where a ray is inside its voxel follows from how far its parameter is from the three faces ahead, (tx - t) |ux| from the
    // x face and so on, and is worked out when the walk ends.  While the rays of an event are walked, (px, py, pz) stay the
    // event's position and the event's direction and cell wait in (eux..euz), (eix, eiy, ek).
    float px = 0, py = 0, pz = 0, ux = 0, uy = 0, uz = 1, iux = 1, iuy = 1, iuz = 1;
    float t = 0, tx = 0, ty = 0, tz = 0;   // ray parameter now / at the next x, y, z face
    int ix = 0, iy = 0, k = 0, stepx = 0, stepy = 0; // stepx/y: column step per crossing (0 under IPA)
    int wrapx = 0, wrapy = 0, stepk = 1;              // the column a step across the domain's edge leads to; layer step per level crossing
    float rem = 0.0f;   // optical depth left: to the photon's collision, or before the local-estimate ray is given up (< 0: given up)
    float w = 0.0f;
    float u1 = 0, u2 = 0, u3 = 0;
    uint64_t id = 0;
    uint32_t draw = 0;
    int mode = M_NEED, kind = E_LAUNCH, dkind = D_LAUNCH;
    bool direct = false, walked = false;   // walked: the lane has just left a walk (position to be reconstructed) or is about to start one (to be set up)
    unsigned long long pool_next = 0, pool_end = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID (speed only)
    unsigned victim = 0;
    int pend_pix = -1;
    float pend_val = 0.0f;
    // the voxel record the walk read last: {total extinction, optical depth above the voxel, omega*ext and apf of the first 3-D
    // constituent}.  A lane that stops walking keeps it: it IS the record of the voxel its event lies in (the loop with the rays
    // marched inside takes a copy: the rays' walks read on).
    float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 evr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float &bt_ev = MLOOP ? evr.x : rec.x, &ev_tab = MLOOP ? evr.y : rec.y, &ev_ks0 = MLOOP ? evr.z : rec.z, &ev_apf0 = MLOOP ? evr.w : rec.w;
    float ev_ksb = 0.0f, ev_apfb = 0.0f;   // the second 3-D constituent of the event's voxel (np3d = 2), else 0
    float &ev_sfc = ev_tab;
    // local-estimate rays (MLOOP): the event they belong to, the ray's own walk origin, what it carries
    float eux = 0, euy = 0, euz = 0, zev = 0;
    int eix = 0, eiy = 0, ek = 0, iv = 0;
    float rox = 0, roy = 0, roz = 0, rpz = 0;   // origin of the ray's walk (frame of the voxel it started in, absolute height); height inside its layer while in uniform layers
    float contrib = 0.0f, tkill = kTauCut, zstop = 0.0f;
    bool emit = false;   // EMIT: this lane's event of the current pass is to be written to the event list
    unsigned long long ev_lo = 0, ev_hi = 0;   // EMIT, wave-uniform: slots of this XCD's list reserved by this wave and not yet used

#ifdef MI3D_MARKS
#define MI3D_MARK(name) asm volatile("; MARK " name)
#else
#define MI3D_MARK(name)
#endif
#ifdef MI3D_CENSUS
#define MI3D_TICK(slot) do { if (COUNT && (slot) < 3) { const long long t_ = clock64(); cnt.cyc[slot] += (uint32_t)((t_ - tick) >> 6); tick = t_; } } while (0)
#else
#define MI3D_TICK(slot) do { if (COUNT) { const long long t_ = clock64(); cnt.cyc[slot] += (uint32_t)((t_ - tick) >> 6); tick = t_; } } while (0)
#endif
    long long tick = COUNT ? clock64() : 0;   // instrumented build: wave clock ticks / 64 spent in A, walk end + B0, B1 + B2, B3 + B4, B5, B6 + B7
    unsigned pass_ctr = 0;
    for (;;) {
        // =================================== phase A: voxel steps ===================================
        // (Reading the next voxel's record one step ahead -- its address does not depend on the record being waited for -- was
        //  tried, with the two records in ping-pong register sets: no gain on the nine-view configuration, 6 % slower on the
        //  nadir one, profiles/r02/lean_walk_prefetch.log: the walk does not wait for memory.)
        MI3D_MARK("A");
        for (;;) {
            const bool flying = (mode <= M_LE);
            const int nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (nfly < MI3D_LEAN_THRESH && __ballot(mode > M_LE && mode != M_DONE) != 0ull) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (!MLOOP) {
#if MI3D_LEAN_PREFETCH
              if (flying) {
                // ---- photons only, the record of the NEXT cell asked for before this cell's record is looked at: where the ray goes
                // next follows from the face parameters alone, so the read of step i+1 travels while step i is worked out (a read
                // too many per walk: the one behind a collision)
                const float tn = fminf(fminf(tx, ty), tz);
                const bool zf = (tz == tn), xf = !zf && (tx == tn);
                int nix = ix, niy = iy, nk = k;
                if (zf) nk = k + stepk;
                else if (xf) { const int c = ix + stepx; nix = (unsigned)c >= (unsigned)S.nx ? wrapx : c; }
                else { const int c = iy + stepy; niy = (unsigned)c >= (unsigned)S.ny ? wrapy : c; }
                const int kk = min(max(nk, S.k3lo), S.k3lo + S.nz3 - 1);     // (inside the voxel table whatever lies beyond the level)
                const float4 recn = *reinterpret_cast<const float4 *>(vbase + ((unsigned)niy * sy_b + (unsigned)nix * sx_b + (unsigned)kk * 16u));
                const float dtau = rec.x * (tn - t);
                if (COUNT) { cnt.steps++; cnt.steps3d++; }
                if (dtau >= rem) { mode = M_COLL; walked = true; }
                else {
                    rem -= dtau;
                    t = tn;
                    ix = nix; iy = niy;
                    if (zf) {
                        k = nk;
                        const float4 Ln = lay4[k * kL4];
                        tz = fmaf(Ln.x, iuz, tz);
                        if (!(__float_as_int(Ln.w) & kLayStep3d)) { mode = M_UNIF; walked = true; }
                    } else if (xf) tx = fmaf(S.dx, iux, tx);
                    else ty = fmaf(S.dy, iuy, ty);
                    rec = recn;
                }
              }
#else
              if (flying) {
                // ---- photons only: one record, min3, one multiply, one compare -- and ONE face parameter moved on
                rec = *reinterpret_cast<const float4 *>(vbase + ((unsigned)iy * sy_b + (unsigned)ix * sx_b + (unsigned)k * 16u));
                const float tn = fminf(fminf(tx, ty), tz);
                const float dtau = rec.x * (tn - t);
                if (COUNT) { cnt.steps++; cnt.steps3d++; }
                if (dtau >= rem) { mode = M_COLL; walked = true; }   // the collision lies inside this voxel: at t + rem / bt (phase B)
                else {
                    rem -= dtau;
                    t = tn;
                    if (tz == tn) {
                        k += stepk;                                  // (-1 and nz: the table's end records, uniform layers)
                        const float4 Ln = lay4[k * kL4];
                        tz = fmaf(Ln.x, iuz, tz);
                        if (!(__float_as_int(Ln.w) & kLayStep3d)) { mode = M_UNIF; walked = true; }
                    } else if (tx == tn) {
                        tx = fmaf(S.dx, iux, tx);
                        const int c = ix + stepx;
                        ix = (unsigned)c >= (unsigned)S.nx ? wrapx : c;
                    } else {
                        ty = fmaf(S.dy, iuy, ty);
                        const int c = iy + stepy;
                        iy = (unsigned)c >= (unsigned)S.ny ? wrapy : c;
                    }
                }
              }
#endif
            } else
            if (flying) {
                const bool is_le = MLOOP && (mode == M_LE);
                rec = *reinterpret_cast<const float4 *>(vbase + ((unsigned)iy * sy_b + (unsigned)ix * sx_b + (unsigned)k * 16u));
                const float4 &r4 = rec;
                const float tn = fminf(fminf(tx, ty), tz);
                float dtau = r4.x * (tn - t);
                if (COUNT) {
                    if (is_le) { cnt.le_steps++; cnt.le_steps3d++; }
                    else { cnt.steps++; cnt.steps3d++; }
                }
                // a sensor plane inside the atmosphere ends the ray inside this voxel
                bool plane = false;
                if (MLOOP && is_le && zstop < INFINITY) {
                    const float zn = fmaf(uz, tn, roz);
                    plane = uz > 0.0f ? zn >= zstop : zn <= zstop;
                    if (plane) dtau = r4.x * fabsf(zstop - fmaf(uz, t, roz)) * iuz;
                }
                if (dtau >= rem) {
                    if (is_le) { rem = -1.0f; mode = M_LEEND; }   // the ray's budget is used up: given up
                    else {
                        // ---- the collision lies inside this voxel: at t + rem / bt (worked out in phase B)
                        if (MLOOP) evr = r4;
                        mode = M_COLL; walked = true;
                    }
                } else if (MLOOP && plane) {
                    rem -= dtau;
                    mode = M_LEEND;
                } else {
                    rem -= dtau;
                    t = tn;
                    if (tz == tn) {
                        const bool up = uz > 0.0f;
                        const int knew = up ? k + 1 : k - 1;
                        if (knew >= S.nz) {
                            if (is_le) mode = M_LEEND;
                            else { if (COUNT) cnt.escaped++; mode = M_NEED; }
                        } else if (knew < 0) {
                            if (is_le) mode = M_LEEND;   // (a ray towards an up-looking sensor on the ground, ended by rounding)
                            else {
                                mode = M_SURF; walked = true;
                                if (MLOOP) evr = r4;
                            }
                        } else {
                            const float4 Ln = lay4[knew * (kLayStride / 4)];
                            tz = fmaf(Ln.x, iuz, tz);
                            if (!(__float_as_int(Ln.w) & kLayStep3d)) {
                                if (is_le) { mode = M_LEUNIF; rpz = up ? 0.0f : Ln.x; }
                                else { mode = M_UNIF; walked = true; }
                            }
                            k = knew;
                        }
                    } else if (tx == tn) {
                        tx = fmaf(S.dx, iux, tx);
                        int c = ix + stepx;
                        c = c >= S.nx ? 0 : (c < 0 ? S.nx - 1 : c);
                        ix = c;
                    } else {
                        ty = fmaf(S.dy, iuy, ty);
                        int c = iy + stepy;
                        c = c >= S.ny ? 0 : (c < 0 ? S.ny - 1 : c);
                        iy = c;
                    }
                }
            }
        }

        // =================================== phase B ===================================
        MI3D_TICK(0);
        MI3D_MARK("B0");
        if (COUNT) { cnt.b_slots++; if (mode > M_LE && mode != M_DONE) cnt.b_lanes++; }
        // Column views only: every MI3D_LEAN_PASS-th pass is a full one, the passes between serve collisions only (see k_transport).
        // With marched views the rays are the common work: every pass serves them, every MI3D_LEAN_PASS_MARCH-th the photons' events.
        bool evt_m = true;
        if (MLOOP) evt_m = MI3D_LEAN_PASS_MARCH <= 1 || ((pass_ctr++ % (unsigned)(MI3D_LEAN_PASS_MARCH)) == 0u) ||
                           __ballot(mode == M_LEEND || mode == M_VIEWS || mode == M_LEUNIF) == 0ull;
        const bool full = MLOOP ? evt_m : (MI3D_LEAN_PASS <= 1 || ((pass_ctr++ % (unsigned)(MI3D_LEAN_PASS)) == 0u) ||
                          __ballot(mode == M_COLL || (mode == M_FINISH && (kind & 15) != E_SURFACE) || (mode == M_DRAW && dkind == D_FLIGHT)) == 0ull);
#define EVT (!MLOOP || evt_m)

        // ---- where a photon's walk has ended: inside its voxel, as far from the faces ahead as its parameter is from theirs
        if (MLOOP ? (walked && (mode == M_COLL || mode == M_SURF || mode == M_UNIF)) : walked) {
            walked = false;
            const float tc = (mode == M_COLL) ? fmaf(rem, frcp(bt_ev), t) : t;
            const float4 L = lay4[k * (kLayStride / 4)];
            // (|u| floored as where the parameters were set up, B7: a photon flying exactly along an axis keeps its place across it)
            const float ax = fminf(fmaxf((tx - tc) * fmaxf(fabsf(ux), 1e-20f), 0.0f), S.dx), ay = fminf(fmaxf((ty - tc) * fmaxf(fabsf(uy), 1e-20f), 0.0f), S.dy);
            px = ux > 0.0f ? S.dx - ax : ax;
            py = uy > 0.0f ? S.dy - ay : ay;
            if (mode == M_COLL) {
                const float az = fminf(fmaxf((tz - tc) * fmaxf(fabsf(uz), 1e-20f), 0.0f), L.x);
                pz = uz > 0.0f ? L.x - az : az;
            } else pz = (mode == M_SURF || uz > 0.0f) ? 0.0f : L.x;   // on a level: bottom of the layer entered going up (and the surface), top going down
        }

        // ---- B0: photons inside runs of horizontally uniform layers: the whole rest of the run at once
        if (!MLOOP && full && mode == M_UNIF && (k < 0 || k >= S.nz)) {
            // the walk has left the atmosphere (the layer table's end records): out through the top, or onto the surface
            if (k < 0) { k = 0; pz = 0.0f; mode = M_SURF; }
            else { if (COUNT) cnt.escaped++; mode = M_NEED; }
        }
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
            // where the flight through the run ends: (layer, height in it), how far it went, what comes next
            int knew, next;
            float pzn, s;
            if (tpath < rem) {
                rem -= tpath;
                s = hv * iuzl;
                knew = up ? kend + 1 : kend - 1;
                pzn = 0.0f;
                next = M_FLY;                                            // (the walk is set up in B7)
                if (knew >= S.nz) { if (COUNT) cnt.escaped++; next = M_NEED; }
                else if (knew < 0) { knew = 0; next = M_SURF; }
                else if (!up) pzn = lay4[knew * kL4].x;
            } else {
                // the collision lies inside the run: bisection on the vertical optical depth below every layer
                const float T = Lk.tauz + Lk.bt * pz + (up ? rem : -rem) * fabsf(uz);
                int lo = up ? k : kend, hi = up ? kend : k;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (lay[mid].tauz <= T) lo = mid; else hi = mid - 1;
                }
                const float4 Lj = lay4[lo * kL4];     // {dz, bt, zlo, flags}
                pzn = fminf(fmaxf((T - lay[lo].tauz) * frcp(fmaxf(Lj.y, 1e-30f)), 0.0f), Lj.x);
                s = fabsf((Lj.z + pzn) - (Lk.zlo + pz)) * iuzl;
                knew = lo;
                bt_ev = Lj.y;
                next = M_COLL;
            }
            if (COUNT) cnt.steps++;
            px += ux * s; py += uy * s;
            k = knew; pz = pzn;
            // one fold for every way out of the run (the event blocks below find the position inside its column)
            fold_xy(S, cold, px, py, ix, iy, IPA_NOW(false));
            mode = next;
            if (next == M_FLY) walked = true;
        }

        // ---- B0': local-estimate rays inside runs of uniform layers
        if (MLOOP && mode == M_LEUNIF) {
            const bool up = uz > 0.0f;
            bool reenter = false;
            if (!(zstop < INFINITY)) {
                // the whole rest of the run at once, from the prefix sums of the layer table
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
                // a sensor plane inside the atmosphere ends the ray somewhere in the run: layer by layer
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
                // into layers that are walked voxel by voxel: where the ray is now becomes the origin of its walk
                // (the ray's horizontal position in the frame of the column it left the voxels in: linear in its parameter, (tx - t) |ux|
                //  short of the x face that was ahead of it there -- beyond it by now, fold_xy brings it home)
                // (|u| floored as where the parameters were set up: an exactly vertical ray keeps its place)
                const float dxo = (tx - t) * fmaxf(fabsf(ux), 1e-20f), dyo = (ty - t) * fmaxf(fabsf(uy), 1e-20f);
                float xo = ux > 0.0f ? S.dx - dxo : dxo;
                float yo = uy > 0.0f ? S.dy - dyo : dyo;
                fold_xy(S, cold, xo, yo, ix, iy, IPA_NOW(true));
                rox = xo; roy = yo; roz = lay[k].zlo + rpz;
                t = 0.0f;
                mode = M_LE; walked = true;
            }
        }

        MI3D_TICK(1);
        MI3D_MARK("B1");
        // ---- B1: a local-estimate ray has arrived (or has been given up): tally it
        if (MLOOP && mode == M_LEEND) {
            if (rem >= 0.0f) {
                const ViewRec V = views[iv];
                const float acc = tkill - rem;
                // pixel = where the line of sight through the event meets zreg
                float xr = (float)eix * S.dx + px, yr = (float)eiy * S.dy + py;
                if (!IPA_NOW(true)) {
                    const float tt = (zev - V.zreg) * frcp(V.vz);
                    xr -= V.vx * tt; yr -= V.vy * tt;
                    xr -= floorf(xr * cold->inv_Lx) * cold->Lx; yr -= floorf(yr * cold->inv_Ly) * cold->Ly;
                }
                const int ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                const int jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                RAD_ADD(&S.rad[(unsigned)((iv * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride],
                        contrib * fexp_neg((V.roulette & 1) ? fminf(acc, cold->le_tau1) : acc) * frcp(fabsf(V.vz)));
            }
            iv += 1;
            mode = M_VIEWS;
        }

        MI3D_MARK("B2");
        // ---- B2: a new event: weight, local estimates answered from the column table
        if (EVT && (mode == M_COLL || (full && mode == M_SURF))) {
            const float4 L = lay4[k * (kLayStride / 4)];              // {dz, bt, zlo, flags}
            const int flags = __float_as_int(L.w);
            const bool in3d = (flags & kLayIn3d) != 0;
            const LayerRec &Lk = lay[k];
            // (an event inside a uniform layer was found by B0, which has folded the position into its column)
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
                float kstot = ks1 + ks3;
                if (TWO) {
                    // er3t's cloud + aerosol scenes, mca_atm.py: a second {omega*ext, apf} pair per voxel
                    ev_ksb = 0.0f;
                    if (in3d) {
                        const float2 cs = cold->csca[(col * (unsigned)S.nz3 + (unsigned)(k - S.k3lo)) * 2u + 1u];
                        ev_ksb = cs.x; ev_apfb = cs.y;
                    }
                    kstot += ev_ksb;
                }
                // (exactly 1 for conservative scattering: the approximate reciprocal must not nudge a weight that sits on
                //  the roulette threshold below it)
                w *= (kstot >= bt_ev) ? 1.0f : kstot * frcp(bt_ev);
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; dead = true; }
                if (any_col) {
                    // mixture phase function towards the zenith (a column view looks straight down): cos(angle) = uz
                    float P = 0.0f;
                    if (plain) {
                        // (a constituent that is not there has a coefficient of 0 and a harmless selector: the last voxel's, or 0)
                        P = ks1 * (0.75f * fmaf(uz, uz, 1.0f)) + ks3 * phase_eval_hg(ev_apf0, uz);
                        if (TWO) P += ev_ksb * phase_eval_hg(ev_apfb, uz);
                    } else {
                        if (ks1 > 0.0f) P = ks1 * phase_eval_analytic(Lk.apf1d[0], uz);
                        if (ks3 > 0.0f) P += ks3 * phase_eval_analytic(ev_apf0, uz);
                        if (TWO && ev_ksb > 0.0f) P += ev_ksb * phase_eval_analytic(ev_apfb, uz);
                    }
                    c = w * P * frcp(kstot) * (0.25f / kPi);
                }
                kind = E_SCATTER;
            }
            if (dead) {
                mode = M_NEED;
            } else {
                if (any_col) {
                    const float tau = bt_ev * (L.x - pz) + tcol_here;
                    // the pixel under the event: its column where the image has one pixel per column (er3t's satellite images:
                    // Rad_nxr = Atm_nx, Rad_nyr = Atm_ny, mcarats.py:360-367)
                    int ir = ix, jr = iy;
                    if (!same_grid) {
                        const float xr = (float)ix * S.dx + px, yr = (float)iy * S.dy + py;
                        ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                        jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                    }
                    const float val = c * fexp_neg(tau);
                    // consecutive tallies of one history into the same pixel are summed in a register (first column view: S.col0)
                    const int jv0 = MIXED ? S.col0 : 0;
                    if (COUNT) { const int nc = MIXED ? S.nview - S.nmarch : S.nview; cnt.le_rays += nc; cnt.le_column += nc; }
                    if (c > 0.0f) {
                        const int pix = (jv0 * S.nyr + jr) * S.rad_row + ir;
                        if (pix == pend_pix) pend_val += val;
                        else {
                            if (pend_pix >= 0) { MI3D_TALLY_CENSUS(&S.rad[(unsigned)pend_pix * (unsigned)S.rad_stride]); RAD_ADD(&S.rad[(unsigned)pend_pix * (unsigned)S.rad_stride], pend_val); }
                            pend_pix = pix; pend_val = val;
                        }
                        if (!MIXED || S.nview - S.nmarch > 1)    // (further column views: none in a nadir + slant set)
                        for (int jv = jv0 + 1; jv < S.nview; ++jv)
                            if (!MIXED || views[jv].column) RAD_ADD(&S.rad[(unsigned)((jv * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride], val);
                    }
                }
                if (MLOOP && S.nmarch > 0) {
                    // the event waits in registers while the rays of its marched views are walked
                    eux = ux; euy = uy; euz = uz; eix = ix; eiy = iy; ek = k; zev = L.z + pz;
                    iv = 0;
                    mode = M_VIEWS;
                } else mode = M_FINISH;
                if (EMIT) emit = true;
            }
        }

        if (EMIT) {
            // ---- the event goes to this XCD's list for k_rays; the photon carries on at once.  A wave reserves room for
            // kEvBlock records at a time (one returning atomic per block instead of one per pass: the wave waits for it) and
            // hands the slots out itself; what it leaves unused is marked empty (weight 0) before it reserves again or ends.
            const unsigned long long em = __ballot(emit);
            if (em != 0ull) {
                const unsigned n = (unsigned)__popcll(em);
                if (ev_lo + n > ev_hi) {
                    for (unsigned long long q = ev_lo + (threadIdx.x & 63); q < ev_hi; q += 64)
                        if (q < (unsigned long long)cold->ev_cap) cold->ev_list[ev_list_f4(cold->ev_cap) * xcc + ev_index((unsigned)q)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    const int leader = __ffsll((long long)em) - 1;
                    unsigned long long base = 0;
                    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(cold->ev_ctr + xcc * kCtrStride, (unsigned long long)kEvBlock);
                    base = __shfl(base, leader, 64);
                    ev_lo = base; ev_hi = base + kEvBlock;
                }
                if (emit) {
                    const unsigned long long slot = ev_lo + __builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, 0u));
                    if (slot < (unsigned long long)cold->ev_cap) {
                        // (plain stores: write-through ones that bypass the XCD's L2, `sc1`, were 10 % slower -- the four 16-byte
                        //  pieces of a record then leave one by one, profiles/r02/mv9_event_stores.log)
                        float4 *lbase = cold->ev_list + ev_list_f4(cold->ev_cap) * xcc;       // (this XCD's list: wave-uniform)
                        float4 *e = lbase + ev_index((unsigned)slot);
#ifdef MI3D_ABL_NOEMITSTORE   // ablation (results wrong): what the stores of the event records cost the photon loop
                        asm volatile("" ::"v"(px), "v"(py), "v"(pz), "v"(w), "v"(ux), "v"(uy), "v"(uz), "v"(ev_ks0), "v"(ev_apf0), "v"(ev_sfc), "v"(e));
#else
                        e[0] = make_float4(px, py, pz, w);
                        e[kEvStride] = make_float4(ux, uy, uz, ev_ks0);
                        e[2 * kEvStride] = make_float4(ev_apf0, ev_sfc, __int_as_float(ix | (iy << 16)), __int_as_float(k | (kind << 16)));
                        reinterpret_cast<uint32_t *>(lbase)[ev_word((unsigned)slot)] = le_hash_base(seed, id, draw);
#endif
                    } else cold->ev_ctr[8 * kCtrStride] = 1ull;   // list full: the launch is reported as failed (mi3d_run), never silently short
                    emit = false;
                }
                ev_lo += n;
            }
        }

        MI3D_TICK(2);
        MI3D_MARK("B3");
        // ---- B3: start the local-estimate ray of the next marched view, if any is left
        // (the rays of a surface event start in the passes that serve the photons' events: their reflectance models are long
        //  and rare -- some lane of the wave would otherwise drag them into half of all passes)
        if (MLOOP && mode == M_VIEWS && (evt_m || (kind & 15) != E_SURFACE)) {
            // skip the views answered from the column table, the sensors on the wrong side of the event, and -- for a surface
            // event -- the up-looking ones
            while (iv < S.nview && (views[iv].column || (views[iv].vz > 0.0f ? zev >= views[iv].zs : (zev <= views[iv].zs || (kind & 15) == E_SURFACE)))) ++iv;
            if (iv >= S.nview) {
                ux = eux; uy = euy; uz = euz; ix = eix; iy = eiy; k = ek;   // the photon takes its direction and cell back
                mode = M_FINISH;
            } else {
                const ViewRec V = views[iv];
                const LayerRec &Lk = lay[ek];
                float c;
                if ((kind & 15) == E_SURFACE) {
                    ix = eix; iy = eiy;
                    const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ix, iy, px, py) : Sfc{kind >> 4, ev_ks0, ev_apf0, ev_sfc, 0.0f, 0.0f};
                    c = w * surface_R(sf, eux, euy, euz, V.vx, V.vy, V.vz) * V.vz * (1.0f / kPi);
                } else {
                    const float mu = eux * V.vx + euy * V.vy + euz * V.vz;
                    const float ks1 = Lk.ks1d[0], ks3 = (Lk.flags & kLayIn3d) ? ev_ks0 : 0.0f;
                    float P = 0.0f;
                    if (ks1 > 0.0f) P = ks1 * phase_eval_analytic(Lk.apf1d[0], mu);
                    if (ks3 > 0.0f) P += ks3 * phase_eval_analytic(ev_apf0, mu);
                    float kst = ks1 + ks3;
                    if (TWO) {
                        if (ev_ksb > 0.0f) P += ev_ksb * phase_eval_analytic(ev_apfb, mu);
                        kst += ev_ksb;
                    }
                    c = w * P * frcp(kst) * (0.25f / kPi);
                }
                if (COUNT) cnt.le_rays++;
                if (V.roulette & 2) c = le_weight_roulette(c, cold->le_cmin, seed, id, draw, iv);
                if (c > 0.0f) {
                    contrib = c;
                    ux = V.vx; uy = V.vy; uz = V.vz;
                    zstop = (uz < 0.0f || V.zs < cold->ztoa) ? V.zs : INFINITY; // a sensor above the atmosphere is never reached
                    // roulette: the ray survives to optical depth tau with probability min(1, exp(-(tau - tau1))) and then carries
                    // exp(-min(tau, tau1)); one hashed uniform number per ray fixes where it ends
                    tkill = (V.roulette & 1) ? cold->le_tau1 - 0.69314718f * __builtin_amdgcn_logf(le_roulette_u(seed, id, draw, iv)) : kTauCut;
                    rem = tkill;
                    ix = eix; iy = eiy; k = ek;
                    rox = px; roy = py; roz = zev; rpz = pz;
                    t = 0.0f;
                    if (Lk.flags & kLayStep3d) { mode = M_LE; walked = true; }
                    else {
                        // (a ray that starts inside uniform layers: its horizontal position is read off the x and y face parameters
                        //  when it enters layers that are walked, so they are set here)
                        mode = M_LEUNIF;
                        iux = frcp(fmaxf(fabsf(ux), 1e-20f)); iuy = frcp(fmaxf(fabsf(uy), 1e-20f)); iuz = frcp(fmaxf(fabsf(uz), 1e-20f));
                        tx = (ux > 0.0f ? S.dx - rox : rox) * iux;
                        ty = (uy > 0.0f ? S.dy - roy : roy) * iuy;
                    }
                } else {
                    iv += 1; // nothing to carry: look at the next view on the next pass
                }
            }
        }

        MI3D_MARK("B4");
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
                    if ((int)(threadIdx.x & 63) == leader) b = atomicAdd(cold->next_photon + x * kCtrStride, (unsigned long long)kChunk);
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

        MI3D_TICK(3);
        MI3D_MARK("B5");
        // ---- B5: finish the event (scattering, surface reflection or launch): new direction and weight
        if (EVT && mode == M_FINISH && (full || (kind & 15) != E_SURFACE)) {
            float bx = ux, by = uy, bz = uz, mu_rot = u2;
            if ((kind & 15) == E_SURFACE) {
                bx = 0.0f; by = 0.0f; bz = 1.0f;
                mu_rot = fsqrt(u2);
            } else if ((kind & 15) == E_SCATTER) {
                const LayerRec &Lk = lay[k];
                const bool in3d = (Lk.flags & kLayIn3d) != 0;
                const float ks1 = Lk.ks1d[0], ks3 = in3d ? ev_ks0 : 0.0f;
                // choose the constituent that scatters: the 1-D one first, then the 3-D ones in their order
                float kst = ks1 + ks3;
                if (TWO) kst += ev_ksb;
                const float target = u1 * kst;
                const bool first = (target < ks1) || !in3d;
                float apf_sel = first ? Lk.apf1d[0] : ev_apf0;
                if (TWO && !first && !(target < ks1 + ks3)) apf_sel = ev_apfb;
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

        MI3D_TICK(4);
        MI3D_MARK("B6");
        // ---- B6: the one Philox block
        if (EVT && mode == M_DRAW && (full || dkind == D_FLIGHT)) {
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

        MI3D_MARK("B7");
        // ---- B7: a lane about to walk: the parameters of the walk's first three faces, seen from its origin
        if (MLOOP ? (walked && mode <= M_LE) : (walked && mode == M_FLY)) {
            walked = false;
            const bool is_le = MLOOP && (mode == M_LE);
            const float4 L = lay4[k * (kLayStride / 4)];
            iux = frcp(fmaxf(fabsf(ux), 1e-20f)); iuy = frcp(fmaxf(fabsf(uy), 1e-20f)); iuz = frcp(fmaxf(fabsf(uz), 1e-20f));
            const float ox = is_le ? rox : px, oy = is_le ? roy : py, oz = is_le ? rpz : pz;   // oz: height inside the layer
            tx = (ux > 0.0f ? S.dx - ox : ox) * iux;
            ty = (uy > 0.0f ? S.dy - oy : oy) * iuy;
            tz = (uz > 0.0f ? L.x - oz : oz) * iuz;
            if (!is_le) t = 0.0f;   // (a ray's origin is set where the ray starts or re-enters)
            const bool ipa = IPA_NOW(is_le);
            stepx = ipa ? 0 : (ux > 0.0f ? 1 : -1);
            stepy = ipa ? 0 : (uy > 0.0f ? 1 : -1);
            if (!MLOOP) { wrapx = ux > 0.0f ? 0 : S.nx - 1; wrapy = uy > 0.0f ? 0 : S.ny - 1; stepk = uz > 0.0f ? 1 : -1; }
#if MI3D_LEAN_PREFETCH
            if (!MLOOP) rec = *reinterpret_cast<const float4 *>(vbase + ((unsigned)iy * sy_b + (unsigned)ix * sx_b + (unsigned)k * 16u));   // the walk's first record, on its way while the pass ends
#endif
        }

        MI3D_TICK(5);
        MI3D_MARK("END");
        if (__ballot(mode != M_DONE) == 0ull) break;
    }
#undef MI3D_TICK
#undef EVT

    if (EMIT) {
        for (unsigned long long q = ev_lo + (threadIdx.x & 63); q < ev_hi; q += 64)
            if (q < (unsigned long long)S.cold->ev_cap) S.cold->ev_list[ev_list_f4(S.cold->ev_cap) * xcc + ev_index((unsigned)q)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // ---- counters: wave reduction, one atomic per wave and counter
    {
        uint32_t vals[24] = {cnt.photons, cnt.steps, cnt.steps3d, cnt.scatter, cnt.surface, cnt.le_rays,
                             cnt.le_steps, cnt.le_steps3d, cnt.le_column, cnt.flux_tally, cnt.roulette,
                             cnt.killed, cnt.escaped, cnt.absorbed, cnt.a_lanes, cnt.a_slots, cnt.b_lanes, cnt.b_slots,
                             cnt.cyc[0], cnt.cyc[1], cnt.cyc[2], cnt.cyc[3], cnt.cyc[4], cnt.cyc[5]};
        const int ncnt = COUNT ? 24 : 1;
        for (int q = 0; q < ncnt; ++q) {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&S.cold->counters[q], v);
        }
    }
#undef IPA_NOW
}

#define MI3D_LEAN_INST(C, P) template __global__ void k_transport_leanloop<C, P, 1, false>(const DevScene, const uint64_t, const uint64_t, const uint64_t);
MI3D_LEAN_INST(false, false) MI3D_LEAN_INST(false, true) MI3D_LEAN_INST(true, false) MI3D_LEAN_INST(true, true)
#undef MI3D_LEAN_INST

} // namespace mi3d
