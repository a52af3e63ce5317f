// mi3d_kernel_rays.hip — the local-estimate rays of marched satellite views, and of cameras, as a kernel of their own.
//
// A local-estimate ray feeds nothing back into the photon it comes from, so the two need not share a lane.  Inside one loop
// (k_transport, k_transport_lean<.,.,1>) a lane walks the eight rays of an event one after the other while its photon waits,
// and the blocks that serve the photons' own events run for the few lanes that happen to need them: on the nine-view
// configuration 25 % of the lanes are active per vector instruction (profiles/r02/pmc_les480_mv9_lean_loop.txt).  Here the
// photon loop (k_transport_lean<.,.,2>, as lean as the nadir build) only WRITES an event record per collision and reflection
// -- 64 bytes, into the list of the XCD it runs on -- and carries on; this kernel then takes (event, view) pairs off those
// lists.  The lists are per XCD and in the order the photon loop met the events, which is the order of the photons' start
// tiles: an XCD marches rays through the voxels its L2 still holds.
//
// A ray is short (five voxel steps on the nine-view configuration) and expensive to start (the event record, the phase
// function towards the view, the roulettes, the pixel), so how the STARTS are scheduled decides the lane utilisation.
// Starting rays only in the lanes that happen to be free ran the ~90 instructions of a start for 40 lanes of 64 at best
// (profiles/r02/pmc_les480_mv9_rays.txt: 45 % of the lanes active).  Now every wave keeps a small pool of started rays in LDS
// (kPool records of 32 bytes, private to the wave: no atomics, no barriers):
//   * a wave takes a CHUNK of 64 events off a list, one event per lane, reads and takes each apart ONCE (round 3: until then
//     every (event, view) pair read and decoded its event again: eight times per event on the nine-view configuration);
//   * a START BATCH goes through ONE view for all the chunk's events at once, in all 64 lanes, whatever those lanes' own rays
//     are doing -- the rays' state and the events stay in registers -- and pushes the rays worth marching onto the pool
//     (neighbouring pool entries are parallel rays from neighbouring events: they make the same kind of step at the same time).
//     Round 5: the batch works out no more than what DECIDES whether a ray is marched -- the weight it would carry (what depends on
//     the event alone is worked out once per chunk: w / (4 pi k), 1 + g^2, -2 g, (1 - g^2) k3; the view's direction sits in scalar
//     registers) and the roulette on it, branch-free -- and pushes the survivors as they are: seven of ten pairs die there, and
//     until round 5 the batch computed pixel, roulette budget and record for all 64 lanes before it looked (176 vector
//     instructions per batch, now 60);
//   * a lane whose ray has ended adds its tally (one exp, one atomic) and POPS a started ray: two 16-byte LDS reads, the pixel
//     its line of sight belongs to and its roulette budget (a hash and a logarithm: for the rays that are marched only), the
//     three face distances, the first cell's extinction asked for.
// The walk itself is the loop of k_transport_lean.  A ray that leaves the top of the voxel region towards a sensor above
// the atmosphere finishes inside the walk (one LDS read: `tup`, the optical depth of the uniform layers above); the few rays
// that start or travel inside uniform layers otherwise wait until a handful can be served together.
//
// Reflections off LSRT and DSM surfaces are three events in a hundred and their reflectance models need 100 registers where
// the rest of the kernel needs 70 -- and this kernel waits on memory: its speed is proportional to the waves a SIMD holds
// (profiles/r02/mv9_rays_by_resident_waves.log).  So there are two builds.  The light one (6 waves per SIMD) leaves those
// events alone and notes where they are (DevCold::hv_list, 8 bytes each, once per event); the HEAVY one (4 waves) is launched
// after it over those notes only.
//
// Same estimator, same numbers: which ray carries what is a function of (photon id, index of the photon's Philox block, view)
// as before; only the order of the sums changes.  Serves what k_transport_lean serves.
#include "mi3d_device.h"

namespace mi3d {

#ifndef MI3D_MARK
#endif

#ifndef MI3D_RAYS_THRESH
#define MI3D_RAYS_THRESH 24   // phase A keeps stepping while at least this many lanes of the wave are walking
#endif
#ifndef MI3D_RAYS_WAVES
// (HEAVY: the build for reflections off LSRT and DSM surfaces, 100+ registers; everything else needs 70-78)
#ifndef MI3D_RAYS_LIGHT_WAVES
#define MI3D_RAYS_LIGHT_WAVES 6
#endif
#define MI3D_RAYS_WAVES(COUNT, HEAVY) ((HEAVY) ? ((COUNT) ? 4 : 5) : ((COUNT) ? 6 : MI3D_RAYS_LIGHT_WAVES))
#endif
#ifndef MI3D_EV_NT_LOAD
#define MI3D_EV_NT_LOAD 1    // 1: the event records are read with non-temporal loads (+1 %)
#endif
#ifndef MI3D_RAYS_UNIBATCH
#define MI3D_RAYS_UNIBATCH 8  // rays inside uniform layers wait until this many of a wave can be served together
#endif
constexpr unsigned kEvChunk = 64;   // events a wave takes from a list at a time
#ifndef MI3D_RAYS_POOL
#define MI3D_RAYS_POOL 128
#endif
constexpr unsigned kPool = MI3D_RAYS_POOL;     // started rays a wave can hold in LDS (a start batch adds up to 64: it runs while there is room for one)
static_assert(kPool >= 64 && kPool % 32 == 0, "pool size");
constexpr unsigned kPoolF4 = 2;     // float4 per pool record: (x, y in the voxel, height in the layer, column) (layer | view, weight, budget, pixel)

// LDS of k_rays beyond what k_transport_lean stages (layers, views, DevCold), in bytes: the list of marched views, per view
// (1/|vx|, 1/|vy|, 1/|vz|, height where its rays end), per layer the optical depth up to the top of the atmosphere, the pools
__host__ __device__ inline size_t rays_lds_extra(int nz, bool cam = false) {
    return (size_t)MI3D_MAX_VIEW * sizeof(int) + (size_t)MI3D_MAX_VIEW * 16 + (size_t)((nz + 3) / 4) * 16 + (size_t)4 * kPool * (cam ? 3 : kPoolF4) * 16;
}   // (+ lean_tab_floats(nang, tab_n) floats where the scene refers to tables)

// CAM: the views are cameras (Rad_mrkind = 1, point sensors: CamRec): every ray has a direction of its own -- towards the nearest
// periodic image of the camera --, which travels in a third float4 of its pool record; its value carries 1 / r^2 and the solid angle
// of its pixel of the polar map, both known where the ray starts.  Every surface model in the one build (four waves per SIMD).
// PLAIN: the 1-D constituent is Rayleigh, every 3-D one Henyey-Greenstein (kTargetPlainPhase, checked on the host): no selector is looked at
template <bool COUNT, bool P3D, bool HEAVY, bool CAM = false, bool PLAIN = false>
__global__ void __launch_bounds__(256, CAM ? 4 : MI3D_RAYS_WAVES(COUNT, HEAVY))
k_rays(const DevScene S, const uint64_t seed) {
    constexpr unsigned PF4 = CAM ? 3u : kPoolF4;   // float4 per pool record
    extern __shared__ float4 smem[];
    // (layer table with an end record below the surface and above the top, as in k_transport_lean: no bounds check per level crossing)
    constexpr int kL4 = kLayStride / 4;
    const int o_view = (S.nz + 2) * kL4, o_cold = o_view + MI3D_MAX_VIEW * 2, o_mview = o_cold + kColdF4,
              o_vinv = o_mview + MI3D_MAX_VIEW / 4, o_tup = o_vinv + MI3D_MAX_VIEW, o_pool = o_tup + (S.nz + 3) / 4;
    const float4 *lay4 = smem + kL4;
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(lay4);
    const ViewRec *views = reinterpret_cast<const ViewRec *>(smem + o_view);
    const DevCold *cold = reinterpret_cast<const DevCold *>(smem + o_cold);
    int *mview = reinterpret_cast<int *>(smem + o_mview);   // [nmarch] marched views
    float4 *vinv = smem + o_vinv;                            // [nview]
    float *tup = reinterpret_cast<float *>(smem + o_tup);    // [nz] see the walk
    float4 *pool = smem + o_pool + (threadIdx.x >> 6) * (kPool * PF4);
    // the phase tables the scene refers to, staged behind the pools (the builds that look at selectors; DevCold::tab_n > 0: the launch has
    // given the kernel the LDS)
    const float *ltab = nullptr;
    if (!PLAIN && S.cold->tab_n > 0) {
        float *dst = reinterpret_cast<float *>(smem + o_pool + 4 * (kPool * PF4));
        stage_tables(S.cold, dst);
        ltab = dst;
    }
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.cold->lay);
        for (int i = threadIdx.x; i < S.nz * kL4; i += blockDim.x) smem[kL4 + i] = src[i];
        if (threadIdx.x < 2 * kL4) smem[threadIdx.x < kL4 ? threadIdx.x : (S.nz + 1) * kL4 + (threadIdx.x - kL4)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 *vsrc = reinterpret_cast<const float4 *>(S.cold->views);
        for (int i = threadIdx.x; i < S.nview * 2; i += blockDim.x) smem[o_view + i] = vsrc[i];
        const float4 *csrc = reinterpret_cast<const float4 *>(S.cold);
        if (threadIdx.x < kColdF4) smem[o_cold + threadIdx.x] = csrc[threadIdx.x];
        if (threadIdx.x == 0) {
            int n = 0;
            for (int v = 0; v < S.nview; ++v)
                if (!S.cold->views[v].column) mview[n++] = v;
        }
        __syncthreads();
        if ((int)threadIdx.x < S.nview) {
            const ViewRec V = views[threadIdx.x];
            vinv[threadIdx.x] = make_float4(frcp(fmaxf(fabsf(V.vx), 1e-20f)), frcp(fmaxf(fabsf(V.vy), 1e-20f)), frcp(fmaxf(fabsf(V.vz), 1e-20f)),
                                            (V.vz < 0.0f || V.zs < cold->ztoa) ? V.zs : INFINITY);
        }
        // tup[k]: vertical optical depth from the bottom of layer k to the top of the atmosphere when every layer from k up
        // is horizontally uniform (the figure block LEUNIF below would compute), -1 otherwise
        for (int kk = threadIdx.x; kk < S.nz; kk += blockDim.x) {
            const LayerRec &Lk = lay[kk];
            float v = -1.0f;
            if (!(Lk.flags & kLayStep3d) && Lk.run_hi == S.nz - 1) {
                const LayerRec &Le = lay[Lk.run_hi];
                v = (Le.tauz + Le.bt * Le.dz - Lk.tauz) - Lk.bt * 0.0f;
            }
            tup[kk] = v;
        }
    }
    __syncthreads();
    // does any view end its rays inside the atmosphere?  (wave-uniform: satellites never do, and the walk skips the test)
    bool any_plane = false;
    for (int v = 0; v < S.nview; ++v) any_plane = any_plane || (!views[v].column && vinv[v].w < INFINITY);
    if (CAM) any_plane = true;

    // where the event lists are: read through the kernel argument before the loop, so that they are scalar values (read from the
    // LDS copy of the cold block they are vector values, and the 64-bit address of every record is worked out lane by lane)
    const float4 *const ev_list_s = S.cold->ev_list;
    const unsigned ev_cap_s = (unsigned)S.cold->ev_cap;
    const size_t ev_lf4 = ev_list_f4(ev_cap_s);
    const bool ipa = (S.solver == MI3D_SOLVER_IPA) || P3D;   // everything scattered stays in its column under both
    const bool plain = PLAIN || (CAM && (S.target & kTargetPlainPhase) != 0);
    const LeanTab T = PLAIN ? LeanTab{} : lean_tab(cold, ltab);
    Counters cnt = {};
    const float *bbase = S.bext3 - S.k3lo;   // (the walk reads the extinction alone: 4 bytes per cell)
    const unsigned nm = (unsigned)S.nmarch;
    const unsigned lane = threadIdx.x & 63u;
    // the lists this build works from: the XCDs' event lists, or (HEAVY) the lists of events the light build left to this one
    constexpr unsigned c_fill = HEAVY ? kCtrHeavyFill : 0u, c_cur = HEAVY ? kCtrHeavyCursor : kCtrCursor;

    // ---- lane state: one ray
    float uz = 1, iux = 1, iuy = 1, iuz = 1;   // (the view's vx, vy are read from LDS where a walk is set up)
    float dux = 0, duy = 0;                    // CAM: the ray's own vx, vy
    float t = 0, tx = 0, ty = 0, tz = 0;       // ray parameter now / at the next x, y, z face: where the ray is inside its voxel follows from them
    int ix = 0, iy = 0, k = 0, stepx = 0, stepy = 0, stepk = 1, wrapx = 0, wrapy = 0;
    float rem = 0.0f, tkill = kTauCut, contrib = 0.0f, zstop = INFINITY;
    float roz = 0, rpz = 0, bext = 0;
    int iv = 0, pix = 0, mode = M_NEED;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    // wave-uniform: events [ev_next, ev_end) of list `list` are this wave's, `sub` rays of them have been started;
    // pool_n started rays wait in the pool; exhausted: every list has been handed out
    unsigned victim = 0, list = xcc;
    unsigned long long ev_next = 0, ev_end = 0;
    unsigned vsub = nm, pool_n = 0;   // vsub: marched views of this wave's chunk of events that are started (nm: all of them: next chunk)
    // CAM: the periodic images of a camera an event contributes to, (2 N + 1)^2 of them (N = DevCold::cam_images; index 0: the nearest
    // one, then the others row by row: the oracle's enumeration, cam_image); a start batch serves ONE image of one camera
    const unsigned cam_n = CAM ? cold->cam_images : 0u, nimg = CAM ? (2u * cam_n + 1u) * (2u * cam_n + 1u) : 1u;
    unsigned isub = 0;
    bool exhausted = false;
    // ---- lane state: the event of this wave's chunk this lane stands for (one event per lane, taken apart once, all its views started from it)
    float4 E0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), E1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // position in the voxel, weight (0: none); direction, first parameter
    float eapf = 0, esfc = 0, ezz = 0, eks1 = 0, eks3 = 0, eksb = 0, eapfb = 0;
    int ecell = 0, ekk = 0;
    uint32_t ehb = 0;
    // what a start batch needs of the event and no view changes (satellite views): the weight the ray would carry but for the phase
    // function, w / (4 pi k) (a Lambertian reflection: w A / pi; 0: no event in this lane); PLAIN: 0.75 k1, 1 + g^2, -2 g, (1 - g^2) k3
    float ewk = 0, era = 0, ega = 1, egb = 0, egc = 0;

    long long rtick = COUNT ? clock64() : 0;
    for (;;) {
        // =================================== phase A: voxel steps ===================================
        MI3D_MARK("RA");
        // (instrumented build: wave clock ticks / 64 in cyc[2] the walk, cyc[3] uniform layers and tallies, cyc[4] chunks and start batches, cyc[5] the pop)
#define MI3D_RTICK(slot) do { if (COUNT) { const long long t_ = clock64(); cnt.cyc[slot] += (uint32_t)((t_ - rtick) >> 6); rtick = t_; } } while (0)
        int nfly = 0;
        for (;;) {
            const bool flying = (mode == M_LE);
            nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (nfly < MI3D_RAYS_THRESH &&
                (__ballot(mode == M_LEEND || mode == M_NEED) != 0ull || __popcll(__ballot(mode == M_LEUNIF)) >= MI3D_RAYS_UNIBATCH)) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (flying) {
                // (bext: the extinction of the ray's cell, asked for when the ray entered it -- at the end of the step before, at the
                //  pop, at the re-entry from uniform layers -- so that the read travels while the wave's other work goes on)
                const float4 r4 = make_float4(bext, 0.0f, 0.0f, 0.0f);
                const float tn = fminf(fminf(tx, ty), tz);
                float dtau = r4.x * (tn - t);
                if (COUNT) { cnt.le_steps++; cnt.le_steps3d++; }
                bool plane = false;
                if (any_plane && zstop < INFINITY) {
                    const float zn = fmaf(uz, tn, roz);
                    plane = uz > 0.0f ? zn >= zstop : zn <= zstop;
                    if (plane) dtau = r4.x * fabsf(zstop - fmaf(uz, t, roz)) * iuz;
                }
                if (dtau >= rem) { rem = -1.0f; mode = M_LEEND; }   // the ray's budget is used up: given up
                else if (any_plane && plane) { rem -= dtau; mode = M_LEEND; }
                else {
                    rem -= dtau;
                    t = tn;
                    if (tz == tn) {
                        k += stepk;                                  // (-1 and nz: the table's end records)
                        const float4 Ln = lay4[k * kL4];
                        tz = fmaf(Ln.x, iuz, tz);
                        if (!(__float_as_int(Ln.w) & kLayStep3d)) {
                            const bool up = uz > 0.0f;
                            if (k < 0 || k >= S.nz) mode = M_LEEND;  // out of the atmosphere: arrived (or, downwards, ended by rounding)
                            else {
                                const float tu = tup[k];
                                if (up && tu >= 0.0f && !(zstop < INFINITY)) {   // nothing but uniform layers between here and the sensor
                                    const float tpath = tu * iuz;
                                    if (COUNT) cnt.le_steps++;
                                    rem = tpath >= rem ? -1.0f : rem - tpath;
                                    mode = M_LEEND;
                                } else { mode = M_LEUNIF; rpz = up ? 0.0f : Ln.x; }
                            }
                        }
                    } else if (tx == tn) {
                        tx = fmaf(S.dx, iux, tx);
                        const int c = ix + stepx;
                        ix = (unsigned)c >= (unsigned)S.nx ? wrapx : c;
                    } else {
                        ty = fmaf(S.dy, iuy, ty);
                        const int c = iy + stepy;
                        iy = (unsigned)c >= (unsigned)S.ny ? wrapy : c;
                    }
                    if (mode == M_LE) bext = bbase[((unsigned)iy * (unsigned)S.nx + (unsigned)ix) * (unsigned)S.nz3 + (unsigned)k];
                }
            }
        }

        // =================================== phase B ===================================
        MI3D_RTICK(2);
        MI3D_MARK("RB0");
        if (COUNT) { cnt.b_slots++; if (mode != M_LE && mode != M_DONE) cnt.b_lanes++; }
        // ---- rays inside runs of uniform layers (as k_transport_lean, block B0'); rare: served a handful at a time
        const int n_uni = __popcll(__ballot(mode == M_LEUNIF));
        if (n_uni != 0 && (n_uni >= MI3D_RAYS_UNIBATCH || nfly == 0 || exhausted)) {
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
                    const float4 L = lay4[k * kL4];
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
                    rpz = up ? 0.0f : lay4[k * kL4].x;
                    if (rem < 0.0f) { mode = M_LEEND; break; }
                }
            }
            if (reenter) {
                // where the ray is, in the frame of the column it left the voxels in (or started in): (tx - t) |ux| short of the x face
                // that was ahead of it there -- beyond it by now, fold_xy brings it home; then the walk's faces from the new origin
                const float ux = CAM ? dux : views[iv].vx, uy = CAM ? duy : views[iv].vy;
                // (|u| floored as where the parameters were set up: an exactly vertical ray keeps its place)
                const float dxo = (tx - t) * fmaxf(fabsf(ux), 1e-20f), dyo = (ty - t) * fmaxf(fabsf(uy), 1e-20f);
                float xo = ux > 0.0f ? S.dx - dxo : dxo;
                float yo = uy > 0.0f ? S.dy - dyo : dyo;
                fold_xy(S, cold, xo, yo, ix, iy, ipa);
                const float4 L = lay4[k * kL4];
                roz = L.z + rpz;
                t = 0.0f;
                tx = (ux > 0.0f ? S.dx - xo : xo) * iux;
                ty = (uy > 0.0f ? S.dy - yo : yo) * iuy;
                tz = (uz > 0.0f ? L.x - rpz : rpz) * iuz;
                mode = M_LE;
                bext = bbase[((unsigned)iy * (unsigned)S.nx + (unsigned)ix) * (unsigned)S.nz3 + (unsigned)k];
            }
          }
        }

        MI3D_MARK("RB1");
        // ---- a ray has arrived (or has been given up): its tally; the lane is free
        if (mode == M_LEEND) {
            if (rem >= 0.0f) {
                float acc = tkill - rem;
                if (views[iv].roulette & 1) acc = fminf(acc, cold->le_tau1);
                RAD_ADD(&S.rad[(unsigned)pix * (unsigned)S.rad_stride], contrib * fexp_neg(acc));
            }
            mode = M_NEED;
        }

        MI3D_RTICK(3);
        // ---- free lanes want started rays: start batches until the pool holds enough, then pop
        const unsigned long long need = __ballot(mode == M_NEED);
        const unsigned nn = (unsigned)__popcll(need);
        if (nn != 0u) {
            while (pool_n < nn && pool_n + 64u <= kPool && !exhausted) {
                if (vsub >= nm) {
                    // every view of this wave's events is started: the next chunk of its XCD's list, then of the others' -- one event per
                    // lane, read and taken apart ONCE; the views are then gone through one at a time, all events at once
                    bool got = false;
                    while (victim < 8u) {
                        list = (xcc + victim) & 7u;
                        const unsigned long long have = cold->ev_ctr[(c_fill + list) * kCtrStride] < (unsigned long long)cold->ev_cap
                                                            ? cold->ev_ctr[(c_fill + list) * kCtrStride] : (unsigned long long)cold->ev_cap;
                        unsigned long long b = 0;
                        if (lane == 0u) b = atomicAdd(cold->ev_ctr + (c_cur + list) * kCtrStride, (unsigned long long)kEvChunk);
                        b = __shfl(b, 0, 64);
                        if (b < have) { ev_next = b; ev_end = b + kEvChunk < have ? b + kEvChunk : have; got = true; break; }
                        victim++;
                    }
                    if (!got) { exhausted = true; break; }
                    vsub = 0; isub = 0;
                    MI3D_MARK("RCHUNK");
                    E0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    bool defer = false;
                    unsigned long long where = 0;   // the record: list << 32 | slot
                    if (lane < (unsigned)(ev_end - ev_next)) {
                        // (the light build reads its wave's own chunk of ONE list: a wave-uniform base and 32-bit offsets)
                        where = HEAVY ? cold->hv_list[(size_t)list * cold->ev_cap + (ev_next + lane)] : (((unsigned long long)list << 32) | (ev_next + lane));
                        const float4 *lbase = ev_list_s + ev_lf4 * (HEAVY ? (size_t)(where >> 32) : (size_t)list);
                        const unsigned slot = HEAVY ? (unsigned)where : (unsigned)ev_next + lane;
                        const float4 *e = lbase + ev_index(slot);
#if MI3D_EV_NT_LOAD
                        // (read once: non-temporal loads leave the voxels' extinction in the caches)
                        typedef float vf4 __attribute__((ext_vector_type(4)));
                        const vf4 *en = reinterpret_cast<const vf4 *>(e);
                        const vf4 n0 = __builtin_nontemporal_load(en), n1 = __builtin_nontemporal_load(en + kEvStride), n2 = __builtin_nontemporal_load(en + 2 * kEvStride);
                        E0 = make_float4(n0.x, n0.y, n0.z, n0.w); E1 = make_float4(n1.x, n1.y, n1.z, n1.w);
                        const float4 e2 = make_float4(n2.x, n2.y, n2.z, n2.w);
                        ehb = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(lbase) + ev_word(slot));
#else
                        E0 = e[0]; E1 = e[kEvStride];
                        const float4 e2 = e[2 * kEvStride];
                        ehb = reinterpret_cast<const uint32_t *>(lbase)[ev_word(slot)];   // le_hash_base of the event
#endif
                        if (!(E0.w > 0.0f)) E0.w = 0.0f;       // (a record a wave of the photon loop reserved and did not use has weight 0)
                        eapf = e2.x; esfc = e2.y; ecell = __float_as_int(e2.z); ekk = E0.w > 0.0f ? __float_as_int(e2.w) : 0;
                        const LayerRec &Lk = lay[ekk & 0xffff];
                        ezz = Lk.zlo + E0.z;
                        eks1 = Lk.ks1d[0];
                        if (!PLAIN) for (int ip = 1; ip < S.np1d; ++ip) eks1 += Lk.ks1d[ip];     // (every 1-D constituent: the mixture's total)
                        eks3 = (Lk.flags & kLayIn3d) ? E1.w : 0.0f;
                        eksb = 0.0f; eapfb = 0.0f;
                        if (S.np3d > 1 && (Lk.flags & kLayIn3d) && ((ekk >> 16) & 15) != E_SURFACE) {   // the voxel's second 3-D constituent
                            const float2 cs = cold->csca[((unsigned)((ecell >> 16) * S.nx + (ecell & 0xffff)) * (unsigned)S.nz3 + (unsigned)((ekk & 0xffff) - S.k3lo)) * 2u + 1u];
                            eksb = cs.x; eapfb = cs.y;
                        }
                        const int kind = ekk >> 16;
                        defer = !HEAVY && !CAM && E0.w > 0.0f && (kind & 15) == E_SURFACE && ((kind >> 4) == MI3D_SFC_LSRT || (kind >> 4) == MI3D_SFC_DSM);
                        if (defer) E0.w = 0.0f;                 // the heavy build's: noted once, no ray from it here
                    }
                    if (!CAM) {
                        const bool surf = ((ekk >> 16) & 15) == E_SURFACE;
                        ewk = E0.w > 0.0f ? (surf ? E0.w * fminf(fmaxf(E1.w, 0.0f), 1.0f) * (1.0f / kPi)       // Lambertian: the albedo (surface_R)
                                                  : E0.w * frcp((eks1 + eks3) + eksb) * (0.25f / kPi)) : 0.0f;
                        if (PLAIN) { era = 0.75f * eks1; ega = fmaf(eapf, eapf, 1.0f); egb = -2.0f * eapf; egc = (1.0f - eapf * eapf) * eks3; }
                    }
                    if (!HEAVY) {
                        // reflections off LSRT / DSM surfaces: where they are goes onto this XCD's list for the heavy build
                        const unsigned long long dm = __ballot(defer);
                        if (dm != 0ull) {
                            const int leader = __ffsll((long long)dm) - 1;
                            unsigned long long base = 0;
                            if ((int)lane == leader) base = atomicAdd(cold->ev_ctr + (kCtrHeavyFill + xcc) * kCtrStride, (unsigned long long)__popcll(dm));
                            base = __shfl(base, leader, 64);
                            if (defer) {
                                const unsigned long long slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u));
                                if (cold->hv_list && slot < (unsigned long long)cold->ev_cap) cold->hv_list[(size_t)xcc * cold->ev_cap + slot] = where;
                                else cold->ev_ctr[8 * kCtrStride] = 1ull;   // (reported by mi3d_run like a full event list)
                            }
                        }
                    }
                }
                MI3D_MARK("RBATCH");
                // a start batch: ONE view for all the events of the chunk, one event per lane
                const int jv = mview[vsub];
                const ViewRec V = views[jv];
                const int kind = ekk >> 16;
                bool push = false;
                float4 q0 = make_float4(0, 0, 0, 0), q1 = make_float4(0, 0, 0, 0), q2 = make_float4(0, 0, 0, 0);
                if (COUNT) { cnt.cyc[0]++; if (E0.w > 0.0f) cnt.cyc[1]++; }
                if (CAM) {
                  if (E0.w > 0.0f) {
                    // the ray goes to the nearest periodic image of the camera, a distance r away (k_transport's B3 and B1 in one place)
                    const CamRec Cm = cold->cams[jv];
                    float rx = Cm.cx - ((float)(ecell & 0xffff) * S.dx + E0.x), ry = Cm.cy - ((float)(ecell >> 16) * S.dy + E0.y);
                    const float rz = Cm.cz - ezz;
                    rx -= cold->Lx * floorf(rx * cold->inv_Lx + 0.5f); ry -= cold->Ly * floorf(ry * cold->inv_Ly + 0.5f);
                    float fimg = 1.0f;
                    if (isub != 0u) {      // a periodic image beyond the nearest one (wave-uniform)
                        // Russian roulette on it: served with probability (r0 / r)^2, r0 the distance of the nearest image, and then carrying
                        // (r / r0)^2 times its contribution (unbiased; the rays per event grow with the logarithm of the number of images);
                        // one hashed number per (event, view, image), as in the oracle (cam_image_roulette)
                        const float r0sq = rx * rx + ry * ry + rz * rz;
                        const unsigned n1 = 2u * cam_n + 1u;
                        unsigned tt = isub - 1u;
                        if (tt >= (n1 * n1 - 1u) / 2u) tt += 1u;
                        rx += (float)((int)(tt % n1) - (int)cam_n) * cold->Lx; ry += (float)((int)(tt / n1) - (int)cam_n) * cold->Ly;
                        const float rsq = rx * rx + ry * ry + rz * rz;
                        fimg = le_roulette_from_base(ehb, jv + 64 * (int)isub + 16) * rsq < r0sq ? rsq * frcp(fmaxf(r0sq, 1e-30f)) : 0.0f;
                    }
                    const float r2 = rx * rx + ry * ry + rz * rz, irr = frsq(fmaxf(r2, 1e-30f));
                    const float vx = rx * irr, vy = ry * irr, vz = rz * irr;
                    const float inv_r2 = frcp(fmaxf(r2, Cm.r2min));
                    // outside the cone of view, a line of sight within 0.06 degrees of the horizontal, the surface seen from below: nothing to carry
                    const bool visible = fimg > 0.0f && r2 > 0.0f && fabsf(vz) >= 1e-3f && -(vx * Cm.zx + vy * Cm.zy + vz * Cm.zz) >= Cm.cos_half &&
                                         !((kind & 15) == E_SURFACE && vz <= 0.0f);
                    if (visible) {
                        float c;
                        if ((kind & 15) == E_SURFACE) {
                            if ((kind >> 4) == MI3D_SFC_LAMBERT) c = E0.w * fminf(fmaxf(E1.w, 0.0f), 1.0f) * vz * (1.0f / kPi);
                            else {   // (LSRT, DSM: this build runs at four waves per SIMD anyway and has the registers)
                                const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ecell & 0xffff, ecell >> 16, E0.x, E0.y) : Sfc{kind >> 4, E1.w, eapf, esfc, 0.0f, 0.0f};
                                c = E0.w * surface_R(sf, E1.x, E1.y, E1.z, vx, vy, vz) * vz * (1.0f / kPi);
                            }
                        }
                        else {
                            const float mu = E1.x * vx + E1.y * vy + E1.z * vz;
                            float P = 0.0f;
                            if (plain) {
                                P = eks1 * (0.75f * fmaf(mu, mu, 1.0f)) + eks3 * phase_eval_hg(eapf, mu);
                                if (S.np3d > 1 && eksb > 0.0f) P += eksb * phase_eval_analytic(eapfb, mu);
                            } else P = lean_mix_phase(T, lay[ekk & 0xffff], S.np1d, eks3, eapf, eksb, eapfb, mu);
                            c = E0.w * P * frcp((eks1 + eks3) + eksb) * (0.25f / kPi);
                        }
                        if (COUNT) cnt.le_rays++;
                        if (V.roulette & 2) c = le_weight_roulette_base(c, cold->le_cmin, ehb, jv);
                        c *= fimg;
                        if (c > 0.0f) {
                            // the pixel: where the direction the camera looks in to see the event falls in the polar map (k_transport's B1)
                            const float dxc = -(vx * Cm.xx + vy * Cm.xy + vz * Cm.xz), dyc = -(vx * Cm.yx + vy * Cm.yy + vz * Cm.yz);
                            const float dzc = fminf(-(vx * Cm.zx + vy * Cm.zy + vz * Cm.zz), 1.0f);
                            const float theta = acosf(dzc), rho2 = dxc * dxc + dyc * dyc;
                            const float sc = rho2 > 1e-24f ? theta * frsq(rho2) : 0.0f;
                            const int ir = (int)floorf(dxc * sc * Cm.inv_du + 0.5f * (float)S.nxr), jr = (int)floorf(dyc * sc * Cm.inv_dv + 0.5f * (float)S.nyr);
                            if (ir >= 0 && ir < S.nxr && jr >= 0 && jr < S.nyr) {
                                const float sinc = theta > 1e-6f ? sinf(theta) / theta : 1.0f;
                                const float tk = (V.roulette & 1) ? cold->le_tau1 - 0.69314718f * __builtin_amdgcn_logf(le_roulette_from_base(ehb, jv + 64 * (int)isub)) : kTauCut;
                                q0 = make_float4(E0.x, E0.y, E0.z, __int_as_float(ecell));
                                q1 = make_float4(__int_as_float((ekk & 0xffff) | (jv << 16)), c * inv_r2 * Cm.inv_du * Cm.inv_dv / sinc, tk, __int_as_float((jv * S.nyr + jr) * S.rad_row + ir));
                                q2 = make_float4(vx, vy, vz, Cm.cz);
                                push = true;
                            }
                        }
                    }
                  }
                } else {
                    // ---- satellite views: what decides whether the ray is marched, and nothing else.  The view's direction, height and
                    // roulette flags in scalar registers (the same in every lane: one LDS read, broadcast)
                    const float vvx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.vx)));
                    const float vvy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.vy)));
                    const float vvz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.vz)));
                    const float vzs = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.zs)));
                    const int vroul = __builtin_amdgcn_readfirstlane(V.roulette);
                    const bool surf = (kind & 15) == E_SURFACE;
                    // the sensor on the wrong side of the event, an up-looking one for a surface event: no ray
                    const bool ok = (HEAVY ? E0.w > 0.0f : ewk > 0.0f) && (vvz > 0.0f ? ezz < vzs : (ezz > vzs && !surf));
                    float c;
                    if (HEAVY) {
                        const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ecell & 0xffff, ecell >> 16, E0.x, E0.y) : Sfc{kind >> 4, E1.w, eapf, esfc, 0.0f, 0.0f};
                        c = E0.w * surface_R(sf, E1.x, E1.y, E1.z, vvx, vvy, vvz) * vvz * (1.0f / kPi);
                    } else {
                        const float mu = fmaf(E1.z, vvz, fmaf(E1.y, vvy, E1.x * vvx));
                        float P;
                        if (PLAIN) {      // Rayleigh + Henyey-Greenstein: no selector looked at
                            const float r = frsq(fmaf(mu, egb, ega));
                            P = fmaf(era, fmaf(mu, mu, 1.0f), egc * r * r * r);
                            if (S.np3d > 1 && eksb > 0.0f) P += eksb * phase_eval_analytic(eapfb, mu);
                        } else P = lean_mix_phase(T, lay[ekk & 0xffff], S.np1d, eks3, eapf, eksb, eapfb, mu);   // (any mixture: selectors, tables)
                        c = surf ? ewk * vvz : ewk * P;     // (a Lambertian reflection: w A cos / pi)
                    }
                    if (COUNT && ok) cnt.le_rays++;
                    if (vroul & 2) {
                        // roulette on the weight the ray would carry (le_weight_roulette_base, branch-free: every lane works its hash out)
                        const float cmin = cold->le_cmin;
                        const float cr = le_roulette_from_base(ehb, jv + 16) * cmin < c ? cmin : 0.0f;
                        c = c < cmin ? cr : c;
                    }
                    if (ok && c > 0.0f) {
                        // the ray as it stands: pixel, budget and 1 / |cos| are worked out by the lane that marches it (the pop below)
                        q0 = make_float4(E0.x, E0.y, E0.z, __int_as_float(ecell));
                        q1 = make_float4(__int_as_float((ekk & 0xffff) | (jv << 16)), c, __uint_as_float(ehb), ezz);
                        push = true;
                    }
                }
                if (CAM) { isub += 1u; if (isub >= nimg) { isub = 0u; vsub += 1u; } }
                else vsub += 1u;
                const unsigned long long pm = __ballot(push);
                if (push) {
                    const unsigned slot = pool_n + __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
                    pool[slot * PF4] = q0;
                    pool[slot * PF4 + 1] = q1;
                    if (CAM) pool[slot * PF4 + 2] = q2;
                }
                pool_n += (unsigned)__popcll(pm);
            }
            MI3D_RTICK(4);
            MI3D_MARK("RPOP");
            // (the pool is the wave's own and a wave's LDS operations complete in order; the fence keeps the compiler from
            //  moving the reads of one lane above the writes of another)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            if (mode == M_NEED) {
                if (rank < pool_n) {
                    const unsigned slot = pool_n - 1u - rank;
                    const float4 q0 = pool[slot * PF4], q1 = pool[slot * PF4 + 1];
                    const int cell = __float_as_int(q0.w), kk = __float_as_int(q1.x);
                    rpz = q0.z;
                    ix = cell & 0xffff; iy = cell >> 16;
                    k = kk & 0xffff; iv = kk >> 16;
                    float4 vi, vd;
                    if (CAM) {
                        contrib = q1.y; tkill = q1.z; rem = q1.z; pix = __float_as_int(q1.w);
                        vd = pool[slot * PF4 + 2];                                         // the ray's own direction, the camera's height
                        vi = make_float4(frcp(fmaxf(fabsf(vd.x), 1e-20f)), frcp(fmaxf(fabsf(vd.y), 1e-20f)), frcp(fmaxf(fabsf(vd.z), 1e-20f)),
                                         (vd.z < 0.0f || vd.w < cold->ztoa) ? vd.w : INFINITY);
                        dux = vd.x; duy = vd.y;
                    } else {
                        vi = vinv[iv];
                        vd = reinterpret_cast<const float4 *>(views)[iv * 2];              // (vx, vy, vz, zs)
                        const float4 ve = reinterpret_cast<const float4 *>(views)[iv * 2 + 1];   // (column, roulette, zreg, point)
                        // the pixel the ray's line of sight belongs to, the roulette budget: all a function of the event and the view
                        const uint32_t hb = __float_as_uint(q1.z);
                        float xr = (float)ix * S.dx + q0.x, yr = (float)iy * S.dy + q0.y;
                        if (!ipa) {
                            const float tt = (q1.w - ve.z) * frcp(vd.z);
                            xr -= vd.x * tt; yr -= vd.y * tt;
                            xr -= floorf(xr * cold->inv_Lx) * cold->Lx; yr -= floorf(yr * cold->inv_Ly) * cold->Ly;
                        }
                        const int ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                        const int jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                        pix = (iv * S.nyr + jr) * S.rad_row + ir;
                        tkill = (__float_as_int(ve.y) & 1) ? cold->le_tau1 - 0.69314718f * __builtin_amdgcn_logf(le_roulette_from_base(hb, iv)) : kTauCut;
                        rem = tkill;
                        contrib = q1.y * vi.z;      // (1 / |cos| of the view)
                    }
                    uz = vd.z;
                    iux = vi.x; iuy = vi.y; iuz = vi.z; zstop = vi.w;
                    const float4 L = lay4[k * kL4];
                    roz = L.z + rpz;
                    t = 0.0f;
                    // the parameters of the first three faces (x and y also for a ray that starts inside uniform layers: its
                    // horizontal position is read off them when it enters layers that are walked)
                    tx = (vd.x > 0.0f ? S.dx - q0.x : q0.x) * iux;
                    ty = (vd.y > 0.0f ? S.dy - q0.y : q0.y) * iuy;
                    tz = (uz > 0.0f ? L.x - rpz : rpz) * iuz;
                    stepx = ipa ? 0 : (vd.x > 0.0f ? 1 : -1);
                    stepy = ipa ? 0 : (vd.y > 0.0f ? 1 : -1);
                    wrapx = vd.x > 0.0f ? 0 : S.nx - 1; wrapy = vd.y > 0.0f ? 0 : S.ny - 1; stepk = uz > 0.0f ? 1 : -1;
                    mode = (__float_as_int(L.w) & kLayStep3d) ? M_LE : M_LEUNIF;
                    if (mode == M_LE) {
                        bext = bbase[((unsigned)iy * (unsigned)S.nx + (unsigned)ix) * (unsigned)S.nz3 + (unsigned)k];
                    }
                } else if (exhausted) mode = M_DONE;
            }
            pool_n -= nn < pool_n ? nn : pool_n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        MI3D_RTICK(5);
        MI3D_MARK("REND");
        if (__ballot(mode != M_DONE) == 0ull) break;
    }

    if (COUNT) {
        // (cyc[0], cyc[1]: start batches and the lanes that had a pair in them)
        uint32_t vals[24] = {0, 0, 0, 0, 0, cnt.le_rays, cnt.le_steps, cnt.le_steps3d, 0, 0, 0, 0, 0, 0,
                             cnt.a_lanes, cnt.a_slots, cnt.b_lanes, cnt.b_slots, cnt.cyc[0], cnt.cyc[1], cnt.cyc[2], cnt.cyc[3], cnt.cyc[4], cnt.cyc[5]};
        for (int q = 0; q < 24; ++q) {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&S.cold->counters[q], v);
        }
    }
}

template __global__ void k_rays<false, false, false, true>(const DevScene, const uint64_t);
template __global__ void k_rays<true, false, false, true>(const DevScene, const uint64_t);
#define MI3D_RAYS_INST(C, P) template __global__ void k_rays<C, P, false, false, false>(const DevScene, const uint64_t); \
                             template __global__ void k_rays<C, P, false, false, true>(const DevScene, const uint64_t);  \
                             template __global__ void k_rays<C, P, true, false, false>(const DevScene, const uint64_t);
MI3D_RAYS_INST(false, false) MI3D_RAYS_INST(false, true) MI3D_RAYS_INST(true, false) MI3D_RAYS_INST(true, true)
#undef MI3D_RAYS_INST

} // namespace mi3d
