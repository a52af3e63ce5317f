// mi3d_kernel_flux.hip — the lean photon loop for er3t's flux and heating-rate jobs (er3t/rtm/mca/mcarats.py:296-304: target='flux' /
// 'heating rate' -> Flx_mflx = 3, Flx_mhrt, no radiance views), and the kernels that add up its tally records.
//
// The photon loop is k_transport_lean's (mi3d_kernel_lean.hip: incremental voxel walk on face parameters, end records in the layer
// table, one shared finish and Philox block) without anything that serves radiance, plus one tally per level crossed: plane
// 0 direct-down, 1 diffuse-down, 2 up of DevScene::flux, exactly the crossings k_transport<.,.,true,.> tallies (same photon id ->
// same history; a crossing inside a run of uniform layers is placed by one multiply-add from where the run was entered instead of
// layer by layer).  One 1-D and at most two 3-D constituents, analytic phase functions; everything else runs through k_transport.
//
// Where the tallies go.  A float64 atomic per crossing is what k_transport does, and on this chip that IS its speed: atomics
// execute at the memory side at 2.35e10 per second chip-wide wherever they point (profiles/r03/atomic_rates.log), 44 crossings
// per photon on the 128 x 128 x 69 flux scene -> 5.3e8 photons/s at best.  Here a crossing (and, for heating rates, what a collision
// absorbs: its cells follow the flux cells in one index space) is an 8-byte RECORD {tally index, weight}:
//   * a wave stages its records in LDS and writes them 192-256 at a time, coalesced, into the chunk of 1024 records it owns in the
//     list (one atomic per chunk); it keeps the list of its chunks and a histogram of its records over the bins (a bin = 16 384
//     consecutive tally cells; beyond 1024 bins the four waves of a workgroup share one histogram: TallyList::hist_wg);
//   * after the launch k_tl_wavescan and k_tl_prefix turn the histograms into where every wave's (workgroup's) share of every bin
//     goes, k_tl_scatter sorts each wave's chunks tile by tile inside LDS and copies the sorted tiles there -- no atomic outside LDS --,
//     and k_tl_sum adds every bin up in LDS (float64, ds_add_f64) and adds the sums to the tally.
// HBM sees 4 x 8 bytes per crossing, streamed, instead of an atomic: 9e8 photons/s on that scene.  A list that runs full costs
// nothing but speed: the wave's tallies turn into atomics from there on.  DESIGN.md section 5 has the measurements behind every choice.
#include "mi3d_device.h"

namespace mi3d {

#ifndef MI3D_FLUX_WAVES
#define MI3D_FLUX_WAVES(COUNT) ((COUNT) ? 4 : 5)   // waves per SIMD the register budget must allow.  Round 6, with run records (block B0 no longer places levels:
                                  // 95-97 registers): 4 / 5 / 6 give 1.35 / 1.43 / 1.00e9 photons/s on the 128 x 128 flux scene (profiles/r06/ab_flux_waves.log); round 3, when B0
                                  // placed every level: 8.96 / 8.80 / 6.50e8 (five waves spilled, six spilled into the walk -- profiles/r03/ab_flux_tuning.log)
#endif
#ifndef MI3D_FLUX_PASS
#define MI3D_FLUX_PASS 2      // every second pass of phase B is a full one (see k_transport); 1 / 2 / 3: 8.56 / 8.80 / 8.81e8
#endif
#ifndef MI3D_FLUX_FAST_PASS
#define MI3D_FLUX_FAST_PASS 8 // every n-th pass of phase B is a full one (block C serves the collisions in between: mi3d_kernel_lean.hip); with run records
                              // 4 / 6 / 8 / 12: 1.40 / 1.42 / 1.44 / 1.43e9 on les128_flux, 7.7 / 8.0 / 8.0 / 8.1e8 on les480_flux (profiles/r06/ab_flux_cadence.log)
#endif
#ifndef MI3D_FLUX_THRESH
#define MI3D_FLUX_THRESH 16   // phase A keeps stepping while at least this many lanes walk; 4 / 8 / 12 / 16 / 24 / 32: 8.1 / 8.9 / 9.1 / 9.1 / 8.6 / 8.2e8
#endif

constexpr unsigned kTlChunk = 1024;   // records a wave of the photon loop reserves at a time
constexpr unsigned kTlNone = 0xffffffffu;
constexpr unsigned kTlIds = 512;       // chunk numbers a workgroup of k_tl_scatter keeps in LDS (its waves' lists; longer lists are read on from memory)
#ifndef MI3D_TL_STAGE
#define MI3D_TL_STAGE 256
#endif
constexpr unsigned kTlStage = MI3D_TL_STAGE;     // records a wave stages in LDS before they leave for its chunk
constexpr unsigned kRunChunk = 1024;   // run records (32 bytes each) a wave of the photon loop reserves at a time
constexpr int kRunMin = 3;             // a flight through uniform layers that crosses at least this many tallied levels leaves ONE run record (fewer: a record per level)

struct TallyList {
    uint2 *rec;                  // [cap] {tally index, weight bits} as the photon loop writes them, a chunk of kTlChunk per wave at a time
    uint2 *binned;               // [cap] the same records bin by bin (k_tl_scatter)
    uint32_t *chunk_fill;        // [cap / kTlChunk] records in use of each chunk
    unsigned long long *cursor;  // [0] records reserved so far (a multiple of kTlChunk; beyond cap: the list ran full)
    uint32_t *wave_chunks;       // [nwave][wcap] the chunks each wave of the photon loop has filled, in order
    uint32_t *wave_nchunk;       // [nwave] how many
    uint32_t *whist;             // [nwave][nbins] that wave's records per bin
    uint32_t *wbase;             // [nwave][nbins] records of the same bin from the waves before it (k_tl_wavescan)
    uint32_t *hist;              // [nbins] records per bin (k_tl_wavescan)
    uint32_t *bin_start;         // [nbins + 1] exclusive prefix sums of hist (k_tl_prefix)
    unsigned cap;                // records (a multiple of kTlChunk); 0: no lists, every tally is an atomic
    int shift, nbins;            // tally index i lies in bin i >> shift
    int nwave, wcap;             // waves of the photon loop's grid; chunks a wave may fill before its tallies turn into atomics
    int hist_wg;                 // 1: whist / wbase hold one row per WORKGROUP of the photon loop (its four waves count into one histogram: tallies of more
                                 // than 1024 bins, where four histograms per workgroup would not leave room in LDS), nrow = nwave / 4 rows; 0: one per wave
    // Run records (round 6): a flight through a run of horizontally uniform layers crosses its levels at places that follow from where it
    // entered by one multiply-add per level -- 60 % of all crossings on the bench scenes.  The photon loop writes ONE 32-byte record per
    // such flight (tl_run_* below); k_tl_runs expands them, level by level, straight into the bins of `binned` (a count pass first: the
    // rows nrow .. 2 nrow - 1 of whist / wbase are the runs' rows, one per wave or workgroup of the photon loop as for the records).
    int run_wcap;                // chunks of run records a wave may fill before its runs go out level by level again
    float4 *runs;                // [run_cap / kRunChunk][2][kRunChunk]: per chunk kRunChunk first parts, then kRunChunk second parts
    uint32_t *run_fill;          // [run_cap / kRunChunk] run records in use of each chunk
    uint32_t *run_chunks;        // [nwave][run_wcap] the chunks each wave has filled
    uint32_t *run_nchunk;        // [nwave] how many
    unsigned run_cap;            // run records (a multiple of kRunChunk); 0: no run records
    unsigned bcap;               // records `binned` holds: a record whose place lies beyond goes to the tally as an atomic (k_tl_scatter, k_tl_runs)
    unsigned long long *stats;   // [3] this launch's copy of the three cursors below (k_tl_prefix): what the host reads -- the cursors themselves are zeroed for the set's next launch
    // hist_wg: the workgroup's LDS histogram is COMPACT (round 6): it covers the up to four ranges of bins [cb_lo[r], cb_hi[r]] its tally records
    // can fall into -- the levels of the layers that are walked voxel by voxel, plane by plane, and their heating cells -- at cb_off[r] ...;
    // ncb entries in all (on 480 x 480 x 117: 1140 of 5000 bins, 4.6 KB instead of 20 KB of LDS: five workgroups per CU instead of three, the
    // photon loop 58 -> 47 ms per 1e8 photons); unused ranges: lo > hi.  Flights through uniform layers are run records there whatever their
    // length (run_min 1): no record of theirs falls outside.
    int cb_lo[4], cb_hi[4], cb_off[4];
    int ncb, run_min;
};
// cursor[0]: records reserved; cursor[1]: records of the launch in all, expanded runs included (k_tl_prefix); cursor[2]: run records reserved
// A run record: [0] px, py (position in the voxel where the run was entered), z (absolute height there), weight
//               [1] ux / |uz|, uy / |uz| (0 under the independent-column approximation), ix | iy << 16, first level | levels << 10 | plane << 20 | ipa << 22
__host__ __device__ inline int tl_nrow(const TallyList &TL) { return (TL.hist_wg ? TL.nwave / 4 : TL.nwave) * (TL.run_cap ? 2 : 1); }

// MIX: as in k_transport_lean: 0 one 1-D and one 3-D constituent, 1 a second 3-D constituent, 2 the general mixture (several 1-D
//      constituents, tabulated phase functions: the tables staged in LDS behind the waves' stages)
template <bool COUNT, bool P3D, int MIX>
__global__ void __launch_bounds__(256, MI3D_FLUX_WAVES(COUNT))
k_transport_flux(const DevScene S, const TallyList *__restrict__ TLp, const uint64_t nphoton, const uint64_t seed, const uint64_t offset) {
    constexpr bool TWO = (MIX == 1), GEN = (MIX == 2);
    // (the list's description stays in memory: only the rare step that writes staged records out reads it, with scalar loads;
    //  as kernel arguments its ten pointers cost the walk spilled registers)
#define TL (*TLp)
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform, said so: what hangs on it stays in scalar registers)
    const unsigned tl_cap = TLp->cap;
    const int tl_nbins = TLp->nbins;
    const int tl_hwg = TLp->hist_wg;
    // (one histogram per wave over all bins, or -- more than 1024 bins -- ONE for the workgroup over the bins its records can fall into: TallyList::cb_lo ...)
    const int hist_f4 = tl_cap ? (tl_hwg ? (TLp->ncb + 3) / 4 : tl_nbins) : 0;   // float4 the histogram(s) of this workgroup take
    extern __shared__ float4 smem[];
    constexpr int kL4 = kLayStride / 4;
    const float4 *lay4 = smem + kL4;
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(lay4);
    const int o_cold = (S.nz + 2) * kL4;
    const DevCold *cold = reinterpret_cast<const DevCold *>(smem + o_cold);
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + o_cold + kColdF4) + (tl_hwg ? 0u : wave_u * (unsigned)(tl_cap ? tl_nbins : 0));   // this wave's (this workgroup's) records per bin
    // per wave: room for 64 run records of two float4 (B0)
    float4 *wq = smem + o_cold + kColdF4 + hist_f4 + wave_u * 128;
    uint2 *stage = reinterpret_cast<uint2 *>(smem + o_cold + kColdF4 + hist_f4 + 4 * 128) + wave_u * kTlStage;   // per wave: kTlStage tally records
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.cold->lay);
        for (int i = threadIdx.x; i < S.nz * kL4; i += blockDim.x) smem[kL4 + i] = src[i];
        // end records (k_transport_lean): layers -1 and nz read as uniform layers of no thickness; the upper one knows the height of the
        // top of the atmosphere, where level nz is crossed
        if (threadIdx.x < 2 * kL4)
            smem[threadIdx.x < kL4 ? threadIdx.x : (S.nz + 1) * kL4 + (threadIdx.x - kL4)] = make_float4(0.0f, 0.0f, threadIdx.x == kL4 ? S.cold->ztoa : 0.0f, 0.0f);
        const float4 *csrc = reinterpret_cast<const float4 *>(S.cold);
        if (threadIdx.x < kColdF4) smem[o_cold + threadIdx.x] = csrc[threadIdx.x];
        if (tl_cap) for (int i = threadIdx.x; i < 4 * hist_f4; i += blockDim.x) reinterpret_cast<uint32_t *>(smem + o_cold + kColdF4)[i] = 0u;
    }
    const float *ltab = nullptr;
    if (GEN && S.cold->tab_n > 0) {     // (the launch has given the kernel the LDS: lean_tab_floats)
        float *dst = reinterpret_cast<float *>(reinterpret_cast<uint2 *>(smem + o_cold + kColdF4 + hist_f4 + 4 * 128) + 4 * kTlStage);
        stage_tables(S.cold, dst);
        ltab = dst;
    }
    __syncthreads();
    const LeanTab T = GEN ? lean_tab(cold, ltab) : LeanTab{};
    const int np1d = GEN ? S.np1d : 1;
    const bool two3 = TWO || (GEN && S.np3d > 1);

    const bool ipa_all = (S.solver == MI3D_SOLVER_IPA);
#define IPA_NOW() (ipa_all || (P3D && !direct))
    Counters cnt = {};
    const unsigned sx_b = S.vcol_f4 * 16u, sy_b = S.vrow_f4 * 16u;
    const char *vbase = reinterpret_cast<const char *>(S.vrec) - (long)S.k3lo * 16;
    const unsigned ncol = (unsigned)(S.nx * S.ny), nlev = (unsigned)(S.nz + 1);
    const unsigned lane = threadIdx.x & 63u;

    // ---- lane state (k_transport_lean's, photons only)
    float px = 0, py = 0, pz = 0, ux = 0, uy = 0, uz = 1, iux = 1, iuy = 1, iuz = 1;
    float t = 0, tx = 0, ty = 0, tz = 0;
    int ix = 0, iy = 0, k = 0, stepx = 0, stepy = 0;
    int wrapx = 0, wrapy = 0, stepk = 1;
    unsigned tbase = 0;   // tally index of a crossing out of layer k in column (ix, iy): tbase + k ncol + iy nx + ix (plane and, going up, the level above, folded in)
    float rem = 0.0f, w = 0.0f;
    float u1 = 0, u2 = 0, u3 = 0;
    uint64_t id = 0;
    uint32_t draw = 0;
    int mode = M_NEED, kind = E_LAUNCH, dkind = D_LAUNCH;
    bool direct = false;
    unsigned long long pool_next = 0, pool_end = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    unsigned victim = 0;
    float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float &bt_ev = rec.x, &ev_tab = rec.y, &ev_ks0 = rec.z, &ev_apf0 = rec.w;
    float ev_ksb = 0.0f, ev_apfb = 0.0f;
    float &ev_sfc = ev_tab;
    // the tally a lane has made in the step just taken, written out by the whole wave at once (TL_FLUSH)
    unsigned pidx = kTlNone;
    float pw = 0.0f;
    unsigned long long tl_pos = 0, tl_end = 0;   // wave-uniform: the part of this wave's chunk that is still free
    bool tl_off = (tl_cap == 0u);                 // wave-uniform: tallies go out as atomics (no lists, or the list has run full)

    // Tallies are staged in LDS (kTlStage records per wave) and leave for the wave's chunk of the list 192 to 256 at a time, with
    // fully coalesced stores: a global store per walk step would sit in the same counter (vmcnt) as the walk's record reads, and
    // with loads and stores mixed the compiler has to wait for ALL of them before it may use a record.
    unsigned st_n = 0;   // wave-uniform: records staged
    unsigned tl_nch = 0; // wave-uniform: chunks this wave has taken
    unsigned rl_pos = 0, rl_end = 0, rl_nch = 0;   // wave-uniform: the free part of this wave's chunk of run records; chunks taken
    bool run_off = (TLp->run_cap == 0u);          // wave-uniform: no run records (none asked for, or the list has run full): a record per level
    const unsigned wid = blockIdx.x * 4u + wave_u;
    // (tally indices below nflux are flux cells; the heating rates' cells [nz][ny][nx] follow them: one index space, one list)
    const unsigned nflux = 3u * nlev * ncol;
#define TL_ATOMIC(idx_, w_) do { if ((idx_) < nflux) atomicAdd(&S.flux[(idx_)], (tally_t)(w_)); else atomicAdd(&cold->heat[(idx_) - nflux], (double)(w_)); } while (0)
#ifdef MI3D_ABL_NOFLUXATOMIC   // ablation (no result): the loop without its tallies
#define TL_FLUSH() do { asm volatile("" ::"v"(pidx), "v"(pw)); pidx = kTlNone; } while (0)
#define TL_DUMP() do { } while (0)
#else
#define TL_DUMP()                                                                                                               \
    do {                                                                                                                        \
        if (tl_pos + st_n > tl_end) {                                                                                           \
            if (lane == 0u && tl_end != 0ull) TL.chunk_fill[(tl_end - kTlChunk) / kTlChunk] = (uint32_t)(tl_pos - (tl_end - kTlChunk)); \
            unsigned long long base_ = 0;                                                                                       \
            if (lane == 0u) base_ = atomicAdd(TL.cursor, (unsigned long long)kTlChunk);                                         \
            /* (every lane is active here: the first one is lane 0; read into scalar registers, the chunk stays wave-uniform for the compiler) */ \
            base_ = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)base_) |                                      \
                    ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base_ >> 32)) << 32);                       \
            if (base_ + kTlChunk > (unsigned long long)TL.cap || tl_nch >= (unsigned)TL.wcap) { tl_off = true; tl_pos = 0; tl_end = 0; } \
            else {                                                                                                              \
                tl_pos = base_; tl_end = base_ + kTlChunk;                                                                      \
                if (lane == 0u) TL.wave_chunks[(size_t)wid * TL.wcap + tl_nch] = (uint32_t)(base_ / kTlChunk);                  \
                tl_nch++;                                                                                                       \
            }                                                                                                                   \
        }                                                                                                                       \
        __builtin_amdgcn_wave_barrier();                                                                                        \
        if (!tl_off) {                                                                                                          \
            uint2 *dst_ = TL.rec + tl_pos;                                                                                      \
            for (unsigned i_ = lane; i_ < st_n; i_ += 64u) {                                                                    \
                uint2 v_ = stage[i_];                                                                                           \
                int hb_ = (int)(v_.x >> TL.shift);                                                                              \
                if (tl_hwg) {   /* the workgroup's compact histogram: the bins its records can fall into (TallyList::cb_lo ...) */ \
                    const int b_ = hb_; hb_ = -1;                                                                               \
                    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) if (b_ >= TL.cb_lo[r_] && b_ <= TL.cb_hi[r_]) hb_ = TL.cb_off[r_] + b_ - TL.cb_lo[r_]; \
                    /* (a record outside them -- the surface's tally, a launch under a wide cone: rare -- is added at once; its slot stays empty) */ \
                    if (hb_ < 0) { TL_ATOMIC(v_.x, __uint_as_float(v_.y)); v_ = make_uint2(kTlNone, 0u); }                      \
                }                                                                                                               \
                if (MI3D_TL_NT & 1) nt_store(dst_ + i_, v_); else dst_[i_] = v_;                                                \
                if (hb_ >= 0) atomicAdd(&lhist[hb_], 1u);                                                                       \
            }                                                                                                                   \
            tl_pos += st_n;                                                                                                     \
        } else {   /* the list has run full: what is staged, and everything from here on, goes out as atomics */               \
            for (unsigned i_ = lane; i_ < st_n; i_ += 64u) {                                                                    \
                const uint2 v_ = stage[i_];                                                                                     \
                TL_ATOMIC(v_.x, __uint_as_float(v_.y));                                                                         \
            }                                                                                                                   \
        }                                                                                                                       \
        __builtin_amdgcn_wave_barrier();                                                                                        \
        st_n = 0;                                                                                                               \
    } while (0)
#ifdef MI3D_RUNLEN   // (diagnostic build: le_rays counts the records that do NOT continue the lane's last one -- same column, same weight, the next level)
    unsigned rl_last = kTlNone; float rl_w = 0.0f;
#define MI3D_RUNLEN_DIAG() do { if (COUNT && pidx < nflux) { const bool mg_ = rl_last != kTlNone && (pidx % ncol) == (rl_last % ncol) && (pidx - rl_last == ncol || rl_last - pidx == ncol) && pw == rl_w; if (!mg_) cnt.le_rays++; rl_last = pidx; rl_w = pw; } } while (0)
#else
#define MI3D_RUNLEN_DIAG() do { } while (0)
#endif
#define TL_FLUSH()                                                                                                              \
    do {                                                                                                                        \
        const unsigned long long m_ = __ballot(pidx != kTlNone);                                                                \
        if (m_ != 0ull) {                                                                                                       \
            if (pidx != kTlNone) {                                                                                              \
                if (!tl_off) stage[st_n + __builtin_amdgcn_mbcnt_hi((unsigned)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_, 0u))] = make_uint2(pidx, __float_as_uint(pw)); \
                else { TL_ATOMIC(pidx, pw); if (COUNT) cnt.le_column++; /* (instrumented build: tallies that went out as atomics) */ } \
                if (COUNT && pidx < nflux) cnt.flux_tally++;                                                                    \
                MI3D_RUNLEN_DIAG();                                                                                             \
                pidx = kTlNone;                                                                                                 \
            }                                                                                                                   \
            if (!tl_off) {                                                                                                      \
                st_n += (unsigned)__popcll(m_);                                                                                 \
                if (st_n > kTlStage - 64u) TL_DUMP();                                                                           \
            }                                                                                                                   \
        }                                                                                                                       \
    } while (0)
#endif

#define MI3D_TICK(slot) MI3D_DIAG_TICK(COUNT, cnt, tick, slot)     // (mi3d_diag.h: instrumented build only)
    long long tick = COUNT ? clock64() : 0;   // instrumented build: wave clock ticks / 64 spent in A, walk end + B0, C + B2, B4, B5, B6 + B7
    unsigned pass_ctr = 0;
    // The loop is k_transport_lean's of round 4 (mi3d_kernel_lean.hip): lane modes, a branch-light voxel step, block C for the
    // collisions the walk finds, the rarer events in full passes, new photons from entry records.
    constexpr int M_COLLU = 12, M_UNIFW = 13, M_SETUP = 14;   // (as in mi3d_kernel_lean.hip)
#define VREC(ix_, iy_, k_) (*reinterpret_cast<const float4 *>(vbase + ((unsigned)(iy_) * sy_b + (unsigned)(ix_) * sx_b + (unsigned)(k_) * 16u)))
    for (;;) {
        // =================================== phase A: voxel steps ===================================
        MI3D_MARK("FA");
        for (;;) {
            const bool flying = (mode == M_FLY);
            const int nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (nfly < MI3D_FLUX_THRESH && __ballot(mode != M_FLY && mode != M_DONE) != 0ull) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (flying) {
                const float tn = fminf(fminf(tx, ty), tz);
                const float dtau = rec.x * (tn - t);
                if (COUNT) { cnt.steps++; cnt.steps3d++; }
                if (dtau >= rem) mode = M_COLL;
                else {
                    rem -= dtau;
                    t = tn;
                    const bool zf = (tz == tn), xf = !zf && (tx == tn), yf = !zf && !xf;
                    if (zf) {
                        // a level crossed: the one above layer k going up, the one below it going down
                        pidx = tbase + (unsigned)k * ncol + (unsigned)(iy * S.nx + ix);
                        pw = w;
                        k += stepk;
                        const float4 Ln = lay4[k * kL4];
                        tz = fmaf(Ln.x, iuz, tz);
                        if (!(__float_as_int(Ln.w) & kLayStep3d)) mode = M_UNIFW;
                    }
                    const int cx = ix + stepx, cy = iy + stepy;
                    const int cxw = (unsigned)cx >= (unsigned)S.nx ? wrapx : cx, cyw = (unsigned)cy >= (unsigned)S.ny ? wrapy : cy;
                    ix = xf ? cxw : ix;
                    iy = yf ? cyw : iy;
                    const float txn = fmaf(S.dx, iux, tx), tyn = fmaf(S.dy, iuy, ty);
                    tx = xf ? txn : tx;
                    ty = yf ? tyn : ty;
                    if (mode == M_FLY) rec = VREC(ix, iy, k);
                }
            }
            TL_FLUSH();
        }

        // =================================== phase B ===================================
        MI3D_TICK(0);
        MI3D_MARK("FBSCHED");
        if (COUNT) { cnt.b_slots++; if (mode != M_FLY && mode != M_DONE) cnt.b_lanes++; }
        const bool full = MI3D_FLUX_FAST_PASS <= 1 || ((pass_ctr++ % (unsigned)(MI3D_FLUX_FAST_PASS)) == 0u) || __ballot(mode == M_COLL) == 0ull;

        // =================================== block C: a collision the voxel walk has found ===================================
        MI3D_MARK("FC");
        if (mode == M_COLL) {
            const float4 L = lay4[k * kL4];
            const LayerRec &Lk = lay[k];
            const float ks1 = Lk.ks1d[0];
            const float ibt = frcp(rec.x);
            const float tc = fmaf(rem, ibt, t);
            const float ax = __builtin_amdgcn_fmed3f((tx - tc) * floor_abs(ux), 0.0f, S.dx);
            const float ay = __builtin_amdgcn_fmed3f((ty - tc) * floor_abs(uy), 0.0f, S.dy);
            const float az = __builtin_amdgcn_fmed3f((tz - tc) * floor_abs(uz), 0.0f, L.x);
            px = ux > 0.0f ? S.dx - ax : ax;
            py = uy > 0.0f ? S.dy - ay : ay;
            pz = uz > 0.0f ? L.x - az : az;
            if (COUNT) cnt.scatter++;
            const float ks3 = rec.z;
            float kstot = ks1 + ks3;
            if (GEN) for (int ip = 1; ip < np1d; ++ip) kstot += Lk.ks1d[ip];
            if (two3) {
                const float2 cs = cold->csca[((unsigned)(iy * S.nx + ix) * (unsigned)S.nz3 + (unsigned)(k - S.k3lo)) * 2u + 1u];
                ev_ksb = cs.x; ev_apfb = cs.y;
                kstot += ev_ksb;
            }
            const float w_in = w;
            w *= (kstot >= rec.x) ? 1.0f : kstot * ibt;
            // heating rates (Flx_mhrt = 1): what the collision takes from the weight stays in this cell -- one more tally record
            if (cold->heat && kstot < rec.x) { pidx = nflux + (unsigned)(k * S.ny + iy) * (unsigned)S.nx + (unsigned)ix; pw = w_in * (rec.x - kstot) * ibt; }
            if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; mode = M_NEED; }
            else {
                float mu_rot;
                if (GEN) {
                    float usel;
                    const float apf_g = lean_mix_select(Lk, np1d, ks3, rec.w, ev_ksb, ev_apfb, two3 ? 2 : 1, u1, kstot, usel);
                    mu_rot = lean_phase_sample(T, apf_g, u2, usel);
                } else {
                    const float target = u1 * kstot;
                    const bool first = target < ks1;
                    float apf_sel = first ? Lk.apf1d[0] : rec.w;
                    if (TWO && !first && !(target < ks1 + ks3)) apf_sel = ev_apfb;
                    mu_rot = phase_sample_analytic(apf_sel, u2);
                }
                rotate_dir(ux, uy, uz, mu_rot, u3);
                direct = false;
                if (w < S.wmin) { if (COUNT) cnt.roulette++; mode = M_DRAW; dkind = D_ROULETTE; }
                else {
                    float r0, r1, r2, r3;
                    draw4_fast(seed, id, draw++, r0, r1, r2, r3);
                    rem = -0.69314718f * __builtin_amdgcn_logf(r0);
                    u1 = r1; u2 = r2; u3 = r3;
                    iux = frcp(floor_abs(ux)); iuy = frcp(floor_abs(uy)); iuz = frcp(floor_abs(uz));
                    tx = (ux > 0.0f ? S.dx - px : px) * iux;
                    ty = (uy > 0.0f ? S.dy - py : py) * iuy;
                    tz = (uz > 0.0f ? L.x - pz : pz) * iuz;
                    t = 0.0f;
                    const bool ipa = IPA_NOW();
                    stepx = ipa ? 0 : (ux > 0.0f ? 1 : -1);
                    stepy = ipa ? 0 : (uy > 0.0f ? 1 : -1);
                    wrapx = ux > 0.0f ? 0 : S.nx - 1; wrapy = uy > 0.0f ? 0 : S.ny - 1; stepk = uz > 0.0f ? 1 : -1;
                    tbase = uz > 0.0f ? (2u * nlev + 1u) * ncol : nlev * ncol;     // (scattered light: never the direct plane)
                    mode = M_FLY;
                }
            }
        }
        if (cold->heat) TL_FLUSH();      // (the heat records of this pass's collisions)
        MI3D_TICK(2);

        // =================================== the rarer events: full passes ===================================
        if (full) {
        MI3D_MARK("FB0");
        // ---- where a walk has ended on a level, in front of a horizontally uniform layer (or out of the atmosphere)
        if (mode == M_UNIFW) {
            const float4 L = lay4[k * kL4];
            const float ax = __builtin_amdgcn_fmed3f((tx - t) * floor_abs(ux), 0.0f, S.dx);
            const float ay = __builtin_amdgcn_fmed3f((ty - t) * floor_abs(uy), 0.0f, S.dy);
            px = ux > 0.0f ? S.dx - ax : ax;
            py = uy > 0.0f ? S.dy - ay : ay;
            pz = uz > 0.0f ? 0.0f : L.x;
            mode = M_UNIF;
        }
        // ---- B0: photons inside runs of horizontally uniform layers: the whole rest of the run at once, then one tally per level crossed
        if (mode == M_UNIF && (k < 0 || k >= S.nz)) {
            if (k < 0) { k = 0; pz = 0.0f; mode = M_SURF; }
            else { if (COUNT) cnt.escaped++; mode = M_NEED; }
        }
        const bool inrun = mode == M_UNIF;
        int knew = 0, next = M_SETUP, la = 1, lb = 0;
        float pzn = 0.0f, s = 0.0f, iuzl = 0.0f;
        if (inrun) {
            const bool up = uz > 0.0f;
            const LayerRec &Lk = lay[k];
            const int kend = up ? Lk.run_hi : Lk.run_lo;
            const LayerRec &Le = lay[kend];
            const float tv = up ? (Le.tauz + Le.bt * Le.dz - Lk.tauz) - Lk.bt * pz
                                : (Lk.tauz - Le.tauz) + Lk.bt * pz;
            const float hv = up ? (Le.zlo + Le.dz) - (Lk.zlo + pz) : (Lk.zlo + pz) - Le.zlo;
            iuzl = frcp(fmaxf(fabsf(uz), 1e-20f));
            const float tpath = tv * iuzl;
            int kraw;   // the layer the flight ends in, before the surface and the top are told apart
            if (tpath < rem) {
                rem -= tpath;
                s = hv * iuzl;
                kraw = knew = up ? kend + 1 : kend - 1;
                next = M_SETUP;
                if (knew >= S.nz) { if (COUNT) cnt.escaped++; next = M_NEED; }
                else if (knew < 0) { knew = 0; next = M_SURF; }
                else if (!up) pzn = lay4[knew * kL4].x;
            } else {
                const float T = Lk.tauz + Lk.bt * pz + (up ? rem : -rem) * fabsf(uz);
                int lo = up ? k : kend, hi = up ? kend : k;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (lay[mid].tauz <= T) lo = mid; else hi = mid - 1;
                }
                const float4 Lj = lay4[lo * kL4];
                pzn = fminf(fmaxf((T - lay[lo].tauz) * frcp(fmaxf(Lj.y, 1e-30f)), 0.0f), Lj.x);
                s = fabsf((Lj.z + pzn) - (Lk.zlo + pz)) * iuzl;
                kraw = knew = lo;
                bt_ev = Lj.y;
                next = M_COLLU;
            }
            if (COUNT) cnt.steps++;
            // levels crossed: going up those above layers k .. kraw - 1; going down those below layers k .. kraw + 1 -- of the direct
            // beam only the ones below kdir (above, its flux is known analytically and added when the result is read)
            if (up) { la = k + 1; lb = kraw; }
            else { la = kraw + 1; lb = direct ? min(k, S.kdir - 1) : k; }
        }
        // The levels of the wave's runs, sixteen levels of four runs at a time: a lane inside a run leaves a record of it in LDS --
        // where it entered the run, its slope, column, weight, plane and up to sixteen levels -- and the wave works the records off,
        // lane (r, l) placing level l of record r.  (A loop of every lane over its own levels kept one lane in six busy and took
        // 60 % of the kernel's time.)
        {
            int nrem = inrun ? lb - la + 1 : 0, lcur = la;
            // Round 6: a run that crosses kRunMin tallied levels or more leaves ONE record of 32 bytes; k_tl_runs places its levels after the
            // launch (the same multiply-add and fold, level by level, straight into the bins).  The wave hands consecutive slots of its chunk
            // to the lanes that have a run: each of the two stores writes one contiguous piece.
            if (!run_off) {
                const bool asrun = nrem >= TL.run_min;
                const unsigned long long mr = __ballot(asrun);
                if (mr != 0ull) {
                    const unsigned nr = (unsigned)__popcll(mr);
                    if (rl_pos + nr > rl_end) {   // (every lane is active here: the block is not under a lane's branch)
                        if (lane == 0u && rl_end != 0u) TL.run_fill[(rl_end - kRunChunk) / kRunChunk] = rl_pos - (rl_end - kRunChunk);
                        unsigned long long base_ = 0;
                        if (lane == 0u) base_ = atomicAdd(TL.cursor + 2, (unsigned long long)kRunChunk);
                        base_ = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)base_) |
                                ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base_ >> 32)) << 32);
                        if (base_ + kRunChunk > (unsigned long long)TL.run_cap || rl_nch >= (unsigned)TL.run_wcap) { run_off = true; rl_pos = 0; rl_end = 0; }
                        else {
                            rl_pos = (unsigned)base_; rl_end = (unsigned)base_ + kRunChunk;
                            if (lane == 0u) TL.run_chunks[(size_t)wid * TL.run_wcap + rl_nch] = (uint32_t)(base_ / kRunChunk);
                            rl_nch++;
                        }
                    }
                    if (!run_off) {
                        if (asrun) {
                            const unsigned slot = rl_pos + __builtin_amdgcn_mbcnt_hi((unsigned)(mr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mr, 0u));
                            float4 *dst = TL.runs + (size_t)(slot / kRunChunk) * (2u * kRunChunk) + (slot % kRunChunk);
                            const bool ipa = IPA_NOW();
                            const unsigned plane = uz > 0.0f ? 2u : (direct ? 0u : 1u);
                            const float4 ra_ = make_float4(px, py, lay4[k * kL4].z + pz, w);
                            const float4 rb_ = make_float4(ipa ? 0.0f : ux * iuzl, ipa ? 0.0f : uy * iuzl, __uint_as_float((unsigned)ix | ((unsigned)iy << 16)),
                                                           __uint_as_float((unsigned)la | ((unsigned)nrem << 10) | (plane << 20) | (ipa ? 1u << 22 : 0u)));
                            if (MI3D_TL_NT & 8) { nt_store(dst, ra_); nt_store(dst + kRunChunk, rb_); } else { dst[0] = ra_; dst[kRunChunk] = rb_; }
                            if (COUNT) cnt.flux_tally += (uint32_t)nrem;
                            nrem = 0;
                        }
                        rl_pos += nr;
                    }
                }
            }
            for (;;) {
                const unsigned long long m = __ballot(nrem > 0);
                if (m == 0ull) break;
                const int R = __popcll(m);
                if (nrem > 0) {
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    const bool ipa = IPA_NOW();
                    const unsigned plane = uz > 0.0f ? 2u : (direct ? 0u : 1u);
                    wq[2 * rank] = make_float4(px, py, lay4[k * kL4].z + pz, w);
                    wq[2 * rank + 1] = make_float4(ipa ? 0.0f : ux * iuzl, ipa ? 0.0f : uy * iuzl, __uint_as_float((unsigned)ix | ((unsigned)iy << 16)),
                                                   __uint_as_float((unsigned)lcur | ((unsigned)min(nrem, 16) << 16) | (plane << 24) | (ipa ? 1u << 26 : 0u)));
                    lcur += 16; nrem -= 16;
                }
                __builtin_amdgcn_wave_barrier();   // (one wave's LDS traffic is served in order: the reads below see the writes above)
                for (int g = 0; g * 4 < R; ++g) {
                    const int r = g * 4 + (int)(lane >> 4), l = (int)(lane & 15u);
                    if (r < R) {
                        const float4 A = wq[2 * r], B = wq[2 * r + 1];
                        const unsigned q = __float_as_uint(B.w), cxy = __float_as_uint(B.z);
                        if (l < (int)((q >> 16) & 31u)) {
                            const int L = (int)(q & 0xffffu) + l;
                            int jx = (int)(cxy & 0xffffu), jy = (int)(cxy >> 16);
                            if (!(q & (1u << 26))) {
                                const float sl = fabsf(lay4[L * kL4].z - A.z);
                                const float fx = floorf(fmaf(B.x, sl, A.x) * cold->inv_dx), fy = floorf(fmaf(B.y, sl, A.y) * cold->inv_dy);
                                jx += (int)fx; jy += (int)fy;
                                if ((unsigned)jx >= (unsigned)S.nx) jx = wrapi(jx, S.nx, cold->inv_nx);
                                if ((unsigned)jy >= (unsigned)S.ny) jy = wrapi(jy, S.ny, cold->inv_ny);
                            }
                            pidx = (((q >> 24) & 3u) * nlev + (unsigned)L) * ncol + (unsigned)(jy * S.nx + jx);
                            pw = A.w;
                        }
                    }
                    TL_FLUSH();
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (inrun) {
            px += ux * s; py += uy * s;
            k = knew; pz = pzn;
            fold_xy(S, cold, px, py, ix, iy, IPA_NOW());
            mode = next;
        }

        MI3D_TICK(1);
        MI3D_MARK("FB2");
        // ---- B2: a collision inside uniform layers, or the surface: the weight (and what it loses, for heating rates)
        if (mode == M_COLLU || mode == M_SURF) {
            const float4 L = lay4[k * kL4];
            const int flags = __float_as_int(L.w);
            const bool in3d = (flags & kLayIn3d) != 0;
            if (!(flags & kLayStep3d)) {
                float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (in3d) r = VREC(ix, iy, k);
                ev_tab = r.y; ev_ks0 = r.z; ev_apf0 = r.w;
            }
            bool dead = false;
            if (mode == M_SURF) {
                if (COUNT) cnt.surface++;
                const Sfc sf = load_sfc(S, cold, ix, iy, px, py);
                if (!(flags & kLayStep3d)) bt_ev = L.y;
                ev_ks0 = sf.p0; ev_apf0 = sf.p1; ev_sfc = sf.p2; kind = E_SURFACE | (sf.type << 4);
            } else {
                if (COUNT) cnt.scatter++;
                float kstot = lay[k].ks1d[0] + (in3d ? ev_ks0 : 0.0f);
                if (GEN) for (int ip = 1; ip < np1d; ++ip) kstot += lay[k].ks1d[ip];
                if (two3) {
                    ev_ksb = 0.0f;
                    if (in3d) {
                        const float2 cs = cold->csca[((unsigned)(iy * S.nx + ix) * (unsigned)S.nz3 + (unsigned)(k - S.k3lo)) * 2u + 1u];
                        ev_ksb = cs.x; ev_apfb = cs.y;
                    }
                    kstot += ev_ksb;
                }
                const float w_in = w;
                w *= (kstot >= bt_ev) ? 1.0f : kstot * frcp(bt_ev);
                if (cold->heat && kstot < bt_ev) { pidx = nflux + (unsigned)(k * S.ny + iy) * (unsigned)S.nx + (unsigned)ix; pw = w_in * (bt_ev - kstot) * frcp(bt_ev); }
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; dead = true; }
                kind = E_SCATTER;
            }
            mode = dead ? M_NEED : M_FINISH;
        }
        // (heating rates: the record goes out now -- a lane whose photon the collision has killed takes a new photon in this very
        //  pass, and a launch or surface tally further down would overwrite what is pending)
        if (cold->heat) TL_FLUSH();

        MI3D_TICK(2);
        MI3D_MARK("FB5");
        // ---- B5: finish the event (scattering inside uniform layers, surface reflection, a launch without entry record)
        if (mode == M_FINISH) {
            float bx = ux, by = uy, bz = uz, mu_rot = u2;
            if ((kind & 15) == E_SURFACE) {
                bx = 0.0f; by = 0.0f; bz = 1.0f;
                mu_rot = fsqrt(u2);
            } else if ((kind & 15) == E_SCATTER) {
                const LayerRec &Lk = lay[k];
                const bool in3d = (Lk.flags & kLayIn3d) != 0;
                const float ks1 = Lk.ks1d[0], ks3 = in3d ? ev_ks0 : 0.0f;
                float kst = ks1 + ks3;
                if (GEN) for (int ip = 1; ip < np1d; ++ip) kst += Lk.ks1d[ip];
                if (two3) kst += in3d ? ev_ksb : 0.0f;
                if (GEN) {
                    float usel;
                    const float apf_g = lean_mix_select(Lk, np1d, ks3, ev_apf0, ev_ksb, ev_apfb, in3d ? (two3 ? 2 : 1) : 0, u1, kst, usel);
                    mu_rot = lean_phase_sample(T, apf_g, u2, usel);
                } else {
                    const float target = u1 * kst;
                    const bool first = (target < ks1) || !in3d;
                    float apf_sel = first ? Lk.apf1d[0] : ev_apf0;
                    if (TWO && !first && !(target < ks1 + ks3)) apf_sel = ev_apfb;
                    mu_rot = phase_sample_analytic(apf_sel, u2);
                }
            }
            if (!(kind == E_LAUNCH && cold->cos_cone >= 1.0f)) rotate_dir(bx, by, bz, mu_rot, u3);
            if ((kind & 15) == E_SURFACE) {
                const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ix, iy, px, py) : Sfc{kind >> 4, ev_ks0, ev_apf0, ev_sfc, 0.0f, 0.0f};
                bz = fmaxf(bz, 1e-9f);
                w *= surface_R(sf, ux, uy, uz, bx, by, bz);
                if (w > 0.0f) { pidx = 2u * nlev * ncol + (unsigned)(iy * S.nx + ix); pw = w; }   // what the surface sends back up through level 0
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
        TL_FLUSH();   // (the surface tallies)

        MI3D_TICK(4);
        MI3D_MARK("FB6");
        // ---- B6: the Philox block of the rarer events
        if (mode == M_DRAW) {
            float r0, r1, r2, r3;
            draw4_fast(seed, id, draw++, r0, r1, r2, r3);
            if (dkind == D_FLIGHT) {
                rem = -0.69314718f * __builtin_amdgcn_logf(r0);
                u1 = r1; u2 = r2; u3 = r3;
                mode = (lay[k].flags & kLayStep3d) ? M_SETUP : M_UNIF;
            } else if (dkind == D_ROULETTE) {
                if (r0 * S.wfac < w) { w = S.wfac; dkind = D_FLIGHT; }
                else { if (COUNT) cnt.killed++; mode = M_NEED; }
            } else { // D_LAUNCH (no entry records)
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
                w = 1.0f;
                direct = true;
                if (S.nz < S.kdir) { pidx = (unsigned)S.nz * ncol + (unsigned)(iy * S.nx + ix); pw = w; }   // (a wide source cone: the top level is tallied too)
                kind = E_LAUNCH;
                mode = M_FINISH;
            }
        }
        MI3D_TICK(5);

        MI3D_MARK("FB4");
        // ---- B4: next photon (k_transport_lean: entry records)
        if (mode == M_NEED && (id != 0 || draw != 0)) { cnt.photons++; id = 0; draw = 0; }
        for (;;) {
            const unsigned long long need = __ballot(mode == M_NEED);
            if (need == 0ull) break;
            if (pool_next >= pool_end) {
                const int leader = __ffsll((long long)need) - 1;
                bool got = false;
                while (victim < 8u) {
                    const unsigned x = (xcc + victim) & 7u;
                    const unsigned long long lo = (nphoton * x) >> 3, hi = (nphoton * (x + 1u)) >> 3;
                    unsigned long long b = 0;
                    if ((int)lane == leader) b = atomicAdd(cold->next_photon + x * kCtrStride, (unsigned long long)kChunk);
                    b = ((unsigned long long)__builtin_amdgcn_readlane((int)(b >> 32), leader) << 32) | (unsigned)__builtin_amdgcn_readlane((int)b, leader);
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
                if (cold->entry) {
                    const float4 *e = cold->entry + entry_index((unsigned)(pool_next + rank));
#if MI3D_ENTRY_NT_LOAD
                    const float4 q0 = nt_load(e), q1 = nt_load(e + 64), q2 = nt_load(e + 128);
#else
                    const float4 q0 = e[0], q1 = e[64], q2 = e[128];
#endif
                    px = q0.x; py = q0.y; pz = q0.z; rem = q0.w;
                    ux = q1.x; uy = q1.y; uz = q1.z; u1 = q1.w;
                    u2 = q2.x; u3 = q2.y;
                    const unsigned cell = __float_as_uint(q2.z), km = __float_as_uint(q2.w);
                    ix = (int)(cell & 0xffffu); iy = (int)(cell >> 16);
                    k = (int)(km & 0xffffu);
                    mode = ((km >> 16) & 0x7fffu) == (unsigned)M_FLY ? M_SETUP : M_UNIF;
                    if (COUNT && (km >> 31)) cnt.steps++;
                    w = 1.0f; direct = true; draw = 2;
                    kind = E_LAUNCH; dkind = D_FLIGHT;
                    // (a wide source cone: the top level is tallied too -- what B6 does for a photon launched inside the loop)
                    if (S.nz < S.kdir) { pidx = (unsigned)S.nz * ncol + (unsigned)(iy * S.nx + ix); pw = w; }
                } else {
                    draw = 0;
                    dkind = D_LAUNCH;
                    mode = M_DRAW;
                }
            }
            pool_next += nn < avail ? nn : avail;
        }
        TL_FLUSH();   // (launch tallies)
        MI3D_TICK(3);

        MI3D_MARK("FB7");
        // ---- B7: a lane about to walk: the parameters of the walk's first three faces, the base of its tallies, the first record
        if (mode == M_SETUP) {
            const float4 L = lay4[k * kL4];
            iux = frcp(floor_abs(ux)); iuy = frcp(floor_abs(uy)); iuz = frcp(floor_abs(uz));
            tx = (ux > 0.0f ? S.dx - px : px) * iux;
            ty = (uy > 0.0f ? S.dy - py : py) * iuy;
            tz = (uz > 0.0f ? L.x - pz : pz) * iuz;
            t = 0.0f;
            const bool ipa = IPA_NOW();
            stepx = ipa ? 0 : (ux > 0.0f ? 1 : -1);
            stepy = ipa ? 0 : (uy > 0.0f ? 1 : -1);
            wrapx = ux > 0.0f ? 0 : S.nx - 1; wrapy = uy > 0.0f ? 0 : S.ny - 1; stepk = uz > 0.0f ? 1 : -1;
            tbase = uz > 0.0f ? (2u * nlev + 1u) * ncol : (direct ? 0u : nlev * ncol);
            rec = VREC(ix, iy, k);
            mode = M_FLY;
        }
        MI3D_TICK(5);
        }   // full

        MI3D_MARK("FEND");
        if (__ballot(mode != M_DONE) == 0ull) break;
    }
#undef VREC
#undef MI3D_TICK
#undef IPA_NOW

    // ---- what is still staged; the wave's last chunk; the workgroup's share of the histogram
    if (st_n) TL_DUMP();
#undef TL_FLUSH
#undef TL_DUMP
#undef TL_ATOMIC
    if (tl_end != 0ull && lane == 0u) TL.chunk_fill[(tl_end - kTlChunk) / kTlChunk] = (uint32_t)(tl_pos - (tl_end - kTlChunk));
    if (rl_end != 0u && lane == 0u) TL.run_fill[(rl_end - kRunChunk) / kRunChunk] = rl_pos - (rl_end - kRunChunk);
    if (tl_cap) {
        if (lane == 0u) { TL.wave_nchunk[wid] = tl_nch; if (TL.run_cap) TL.run_nchunk[wid] = rl_nch; }
        if (tl_hwg) {
            __syncthreads();   // (every thread of the workgroup arrives here: no wave leaves the loop any other way)
            // (the workgroup's row of the histogram over ALL bins from its compact one: a bin outside the ranges holds none of its records)
            for (unsigned i = threadIdx.x; i < (unsigned)tl_nbins; i += blockDim.x) {
                uint32_t v = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) if ((int)i >= TL.cb_lo[r] && (int)i <= TL.cb_hi[r]) v = lhist[TL.cb_off[r] + (int)i - TL.cb_lo[r]];
                TL.whist[(size_t)blockIdx.x * tl_nbins + i] = v;
            }
        } else {
            __builtin_amdgcn_wave_barrier();
            for (unsigned i = lane; i < (unsigned)tl_nbins; i += 64u) TL.whist[(size_t)wid * tl_nbins + i] = lhist[i];
        }
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
            if (lane == 0u && v) atomicAdd(&S.cold->counters[q], v);
        }
    }
#undef TL
}

// ---- the records of a launch, added up ---------------------------------------------------------------------------------------

// One workgroup per bin: how many records of the bin the waves before each wave have written (exclusive scan over the waves of
// the photon loop, in the order of their numbers), and the bin's total.
// (256 threads: a workgroup of 1024 finds no room on a CU whose registers the photon loop of the next launch holds, and waited for that loop's tail)
template <int NT>
__global__ void __launch_bounds__(NT)
k_tl_wavescan(const TallyList TL) {
    __shared__ uint32_t part[NT / 64];
    const int bin = blockIdx.x;
    const unsigned tid = threadIdx.x;
    const int nrow = tl_nrow(TL);   // (the records' rows, then the runs')
    const int per = (nrow + NT - 1) / NT;
    const int lo = min((int)tid * per, nrow), hi = min(lo + per, nrow);
    uint32_t sum = 0;
    for (int w = lo; w < hi; ++w) sum += TL.whist[(size_t)w * TL.nbins + bin];
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(incl, off, 64);
        if ((tid & 63u) >= (unsigned)off) incl += x;
    }
    if ((tid & 63u) == 63u) part[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (unsigned wv = 0; wv < (unsigned)(NT / 64); ++wv) { const uint32_t x = part[wv]; total += x; if (wv < (tid >> 6)) before += x; }
    uint32_t run = before + incl - sum;
    for (int w = lo; w < hi; ++w) { TL.wbase[(size_t)w * TL.nbins + bin] = run; run += TL.whist[(size_t)w * TL.nbins + bin]; }
    if (tid == 0) TL.hist[bin] = total;
}

// exclusive prefix sums of the bins' totals (one workgroup)
template <int NT>
__global__ void __launch_bounds__(NT)
k_tl_prefix(const TallyList TL) {
    __shared__ uint32_t part[NT];
    const int per = (TL.nbins + NT - 1) / NT;
    const int lo = min((int)threadIdx.x * per, TL.nbins), hi = min(lo + per, TL.nbins);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += TL.hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) { TL.bin_start[i] = run; run += TL.hist[i]; }
    if (threadIdx.x == NT - 1) {     // (the launch's records in all, beside what its loop reserved: they size the launches to come)
        TL.bin_start[TL.nbins] = part[NT - 1]; TL.cursor[1] = part[NT - 1];
        if (TL.stats) { TL.stats[0] = TL.cursor[0]; TL.stats[1] = part[NT - 1]; TL.stats[2] = TL.cursor[2]; }
    }
}

// Counting sort of the records into their bins, without an atomic outside LDS.  Workgroup g (NT threads) sorts what the NT / 256
// waves g NT/256 ... of the photon loop have written, NT / 64 chunks at a time (sixteen records per thread, held in registers): counts
// per bin, scan, places inside LDS, then the sorted tile is copied out, the records of a bin as one contiguous piece.  Where a
// bin's pieces go is known beforehand: the photon loop counts every wave's records per bin, k_tl_wavescan and k_tl_prefix turn
// the counts into positions, and the workgroup keeps a cursor per bin in LDS.  (With a returning atomic per tile and bin on one
// cursor per bin in memory -- 210 hot addresses for the whole chip -- the kernel took 40 ms per 4.4e9 records and got slower with
// smaller tiles: the atomics' answers were what it waited for.)
#ifndef MI3D_TLS_WAVES
#define MI3D_TLS_WAVES 4
#endif
#ifndef MI3D_TLS_R
#define MI3D_TLS_R 8      // records per thread and tile of the 256-thread sort
#endif
#ifndef MI3D_TLS_NT
#define MI3D_TLS_NT 256
#endif
template <int NT, int R, int G>   // R: records per thread; G: waves of the photon loop a workgroup serves (1: one wave; 4: one of ITS workgroups, TallyList::hist_wg)
__global__ void __launch_bounds__(NT, G == 1 ? MI3D_TLS_WAVES : 4)
k_tl_scatter(const TallyList TL, tally_t *__restrict__ flux, const unsigned nflux, double *__restrict__ heat, const unsigned nheat) {
    constexpr unsigned T = (unsigned)(R * NT) / kTlChunk;  // chunks per tile
    constexpr int NW = NT / 64;                            // waves of this workgroup
    extern __shared__ uint32_t lds_u32[];
    uint32_t *lcount = lds_u32;                       // [nbins] records of the tile per bin, then the running position inside the sorted tile
    uint32_t *lstart = lcount + TL.nbins;             // [nbins] where the bin starts in the sorted tile
    uint32_t *gcur = lstart + TL.nbins;               // [nbins] where the bin's next piece goes in TL.binned
    constexpr bool SEP = (G == 1);                    // a table of its own for the bins' offsets: two barriers per tile less (not where 5000 bins share the LDS with 64-KB tiles)
    uint32_t *ldelta = SEP ? gcur + TL.nbins : lcount; // [nbins] sorted tile -> TL.binned: what to add to a record's position
    uint32_t *part = (SEP ? ldelta : gcur) + TL.nbins; // [16] scan partials
    uint32_t *cid = part + 16;                        // [kTlIds] the chunks of this workgroup's waves, one list after the other ...
    uint32_t *cfill = cid + kTlIds;                   // [kTlIds] ... and how full each is
    uint2 *sorted = reinterpret_cast<uint2 *>(cfill + kTlIds);   // [T * kTlChunk]
    const unsigned tid = threadIdx.x;
    const int w0 = blockIdx.x * G;
    if (w0 >= TL.nwave) return;
    unsigned nch[4] = {0u, 0u, 0u, 0u}, ntot = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) { nch[g] = w0 + g < TL.nwave ? TL.wave_nchunk[w0 + g] : 0u; ntot += nch[g]; }
    // chunk m of the concatenated lists: the lists are read once, up front (a tile's record loads then depend on nothing that is
    // still in memory: with the list and the fill looked up per tile every tile waited three memory latencies in a row)
    auto chunk_at = [&](unsigned m) -> unsigned {
        int g = 0;
#pragma unroll
        for (int q = 0; q < G - 1; ++q) if (g == q && m >= nch[q]) { m -= nch[q]; g = q + 1; }
        return TL.wave_chunks[(size_t)(w0 + g) * TL.wcap + m];
    };
    for (unsigned m = tid; m < ntot && m < kTlIds; m += NT) { const unsigned c = chunk_at(m); cid[m] = c; cfill[m] = TL.chunk_fill[c]; }
    for (int i = tid; i < TL.nbins; i += NT) gcur[i] = TL.bin_start[i] + TL.wbase[(size_t)(TL.hist_wg ? (int)blockIdx.x : w0) * TL.nbins + i];
    // (every thread owns `per` consecutive bins, [lo, hi): it zeroes, scans and moves on the counters of those and of no others)
    const int per = (TL.nbins + NT - 1) / NT;
    const int lo = min((int)tid * per, TL.nbins), hi = min(lo + per, TL.nbins);
    if (SEP) for (int i = lo; i < hi; ++i) lcount[i] = 0u;
    __syncthreads();
    uint2 v[R], vn[R];
    auto load_tile = [&](unsigned tile, uint2 (&dst)[R]) {
        // (two consecutive records of a chunk per load, 16 bytes a lane)
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const unsigned f = 2u * ((unsigned)(r * NT) + tid), j = f % kTlChunk, m = tile + f / kTlChunk;
            dst[2 * r] = make_uint2(kTlNone, 0u); dst[2 * r + 1] = make_uint2(kTlNone, 0u);
            if (m < ntot) {
                unsigned c, fill;
                if (m < kTlIds) { c = cid[m]; fill = cfill[m]; } else { c = chunk_at(m); fill = TL.chunk_fill[c]; }
                if (j + 1u < fill) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(TL.rec + (size_t)c * kTlChunk + j);
                    dst[2 * r] = make_uint2(q.x, q.y); dst[2 * r + 1] = make_uint2(q.z, q.w);
                } else if (j < fill) dst[2 * r] = TL.rec[(size_t)c * kTlChunk + j];
            }
        }
    };
    load_tile(0u, vn);
    for (unsigned tile = 0; tile < ntot; tile += T) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = vn[r];
        if (tile + T < ntot) load_tile(tile + T, vn);   // the next tile's records travel while this one is sorted
        if (!SEP) {
            for (int i = lo; i < hi; ++i) lcount[i] = 0u;
            __syncthreads();
        }
        // one returning atomic per record: its rank among the tile's records of the same bin (kept in a register)
        unsigned rank[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rank[r] = v[r].x != kTlNone ? atomicAdd(&lcount[v[r].x >> TL.shift], 1u) : 0u;
        __syncthreads();
        // exclusive scan of the counts: `per` bins per thread, inside the wave by shuffles, across the waves through LDS
        uint32_t sum = 0;
        for (int i = lo; i < hi; ++i) sum += lcount[i];
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t x = __shfl_up(incl, off, 64);
            if ((tid & 63u) >= (unsigned)off) incl += x;
        }
        if ((tid & 63u) == 63u) part[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (unsigned wv = 0; wv < (unsigned)NW; ++wv) { const uint32_t x = part[wv]; total += x; if (wv < (tid >> 6)) before += x; }
        uint32_t run = before + incl - sum;
        for (int i = lo; i < hi; ++i) {
            // lstart: where the bin starts in the sorted tile; ldelta: what to add to a position in the sorted tile to get the record's
            // place in TL.binned (modulo 2^32); the bin's cursor moves on by its count; the counter is ready for the next tile
            const uint32_t n = lcount[i], g = gcur[i];
            if (SEP) lcount[i] = 0u;
            lstart[i] = run; ldelta[i] = g - run; gcur[i] = g + n;
            run += n;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (v[r].x != kTlNone) sorted[lstart[v[r].x >> TL.shift] + rank[r]] = v[r];
        __syncthreads();
        // the sorted tile leaves, the records of a bin as one contiguous piece (reads of all R records of a thread in flight together)
        uint2 rr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { const unsigned i = (unsigned)(r * NT) + tid; rr[r] = i < total ? sorted[i] : make_uint2(kTlNone, 0u); }
        unsigned dl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dl[r] = rr[r].x != kTlNone ? ldelta[rr[r].x >> TL.shift] : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned i = (unsigned)(r * NT) + tid;
            if (rr[r].x != kTlNone) {
                const unsigned pos = dl[r] + i;
                if (pos < TL.bcap) { if (MI3D_TL_NT & 2) nt_store(TL.binned + pos, rr[r]); else TL.binned[pos] = rr[r]; }
                else if (rr[r].x < nflux) atomicAdd(&flux[rr[r].x], (tally_t)__uint_as_float(rr[r].y));   // (the sorted copy has run full: nothing is lost, k_tl_sum stops at bcap)
                else if (rr[r].x - nflux < nheat) atomicAdd(&heat[rr[r].x - nflux], (double)__uint_as_float(rr[r].y));
            }
        }
        // (without a table of their own the offsets sit where the next tile counts: a barrier before the owners zero them.  With one,
        //  the next tile's writes to ldelta, sorted and part come after its barriers, which every thread reaches after these reads)
        if (!SEP) __syncthreads();
    }
}

// The run records of a launch, expanded: workgroup `row` takes the run records the wave(s) of row `row` of the photon loop have written (G = 1: one
// wave, 4: the four waves of a workgroup, TallyList::hist_wg), a thread a run, 256 runs at a time, and every wave walks the levels its 64 runs cross:
// a lane whose run crosses level L works out the column -- the multiply-add and fold the photon loop's block B0 applies when it places the
// levels itself: same float32 operations, same cell -- and the record {cell, weight} goes straight to its bin of TL.binned.  No sort: the lanes
// of a wave are at the SAME level at the same time, so their records fall into a few bins (one per plane where a level is one bin), and lanes
// of one bin take consecutive places (one LDS add per wave and bin, the lanes' ranks by ballot / mbcnt).
//   WRITE = false, the count pass (before k_tl_wavescan): how many records of every bin the row's runs hold -> whist row nrow + row;
//   WRITE = true (after k_tl_prefix): the records, from bin_start + wbase of that row on.
struct RunGeom {
    const LayerRec *lay;      // [nz] (zlo: the height of level k)
    float ztoa, inv_dx, inv_dy, inv_nx, inv_ny;
    int nz, nx, ny;
    unsigned long long *diag;   // the handle's counter vector (-DMI3D_RUNS_DIAG: slots 18 ... 23 take the write pass's wave clocks by phase, its level steps and the lanes at work in them)
};
constexpr unsigned kRunClasses = 64;   // distinct (plane, first level, levels) among the runs of a tile that get a class of their own (the rest share one)
constexpr unsigned kRunEmpty = 0xffffffffu;
constexpr unsigned kRunPure = 24;      // runs of one class in a tile from which on the class is walked by waves of its own
constexpr unsigned kRunWork = kRunChunk / 64u + kRunClasses + 2u;   // pieces of at most 64 runs of one class a tile falls into, at most
// LDS of k_tl_runs in bytes: a tile of run records sorted by class, the bins' table, the level heights, the row's chunk list, the class tables, the pieces
__host__ __device__ inline size_t tl_runs_lds(int nbins, int nz) {
    return (size_t)2 * kRunChunk * sizeof(float4) + ((size_t)nbins + (size_t)nz + 1 + 2 * kTlIds + 3 * (kRunClasses + 1) + kRunWork + 4) * sizeof(uint32_t);
}
// NTR: threads per workgroup: 256, or 512 where the bins' table is large (5000 bins: 58 KB of LDS with the tile, two workgroups per CU -- eight waves
// of 512 threads then walk the pieces of a tile where four would, and the LDS round trips of a level step find other waves to run)
template <bool WRITE, int G, int NTR = 256>
__global__ void __launch_bounds__(NTR)
k_tl_runs(const TallyList TL, const RunGeom Gm, tally_t *__restrict__ flux) {
    extern __shared__ float4 lds_f4[];
    float4 *sA = lds_f4, *sB = lds_f4 + kRunChunk;                  // [kRunChunk] the tile's runs, class by class
    uint32_t *tbl = reinterpret_cast<uint32_t *>(sB + kRunChunk);   // [nbins] records counted (count pass) / the next place in TL.binned (write pass)
    float *zlev = reinterpret_cast<float *>(tbl + TL.nbins);        // [nz + 1] level heights
    uint32_t *cid = reinterpret_cast<uint32_t *>(zlev + Gm.nz + 1); // [kTlIds] this row's chunks of run records ...
    uint32_t *cfill = cid + kTlIds;                                 // [kTlIds] ... and how full each is
    uint32_t *hkey = cfill + kTlIds;                                // [kRunClasses + 1] the classes of the tile: (plane, first level, levels)
    uint32_t *hcnt = hkey + kRunClasses + 1;                        // [kRunClasses + 1] runs per class
    uint32_t *hstart = hcnt + kRunClasses + 1;                      // [kRunClasses + 1] where each class starts in sA / sB
    uint32_t *work = hstart + kRunClasses + 1;                      // [kRunWork] the tile in pieces of at most 64 runs of ONE class: start | count << 16 | pure << 31
    uint32_t *nwork = work + kRunWork;                              // [1]
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrow_d = TL.hist_wg ? TL.nwave / 4 : TL.nwave;
    const int row = blockIdx.x, w0 = row * G;
    if (w0 >= TL.nwave) return;
    unsigned nch[4] = {0u, 0u, 0u, 0u}, ntot = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) { nch[g] = w0 + g < TL.nwave ? TL.run_nchunk[w0 + g] : 0u; ntot += nch[g]; }
    auto chunk_at = [&](unsigned m) -> unsigned {
        int g = 0;
#pragma unroll
        for (int q = 0; q < G - 1; ++q) if (g == q && m >= nch[q]) { m -= nch[q]; g = q + 1; }
        return TL.run_chunks[(size_t)(w0 + g) * TL.run_wcap + m];
    };
    for (unsigned m = tid; m < ntot && m < kTlIds; m += (unsigned)NTR) { const unsigned c = chunk_at(m); cid[m] = c; cfill[m] = TL.run_fill[c]; }
    const size_t rbase = (size_t)(nrow_d + row) * TL.nbins;
    for (int i = tid; i < TL.nbins; i += NTR) tbl[i] = WRITE ? TL.bin_start[i] + TL.wbase[rbase + i] : 0u;
    for (int i = tid; i <= Gm.nz; i += NTR) zlev[i] = i < Gm.nz ? Gm.lay[i].zlo : Gm.ztoa;
    const unsigned nlev = (unsigned)(Gm.nz + 1), ncol = (unsigned)(Gm.nx * Gm.ny);
    const bool aligned = ncol == (1u << TL.shift);     // a level of a plane is exactly one bin: no column needed to count
    constexpr int RPT = (int)(kRunChunk / (unsigned)NTR);       // runs per thread and tile (a tile: one chunk)
#ifdef MI3D_RUNS_DIAG
    unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};
    long long tk = clock64();
#define RUNS_TICK(i) do { const long long t_ = clock64(); dg[i] += (unsigned long long)(t_ - tk); tk = t_; } while (0)
#else
#define RUNS_TICK(i) do { } while (0)
#endif
    __syncthreads();
    float4 A[RPT], B[RPT];
    unsigned fill_n = 0;
    auto load_tile = [&](unsigned tile) {   // (read once: non-temporal)
        unsigned c;
        if (tile < kTlIds) { c = cid[tile]; fill_n = cfill[tile]; } else { c = chunk_at(tile); fill_n = TL.run_fill[c]; }
        if (c >= TL.run_cap / kRunChunk || fill_n > kRunChunk) { c = 0u; fill_n = 0u; }     // (a chunk number outside the list: nothing is read)
        const float4 *src = TL.runs + (size_t)c * (2u * kRunChunk);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const unsigned j = (unsigned)r * (unsigned)NTR + tid;
            if (j < fill_n) { A[r] = nt_load(src + j); B[r] = nt_load(src + kRunChunk + j); }
        }
    };
    // the cell of run (Ar, Br) at level L: block B0 of the photon loop, operation for operation
    auto cell_of = [&](const float4 &Ar, const float4 &Br, const int jx0, const int jy0, const bool ipa, const unsigned pbase, const int L) -> unsigned {
        int jx = jx0, jy = jy0;
        if (!ipa) {
            const float sl = fabsf(zlev[L] - Ar.z);
            const float fx = floorf(fmaf(Br.x, sl, Ar.x) * Gm.inv_dx), fy = floorf(fmaf(Br.y, sl, Ar.y) * Gm.inv_dy);
            jx += (int)fx; jy += (int)fy;
            if ((unsigned)jx >= (unsigned)Gm.nx) jx = wrapi(jx, Gm.nx, Gm.inv_nx);
            if ((unsigned)jy >= (unsigned)Gm.ny) jy = wrapi(jy, Gm.ny, Gm.inv_ny);
        }
        return (pbase + (unsigned)L) * ncol + (unsigned)(jy * Gm.nx + jx);
    };
    if (ntot) load_tile(0u);
    for (unsigned tile = 0; tile < ntot; ++tile) {
        const unsigned fill = fill_n;
        // ---- the tile's runs into LDS class by class: a wave then takes 64 runs (or fewer) of ONE (plane, first level, levels) and walks their
        //      levels with every lane at work -- taken as they come, a wave holds flights up to the top, down to the surface and back up
        //      side by side and walks the union of their levels with one lane in nine busy
        if (tid <= kRunClasses) { hkey[tid] = kRunEmpty; hcnt[tid] = 0u; }
        __syncthreads();     // (... and the waves are through the tile before)
        RUNS_TICK(0);
        unsigned cls[RPT], rank[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const unsigned j = (unsigned)r * (unsigned)NTR + tid;
            cls[r] = kRunEmpty;
            if (j < fill) {
                const unsigned key = __float_as_uint(B[r].w) & 0x3fffffu;
                unsigned slot = (key * 0x9E3779B1u) >> 26;     // (six bits)
                unsigned cl = kRunClasses;                     // (no class of its own left: the shared one)
                for (int probe = 0; probe < 8; ++probe) {
                    const unsigned old = atomicCAS(&hkey[slot], kRunEmpty, key);
                    if (old == kRunEmpty || old == key) { cl = slot; break; }
                    slot = (slot + 1u) & (kRunClasses - 1u);
                }
                cls[r] = cl;
                rank[r] = atomicAdd(&hcnt[cl], 1u);
            }
        }
        __syncthreads();
        if (tid < 64u) {
            // one wave: where every class starts in the sorted tile -- the classes of kRunPure runs or more first, in pieces of at most 64 runs
            // of ONE class; the small ones (a collision half way through a run, a flight that starts inside one: a fifth of the runs in four
            // dozen classes per tile, each walked by a handful of lanes in the first build) pooled behind them with the runs that found no class
            const uint32_t v = hcnt[tid], vm = hcnt[kRunClasses];
            const bool big = v >= kRunPure;
            uint32_t ib = big ? v : 0u, is = big ? 0u : v, ne = big ? (v + 63u) >> 6 : 0u, einc = ne;
            const uint32_t vb = ib, vs = is;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t x = __shfl_up(ib, off, 64), y = __shfl_up(einc, off, 64), z = __shfl_up(is, off, 64);
                if (lane >= (unsigned)off) { ib += x; einc += y; is += z; }
            }
            const uint32_t totb = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63), tots = (uint32_t)__builtin_amdgcn_readlane((int)is, 63);
            const uint32_t st = big ? ib - vb : totb + (is - vs);
            hstart[tid] = st;
            for (uint32_t k = 0; k < ne; ++k) work[einc - ne + k] = (st + 64u * k) | (min(64u, v - 64u * k) << 16) | (1u << 31);
            if (tid == 63u) {
                hstart[kRunClasses] = totb + tots;
                const uint32_t pool = tots + vm, nm = (pool + 63u) >> 6;
                for (uint32_t k = 0; k < nm; ++k) work[einc + k] = (totb + 64u * k) | (min(64u, pool - 64u * k) << 16);
                nwork[0] = einc + nm;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (cls[r] != kRunEmpty) { const unsigned sl = hstart[cls[r]] + rank[r]; if (sl < kRunChunk) { sA[sl] = A[r]; sB[sl] = B[r]; } }
        if (tile + 1u < ntot) load_tile(tile + 1u);     // the next tile's runs travel while this one's levels are walked
        __syncthreads();
        RUNS_TICK(1);
        // ---- the pieces, a wave each
        const unsigned nw = nwork[0];
#ifdef MI3D_RUNS_DIAG
        if (WRITE && tid == 0u) {
            unsigned ncl = 0, nfull = 0, big = 0;
            for (unsigned i = 0; i < kRunClasses; ++i) { ncl += hcnt[i] ? 1u : 0u; big = max(big, hcnt[i]); }
            for (unsigned i = 0; i < nw; ++i) nfull += ((work[i] >> 16) & 0x7fffu) == 64u ? 1u : 0u;
            atomicAdd(&Gm.diag[21], (unsigned long long)nw); atomicAdd(&Gm.diag[16], (unsigned long long)ncl); atomicAdd(&Gm.diag[15], (unsigned long long)hcnt[kRunClasses]);
            atomicAdd(&Gm.diag[14], (unsigned long long)nfull); atomicAdd(&Gm.diag[17], 1ull); atomicAdd(&Gm.diag[13], (unsigned long long)big); atomicAdd(&Gm.diag[12], (unsigned long long)fill);
        }
#endif
        for (unsigned e = wave; e < nw; e += (unsigned)(NTR / 64)) {
            const unsigned wk = work[e];
            const unsigned start = wk & 0xffffu, cntp = (wk >> 16) & 0x7fffu;
            const bool pure = (wk >> 31) != 0u;
            const bool valid = lane < cntp;
            const float4 Ar = sA[start + (valid ? lane : 0u)], Br = sB[start + (valid ? lane : 0u)];
            const unsigned q = __float_as_uint(Br.w), cxy = __float_as_uint(Br.z);
            const bool ipa = (q & (1u << 22)) != 0u;
            const int jx0 = (int)(cxy & 0xffffu), jy0 = (int)(cxy >> 16);
            if (pure && aligned) {
                // one class, a level of a plane one bin: lane l reserves the places of level la + l for the whole piece (one LDS add per level,
                // none in the loop), the runs take consecutive places at every level: each store of the walk writes ONE contiguous piece
                const unsigned qu = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
                const int la = (int)(qu & 0x3ffu), n = (int)((qu >> 10) & 0x3ffu);
                const unsigned pbase = ((qu >> 20) & 3u) * nlev;
                for (int done = 0; done < n; done += 64) {
                    const int nl = min(64, n - done);
                    unsigned basev = 0u;
                    if ((int)lane < nl) { if (WRITE) basev = atomicAdd(&tbl[pbase + (unsigned)(la + done) + lane], cntp); else atomicAdd(&tbl[pbase + (unsigned)(la + done) + lane], cntp); }
                    if (WRITE) {
                        for (int l = 0; l < nl; ++l) {
                            const int L = la + done + l;
                            const unsigned pos = (unsigned)__builtin_amdgcn_readlane((int)basev, l) + lane;
#ifdef MI3D_RUNS_DIAG
                            if (lane == 0u) dg[4]++;
                            if (valid) dg[5]++;
#endif
                            if (valid) {
                                const unsigned idx = cell_of(Ar, Br, jx0, jy0, ipa, pbase, L);
                                if (pos < TL.bcap) TL.binned[pos] = make_uint2(idx, __float_as_uint(Ar.w));
                                else atomicAdd(&flux[idx], (tally_t)Ar.w);     // (the sorted copy has run full: nothing is lost)
                            }
                        }
                    }
                }
            } else if (pure) {
                // one class on a grid whose bins cut across the levels (strips of rows of a level): every lane at work at every level, each finds its own place
                const unsigned qu = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
                const int la = (int)(qu & 0x3ffu), n = (int)((qu >> 10) & 0x3ffu);
                const unsigned pbase = ((qu >> 20) & 3u) * nlev;
                // (four levels at a time: four cells worked out, four places asked for, four records written -- the LDS round trip of a place is
                //  a hundred cycles, and a workgroup of this kernel has few waves to hide it behind)
                for (int L0 = la; L0 < la + n; L0 += 4) {
                    unsigned idx4[4], pos4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int L = min(L0 + j, la + n - 1);
#ifdef MI3D_RUNS_DIAG
                        if (L0 + j < la + n) { if (lane == 0u) dg[4]++; if (valid) dg[5]++; }
#endif
                        idx4[j] = valid ? cell_of(Ar, Br, jx0, jy0, ipa, pbase, L) : 0u;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool onj = valid && L0 + j < la + n;
                        pos4[j] = 0u;
                        if (onj) { if (WRITE) pos4[j] = atomicAdd(&tbl[idx4[j] >> TL.shift], 1u); else atomicAdd(&tbl[idx4[j] >> TL.shift], 1u); }
                    }
                    if (WRITE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (valid && L0 + j < la + n) {
                                if (pos4[j] < TL.bcap) TL.binned[pos4[j]] = make_uint2(idx4[j], __float_as_uint(Ar.w));
                                else atomicAdd(&flux[idx4[j]], (tally_t)Ar.w);
                            }
                        }
                    }
                }
            } else {
                // mixed runs: the piece's (run, level) pairs counted through, 64 at a time -- lane e finds the run its pair belongs to (the runs'
                // level counts summed up along the wave, a search of six steps), reads it from the sorted tile and finds its own place
                const unsigned nrun = valid ? (q >> 10) & 0x3ffu : 0u;
                unsigned incl = nrun;
                for (int off = 1; off < 64; off <<= 1) { const unsigned x = __shfl_up(incl, off, 64); if (lane >= (unsigned)off) incl += x; }
                const unsigned T = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                for (unsigned e0 = 0; e0 < T; e0 += 64u) {
                    const unsigned e = e0 + lane;
                    unsigned r = 0u;     // the first run whose inclusive sum exceeds e
#pragma unroll
                    for (unsigned sp = 32u; sp > 0u; sp >>= 1) { const unsigned val = (unsigned)__shfl((int)incl, (int)(r + sp - 1u), 64); if (val <= e) r += sp; }
                    const bool on = e < T;
                    // (with every lane at work: a lane that is switched off hands out 0 through the crossbar -- the first build of this search
                    //  read the sum under `on`, placed the last pairs of a piece at levels far outside the table and faulted)
                    const unsigned inc_r = (unsigned)__shfl((int)incl, (int)r, 64);
#ifdef MI3D_RUNS_DIAG
                    if (lane == 0u) dg[4]++;
                    if (on) dg[5]++;
#endif
                    if (on) {
                        const float4 A2 = sA[start + r], B2 = sB[start + r];
                        const unsigned q2 = __float_as_uint(B2.w), cxy2 = __float_as_uint(B2.z);
                        const unsigned n2 = (q2 >> 10) & 0x3ffu;
                        const int L = (int)(q2 & 0x3ffu) + (int)(e - (inc_r - n2));
                        const unsigned pb2 = ((q2 >> 20) & 3u) * nlev;
                        unsigned idx = 0u, bin = pb2 + (unsigned)L;
                        if ((unsigned)L >= nlev) continue;     // (cannot happen: a level outside the grid would be a cell outside the tally)
                        if (WRITE || !aligned) { idx = cell_of(A2, B2, (int)(cxy2 & 0xffffu), (int)(cxy2 >> 16), (q2 & (1u << 22)) != 0u, pb2, L); bin = idx >> TL.shift; }
                        if (WRITE) {
                            const unsigned pos = atomicAdd(&tbl[bin], 1u);
                            if (pos < TL.bcap) TL.binned[pos] = make_uint2(idx, __float_as_uint(A2.w));
                            else atomicAdd(&flux[idx], (tally_t)A2.w);
                        } else atomicAdd(&tbl[bin], 1u);
                    }
                }
            }
        }
        RUNS_TICK(2);
    }
#ifdef MI3D_RUNS_DIAG
    if (WRITE) {
        if (lane == 0u) { atomicAdd(&Gm.diag[18], dg[0] >> 6); atomicAdd(&Gm.diag[19], dg[1] >> 6); atomicAdd(&Gm.diag[20], dg[2] >> 6); atomicAdd(&Gm.diag[22], dg[4]); }
        unsigned long long v = dg[5];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0u) atomicAdd(&Gm.diag[23], v);
    }
#endif
#undef RUNS_TICK
    if (!WRITE) {
        __syncthreads();
        for (int i = tid; i < TL.nbins; i += NTR) TL.whist[rbase + i] = tbl[i];
    }
}

// One bin of 2^shift consecutive tally cells summed in LDS (float64) by `split` workgroups, each over its share of the bin's
// records; what a workgroup has gathered goes to the tally with one atomic per cell it touched.
__global__ void __launch_bounds__(1024)
k_tl_sum(const TallyList TL, tally_t *__restrict__ flux, const unsigned nflux, double *__restrict__ heat, const unsigned nheat, const int split) {
    extern __shared__ double lacc[];
    const int bin = blockIdx.x / split, part = blockIdx.x % split;
    const unsigned lo = TL.bin_start[bin], n = TL.bin_start[bin + 1] - lo;
    unsigned a = lo + (unsigned)(((unsigned long long)n * part) / split), b = lo + (unsigned)(((unsigned long long)n * (part + 1)) / split);
    a = min(a, TL.bcap); b = min(b, TL.bcap);     // (records whose place lay beyond the sorted copy went to the tally as atomics)
    if (a == b) return;
    const unsigned ncell = 1u << TL.shift, mask = ncell - 1u;
    for (unsigned i = threadIdx.x; i < ncell; i += blockDim.x) lacc[i] = 0.0;
    __syncthreads();
    const unsigned nt = blockDim.x;
    unsigned i = a + threadIdx.x;
    for (; i + 3u * nt < b; i += 4u * nt) {   // four records per lane in flight
        const uint2 v0 = (MI3D_TL_NT & 4) ? nt_load(TL.binned + i) : TL.binned[i], v1 = (MI3D_TL_NT & 4) ? nt_load(TL.binned + i + nt) : TL.binned[i + nt],
                    v2 = (MI3D_TL_NT & 4) ? nt_load(TL.binned + i + 2u * nt) : TL.binned[i + 2u * nt], v3 = (MI3D_TL_NT & 4) ? nt_load(TL.binned + i + 3u * nt) : TL.binned[i + 3u * nt];
        atomicAdd(&lacc[v0.x & mask], (double)__uint_as_float(v0.y));
        atomicAdd(&lacc[v1.x & mask], (double)__uint_as_float(v1.y));
        atomicAdd(&lacc[v2.x & mask], (double)__uint_as_float(v2.y));
        atomicAdd(&lacc[v3.x & mask], (double)__uint_as_float(v3.y));
    }
    for (; i < b; i += nt) {
        const uint2 v = TL.binned[i];
        atomicAdd(&lacc[v.x & mask], (double)__uint_as_float(v.y));
    }
    __syncthreads();
    const unsigned base = (unsigned)bin << TL.shift;
    for (unsigned c = threadIdx.x; c < ncell; c += blockDim.x) {
        const double v = lacc[c];
        if (v != 0.0) {
            if (base + c < nflux) atomicAdd(&flux[base + c], v);
            else if (base + c - nflux < nheat) atomicAdd(&heat[base + c - nflux], v);
        }
    }
}

#define MI3D_FLUX_INST(C, P) template __global__ void k_transport_flux<C, P, 0>(const DevScene, const TallyList *, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_flux<C, P, 1>(const DevScene, const TallyList *, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_flux<C, P, 2>(const DevScene, const TallyList *, const uint64_t, const uint64_t, const uint64_t);
MI3D_FLUX_INST(false, false) MI3D_FLUX_INST(false, true) MI3D_FLUX_INST(true, false) MI3D_FLUX_INST(true, true)
#undef MI3D_FLUX_INST

} // namespace mi3d
