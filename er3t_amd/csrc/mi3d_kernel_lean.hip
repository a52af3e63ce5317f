// mi3d_kernel_lean.hip — the lean build of the photon loop for the configurations er3t runs by default
// (er3t/rtm/mca/mcarats.py:76-77,285-307: target='radiance' from a satellite; mca_atm.py:95-102,318-337: Rayleigh as the one
// 1-D constituent, the cloud as the one 3-D constituent): radiance only, satellite views (Rad_mrkind = 2) -- exactly vertical
// ones from above the atmosphere answered from the column table, all others through event records that k_rays marches
// (mi3d_kernel_rays.hip) -- one 1-D and at most two 3-D constituents (cloud, aerosol), analytic phase functions (isotropic /
// Rayleigh / Henyey-Greenstein; the Mie branch of mca_atm.py:299-303 stores the asymmetry parameter, so it is Henyey-Greenstein
// too), any surface model, any solver.  Flux jobs have a lean loop of their own (mi3d_kernel_flux.hip); cameras use this loop's
// event-writing build and the ray kernel's camera build; everything else (several constituents, flux together with radiance) runs
// through k_transport (mi3d_kernels.hip).  (Round 3's build with the rays walked inside this loop, k_transport_leanloop, was retired in
// round 5: a second copy of the walk and of every block, kept as a fall-back that the general kernel provides as well.)  Same random-number protocol, same estimator, same sampling formulas: photon id ->
// history is the function DESIGN.md §3 specifies, whichever kernel serves the launch (tests hold all of them against the oracle).
//
// The loop (round 4).  A lane owns a photon.  Three kinds of work take turns in a wave:
//   A  the voxel walk, an incremental DDA: a photon keeps the parameters tx, ty, tz at which it meets the next x, y and z face;
//      a step is min3, one multiply with the extinction read a step ago, one compare against the optical depth the photon has
//      left and ONE of the three parameters moved on -- the x and y candidates computed for all lanes and selected, only the level
//      crossing (a read of the layer table in LDS) under a branch.  No position is carried: where a photon is inside its voxel
//      follows from how far its parameter is from the three faces ahead.  Repeated while enough lanes are walking.
//   C  a collision the walk has found -- 21 of a photon's 27 stops on the 480 x 480 x 100 bench scene -- from the end of the walk
//      to the start of the next one in ONE straight block: position, weight, the local estimates answered from the column table,
//      (an event record for k_rays,) scattering constituent and angle, new direction, next Philox block, free path, the face
//      parameters of the new walk.  The voxel has not changed: the record the walk read last is the record of the event AND the
//      first record of the next walk.  Every pass.
//   R  everything rarer -- runs of horizontally uniform layers and collisions inside them, the surface, roulette, the end of a
//      history and the next photon -- in the shared blocks B0 ... B7 of rounds 1-3, every MI3D_LEAN_FAST_PASS-th pass only (a pass
//      with no collision pending is always such a full pass).  Ordered so that a history that ends in a full pass is replaced in
//      the same pass, and new photons arrive as ENTRY RECORDS (k_entry below: launch, solar-cone jitter, first free path and the
//      uniform layers above the clouds worked out in a kernel of its own where every lane has a photon): B4 reads 48 bytes and B7
//      sets up the first walk.
// A lane's mode says what it waits for; there is no other bookkeeping between the blocks.
#include "mi3d_device.h"
#include "mi3d_diag.h"

namespace mi3d {

#ifndef MI3D_LEAN_THRESH
#define MI3D_LEAN_THRESH 16   // phase A keeps stepping while at least this many lanes of the wave are walking
#endif
#ifndef MI3D_LEAN_PASS
#define MI3D_LEAN_PASS 2      // (mi3d_kernel_flux.hip: every second pass of phase B is a full one)
#endif
#ifndef MI3D_LEAN_EMIT4
#define MI3D_LEAN_EMIT4 1     // 1: the build that writes event records gets the register budget of four waves per SIMD
#endif
#ifndef MI3D_LEAN_EMIT_GRID
#define MI3D_LEAN_EMIT_GRID 5 // workgroups per CU of the build that writes event records: compiled for four waves per SIMD it takes 96 registers
                              // without a spill, which lets five be resident (compiled for five the allocator spills four dwords)
#endif
#ifndef MI3D_LEAN_FAST
#define MI3D_LEAN_FAST 1      // (entry records exist: mi3d_api.hip asks)
#endif
#ifndef MI3D_LEAN_THRESH_EMIT
#define MI3D_LEAN_THRESH_EMIT 24   // ... in the build that writes event records (five waves per SIMD, the walk a smaller share of its passes):
                                   // 16 / 24 / 32 / 40: 3.06 / 3.12 / 3.06 / 3.10e8 photons/s with nine views (ab_mv9_thresh_cadence.log)
#endif
#ifndef MI3D_LEAN_FAST_PASS
#define MI3D_LEAN_FAST_PASS 8 // every n-th pass of phase B is a full one: 4 / 6 / 8 / 12 -> 2.14 / 2.17 / 2.18 / 2.17e9 photons/s at five waves per SIMD
                              // (profiles/r04/ab_lean_modes_cadence.log), 6 / 8 / 12 -> 2.22 / 2.26 / 2.22e9 at six (ab_lean_waves.log)
#endif
#ifndef MI3D_LEAN_RARE_T
#define MI3D_LEAN_RARE_T 0    // > 0: a pass is also a full one when at least this many lanes wait for the rarer kinds of work (8: -3 %)
#endif
#ifndef MI3D_LEAN_WIN_EMIT
#define MI3D_LEAN_WIN_EMIT 0   // 1: the tally window also in the build that writes event records (105 registers: four waves per SIMD)
#endif
#ifndef MI3D_EV_NT_STORE
#define MI3D_EV_NT_STORE 1   // 1: the event records leave through non-temporal stores (0 / 1: 3.37 / 3.68e8 photons/s with nine views, profiles/r05/ab_nt_event_records.log)
#endif
#ifndef MI3D_LEAN_PEND
#define MI3D_LEAN_PEND 1   // 1: consecutive tallies of one history into the same pixel are summed in a register before they leave
#endif
// The tally window (DevCold::tile_end ...): kWin x kWin float sums and a few control words per workgroup, in LDS behind the tables.
//   ctl[0] origin of the window the waves may add into (x | y << 16; kWinNone: none, tallies are atomics on the image)
//   ctl[1] origin the sums in the window belong to       ctl[2] the wave that is moving the window (its number + 1; 0: nobody)
//   ctl[3] waves of the workgroup that have left          ctl[4], ctl[5] the window's tile: its piece of the photon order
//   ctl[6] the place in the order the mover wants the window for
//   ctl[8 .. 8 + NW) passes each of the workgroup's NW waves has begun (kWinNone: it has left)
//   ctl[8 + NW .. 8 + 2 NW) what the mover saw there when it closed the window      (NW <= 8: workgroups of 256 or 512 threads)
constexpr unsigned kWinNone = 0xffffffffu;
constexpr int kWinCtl = 24;
constexpr size_t kWinLds = (size_t)kWin * kWin * sizeof(float) + kWinCtl * sizeof(unsigned);
#ifndef MI3D_LEAN_WAVES
#define MI3D_LEAN_WAVES(COUNT, MARCH) (((COUNT) || (MARCH)) ? 4 : 6)   // waves per SIMD the register budget must allow: 80 registers hold the
                              // column-view build without a spill (5 / 6 / 7 / 8 waves: 2.18 / 2.26 / 1.57 / 0.86e9 photons/s -- 7 and 8 spill; ab_lean_waves.log)
#endif

// k_transport_lean<.,.,2>: the events of this pass go to this XCD's list for k_rays; the photons carry on at once.  A wave reserves
// room for kEvBlock records at a time (one returning atomic per block instead of one per pass: the wave waits for it) and hands the
// slots out itself; what it leaves unused is marked empty (weight 0) before it reserves again or ends.  Wave-level: to be called
// where the whole wave passes (ev_lo, ev_hi are wave-uniform).
__device__ __forceinline__ void emit_events(const DevCold *cold, const unsigned xcc, bool &emit, unsigned long long &ev_lo, unsigned long long &ev_hi,
                                            const float px, const float py, const float pz, const float w, const float ux, const float uy, const float uz,
                                            const float ev_ks0, const float ev_apf0, const float ev_sfc, const int ix, const int iy, const int k, const int kind,
                                            const uint64_t seed, const uint64_t id, const uint32_t draw) {
    const unsigned long long em = __ballot(emit);
    if (em != 0ull) {
        const unsigned n = (unsigned)__popcll(em);
        if (ev_lo + n > ev_hi) {
            for (unsigned long long q = ev_lo + (threadIdx.x & 63); q < ev_hi; q += 64)
                if (q < (unsigned long long)cold->ev_cap) cold->ev_list[ev_list_f4(cold->ev_cap) * xcc + ev_index((unsigned)q)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const int leader = __ffsll((long long)em) - 1;
            unsigned long long base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(cold->ev_ctr + xcc * kCtrStride, (unsigned long long)kEvBlock);
            // (through scalar registers: the reservation is the same in every lane, and as per-lane values ev_lo / ev_hi cost four registers)
            base = ((unsigned long long)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) | (unsigned)__builtin_amdgcn_readlane((int)base, leader);
            ev_lo = base; ev_hi = base + kEvBlock;
        }
        if (emit) {
            const unsigned long long slot = ev_lo + __builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, 0u));
            if (slot < (unsigned long long)cold->ev_cap) {
                // (plain stores: write-through ones that bypass the XCD's L2, `sc1`, were 10 % slower -- the four 16-byte
                //  pieces of a record then leave one by one, profiles/r02/mv9_event_stores.log)
                float4 *lbase = cold->ev_list + ev_list_f4(cold->ev_cap) * xcc;       // (this XCD's list: wave-uniform)
                float4 *e = lbase + ev_index((unsigned)slot);
#if MI3D_DIAG_NOEMITSTORE   // (mi3d_diag.h: ablation build, results wrong)
                MI3D_DIAG_KEEP11(px, py, pz, w, ux, uy, uz, ev_ks0, ev_apf0, ev_sfc, e);
#elif MI3D_EV_NT_STORE
                // (written once, read once by another kernel: non-temporal stores, so that 1.1 KB of records per photon do not push the
                //  voxel records out of the XCD's L2)
                typedef float vf4 __attribute__((ext_vector_type(4)));
                vf4 *en = reinterpret_cast<vf4 *>(e);
                __builtin_nontemporal_store((vf4){px, py, pz, w}, en);
                __builtin_nontemporal_store((vf4){ux, uy, uz, ev_ks0}, en + kEvStride);
                __builtin_nontemporal_store((vf4){ev_apf0, ev_sfc, __int_as_float(ix | (iy << 16)), __int_as_float(k | (kind << 16))}, en + 2 * kEvStride);
                __builtin_nontemporal_store(le_hash_base(seed, id, draw), reinterpret_cast<uint32_t *>(lbase) + ev_word((unsigned)slot));
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

// MARCH: 0 every view is answered from the column table; 2 the other views are marched by k_rays: this kernel writes an event record
//        for every collision and reflection (k_rays' header).
// MIX: 0 one 1-D and one 3-D constituent with analytic phase functions (er3t's default scene);
//      1 the voxels carry a second 3-D constituent (er3t's cloud + aerosol scenes); a build of its own because even wave-uniform
//        branches around it cost the one-constituent bench 0.8 % (profiles/r02/ab_second_constituent_cost.log);
//      2 (round 5) the general mixture: up to MI3D_MAX_NP1D 1-D constituents, up to two 3-D ones, TABULATED phase functions
//        (selector >= 1: er3t's Mie tables, mca_sca.py:76-91, chosen per layer or per voxel as func_ref_vs_cot does, rtm/mca/util.py:153) --
//        the tables the scene refers to staged in LDS behind the tally window, the node of a look-up found through a bucket index
//        (mi3d_device.h: lean_table_eval, lean_table_sample)
// NT:  threads per workgroup.  512 where the staged tables and the tally window together leave room for three workgroups per CU only:
//      three of 512 threads keep six waves per SIMD
// Lane modes beyond those of k_transport (mi3d_kernels.hip):
constexpr int M_COLLU = 12;   // a collision inside a run of uniform layers, found by B0 (M_COLL here: found by the voxel walk -> block C)
constexpr int M_UNIFW = 13;   // the walk has crossed a level into a uniform layer: position to be worked out, then M_UNIF
constexpr int M_SETUP = 14;   // about to walk voxels: B7 sets up the face parameters and asks for the first record
constexpr int M_DRAWR = 15;   // needs a Philox block for its roulette (M_DRAW here: for its next flight)
constexpr int M_DRAWL = 16;   // ... for its launch (no entry records)

template <bool COUNT, bool P3D, int MARCH, int MIX, int NT = 256>
#ifndef MI3D_LEAN_REG_WAVES
#define MI3D_LEAN_REG_WAVES(COUNT, MARCH) MI3D_LEAN_WAVES(COUNT, MARCH)
#endif
// (the general mixture: the register budget of six waves per SIMD, 80 registers, which it holds with three spilled values, in workgroups
//  of 256 threads -- where its LDS leaves room for six -- and of 512)
#ifndef MI3D_GEN_NARROW_WAVES
#define MI3D_GEN_NARROW_WAVES 6   // (5 / 6: 2.61 / 2.90e9 photons/s for the general-mixture build on the table-free les128: 80 registers, three spilled values)
#endif
__global__ void __launch_bounds__(NT, (MIX >= 2 && NT == 256 && MARCH == 0 && !COUNT) ? MI3D_GEN_NARROW_WAVES : MI3D_LEAN_REG_WAVES(COUNT, MARCH == 2 && MI3D_LEAN_EMIT4))
k_transport_lean(const DevScene S, const uint64_t nphoton, const uint64_t seed, const uint64_t offset) {
    static_assert(MARCH == 0 || MARCH == 2, "marched views go through event records and k_rays");
    static_assert(MIX >= 0 && MIX <= 3 && (NT == 256 || NT == 512), "builds");
    // MIX 3 (round 6): the general mixture's COMMON scene known at compile time -- ONE 1-D constituent, Rayleigh (er3t's mca_atm_1d), and one
    // 3-D constituent whose selectors may name tables (the Mie branch as mca_atm.py:275-277 would write it): no loop over constituents, no second
    // voxel constituent, no run-time flags for either in the registers (+3 % on les128_mie over the same code with the flags read at run time)
    constexpr bool MIXED = (MARCH != 0), EMIT = (MARCH == 2), TWO = (MIX == 1), GEN = (MIX >= 2), RAY1 = (MIX == 3);
    constexpr unsigned NW = NT / 64;
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
    float *wbuf = reinterpret_cast<float *>(smem + o_view + MI3D_MAX_VIEW * 2 + kColdF4);
    // (the control words are read and written as workgroup-scope atomics, sequentially consistent: LDS instructions, in program order --
    //  through a volatile pointer the compiler loses the address space and every access becomes a FLAT instruction on a 64-bit address)
    unsigned *wctl = reinterpret_cast<unsigned *>(wbuf + kWin * kWin);
#define WLD(i_) __hip_atomic_load(wctl + (i_), __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP)
#define WST(i_, v_) __hip_atomic_store(wctl + (i_), (unsigned)(v_), __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP)
    // (not in the build that writes event records: it has no registers to spare -- 93 hold five waves per SIMD, 105 would hold four --
    //  and its column view is one view in nine)
    const bool win_on = (!EMIT || MI3D_LEAN_WIN_EMIT) && S.cold->tile_end != nullptr;   // (the launch has given the kernel the LDS for it: kWinLds)
    if (win_on) {
        for (int i = threadIdx.x; i < kWin * kWin; i += blockDim.x) wbuf[i] = 0.0f;
        if (threadIdx.x < kWinCtl) WST(threadIdx.x, (threadIdx.x < 2) ? kWinNone : 0u);
    }
    // GEN: the phase tables the scene refers to, staged behind the window (DevCold::tab_n > 0: the launch has given the kernel the LDS)
    const float *ltab = nullptr;
    if (GEN && S.cold->tab_n > 0) {
        float *dst = win_on ? reinterpret_cast<float *>(wctl + kWinCtl) : wbuf;      // (no window: the tables stand where it would)
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
        if (threadIdx.x < kColdF4) smem[o_view + MI3D_MAX_VIEW * 2 + threadIdx.x] = csrc[threadIdx.x];
    }
    __syncthreads();

    const bool ipa_all = (S.solver == MI3D_SOLVER_IPA);
    const bool same_grid = (S.nxr == S.nx) && (S.nyr == S.ny);
    const bool plain = !GEN && (S.target & kTargetPlainPhase) != 0;   // Rayleigh + Henyey-Greenstein: no selector is looked at
    const LeanTab T = GEN ? lean_tab(cold, ltab) : LeanTab{};
    const int np1d = (GEN && !RAY1) ? S.np1d : 1;
    const bool two3 = TWO || (GEN && !RAY1 && S.np3d > 1);     // the voxels carry a second 3-D constituent
    // GEN, the common case of a scene with tables: ONE 1-D constituent, Rayleigh (er3t's mca_atm_1d) -- its share of the mixture then costs
    // what it costs the plain build (no selector looked at, no loop over constituents); the tables are the cloud's, voxel by voxel
    const bool ray1 = RAY1 || (GEN && (S.target & kTargetRayleigh1d) != 0);
#define IPA_NOW() (ipa_all || (P3D && !direct))
    Counters cnt = {};
    // byte offsets into the voxel records: record of (ix, iy, k) at vbase + iy*sy_b + ix*sx_b + k*16
    const unsigned sx_b = S.vcol_f4 * 16u, sy_b = S.vrow_f4 * 16u;
    const char *vbase = reinterpret_cast<const char *>(S.vrec) - (long)S.k3lo * 16;
#define VREC(ix_, iy_, k_) (*reinterpret_cast<const float4 *>(vbase + ((unsigned)(iy_) * sy_b + (unsigned)(ix_) * sx_b + (unsigned)(k_) * 16u)))

    // ---- lane state
    // Outside its walk (px, py, pz) is the photon's position inside the voxel (ix, iy, k).  During a walk no position is carried at
    // all: where a photon is inside its voxel follows from how far its parameter is from the three faces ahead, (tx - t) |ux| from
    // the x face and so on, and is worked out when the walk ends.
    float px = 0, py = 0, pz = 0, ux = 0, uy = 0, uz = 1, iux = 1, iuy = 1, iuz = 1;
    float t = 0, tx = 0, ty = 0, tz = 0;   // ray parameter now / at the next x, y, z face
    int ix = 0, iy = 0, k = 0, stepx = 0, stepy = 0; // stepx/y: column step per crossing (0 under IPA)
    int wrapx = 0, wrapy = 0;                         // the column a step across the domain's edge leads to
    float rem = 0.0f;   // optical depth left to the photon's collision
    float w = 0.0f;
    float u1 = 0, u2 = 0, u3 = 0;
    uint32_t lid = 0;   // the photon's place in the launch's id range: its id is offset + lid (32 bits carried, 64 formed where a Philox block is drawn)
#define PHOTON_ID() (offset + (uint64_t)lid)
    uint32_t draw = 0;
    int mode = M_NEED, kind = E_LAUNCH;
    bool direct = false;
    unsigned long long pool_next = 0, pool_end = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID (speed only)
    unsigned victim = 0;
    unsigned nphot_wave = 0;   // wave-uniform: histories this wave has ended
    int pend_pix = -1;     // (ir | jr << 16 of the pending tally's pixel, -1: none)
    float pend_val = 0.0f;
    // ---- the tally window.  The chip does 2.4-2.7e10 float64 atomics a second when it does nothing else (profiles/r04/atomic_rates.log)
    // and this loop, at ten tallies per photon, wanted 2.3e10: without its tallies it ran 36 % faster (ab_no_tally_ablation.log).  Most
    // tallies of a workgroup fall near the tile of columns its photons started above (the launch is worked through tile by tile),
    // so the workgroup sums them in LDS -- kWin x kWin pixels of the column view around where the direct beam from the tile meets the
    // clouds, float32, ds_add_f32 -- and adds the sums to the image when its photons come from the next tile.  Tallies outside the
    // window, and all of them while a window is being moved, are atomics on the image as before.
    // Moving needs no barrier.  Every wave counts its passes in LDS and reads the window's origin ONCE, when a pass begins.  The wave
    // that finds its new photons outside the window's tile closes the window (origin: none), notes the others' pass counts, and
    // carries on; when each of the others has begun another pass (or left) nobody can still hold the old origin: it then adds
    // the sums to the image, clears them and opens the window over the new tile.  The last wave to leave empties the window.
    const unsigned wave_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (a scalar: the addresses made of it are scalar arithmetic)
    unsigned my_pass = 0;      // wave-uniform
    static_assert(kWin == 64, "the window's place arithmetic");
    unsigned worg = 0x80008000u, worg2 = 0x80008000u;  // wave-uniform: the window's origin as read when this pass began (x | y << 16), and the
                                                       // same less the image's size (mod 2^16); no window: a pixel no image of < 32768 has
    const int jv_col = MIXED ? S.col0 : 0;       // (first column view: the one the window serves)
    // (both pixel coordinates at once, as a pair of 16-bit numbers: pixel - origin, and pixel - (origin - image size) for the part of a
    //  window that lies across the image's cyclic edge; the smaller of the two is the place inside the window if there is one)
    typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
#define RAD_TALLY(key_, val_)                                                                                                    \
    do {                                                                                                                         \
        MI3D_WIN_ANY(COUNT, cnt);                                                                                                    \
        const us2_t k2_ = __builtin_bit_cast(us2_t, (unsigned)(key_));                                                           \
        const us2_t d2_ = __builtin_elementwise_min(k2_ - __builtin_bit_cast(us2_t, worg), k2_ - __builtin_bit_cast(us2_t, worg2)); \
        const unsigned d_ = __builtin_bit_cast(unsigned, d2_);                                                                   \
        if ((d_ & ~((unsigned)(kWin - 1) * 0x10001u)) == 0u) {                                                                   \
            atomicAdd(&wbuf[(d_ >> (16 - 6)) | (d_ & (unsigned)(kWin - 1))], (val_));                                            \
            MI3D_WIN_HIT(COUNT, cnt);                                                                                                      \
        } else {                                                                                                                 \
            const int ir_ = (key_) & 0xffff, jr_ = (int)((unsigned)(key_) >> 16);                                                \
            RAD_ADD(&S.rad[(unsigned)((jv_col * S.nyr + jr_) * S.rad_row + ir_) * (unsigned)S.rad_stride], (val_));              \
        }                                                                                                                        \
    } while (0)
    // the window's sums go to the image (whole wave; nobody adds to them meanwhile)
    auto win_flush = [&]() {
        const unsigned bo = WLD(1);
        if (bo == kWinNone) return;
        const int bx = (int)(bo & 0xffffu), by = (int)(bo >> 16);
        for (int i = threadIdx.x & 63; i < kWin * kWin; i += 64) {
            const float v = __hip_atomic_exchange(&wbuf[i], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v != 0.0f) {
                int ir = bx + (i & (kWin - 1)), jr = by + i / kWin;
                ir -= ir >= S.nxr ? S.nxr : 0; jr -= jr >= S.nyr ? S.nyr : 0;
                RAD_ADD(&S.rad[(unsigned)((jv_col * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride], v);
            }
        }
    };
    // the voxel record the walk read last: {total extinction, optical depth above the voxel, omega*ext and apf of the first 3-D
    // constituent}.  A lane that stops walking keeps it: it IS the record of the voxel its event lies in.
    float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float &bt_ev = rec.x, &ev_tab = rec.y, &ev_ks0 = rec.z, &ev_apf0 = rec.w;
    float ev_ksb = 0.0f, ev_apfb = 0.0f;   // the second 3-D constituent of the event's voxel (np3d = 2), else 0
    float &ev_sfc = ev_tab;
    bool emit = false;   // EMIT: this lane's event of the current pass is to be written to the event list
    unsigned long long ev_lo = 0, ev_hi = 0;   // EMIT, wave-uniform: slots of this XCD's list reserved by this wave and not yet used

#define MI3D_TICK(slot) MI3D_DIAG_TICK(COUNT, cnt, tick, slot)     // (mi3d_diag.h: instrumented build only)
    long long tick = COUNT ? clock64() : 0; (void)tick;   // instrumented build: wave clock ticks / 64 spent in A, walk end + B0, C + B2, B4, B5, B6 + B7
    unsigned pass_ctr = 0;
    for (;;) {
        if (win_on) {   // (a pass begins: counted where the others see it, THEN the window's origin read -- in this order)
            my_pass++;
            // (relaxed atomics between compiler barriers: two LDS instructions in program order, which the LDS serves in that order --
            //  a sequentially consistent pair would also wait for every load of the voxel walk that is still on its way)
            asm volatile("" ::: "memory");
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(wctl + 8 + wave_w, my_pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            const unsigned o_ = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(wctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            asm volatile("" ::: "memory");
            worg = o_ == kWinNone ? 0x80008000u : o_;
            worg2 = o_ == kWinNone ? 0x80008000u : (((o_ & 0xffffu) - (unsigned)S.nxr) & 0xffffu) | (((o_ >> 16) - (unsigned)S.nyr) << 16);
        }
        // =================================== phase A: voxel steps ===================================
        // The record of the cell a photon is in arrived a step ago (or with the event that started the walk): the step multiplies its
        // extinction with the way to the nearest face, ends the walk where the collision lies inside the cell, and otherwise moves to
        // the next cell and asks for ITS record before it loops back.
        MI3D_MARK("A");
        for (;;) {
            const bool flying = (mode == M_FLY);
            const int nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (nfly < (EMIT ? MI3D_LEAN_THRESH_EMIT : MI3D_LEAN_THRESH) && __ballot(mode != M_FLY && mode != M_DONE) != 0ull) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (flying) {
                const float tn = fminf(fminf(tx, ty), tz);
                const float dtau = rec.x * (tn - t);
                if (COUNT) { cnt.steps++; cnt.steps3d++; }
                MI3D_DIAG_CLEAR_STEP(COUNT, cnt, rec.z, dtau >= rem);
                if (dtau >= rem) mode = M_COLL;     // the collision lies inside this voxel: at t + rem / bt (block C)
                else {
                    rem -= dtau;
                    t = tn;
                    const bool zf = (tz == tn), xf = !zf && (tx == tn), yf = !zf && !xf;
                    // the x and y candidates for every lane, selected without a branch; the column wraps with one unsigned compare
                    const int cx = ix + stepx, cy = iy + stepy;
                    const int cxw = (unsigned)cx >= (unsigned)S.nx ? wrapx : cx, cyw = (unsigned)cy >= (unsigned)S.ny ? wrapy : cy;
                    ix = xf ? cxw : ix;
                    iy = yf ? cyw : iy;
                    const float txn = fmaf(S.dx, iux, tx), tyn = fmaf(S.dy, iuy, ty);
                    tx = xf ? txn : tx;
                    ty = yf ? tyn : ty;
                    if (zf) {
                        k += uz > 0.0f ? 1 : -1;                     // (-1 and nz: the table's end records, uniform layers; the sign of uz: no register for the step)
                        const float4 Ln = lay4[k * kL4];
                        tz = fmaf(Ln.x, iuz, tz);
                        if (!(__float_as_int(Ln.w) & kLayStep3d)) mode = M_UNIFW;
                    }
                    if (mode == M_FLY) rec = VREC(ix, iy, k);
                }
            }
        }

        // =================================== phase B ===================================
        MI3D_TICK(0);
        MI3D_MARK("BSCHED");
        if (COUNT) { cnt.b_slots++; if (mode != M_FLY && mode != M_DONE) cnt.b_lanes++; }
        // Every MI3D_LEAN_FAST_PASS-th pass is a full one; the passes between serve nothing but the collisions the voxel walk has
        // found.  A pass with none of them pending is a full one, so nothing waits for ever.
        bool full_ = MI3D_LEAN_FAST_PASS <= 1 || ((pass_ctr++ % (unsigned)(MI3D_LEAN_FAST_PASS)) == 0u) || __ballot(mode == M_COLL) == 0ull;
        if (MI3D_LEAN_RARE_T > 0 && !full_) full_ = __popcll(__ballot(mode != M_FLY && mode != M_DONE && mode != M_COLL)) >= MI3D_LEAN_RARE_T;
        const bool full = full_;

        // =================================== block C ===================================
        // The same formulas in the same order as the shared blocks below (B2, B5, B6, B7), which serve the rarer events: one lane in a
        // state of its own no longer drags their mode checks and register copies through every pass.
        MI3D_MARK("C");
        bool fastc = (mode == M_COLL);
        float c_kstot = 0.0f, c_ks1 = 0.0f;
        if (fastc) {
            const float4 L = lay4[k * kL4];              // {dz, bt, zlo, flags}: a layer that is walked voxel by voxel
            const LayerRec &Lk = lay[k];
            const float ks1 = Lk.ks1d[0];
            const float ibt = frcp(rec.x);
            const float tc = fmaf(rem, ibt, t);
            // (|u| floored as where the parameters were set up: a photon flying exactly along an axis keeps its place across it)
            const float ax = __builtin_amdgcn_fmed3f((tx - tc) * floor_abs(ux), 0.0f, S.dx);
            const float ay = __builtin_amdgcn_fmed3f((ty - tc) * floor_abs(uy), 0.0f, S.dy);
            const float az = __builtin_amdgcn_fmed3f((tz - tc) * floor_abs(uz), 0.0f, L.x);
            px = ux > 0.0f ? S.dx - ax : ax;
            py = uy > 0.0f ? S.dy - ay : ay;
            pz = uz > 0.0f ? L.x - az : az;
            if (COUNT) cnt.scatter++;
            const float ks3 = rec.z;
            float kstot = ks1 + ks3;
            if (GEN && !ray1) for (int ip = 1; ip < np1d; ++ip) kstot += Lk.ks1d[ip];
            if (two3) {
                // er3t's cloud + aerosol scenes, mca_atm.py: a second {omega*ext, apf} pair per voxel
                const float2 cs = cold->csca[((unsigned)(iy * S.nx + ix) * (unsigned)S.nz3 + (unsigned)(k - S.k3lo)) * 2u + 1u];
                ev_ksb = cs.x; ev_apfb = cs.y;
                kstot += ev_ksb;
            }
            // (exactly 1 for conservative scattering: the approximate reciprocal must not nudge a weight that sits on the roulette
            //  threshold below it)
            w *= (kstot >= rec.x) ? 1.0f : kstot * ibt;
            c_kstot = kstot; c_ks1 = ks1;
            if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; mode = M_NEED; fastc = false; }
            else {
                const bool any_col = !MIXED || S.nmarch < S.nview;
                if (any_col) {
                    // mixture phase function towards the zenith (a column view looks straight down): cos(angle) = uz
                    float P = 0.0f;
                    if (GEN) {
                        if (ray1) {
                            P = ks1 * (0.75f * fmaf(uz, uz, 1.0f));
                            if (ks3 > 0.0f) P += ks3 * lean_phase_eval(T, rec.w, uz);
                            if (two3 && ev_ksb > 0.0f) P += ev_ksb * lean_phase_eval(T, ev_apfb, uz);
                        } else P = lean_mix_phase(T, Lk, np1d, ks3, rec.w, two3 ? ev_ksb : 0.0f, ev_apfb, uz);
                    } else if (plain) {
                        // (a constituent that is not there has a coefficient of 0 and a harmless selector)
                        P = ks1 * (0.75f * fmaf(uz, uz, 1.0f)) + ks3 * phase_eval_hg(rec.w, uz);
                        if (TWO) P += ev_ksb * phase_eval_hg(ev_apfb, uz);
                    } else {
                        if (ks1 > 0.0f) P = ks1 * phase_eval_analytic(Lk.apf1d[0], uz);
                        if (ks3 > 0.0f) P += ks3 * phase_eval_analytic(rec.w, uz);
                        if (TWO && ev_ksb > 0.0f) P += ev_ksb * phase_eval_analytic(ev_apfb, uz);
                    }
                    const float c = w * P * frcp(kstot) * (0.25f / kPi);
                    const float tau = rec.x * (L.x - pz) + rec.y;
                    // the pixel under the event: its column where the image has one pixel per column (er3t's satellite images:
                    // Rad_nxr = Atm_nx, Rad_nyr = Atm_ny, mcarats.py:360-367)
                    int ir = ix, jr = iy;
                    if (!same_grid) {
                        const float xr = (float)ix * S.dx + px, yr = (float)iy * S.dy + py;
                        ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                        jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                    }
                    const float val = c * fexp_neg(tau);
                    const int jv0 = MIXED ? S.col0 : 0;       // (first column view)
                    if (COUNT) { const int nc = MIXED ? S.nview - S.nmarch : S.nview; cnt.le_rays += nc; cnt.le_column += nc; }
                    if (c > 0.0f) {
                        // consecutive tallies of one history into the same pixel are summed in a register
                        const int pix = ir | (jr << 16);
#if MI3D_LEAN_PEND
                        if (pix != pend_pix && pend_pix >= 0) RAD_TALLY(pend_pix, pend_val);
                        pend_val = (pix == pend_pix) ? pend_val + val : val;
                        pend_pix = pix;
#else
                        RAD_TALLY(pix, val);
#endif
                        if (!MIXED || S.nview - S.nmarch > 1)    // (further column views: none in a nadir + slant set)
                        for (int jv = jv0 + 1; jv < S.nview; ++jv)
                            if (!MIXED || views[jv].column) RAD_ADD(&S.rad[(unsigned)((jv * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride], val);
                    }
                }
                if (EMIT) emit = true;
            }
        }
        if (EMIT) emit_events(cold, xcc, emit, ev_lo, ev_hi, px, py, pz, w, ux, uy, uz, rec.z, rec.w, rec.y, ix, iy, k, (int)E_SCATTER, seed, PHOTON_ID(), draw);
        if (fastc) {
            // ---- the constituent that scatters (the 1-D one first, then the 3-D ones in their order), the angle, the new direction
            const LayerRec &Lk = lay[k];
            float mu_rot;
            if (GEN && ray1) {
                // (the rule of lean_mix_select for this mixture: Rayleigh first, then the 3-D constituents in their order)
                const float target = u1 * c_kstot;
                if (target < c_ks1) mu_rot = phase_sample_analytic(-1.0f, u2);
                else {
                    const bool second = two3 && !(target < c_ks1 + rec.z);
                    const float ksel = second ? ev_ksb : rec.z, cum = second ? c_ks1 + rec.z : c_ks1;
                    const float usel = fminf(fmaxf(ksel > 0.0f ? (target - cum) * frcp(ksel) : 0.0f, 0.0f), 1.0f);
                    mu_rot = lean_phase_sample(T, second ? ev_apfb : rec.w, u2, usel);
                }
            } else if (GEN) {
                float usel;
                const float apf_g = lean_mix_select(Lk, np1d, rec.z, rec.w, ev_ksb, ev_apfb, two3 ? 2 : 1, u1, c_kstot, usel);
                mu_rot = lean_phase_sample(T, apf_g, u2, usel);
            } else {
                const float target = u1 * c_kstot;
                const bool first = target < c_ks1;
                float apf_sel = first ? Lk.apf1d[0] : rec.w;
                if (TWO && !first && !(target < c_ks1 + rec.z)) apf_sel = ev_apfb;
                mu_rot = phase_sample_analytic(apf_sel, u2);
            }
            rotate_dir(ux, uy, uz, mu_rot, u3);
            direct = false;
            if (w < S.wmin) { if (COUNT) cnt.roulette++; mode = M_DRAWR; }    // (a full pass plays it: B6)
            else {
                // ---- the next Philox block: free path and the numbers of the event at its end; then the walk's first three faces
                float r0, r1, r2, r3;
                draw4_fast<!EMIT>(seed, PHOTON_ID(), draw++, r0, r1, r2, r3);
                rem = -0.69314718f * __builtin_amdgcn_logf(r0);
                u1 = r1; u2 = r2; u3 = r3;
                const float4 L = lay4[k * kL4];
                iux = frcp(floor_abs(ux)); iuy = frcp(floor_abs(uy)); iuz = frcp(floor_abs(uz));
                tx = (ux > 0.0f ? S.dx - px : px) * iux;
                ty = (uy > 0.0f ? S.dy - py : py) * iuy;
                tz = (uz > 0.0f ? L.x - pz : pz) * iuz;
                t = 0.0f;
                const bool ipa = IPA_NOW();
                stepx = ipa ? 0 : (ux > 0.0f ? 1 : -1);
                stepy = ipa ? 0 : (uy > 0.0f ? 1 : -1);
                wrapx = ux > 0.0f ? 0 : S.nx - 1; wrapy = uy > 0.0f ? 0 : S.ny - 1;
                mode = M_FLY;
            }
        }
        MI3D_TICK(2);

        // =================================== the rarer events: full passes ===================================
        if (full) {
        MI3D_MARK("B0");
        // ---- where a walk has ended on a level, in front of a horizontally uniform layer (or out of the atmosphere)
        if (mode == M_UNIFW) {
            const float4 L = lay4[k * kL4];
            const float ax = __builtin_amdgcn_fmed3f((tx - t) * floor_abs(ux), 0.0f, S.dx);
            const float ay = __builtin_amdgcn_fmed3f((ty - t) * floor_abs(uy), 0.0f, S.dy);
            px = ux > 0.0f ? S.dx - ax : ax;
            py = uy > 0.0f ? S.dy - ay : ay;
            pz = uz > 0.0f ? 0.0f : L.x;   // bottom of the layer entered going up, top going down
            mode = M_UNIF;
        }
        // ---- B0: photons inside runs of horizontally uniform layers: the whole rest of the run at once
        if (mode == M_UNIF && (k < 0 || k >= S.nz)) {
            // the walk has left the atmosphere (the layer table's end records): out through the top, or onto the surface
            if (k < 0) { k = 0; pz = 0.0f; mode = M_SURF; }
            else { if (COUNT) cnt.escaped++; mode = M_NEED; }
        }
        if (mode == M_UNIF) {
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
                next = M_SETUP;                                          // (the walk is set up in B7)
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
                next = M_COLLU;
            }
            if (COUNT) cnt.steps++;
            px += ux * s; py += uy * s;
            k = knew; pz = pzn;
            // one fold for every way out of the run (the event blocks below find the position inside its column)
            fold_xy(S, cold, px, py, ix, iy, IPA_NOW());
            mode = next;
        }

        MI3D_TICK(1);
        MI3D_MARK("B2");
        // ---- B2: a collision inside uniform layers, or the surface: weight, local estimates answered from the column table
        if (mode == M_COLLU || mode == M_SURF) {
            const float4 L = lay4[k * kL4];              // {dz, bt, zlo, flags}
            const int flags = __float_as_int(L.w);
            const bool in3d = (flags & kLayIn3d) != 0;
            const LayerRec &Lk = lay[k];
            // (an event inside a uniform layer was found by B0, which has folded the position into its column)
            const unsigned col = (unsigned)(iy * S.nx + ix);
            if (!(flags & kLayStep3d)) {
                // the event was found by the uniform-layer code: no voxel step has brought the record
                float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (in3d) r = VREC(ix, iy, k);
                ev_tab = r.y; ev_ks0 = r.z; ev_apf0 = r.w;
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
                if (GEN) for (int ip = 1; ip < np1d; ++ip) kstot += Lk.ks1d[ip];
                if (two3) {
                    ev_ksb = 0.0f;
                    if (in3d) {
                        const float2 cs = cold->csca[(col * (unsigned)S.nz3 + (unsigned)(k - S.k3lo)) * 2u + 1u];
                        ev_ksb = cs.x; ev_apfb = cs.y;
                    }
                    kstot += ev_ksb;
                }
                w *= (kstot >= bt_ev) ? 1.0f : kstot * frcp(bt_ev);
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; dead = true; }
                if (any_col) {
                    float P = 0.0f;
                    if (GEN) P = lean_mix_phase(T, Lk, np1d, ks3, ev_apf0, two3 ? ev_ksb : 0.0f, ev_apfb, uz);
                    else if (plain) {
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
                    int ir = ix, jr = iy;
                    if (!same_grid) {
                        const float xr = (float)ix * S.dx + px, yr = (float)iy * S.dy + py;
                        ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                        jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                    }
                    const float val = c * fexp_neg(tau);
                    const int jv0 = MIXED ? S.col0 : 0;
                    if (COUNT) { const int nc = MIXED ? S.nview - S.nmarch : S.nview; cnt.le_rays += nc; cnt.le_column += nc; }
                    if (c > 0.0f) {
                        const int pix = ir | (jr << 16);
#if MI3D_LEAN_PEND
                        if (pix == pend_pix) pend_val += val;
                        else {
                            if (pend_pix >= 0) RAD_TALLY(pend_pix, pend_val);
                            pend_pix = pix; pend_val = val;
                        }
#else
                        RAD_TALLY(pix, val);
#endif
                        if (!MIXED || S.nview - S.nmarch > 1)
                        for (int jv = jv0 + 1; jv < S.nview; ++jv)
                            if (!MIXED || views[jv].column) RAD_ADD(&S.rad[(unsigned)((jv * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride], val);
                    }
                }
                mode = M_FINISH;
                if (EMIT) emit = true;
            }
        }
        if (EMIT) emit_events(cold, xcc, emit, ev_lo, ev_hi, px, py, pz, w, ux, uy, uz, ev_ks0, ev_apf0, ev_sfc, ix, iy, k, kind, seed, PHOTON_ID(), draw);

        MI3D_TICK(2);
        MI3D_MARK("B5");
        // ---- B5: finish the event (scattering inside uniform layers, surface reflection, or a launch without entry record)
        if (mode == M_FINISH) {
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
            }
            ux = bx; uy = by; uz = bz;
            if (kind != E_LAUNCH) direct = false;
            if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; mode = M_NEED; }
            else {
                mode = M_DRAW;
                if (w < S.wmin) { if (COUNT) cnt.roulette++; mode = M_DRAWR; }
            }
        }

        MI3D_TICK(4);
        MI3D_MARK("B6");
        // ---- B6: the Philox block of the rarer events
        if (mode == M_DRAW || mode == M_DRAWR || mode == M_DRAWL) {
            float r0, r1, r2, r3;
            draw4_fast<!EMIT>(seed, PHOTON_ID(), draw++, r0, r1, r2, r3);
            if (mode == M_DRAW) {
                rem = -0.69314718f * __builtin_amdgcn_logf(r0);
                u1 = r1; u2 = r2; u3 = r3;
                mode = (lay[k].flags & kLayStep3d) ? M_SETUP : M_UNIF;
            } else if (mode == M_DRAWR) {
                if (r0 * S.wfac < w) { w = S.wfac; mode = M_DRAW; }    // (survived: its flight is drawn in the next full pass)
                else { if (COUNT) cnt.killed++; mode = M_NEED; }
            } else { // the launch of a photon without entry record
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
                kind = E_LAUNCH;
                mode = M_FINISH;
            }
        }
        MI3D_TICK(5);

        // ---- B4: next photon
        MI3D_MARK("B4");
        // A history that has just ended hands in its pending tally; then the lanes without a photon take the next ids of the launch's
        // order -- a wave takes kChunk places at a time from the cursor of the XCD it runs on (speed only: an XCD then works on one
        // tile of the domain at a time, whose records stay in its L2) and, once that piece is used up, from the next XCD's -- and with
        // them their entry records (k_entry): launch, cone jitter, first free path and the uniform layers above the clouds are behind
        // such a photon, and B7 sets up its first voxel walk in this very pass.
        {
            const bool ended = mode == M_NEED && draw != 0;   // a history just ended (a lane that has had a photon has drawn for it)
            nphot_wave += (unsigned)__popcll(__ballot(ended));            // (counted per wave, in a scalar register)
            if (ended) {
                draw = 0;
                if (pend_pix >= 0) { RAD_TALLY(pend_pix, pend_val); pend_pix = -1; }
            }
        }
        bool took = false;   // wave-uniform: this pass has taken photons off the launch's order
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
                    if ((int)(threadIdx.x & 63) == leader) b = atomicAdd(cold->next_photon + x * kCtrStride, (unsigned long long)kChunk);
                    // (through scalar registers: the cursors are the same in every lane, and as per-lane values they cost five registers)
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
                lid = order ? order[pool_next + rank] : (uint32_t)(pool_next + rank);
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
                    if (COUNT && (km >> 31)) cnt.steps++;      // (the run of uniform layers k_entry has crossed)
                    w = 1.0f; direct = true; draw = 2;
                    kind = E_LAUNCH;
                } else {   // (no entry records: launched by B6 and B5 in the full passes to come)
                    draw = 0;
                    mode = M_DRAWL;
                }
            }
            took = took || avail != 0ull;
            pool_next += nn < avail ? nn : avail;
        }
        // ---- the tally window follows the photons (see RAD_TALLY above)
        if (win_on) {
            const bool lane0 = (threadIdx.x & 63) == 0;
            if (WLD(2) == wave_w + 1u) {
                // this wave has closed the window: has every other wave begun a pass since (or left)?
                bool clear = true;
                for (unsigned q = 0; q < NW; ++q) {
                    const unsigned c = WLD(8 + q);
                    if (q != wave_w && !(c != WLD(8 + NW + q) || c == kWinNone)) clear = false;
                }
                if (clear) {
                    win_flush();
                    // the tile the wanted place of the order lies in: tiles whose pieces end at or before it
                    const unsigned pos = WLD(6);
                    const uint32_t *tend = cold->tile_end;
                    const int ntile = cold->win_ntile;
                    int t = 0;
                    for (int j = 0; j < ntile; j += 64) {
                        const int q = j + (int)(threadIdx.x & 63);
                        t += __popcll(__ballot(q < ntile && tend[q] <= pos));
                    }
                    t = min(t, ntile - 1);
                    const unsigned lo = t > 0 ? tend[t - 1] : 0u, hi = tend[t];
                    const int ntx = cold->win_ntx, tc = cold->win_tc;
                    const int ty = t / ntx, tb = t - ty * ntx, tx = (ty & 1) ? ntx - 1 - tb : tb;   // (boustrophedon: launch_tile)
                    int ox = tx * tc + (int)(cold->win_off & 0xffffu), oy = ty * tc + (int)(cold->win_off >> 16);
                    while (ox >= S.nxr) ox -= S.nxr;
                    while (oy >= S.nyr) oy -= S.nyr;
                    if (lane0) {
                        WST(4, lo); WST(5, hi);
                        WST(1, (unsigned)ox | ((unsigned)oy << 16));
                        WST(0, (unsigned)ox | ((unsigned)oy << 16));
                        WST(2, 0u);
                    }
                }
            } else if (took) {
                const unsigned pos = (unsigned)(pool_next - 1ull);
                if ((pos < WLD(4) || pos >= WLD(5)) && WLD(2) == 0u) {
                    unsigned won = 0u;
                    if (lane0) won = atomicCAS(wctl + 2, 0u, wave_w + 1u) == 0u ? 1u : 0u;
                    won = (unsigned)__builtin_amdgcn_readfirstlane((int)won);
                    if (won) {
                        if (lane0) {
                            WST(0, kWinNone);                                             // closed: from their next pass on nobody adds
                            for (unsigned q = 0; q < NW; ++q) WST(8 + NW + q, WLD(8 + q));   // (read AFTER the origin was taken away)
                            WST(6, pos);
                        }
                    }
                }
            }
        }
        MI3D_TICK(3);

        MI3D_MARK("B7");
        // ---- B7: a lane about to walk: the parameters of the walk's first three faces, seen from where it is; the first record
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
            wrapx = ux > 0.0f ? 0 : S.nx - 1; wrapy = uy > 0.0f ? 0 : S.ny - 1;
            rec = VREC(ix, iy, k);
            mode = M_FLY;
        }
        MI3D_TICK(5);
        }   // full

        MI3D_MARK("END");
        if (__ballot(mode != M_DONE) == 0ull) break;
    }
#undef MI3D_TICK
    if (win_on) {
        // this wave adds to the window no more; the last one to leave empties it (a wave that leaves in the middle of a move leaves
        // the window closed: the sums wait for the last one)
        unsigned nd = 0u;
        if ((threadIdx.x & 63) == 0) { WST(8 + wave_w, kWinNone); nd = atomicAdd(wctl + 3, 1u); }
        nd = (unsigned)__builtin_amdgcn_readfirstlane((int)nd);
        if (nd == NW - 1u) win_flush();
    }

    if (EMIT) {
        for (unsigned long long q = ev_lo + (threadIdx.x & 63); q < ev_hi; q += 64)
            if (q < (unsigned long long)S.cold->ev_cap) S.cold->ev_list[ev_list_f4(S.cold->ev_cap) * xcc + ev_index((unsigned)q)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // ---- counters: wave reduction, one atomic per wave and counter
    {
        uint32_t vals[24] = {(threadIdx.x & 63) == 0 ? nphot_wave : 0u, cnt.steps, cnt.steps3d, cnt.scatter, cnt.surface, cnt.le_rays,
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
#undef VREC
#undef PHOTON_ID
}

#ifdef MI3D_ONLY_HEADLINE    // (a quick listing of the headline build alone: tools/quick_listing.sh)
template __global__ void k_transport_lean<false, false, 0, 0>(const DevScene, const uint64_t, const uint64_t, const uint64_t);
#else
#define MI3D_LEAN_INST(C, P) template __global__ void k_transport_lean<C, P, 0, 0>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 2, 0>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 0, 1>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 2, 1>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 0, 2>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 2, 2>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 0, 2, 512>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 0, 3>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                             template __global__ void k_transport_lean<C, P, 0, 3, 512>(const DevScene, const uint64_t, const uint64_t, const uint64_t);
MI3D_LEAN_INST(false, false) MI3D_LEAN_INST(false, true) MI3D_LEAN_INST(true, false) MI3D_LEAN_INST(true, true)
#undef MI3D_LEAN_INST
#endif

// Entry records (DevCold::entry).  What a new photon does before its first voxel walk -- launch (Philox block 0), the jitter of the
// solar cone, the first free path (Philox block 1), the flight through the uniform layers above the clouds -- is the same handful
// of steps for every photon, and inside the photon loop it is the worst kind of work: needed by one lane in thirty at a time, in
// blocks the loop shares with rarer events still, three full passes long.  Here every lane has a photon.  The same device functions
// in the same order as blocks B6 (launch), B5 (launch), B6 (flight) and B0 of k_transport_lean; a first flight that ends inside the
// uniform layers (a collision with the thin air up there, a scene without clouds) is handed over as it stands at the top of the
// atmosphere and the loop's own blocks finish it.
// What k_entry needs of the scene, by value: the kernel reads no device memory but the layer table, the launch's photon order and
// its own output (it does not wait for the launch's DevCold).
struct EntryArgs {
    const LayerRec *lay;      // [nz]
    float Lx, Ly, dx, dy, inv_dx, inv_dy, inv_nx, inv_ny;
    float sdx, sdy, sdz, cos_cone;
    int nx, ny, nz, solver, target, kdir;
};

// (one photon per thread, no loop over photons: 28 registers.  Running it and the photon order of launch i + 1 on a second stream
//  beside the photon loop of launch i was tried in round 4 -- they do run side by side then, and the loop loses more than the
//  pre-pass takes alone: profiles/r04/prepass_beside_the_photon_loop_tried.log, tools/experiments/prepass_overlap.patch)
__global__ void __launch_bounds__(256)
k_entry(const EntryArgs A, const uint64_t nphoton, const uint64_t seed, const uint64_t offset, const uint32_t *__restrict__ order, float4 *__restrict__ entry) {
    const LayerRec *lay = A.lay;
    const bool ipa = (A.solver == MI3D_SOLVER_IPA);     // (IPA_NOW of a direct beam: the partial 3-D solver moves it in 3-D)
    {   // one photon per thread (launch_entry: the grid covers the launch)
        const unsigned i = blockIdx.x * 256u + threadIdx.x;
        if (i >= (unsigned)nphoton) return;
        const uint64_t id = offset + (uint64_t)(order ? order[i] : i);
        float r0, r1, r2, r3;
        draw4_fast(seed, id, 0u, r0, r1, r2, r3);
        // ---- B6, the launch
        float x = r0 * A.Lx, y = r1 * A.Ly;
        if (x >= A.Lx) x = 0.0f;
        if (y >= A.Ly) y = 0.0f;
        int ix = min((int)(x * A.inv_dx), A.nx - 1);
        int iy = min((int)(y * A.inv_dy), A.ny - 1);
        float px = fminf(fmaxf(x - (float)ix * A.dx, 0.0f), A.dx);
        float py = fminf(fmaxf(y - (float)iy * A.dy, 0.0f), A.dy);
        int k = A.nz - 1;
        float pz = lay[k].dz;
        float ux = A.sdx, uy = A.sdy, uz = A.sdz;
        const float mu_cone = 1.0f - r2 * (1.0f - A.cos_cone);
        // ---- B5, the launch
        if (!(A.cos_cone >= 1.0f)) rotate_dir(ux, uy, uz, mu_cone, r3);
        // ---- B6, the first flight
        draw4_fast(seed, id, 1u, r0, r1, r2, r3);
        float rem = -0.69314718f * __builtin_amdgcn_logf(r0);
        int mode = (lay[k].flags & kLayStep3d) ? M_FLY : M_UNIF;
        unsigned ran = 0u;
        if (mode == M_UNIF) {
            // ---- B0: the run of uniform layers the photon starts in; taken here when the flight comes out at its far end into
            // layers that are walked voxel by voxel
            const bool up = uz > 0.0f;
            const LayerRec Lk = lay[k];
            const int kend = up ? Lk.run_hi : Lk.run_lo;
            const LayerRec Le = lay[kend];
            const float tv = up ? (Le.tauz + Le.bt * Le.dz - Lk.tauz) - Lk.bt * pz
                                : (Lk.tauz - Le.tauz) + Lk.bt * pz;
            const float hv = up ? (Le.zlo + Le.dz) - (Lk.zlo + pz) : (Lk.zlo + pz) - Le.zlo;
            const float iuzl = frcp(fmaxf(fabsf(uz), 1e-20f));
            const float tpath = tv * iuzl;
            const int knew = up ? kend + 1 : kend - 1;
            // (a flux job tallies the levels a flight crosses -- of the direct beam those below kdir: a run that holds such a level is
            //  left to the loop, whose block B0 makes the tallies)
            const bool no_tally = !(A.target & MI3D_TARGET_FLUX) || (!up && min(k, A.kdir - 1) < knew + 1);
            if (tpath < rem && knew >= 0 && knew < A.nz && no_tally) {
                rem -= tpath;
                const float s = hv * iuzl;
                px += ux * s; py += uy * s;
                k = knew;
                pz = up ? 0.0f : lay[knew].dz;
                fold_xy_raw(A.dx, A.dy, A.nx, A.ny, A.inv_dx, A.inv_dy, A.inv_nx, A.inv_ny, px, py, ix, iy, ipa);
                mode = M_FLY;
                ran = 1u;
            }
        }
        // (written once, read once by another kernel: non-temporal stores -- a streaming write of 24 GB per 5e8 photons)
        typedef float vf4 __attribute__((ext_vector_type(4)));
        vf4 *e = reinterpret_cast<vf4 *>(entry) + entry_index(i);
        __builtin_nontemporal_store((vf4){px, py, pz, rem}, e);
        __builtin_nontemporal_store((vf4){ux, uy, uz, r1}, e + 64);
        __builtin_nontemporal_store((vf4){r2, r3, __uint_as_float((unsigned)ix | ((unsigned)iy << 16)), __uint_as_float((unsigned)k | ((unsigned)mode << 16) | (ran << 31))}, e + 128);
    }
}

} // namespace mi3d
