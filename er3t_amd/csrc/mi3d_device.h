// mi3d_device.h — device-side scene description and physics helpers shared by the kernels.
//
// What the functions implement is defined by the input contract of the reference toolbox
// (hong-chen/er3t): phase-function selector `apf` (er3t/rtm/mca/mca_atm.py:101,262,276;
// er3t/rtm/mca/util.py:153), phase tables (er3t/rtm/mca/mca_sca.py:82-92), surface parameter
// packing (er3t/rtm/mca/mca_sfc.py:94-128) and the source / camera angles
// (er3t/rtm/mca/mcarats.py:285-307,374-383).  The transport algorithm itself (forward Monte Carlo
// with local-estimate radiance) is the published one the reference cites at mcarats.py:59.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi3d.h"

namespace mi3d {

constexpr float kPi = 3.14159265358979323846f;
constexpr int kLayStride = 16;   // floats per layer record (see LayerRec)
constexpr int kMaxLayers = 512;  // layer table capacity (staged in LDS)
constexpr float kTauCut = 16.0f; // a local-estimate ray beyond this optical depth adds < 1.2e-7 of its weight (dropped: five orders below the
                                 // noise of any affordable run; 32 instead of 16 costs the nine-view configuration 14 % of its speed)

constexpr int kTargetPlainPhase = 0x100;   // DevScene::target: the 1-D constituent is Rayleigh, every 3-D one Henyey-Greenstein (checked on the host)
constexpr int kTargetRayleigh1d = 0x200;   // ... there is ONE 1-D constituent and it is Rayleigh in every layer (whatever the voxels scatter by)
constexpr int kLayIn3d = 1;   // LayerRec.flags: the layer lies in the 3-D region (voxel tables exist)
constexpr int kLayStep3d = 2; // ... and its total extinction varies horizontally: march voxel by voxel

// One record per layer of the 1-D grid, staged in LDS (64 B per layer).  The first 16 bytes are
// what every cell step reads (one ds_read_b128).
struct LayerRec {
    float dz;         // thickness [m]
    float bt;         // total extinction while flying through the layer when it is horizontally
                      // uniform (1-D layers and uniform layers of the 3-D region) [1/m]
    float zlo;        // height of the layer bottom [m]
    int flags;        // kLayIn3d | kLayStep3d
    float ks1d[MI3D_MAX_NP1D];   // scattering coefficient of each 1-D constituent
    float apf1d[MI3D_MAX_NP1D];  // its phase-function selector
    float tabove;     // vertical optical depth from the top of this layer up to the top of the
                      // atmosphere (layers above the 3-D region) or up to the bottom of the 3-D
                      // region (layers below it); unused inside the 3-D region
    float tauz;       // vertical optical depth of the uniform layers below this one (0 at the
                      // surface; non-uniform layers count as 0)
    int run_lo, run_hi; // first / last layer of the run of consecutive uniform layers this layer
                        // belongs to (run_lo > run_hi for a non-uniform layer)
};
static_assert(sizeof(LayerRec) == kLayStride * sizeof(float), "LayerRec layout");

// One record per radiance view, staged in LDS (32 B).
struct ViewRec {
    float vx, vy, vz; // unit vector from the scene towards the sensor (vz > 0: down-looking sensor, vz < 0: up-looking)
    float zs;         // height of the sensor plane, clamped into the atmosphere
    int column;       // 1: exactly vertical view of a sensor above the atmosphere -> column table
    int roulette;     // bit 0: the view's local-estimate rays play Russian roulette beyond DevCold::le_tau1; bit 1: ... and on their weight
                      // below DevCold::le_cmin (le_weight_roulette)
    float zreg;       // height at which the line of sight is registered to a pixel: Rad_zref (down-looking), zs (up-looking)
    int point;        // 1: a camera (Rad_mrkind = 1): a point sensor, described by CamRec[view]; vx..vz, zreg unused
};
static_assert(sizeof(ViewRec) == 32, "ViewRec layout");

// One record per camera (mi3d_set_cameras), read from global memory when a ray to it starts or ends.
struct CamRec {
    float cx, cy, cz;        // position [m]
    float r2min;             // square of the nearest distance counted (Rad_apsize)
    float zx, zy, zz;        // the axis the camera looks along, world coordinates
    float cos_half;          // cosine of half the cone of view (Rad_qmax / 2)
    float xx, xy, xz;        // image x axis
    float inv_du;            // pixels per radian along U (nxr / Rad_umax)
    float yx, yy, yz;        // image y axis
    float inv_dv;            // pixels per radian along V (nyr / Rad_vmax)
};
static_assert(sizeof(CamRec) == 64, "CamRec layout");

// Rarely used scene data lives in device memory behind a pointer and is copied to LDS when the transport kernel
// starts; only what the voxel walk and the common collision path touch travels in SGPRs as kernel arguments.  (The
// full set in SGPRs exceeded the register file and cost ~170 v_readlane/v_writelane spill moves per pass; read
// through the global pointer inside the loop the fields became VECTOR loads -- the kernel writes to memory, so the
// compiler cannot treat them as invariant -- each a round trip queued behind the tally atomics.)
struct DevCold {
    float ztoa, zref, inv_Lx, inv_Ly;
    // phase tables (ascending mu)
    int nang, npf;
    const float *tmu, *tp, *tcdf;
    int tab_lo, tab_n;     // tables tab_lo .. tab_lo+tab_n-1 are staged in LDS by the transport kernel (tab_n = 0: none)
    // surface
    int sfc_mtype, nxb, nyb;
    float sfc_p0, sfc_p1, sfc_p2, sfc_sx, sfc_sy; // sfc_sx/sy = nxb/Lx, nyb/Ly: position -> surface cell
    const float *sfc2d;    // [(jb*nxb+ib)*8] {type, p0..p4, pad, pad}
    const LayerRec *lay;   // [nz]
    const ViewRec *views;  // [nview]
    unsigned long long *counters;  // [MI3D_NCOUNTER]
    // domain, source and the tables only rarer paths touch (launch, position folding in 1-D layers, events below the
    // 3-D region, further 3-D constituents, the photon-id pool)
    float Lx, Ly;
    float inv_nx, inv_ny;                 // 1/nx, 1/ny (column index wrap without integer division)
    float inv_dx, inv_dy;                 // reciprocals (multiplications instead of divisions in the loop)
    float sdx, sdy, sdz, cos_cone;
    const float2 *csca;    // [((iy*nx+ix)*nz3 + k3)*np3d + ip] {omega*ext, apf}: further constituents (ip >= 1)
    const float *tcol0;    // [iy*nx+ix]  vertical optical depth from the bottom of the 3-D region to TOA
    unsigned long long *next_photon;      // [8][kCtrStride]: one cursor per XCD into its eighth of the launch's photon order
    float le_tau1;         // > 0: local-estimate rays survive beyond this optical depth with probability exp(-(tau - le_tau1))
    float sfc_p3;          // fourth and fifth surface parameter (the diffuse-specular mixture has five)
    const uint32_t *order; // [nphoton of the launch] photon indices sorted by launch tile (k_bin_*), or nullptr: identity
    float sfc_p4;
    int ev_cap;            // capacity of each XCD's event list
    const CamRec *cams;    // [nview] cameras (views with ViewRec::point), else nullptr
    float4 *ev_list;       // [8][ev_list_f4(ev_cap)] event records in blocks of 64, one list per XCD (k_transport_lean<.,.,2> writes, k_rays reads): ev_index, ev_word
    unsigned long long *ev_ctr;   // [kCtrWords][kCtrStride]: [x] events in list x; [8]: set when a list ran full; [kCtrCursor + x]: k_rays'
                                  // cursor into list x; [kCtrHeavyFill + x], [kCtrHeavyCursor + x]: the same for hv_list
    unsigned long long *hv_list;  // [8][ev_cap] list << 32 | slot of the events k_rays' light build leaves to the heavy one
                                  // (reflections off LSRT / DSM surfaces), written by the former; nullptr: the scene has none
    double *heat;                 // [nz][ny][nx] weight absorbed per cell (heating rates, Flx_mhrt = 1), or nullptr
    float le_cmin;                // > 0: local-estimate rays of marched satellite views that would carry less are marched with probability c / le_cmin
    unsigned cam_images;          // cameras: contributions go to the periodic images of the camera within this many domain lengths of the nearest one (0: nearest only)
    const float4 *entry;          // [entry_f4(photons of the launch)] entry records (k_entry -> k_transport_lean, block B4), or nullptr: none
    // The tally window of the lean loop (mi3d_kernel_lean.hip): a workgroup sums the tallies of the column view in LDS for the
    // kWin x kWin pixels around the tile of columns its photons started in, and adds them to the image when it moves on to
    // the next tile -- the chip does 2.4e10 float64 atomics a second, the loop wanted 2.3e10.
    const uint32_t *tile_end;     // [win_ntile] where every tile's piece of the launch's photon order ends (k_bin_scatter leaves it in its cursors), or nullptr: no window
    int win_tc, win_ntx;          // tile edge in columns, tiles per row of tiles
    int win_ntile;
    unsigned win_off;             // x | y << 16: from a tile's first column / row to the window's, modulo the domain: where the direct
                                  // beam from the tile's columns at the top of the atmosphere meets the clouds, less the margin
    // bucket indices into the phase tables (round 5, the lean kernels' table look-ups): kTabNB buckets over mu in [-1, 1] (resp. over the
    // cumulative probability in [0, 1], per table); entry b: the last node that falls into a bucket below b (tab_bucket_mu / tab_bucket_u of
    // its value < b), 0 where there is none -- the node a look-up in bucket b starts from
    const uint16_t *tmu_idx;      // [kTabIdxN]
    const uint16_t *tcdf_idx;     // [npf][kTabIdxN]
};
static_assert(sizeof(DevCold) == 304, "DevCold is staged in LDS as 19 float4");
constexpr int kTabNB = 512;       // buckets of the table indices
constexpr int kTabIdxN = kTabNB + 2;   // entries per index (one per bucket and one beyond, padded to an even count)
// The bucket of a value, worked out with the SAME float32 operations by the host that builds the indices and by the kernels that use them
// (one fused multiply-add, one multiplication, one truncation): monotone in the value, so every node in a lower bucket lies below
// the value looked up and every node in a higher one above it -- no slack bucket on either side.
__host__ __device__ inline int tab_bucket_mu(float mu) { const int b = (int)(fmaf(mu, 0.5f, 0.5f) * (float)kTabNB); return b < 0 ? 0 : (b > kTabNB - 1 ? kTabNB - 1 : b); }
__host__ __device__ inline int tab_bucket_u(float u) { const int b = (int)(u * (float)kTabNB); return b < 0 ? 0 : (b > kTabNB - 1 ? kTabNB - 1 : b); }
constexpr int kWin = 64;          // edge of the tally window in pixels (kWin * kWin floats of LDS per workgroup)
// Entry record (k_entry -> k_transport_lean, block B4): the state of a photon of the launch where its first voxel walk begins -- the
// launch, the solar-cone jitter, the first free path and the flight through the uniform layers above the clouds worked out by a
// kernel of its own in which every lane has a photon -- 48 bytes, at the photon's place in the launch's order:
//   [0] px, py, pz, rem        position inside the voxel, optical depth left of the first free path
//   [1] ux, uy, uz, u1         direction; the three numbers of the event at the end of the flight
//   [2] u2, u3, ix | iy<<16, k | mode<<16 | ran<<31      cell; M_FLY: ready to walk (ran: a run of uniform layers was crossed on the way),
//                                                        M_UNIF: still at the top of the atmosphere (the rare ways a first flight ends)
// In blocks of 64 records, part by part, like the event records: lanes with consecutive places read consecutive 16-byte pieces.
constexpr int kEntryF4 = 3;
__host__ __device__ inline size_t entry_f4(size_t n) { return ((n + 63) / 64) * 64 * (size_t)kEntryF4; }
__host__ __device__ inline unsigned entry_index(unsigned pos) { return (pos >> 6) * (64u * (unsigned)kEntryF4) + (pos & 63u); }
// Event record (k_transport_lean<.,.,2> -> k_rays): a collision or surface reflection whose marched views are still to be served,
// 52 bytes:
//   [0] px, py, pz, w          position inside the voxel, weight after the event
//   [1] ux, uy, uz, ks0        incoming direction; scattering coefficient of the 3-D constituent (surface: first parameter)
//   [2] apf0, p2, ix | iy<<16, k | kind<<16     phase selector (surface: second, third parameter); cell; kind as in the loop
//   [3] one word: le_hash_base(seed, photon id, index of its next Philox block): what the roulettes of the event's rays hash
// In memory the records of a list stand in blocks of 64, part by part: 64 first parts, 64 second ones, 64 third ones, 64 words.  A
// wave of the photon loop hands consecutive slots to its lanes, so each store of an event batch writes one contiguous piece
// instead of 64 pieces at a stride of a record.
constexpr int kEvStride = 64;             // float4 between two 16-byte parts of one record
constexpr int kEvBlockF4 = 3 * 64 + 16;   // float4 per block of 64 records (3328 bytes)
__host__ __device__ inline size_t ev_list_f4(size_t cap) { return (cap / 64) * kEvBlockF4; }   // float4 per list of `cap` records (a multiple of 64)
__host__ __device__ inline unsigned ev_index(unsigned slot) { return (slot >> 6) * (unsigned)kEvBlockF4 + (slot & 63u); }   // float4 index of part 0 within its list
__host__ __device__ inline unsigned ev_word(unsigned slot) { return ((slot >> 6) * (unsigned)kEvBlockF4 + 192u) * 4u + (slot & 63u); }   // uint32 index of part 3
constexpr unsigned kEvBlock = 512;   // records a wave of the photon loop reserves at a time; unused ones are marked empty (w = 0)
constexpr unsigned kCtrCursor = 9, kCtrHeavyFill = 17, kCtrHeavyCursor = 25, kCtrWords = 33;   // rows of DevCold::ev_ctr
constexpr int kCtrStride = 16; // unsigned long long words between two XCD cursors: one 128-byte line each
constexpr int kColdF4 = sizeof(DevCold) / 16;

// Tallies are float64: a float32 accumulator stops growing once it exceeds 2^24 times a contribution (one pixel fed by 4e6
// photons through marched views lost 7 % that way; small grids and long runs are exactly where that happens), and
// global_atomic_add_f64 is a native instruction on this chip.
typedef double tally_t;
// Float atomics execute at the memory side, one after the other for adds to the same 128-byte line
// (profiles/r02/atomic_contention_les480.log): with the photon order sorted by start tile the 41 000 photons an XCD has in
// flight tally into a patch of a few thousand pixels, i.e. a few hundred lines of a dense image.  The kernels therefore add
// into an accumulation image that gives every pixel a line of its own.
constexpr int kRadLine = 16;   // tally_t elements per 128-byte line

struct DevScene {
    // grid
    int nz, k3lo, nx, ny, nz3, np1d, np3d;
    int kdir;                             // flux: direct-beam crossings of levels >= kdir are not tallied (analytic, added on read-out)
    float dx, dy;
    float pix_sx, pix_sy;                 // nxr/Lx, nyr/Ly: position -> radiance pixel
    unsigned vcol_f4, vrow_f4;   // records between two columns of a row (>= nz3) and between two rows (>= nx vcol_f4): mi3d_prepare pads
                                 // them so that neighbouring columns and rows do not land on the same memory channels
    const float *bext3;    // [(iy*nx + ix)*nz3 + k3] the total extinction alone, 4 bytes per voxel: what the ray kernel's walk reads (sixteen
                           // cells of a column per 64 bytes instead of four: its rays cross five cells on average)
    const float4 *vrec;    // [iy*vrow_f4 + ix*vcol_f4 + k3]  one 16-byte record per voxel, z fastest:
                           //   .x total extinction, .y vertical optical depth from the voxel's top face to TOA,
                           //   .z omega*ext and .w apf of the first 3-D constituent.  Everything a collision in
                           //   the voxel needs sits in the cache line the voxel walk has just touched.
    // views
    int nview, nmarch, nxr, nyr; // nmarch: views whose local-estimate ray is marched cell by cell
    int col0;                    // first view answered from the column table (0 when there is none)
    // job
    int target, solver;
    float wmin, wfac;
    // outputs
    int rad_row;                   // pixels between two rows of `rad`: nxr, or more in the accumulation image (its rows are padded: mi3d_run)
    int rad_stride;                // tally elements between two pixels of `rad`: kRadLine when `rad` is the accumulation image
                                   // (one pixel per 128-byte line, folded into the caller's tally by k_fold_rad), else 1
    tally_t *rad;                  // [nview][nyr][nxr] x rad_stride raw sums
    tally_t *flux;                 // [3][nz+1][ny][nx] raw sums
    const DevCold *cold;
};

// Non-temporal accesses for what is written once and read once by another kernel (entry records, event records, tally records): such a
// stream through an XCD's 4 MiB L2 pushes out the voxel records the photon loop and the ray kernel live on -- 1.1 KB of event records per
// photon cost the nine-view workload 9 % (profiles/r05/ab_nt_event_records.log).
typedef float vf4_t __attribute__((ext_vector_type(4)));
typedef unsigned vu2_t __attribute__((ext_vector_type(2)));
__device__ inline float4 nt_load(const float4 *p) { const vf4_t v = __builtin_nontemporal_load(reinterpret_cast<const vf4_t *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ inline void nt_store(float4 *p, const float4 v) { __builtin_nontemporal_store((vf4_t){v.x, v.y, v.z, v.w}, reinterpret_cast<vf4_t *>(p)); }
__device__ inline uint2 nt_load(const uint2 *p) { const vu2_t v = __builtin_nontemporal_load(reinterpret_cast<const vu2_t *>(p)); return make_uint2(v.x, v.y); }
__device__ inline void nt_store(uint2 *p, const uint2 v) { __builtin_nontemporal_store((vu2_t){v.x, v.y}, reinterpret_cast<vu2_t *>(p)); }
#ifndef MI3D_ENTRY_NT_LOAD
#define MI3D_ENTRY_NT_LOAD 1   // 1: the photon loops read their entry records with non-temporal loads (+1.1 % on the 480 x 480 nadir bench, profiles/r05/ab_nt_entry_tally_records.log)
#endif
#ifndef MI3D_TL_NT
#define MI3D_TL_NT 9           // bit 0: the flux loop writes its tally records with non-temporal stores; bit 1: the sort reads them so and writes the binned
                               // records so; bit 2: the sum reads the binned records so; bit 3: the loop writes its run records so.  Round 6, with run records:
                               // 0 / 1 / 8 / 9 / 11: 1.438 / 1.457 / 1.449 / 1.455 / 1.405e9 on les128_flux, 8.06 / 8.03 / 8.00 / 8.10 / 7.87e8 on les480_flux
                               // (profiles/r06/ab_flux_nt_records.log); round 5 (every crossing a record): +0.2 % in the loop, -12 % in the sort
#endif

// 1-ulp hardware reciprocal / square root / reciprocal square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32): a plain `/`
// or sqrtf() expands to a ~12-instruction IEEE sequence, far more than Monte-Carlo noise can make use of.
__device__ inline float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ inline float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10, counter = (id lo, id hi, draw, 0), key = (seed lo, seed hi)
// ---------------------------------------------------------------------------------------------
// a ^ b ^ k in ONE vector instruction (v_bitop3_b32, truth table 0x96), k wave-uniform: the compiler emits two v_xor_b32 per
// three-way xor (40 per Philox block where 20 will do; the key schedule stays on the scalar unit)
__device__ inline uint32_t xor3_key(uint32_t a, uint32_t b, uint32_t k) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

// (the key is the job's seed: the same in every lane -- a kernel argument)
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3_key((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = xor3_key((uint32_t)(p0 >> 32), c3, k1);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// u = ((word >> 9) + 0.5) * 2^-23: 24 significant bits, exact in float, never 0 or 1
__device__ inline float u01(uint32_t w) { return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f); }
// The same number in two instructions instead of four: the word's upper 23 bits as the mantissa of a float in [1, 2)
// (v_alignbit_b32 shifts them in under the exponent of 1.0), then ONE add of -(1 - 2^-24): 1 + m 2^-23 - 1 + 2^-24 = (m + 0.5) 2^-23 is
// representable, so the add is exact and the result equals u01(w) bit for bit (tests/test_host_properties.py holds the two forms
// against each other over all 2^23 mantissas; the single-history tests of tests/test_gpu_parity.py would show a slip on the GPU).
__device__ inline float u01_fast(uint32_t w) {
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, w, 9u)) + (-0.99999994f);
}
// max(|x|, 1e-20) in ONE instruction, where fmaxf(fabsf(x), 1e-20f) costs two in IEEE mode (the compiler puts a canonicalising
// v_max in front, also when the clamp is written as a median)
__device__ inline float floor_abs(float x) {
    float r;
    asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(1e-20f));
    return r;
}

__device__ inline void draw4(uint64_t seed, uint64_t id, uint32_t draw, float &u0, float &u1, float &u2,
                             float &u3) {
    uint32_t w[4];
    philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), draw, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    u0 = u01(w[0]); u1 = u01(w[1]); u2 = u01(w[2]); u3 = u01(w[3]);
}
template <bool OPAQUE_KEY = false>
__device__ inline void draw4_fast(uint64_t seed, uint64_t id, uint32_t draw, float &u0, float &u1, float &u2,
                                  float &u3) {
    uint32_t w[4];
    // (OPAQUE_KEY, the plain lean photon loop: the key words made opaque where the block is drawn: the nine later round keys are then worked out by the scalar unit inside
    //  every call, eighteen s_add -- hoisted out of the photon loop they hold eighteen scalar registers for its whole length, and
    //  the loop spills scalars into vector lanes that it reads back with v_readlane on the vector unit, which is the busy one)
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (OPAQUE_KEY) asm volatile("" : "+s"(k0), "+s"(k1));
    philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), draw, 0u, k0, k1, w);
    u0 = u01_fast(w[0]); u1 = u01_fast(w[1]); u2 = u01_fast(w[2]); u3 = u01_fast(w[3]);
}

// ---------------------------------------------------------------------------------------------
// phase functions (∫P dΩ = 4π)
// ---------------------------------------------------------------------------------------------
// Table pointers travel BY VALUE into the out-of-line table routines: taking the address of the
// kernel-argument struct would force the whole of it into scratch memory.
// exp(-t) for t >= 0 as one v_exp_f32 (the library form adds range handling these arguments do not need)
__device__ inline float fexp_neg(float t) { return __builtin_amdgcn_exp2f(t * -1.44269504f); }

// One uniform number per local-estimate ray for its roulette: a hash (lowbias32 finaliser) of the seed, the photon id, the
// index of the photon's next Philox block and the view.  Restated bit for bit in oracle/mi3d_oracle.c.  In two steps: what depends
// on the event only (le_hash_base: the photon loop that writes event records for k_rays works it out once per event and puts it
// into the record) and the rest per view.
__device__ inline uint32_t le_hash_base(uint64_t seed, uint64_t id, uint32_t draw) {
    return (uint32_t)id ^ ((uint32_t)(id >> 32) * 0x9E3779B9u) ^ (draw * 0x85EBCA6Bu) ^ (uint32_t)seed;
}
__device__ inline float le_roulette_from_base(uint32_t base, int iv) {
    uint32_t h = base ^ ((uint32_t)(iv + 1) * 0xC2B2AE35u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
}
__device__ inline float le_roulette_u(uint64_t seed, uint64_t id, uint32_t draw, int iv) {
    return le_roulette_from_base(le_hash_base(seed, id, draw), iv);
}

// Russian roulette on the weight c a marched local-estimate ray would carry (w P / 4 pi, surface: w R cos / pi): below cmin it is
// marched with probability c / cmin and then carries cmin.  Unbiased; its uniform number is the hash above with the view moved on by
// 16.  Returns the weight to march with (0: no ray).  Restated in oracle/mi3d_oracle.c, radiance_tally.
__device__ inline float le_weight_roulette_base(float c, float cmin, uint32_t base, int iv) {
    if (!(c < cmin) || !(c > 0.0f)) return c;
    return le_roulette_from_base(base, iv + 16) * cmin < c ? cmin : 0.0f;
}
__device__ inline float le_weight_roulette(float c, float cmin, uint64_t seed, uint64_t id, uint32_t draw, int iv) {
    return le_weight_roulette_base(c, cmin, le_hash_base(seed, id, draw), iv);
}

struct PhaseTab {
    const float *tmu, *tp, *tcdf; // tp/tcdf already offset so that index `it` is the absolute table number
    int nang, npf;
};

// `ltab` = LDS copy of the tables the scene actually uses (tables tab_lo .. tab_lo+tab_n-1 laid out
// [mu(nang)][p(tab_n*nang)][cdf(tab_n*nang)]), or nullptr when they are read from global memory.
__device__ inline PhaseTab phase_tab(const DevCold *C, const float *ltab) {
    PhaseTab T;
    T.nang = C->nang; T.npf = C->npf;
    if (ltab) {
        const long shift = (long)C->tab_lo * C->nang;
        T.tmu = ltab;
        T.tp = ltab + C->nang - shift;
        T.tcdf = ltab + C->nang + (long)C->tab_n * C->nang - shift;
    } else {
        T.tmu = C->tmu; T.tp = C->tp; T.tcdf = C->tcdf;
    }
    return T;
}

__device__ inline float table_eval(const PhaseTab S, int it, float mu) {
    const float *m = S.tmu, *p = S.tp + (long)it * S.nang;
    int lo = 0, hi = S.nang - 1;
    if (mu <= m[0]) return p[0];
    if (mu >= m[hi]) return p[hi];
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (m[mid] <= mu) lo = mid; else hi = mid;
    }
    const float f = (mu - m[lo]) / (m[hi] - m[lo]);
    return p[lo] + f * (p[hi] - p[lo]);
}

__device__ inline float table_sample(const PhaseTab S, int it, float u) {
    const float *m = S.tmu, *p = S.tp + (long)it * S.nang, *cdf = S.tcdf + (long)it * S.nang;
    int lo = 0, hi = S.nang - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid; else hi = mid;
    }
    const float r = 2.0f * (u - cdf[lo]);
    const float sl = (p[hi] - p[lo]) / (m[hi] - m[lo]);
    const float disc = fmaxf(p[lo] * p[lo] + 2.0f * sl * r, 0.0f);
    const float den = p[lo] + sqrtf(disc);
    const float t = den > 0.0f ? 2.0f * r / den : 0.0f;
    return fminf(m[lo] + t, m[hi]);
}

__device__ inline void table_pick(const PhaseTab S, float apf, int &i0, float &f) {
    const float t = apf - 1.0f;
    int i = (int)floorf(t);
    float fr = t - (float)i;
    if (i < 0) { i = 0; fr = 0.0f; }
    if (i >= S.npf - 1) { i = S.npf - 1; fr = 0.0f; }
    i0 = i; f = fr;
}

__device__ __noinline__ float phase_eval_table(const PhaseTab S, float apf, float mu) {
    if (S.npf <= 0) return 1.0f;
    int i0; float f;
    table_pick(S, apf, i0, f);
    float p = table_eval(S, i0, mu);
    if (f > 0.0f) p = (1.0f - f) * p + f * table_eval(S, i0 + 1, mu);
    return p;
}

__device__ inline float phase_eval(const DevCold *C, const float *ltab, float apf, float mu) {
    if (apf >= 1.0f) return phase_eval_table(phase_tab(C, ltab), apf, mu);
    if (apf <= -1.5f) return 1.0f;
    if (apf <= -1.0f) return 0.75f * (1.0f + mu * mu);
    const float g = apf, r = frsq(1.0f + g * g - 2.0f * g * mu);
    return (1.0f - g * g) * r * r * r;
}

__device__ __noinline__ float phase_sample_table(const PhaseTab S, float apf, float u, float usel) {
    if (S.npf <= 0) return 2.0f * u - 1.0f;
    int i0; float f;
    table_pick(S, apf, i0, f);
    if (f > 0.0f && usel < f) i0 += 1;
    return table_sample(S, i0, u);
}

__device__ inline float phase_sample(const DevCold *C, const float *ltab, float apf, float u, float usel) {
    if (apf >= 1.0f) return phase_sample_table(phase_tab(C, ltab), apf, u, usel);
    if (apf <= -1.5f) return 2.0f * u - 1.0f;
    if (apf <= -1.0f) {
        const float q = 8.0f * u - 4.0f;
        const float a = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(0.5f * q + fsqrt(0.25f * q * q + 1.0f)) * (1.0f / 3.0f)); // argument > 0
        return a - frcp(a);
    }
    const float g = apf;
    if (fabsf(g) < 1e-3f) return 2.0f * u - 1.0f;
    const float t = (1.0f - g * g) * frcp(1.0f - g + 2.0f * g * u);
    const float mu = (1.0f + g * g - t * t) * frcp(2.0f * g);
    return fminf(fmaxf(mu, -1.0f), 1.0f);
}

// The same two functions for scenes that refer to no tabulated phase function (apf < 1 everywhere: checked on the host
// before a kernel that uses them is chosen): no table branch, no call.
__device__ inline float phase_eval_analytic(float apf, float mu) {
    if (apf <= -1.5f) return 1.0f;
    if (apf <= -1.0f) return 0.75f * (1.0f + mu * mu);
    const float g = apf, r = frsq(1.0f + g * g - 2.0f * g * mu);
    return (1.0f - g * g) * r * r * r;
}

__device__ inline float phase_sample_analytic(float apf, float u) {
    if (apf <= -1.5f) return 2.0f * u - 1.0f;
    if (apf <= -1.0f) {
        const float q = 8.0f * u - 4.0f;
        const float a = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(0.5f * q + fsqrt(0.25f * q * q + 1.0f)) * (1.0f / 3.0f)); // argument > 0
        return a - frcp(a);
    }
    const float g = apf;
    if (fabsf(g) < 1e-3f) return 2.0f * u - 1.0f;
    const float t = (1.0f - g * g) * frcp(1.0f - g + 2.0f * g * u);
    const float mu = (1.0f + g * g - t * t) * frcp(2.0f * g);
    return fminf(fmaxf(mu, -1.0f), 1.0f);
}

// ---------------------------------------------------------------------------------------------
// Tabulated phase functions in the lean kernels (round 5): the same tables, the same piecewise-linear function of mu and the same
// exact inversion of its CDF as table_eval / table_sample above (and as oracle/mi3d_oracle.c), with the node found through a bucket
// index instead of a bisection over the whole table: two or three dependent reads instead of nine (498 angles), inlined -- no call,
// no spill around it.  The bisection that remains runs over the nodes of three neighbouring buckets (the bucket of a value worked out
// in float32 may be off by one) with the comparisons of the full bisection: it ends on the same node.
// ---------------------------------------------------------------------------------------------
struct LeanTab {
    const float *mu;                      // the LDS copy (stage_tables): mu[nang], p and cdf of the staged tables, the indices
    int nang, npf;
    int op, oc;                           // floats from mu to where p / cdf of table number 0 WOULD stand: table `it` at mu + op + it * nang
    int oi, ilo;                          // floats from mu to the indices (mu's first, then the staged tables'); the first staged table
    __device__ const float *p(int it) const { return mu + op + it * nang; }
    __device__ const float *cdf(int it) const { return mu + oc + it * nang; }
    __device__ const uint16_t *mu_idx() const { return reinterpret_cast<const uint16_t *>(mu + oi); }
    __device__ const uint16_t *cdf_idx(int it) const { return reinterpret_cast<const uint16_t *>(mu + oi) + (1 + it - ilo) * kTabIdxN; }
};

// where the tables are: the LDS copy made by stage_tables (tables tab_lo .. tab_lo + tab_n - 1).  ALWAYS LDS: the lean kernels serve a
// scene that refers to tables only when they fit (mi3d_run; else the general kernel reads them from global memory) -- every pointer
// below then derives from the kernel's LDS array, the look-ups compile to ds_read instructions and their addresses are 32-bit
__device__ inline LeanTab lean_tab(const DevCold *C, const float *ltab) {
    // (five wave-uniform integers, said so: read from the LDS copy of the cold block they are vector values, and seven vector registers
    //  held the same number in every lane for the length of the photon loop)
    LeanTab T;
    const int nang = __builtin_amdgcn_readfirstlane(C->nang), lo = __builtin_amdgcn_readfirstlane(C->tab_lo), n = __builtin_amdgcn_readfirstlane(C->tab_n);
    T.nang = nang; T.npf = ltab ? __builtin_amdgcn_readfirstlane(C->npf) : 0;      // (no table staged: a selector >= 1 reads as isotropic, as where no table is loaded)
    T.mu = ltab;
    T.op = nang - lo * nang; T.oc = nang + n * nang - lo * nang;
    T.oi = nang + 2 * n * nang; T.ilo = lo;
    return T;
}
// floats of LDS the staged tables take (mu, p, cdf of tab_n tables, the indices behind them)
__host__ __device__ inline size_t lean_tab_floats(int nang, int tab_n) {
    return (size_t)(1 + 2 * tab_n) * nang + ((size_t)(1 + tab_n) * kTabIdxN * sizeof(uint16_t) + 3) / 4;
}
// workgroup-wide copy of the tables the scene refers to into LDS (call before the kernel's first __syncthreads)
__device__ inline void stage_tables(const DevCold *C, float *dst) {
    const int nang = C->nang, nt = C->tab_n * nang;
    for (int i = threadIdx.x; i < nang; i += blockDim.x) dst[i] = C->tmu[i];
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
        dst[nang + i] = C->tp[(long)C->tab_lo * nang + i];
        dst[nang + nt + i] = C->tcdf[(long)C->tab_lo * nang + i];
    }
    uint16_t *ib = reinterpret_cast<uint16_t *>(dst + nang + 2 * nt);
    for (int i = threadIdx.x; i < kTabIdxN; i += blockDim.x) ib[i] = C->tmu_idx[i];
    for (int i = threadIdx.x; i < C->tab_n * kTabIdxN; i += blockDim.x) ib[kTabIdxN + i] = C->tcdf_idx[(long)C->tab_lo * kTabIdxN + i];
}

// largest node lo in [0, n - 2] with a[lo] <= x, for a[0] <= x < a[n - 1]; b: the bucket of x.  The nodes of the lower buckets lie below x:
// the search starts at the last of them and looks at the nodes of x's own bucket -- at most two in nearly every bucket of er3t's angle
// grids (two reads at fixed offsets, no loop); the crowded buckets (the diffraction peak's hundred nodes per bucket of mu) finish by bisection.
__device__ inline int lean_tab_find(const float *a, const uint16_t *idx, const int n, const float x, const int b) {
    int lo = idx[b];
    const int hi = min((int)idx[b + 1] + 1, n - 1);      // (the first node of a higher bucket, or the last node: above x)
    const float a1 = a[min(lo + 1, n - 1)], a2 = a[min(lo + 2, n - 1)];
    lo += (lo + 1 < hi && a1 <= x) ? ((lo + 2 < hi && a2 <= x) ? 2 : 1) : 0;
    // (more than two nodes of the bucket below x?  A third probe: deciding it from the index alone -- no read -- was built and cost the photon
    //  loop four more spilled registers and 8 %: profiles/r05/ab_tables_builds.log)
    if (hi - lo > 1 && a[lo + 1] <= x) {
        int h2 = hi;
        lo += 1;
        while (h2 - lo > 1) {
            const int mid = (lo + h2) >> 1;
            if (a[mid] <= x) lo = mid; else h2 = mid;
        }
    }
    return lo;
}

// the phase function of selector apf towards cosine mu (apf >= 1: table apf - 1, a fractional part mixes it with the next one): one
// search of the shared mu grid serves both tables of a mixture
__device__ inline float lean_phase_eval(const LeanTab &T, const float apf, const float mu) {
    if (apf >= 1.0f) {
        if (T.npf <= 0) return 1.0f;
        const float t = apf - 1.0f;
        int i = (int)t;
        float fr = t - (float)i;
        if (i >= T.npf - 1) { i = T.npf - 1; fr = 0.0f; }
        const float *m = T.mu, *p = T.p(i);
        // (the grid's ends are -1 and 1 exactly, build_tables: a cosine clamped just inside them needs no branch for the ends -- at -1
        //  the search ends on node 0 with f = 0, just below 1 on the last interval with f = 1 to seven digits)
        const float mc = fminf(fmaxf(mu, -1.0f), 0.99999994f);
        const int lo = lean_tab_find(m, T.mu_idx(), T.nang, mc, tab_bucket_mu(mc));
        const float f = (mc - m[lo]) * frcp(m[lo + 1] - m[lo]);
        float pv = fmaf(f, p[lo + 1] - p[lo], p[lo]);
        if (fr > 0.0f) { const float *q = p + T.nang; pv = (1.0f - fr) * pv + fr * fmaf(f, q[lo + 1] - q[lo], q[lo]); }
        return pv;
    }
    return phase_eval_analytic(apf, mu);
}

__device__ inline float lean_table_sample(const LeanTab &T, const int it, const float u) {
    const float *m = T.mu, *p = T.p(it), *cdf = T.cdf(it);
    const int lo = lean_tab_find(cdf, T.cdf_idx(it), T.nang, u, tab_bucket_u(u));
    const float r = 2.0f * (u - cdf[lo]);
    const float dm = m[lo + 1] - m[lo];
    const float sl = (p[lo + 1] - p[lo]) * frcp(dm);
    const float disc = fmaxf(fmaf(p[lo], p[lo], 2.0f * sl * r), 0.0f);
    const float den = p[lo] + fsqrt(disc);
    const float t = den > 0.0f ? 2.0f * r * frcp(den) : 0.0f;
    return fminf(m[lo] + t, m[lo + 1]);
}

// the cosine of the scattering angle drawn from selector apf with the uniform number u (usel: which of two mixed tables)
__device__ inline float lean_phase_sample(const LeanTab &T, const float apf, const float u, const float usel) {
    if (apf >= 1.0f) {
        if (T.npf <= 0) return 2.0f * u - 1.0f;
        const float t = apf - 1.0f;
        int i = (int)t;
        float fr = t - (float)i;
        if (i >= T.npf - 1) { i = T.npf - 1; fr = 0.0f; }
        if (fr > 0.0f && usel < fr) i += 1;
        return lean_table_sample(T, i, u);
    }
    return phase_sample_analytic(apf, u);
}

// A collision in a mixture of np1d 1-D constituents (layer record Lk) and up to two 3-D ones (k3, apf3; kb, apfb; 0 where there is
// none): the mixture's phase function towards cosine mu (sum of k_i P_i: the caller divides by the total scattering coefficient) ...
__device__ inline float lean_mix_phase(const LeanTab &T, const LayerRec &Lk, const int np1d, const float k3, const float apf3, const float kb, const float apfb,
                                       const float mu) {
    float P = 0.0f;
    for (int ip = 0; ip < np1d; ++ip) { const float ks = Lk.ks1d[ip]; if (ks > 0.0f) P += ks * lean_phase_eval(T, Lk.apf1d[ip], mu); }
    if (k3 > 0.0f) P += k3 * lean_phase_eval(T, apf3, mu);
    if (kb > 0.0f) P += kb * lean_phase_eval(T, apfb, mu);
    return P;
}
// ... and the constituent that scatters, chosen by u1 in proportion to the scattering coefficients, the 1-D ones first, then the 3-D ones in
// their order (the rule of k_transport, block B5, and of the oracle): its selector; usel: where u1 fell inside its share (mixed tables)
__device__ inline float lean_mix_select(const LayerRec &Lk, const int np1d, const float k3, const float apf3, const float kb, const float apfb, const int n3,
                                        const float u1, const float kstot, float &usel) {
    const float target = u1 * kstot;
    float cum = 0.0f, apf_sel = -2.0f;
    bool found = false;
    usel = 0.0f;
    const int ncomp = np1d + n3;
    for (int q = 0; q < ncomp; ++q) {
        const float ks = q < np1d ? Lk.ks1d[q] : (q == np1d ? k3 : kb);
        const float apf = q < np1d ? Lk.apf1d[q] : (q == np1d ? apf3 : apfb);
        if (!found && (target < cum + ks || q == ncomp - 1)) {
            found = true;
            apf_sel = apf;
            usel = ks > 0.0f ? (target - cum) * frcp(ks) : 0.0f;
        }
        cum += ks;
    }
    usel = fminf(fmaxf(usel, 0.0f), 1.0f);
    return apf_sel;
}

// Henyey-Greenstein alone (the selector is known to lie in (-1, 1))
__device__ inline float phase_eval_hg(float g, float mu) {
    const float r = frsq(1.0f + g * g - 2.0f * g * mu);
    return (1.0f - g * g) * r * r * r;
}

// sin and cos of 2*pi*u for u in (0,1): hardware v_sin_f32 / v_cos_f32 take the angle in turns
__device__ inline void sincos_turns(float u, float &s, float &c) {
    s = __builtin_amdgcn_sinf(u);
    c = __builtin_amdgcn_cosf(u);
}

// rotate (ux,uy,uz) by polar cosine mu and azimuth 2*pi*uphi
__device__ inline void rotate_dir(float &ux, float &uy, float &uz, float mu, float uphi) {
    const float st = fsqrt(fmaxf(0.0f, 1.0f - mu * mu));
    float sp, cp;
    sincos_turns(uphi, sp, cp);
    const float den2 = 1.0f - uz * uz;
    float nx, ny, nz;
    if (den2 < 1e-10f) {
        const float sg = uz >= 0.0f ? 1.0f : -1.0f;
        nx = st * cp; ny = st * sp; nz = mu * sg;
    } else {
        const float iden = frsq(den2), den = den2 * iden;
        nx = st * (ux * uz * cp - uy * sp) * iden + ux * mu;
        ny = st * (uy * uz * cp + ux * sp) * iden + uy * mu;
        nz = -st * cp * den + uz * mu;
    }
    const float n = frsq(nx * nx + ny * ny + nz * nz);
    ux = nx * n; uy = ny * n; uz = nz * n;
}

// ---------------------------------------------------------------------------------------------
// surface: Ross-Thick / Li-Sparse-Reciprocal reflectance factor (BRDF = R/pi)
// ---------------------------------------------------------------------------------------------
// (hardware reciprocal / square root, 1 ulp, instead of the IEEE sequences; sin(acos x) written as sqrt(1 - x^2): the function is
//  ~400 instructions otherwise, and every reflection off such a surface evaluates it once per view and once for the photon)
__device__ __noinline__ float lsrt_R(float fiso, float fgeo, float fvol, float dix, float diy, float diz,
                                     float dox, float doy, float doz) {
    const float ci = fmaxf(-diz, 1e-6f), cv = fmaxf(doz, 1e-6f);
    const float si = fsqrt(fmaxf(0.0f, 1.0f - ci * ci)), sv = fsqrt(fmaxf(0.0f, 1.0f - cv * cv));
    float cphi = 1.0f;
    const float hi2 = dix * dix + diy * diy, hv2 = dox * dox + doy * doy;
    if (hi2 > 1e-24f && hv2 > 1e-24f) cphi = (-dix * dox - diy * doy) * frsq(hi2) * frsq(hv2);
    cphi = fminf(fmaxf(cphi, -1.0f), 1.0f);
    const float sphi2 = 1.0f - cphi * cphi;
    const float cxi = fminf(fmaxf(ci * cv + si * sv * cphi, -1.0f), 1.0f);
    const float xi = acosf(cxi), sxi = fsqrt(fmaxf(1.0f - cxi * cxi, 0.0f));
    const float seci = frcp(ci), secv = frcp(cv);
    const float kvol = ((0.5f * kPi - xi) * cxi + sxi) * frcp(ci + cv) - 0.25f * kPi;
    const float ti = si * seci, tv = sv * secv;
    const float D2 = fmaxf(ti * ti + tv * tv - 2.0f * ti * tv * cphi, 0.0f);
    const float cost = fminf(2.0f * fsqrt(D2 + ti * ti * tv * tv * sphi2) * frcp(seci + secv), 1.0f);
    const float t = acosf(cost), sint = fsqrt(fmaxf(1.0f - cost * cost, 0.0f));
    const float O = (t - sint * cost) * (seci + secv) * (1.0f / kPi);
    const float kgeo = O - seci - secv + 0.5f * (1.0f + cxi) * seci * secv;
    return fmaxf(fiso + fgeo * kgeo + fvol * kvol, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// surface: diffuse-specular mixture (jsfc = 2), parameters (diffuse albedo, diffuse fraction, Re m, Im m, slope variance)
// as packed by er3t/rtm/mca/mca_sfc.py:119-128.  Formulation: oracle/mi3d_oracle.c, dsm_R (Cox-Munk facets, Fresnel,
// Smith shadowing).
// ---------------------------------------------------------------------------------------------
__device__ inline float fresnel_unpolarised(float nr, float ni, float c) {
    c = fminf(fmaxf(c, 1e-6f), 1.0f);
    const float s2 = 1.0f - c * c;
    const float u = nr * nr - ni * ni - s2, v = sqrtf(u * u + 4.0f * nr * nr * ni * ni);
    const float a2 = fmaxf(0.5f * (v + u), 0.0f), b2 = fmaxf(0.5f * (v - u), 0.0f);
    const float a = sqrtf(a2);
    const float rs = ((a - c) * (a - c) + b2) / ((a + c) * (a + c) + b2);
    const float q = s2 / c;
    const float rp = rs * ((a - q) * (a - q) + b2) / ((a + q) * (a + q) + b2);
    return 0.5f * (rs + rp);
}

__device__ inline float dsm_shadow_lambda(float mu, float sigma) {
    if (mu >= 1.0f) return 0.0f;
    const float nu = mu / (sigma * sqrtf(1.0f - mu * mu));
    return 0.5f * (expf(-nu * nu) / (1.7724539f * nu) - erfcf(nu));
}

__device__ __noinline__ float dsm_R(float ad, float fd, float nr, float ni, float s2, float dix, float diy, float diz,
                                    float dox, float doy, float doz) {
    fd = fminf(fmaxf(fd, 0.0f), 1.0f);
    ad = fminf(fmaxf(ad, 0.0f), 1.0f);
    float R = fd * ad;
    const float mi = -diz, mv = doz;
    if (s2 > 0.0f && fd < 1.0f && mi > 1e-6f && mv > 1e-6f) {
        const float hx = dox - dix, hy = doy - diy, hz = doz - diz;     // towards the viewer + towards the source
        const float hn = sqrtf(hx * hx + hy * hy + hz * hz);
        if (hn > 1e-12f) {
            const float mun = hz / hn;
            const float cchi = (-dix * hx - diy * hy - diz * hz) / hn;
            if (mun > 1e-6f) {
                const float mun2 = mun * mun;
                const float P = expf(-(1.0f - mun2) / (mun2 * s2)) / (kPi * s2);
                const float sg = sqrtf(s2);
                const float S = 1.0f / (1.0f + dsm_shadow_lambda(mi, sg) + dsm_shadow_lambda(mv, sg));
                R += (1.0f - fd) * kPi * fresnel_unpolarised(nr, ni, cchi) * P * S / (4.0f * mi * mv * mun2 * mun2);
            }
        }
    }
    return fmaxf(R, 0.0f);
}

struct Sfc { int type; float p0, p1, p2, p3, p4; };

__device__ inline float surface_R(const Sfc &sf, float dix, float diy, float diz, float dox, float doy, float doz) {
    if (sf.type == MI3D_SFC_LSRT) return lsrt_R(sf.p0, sf.p1, sf.p2, dix, diy, diz, dox, doy, doz);
    if (sf.type == MI3D_SFC_DSM) return dsm_R(sf.p0, sf.p1, sf.p2, sf.p3, sf.p4, dix, diy, diz, dox, doy, doz);
    return fminf(fmaxf(sf.p0, 0.0f), 1.0f);
}

} // namespace mi3d
