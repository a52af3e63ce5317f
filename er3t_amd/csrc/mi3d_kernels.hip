// mi3d_kernels.hip — CDNA4 (gfx950) kernels of the photon-transport hot path.
//
// Replaces the main loop of the external solver the reference launches as
// "<exe> <Nphoton> <solver> <inp> <out>" (er3t/rtm/mca/mca_run.py:113).  No reference kernels
// exist to mirror; the algorithm is the forward Monte Carlo / local-estimate method the
// reference cites (er3t/rtm/mca/mcarats.py:59) on er3t's input contract (see mi3d_device.h).
//
// Kernels
//   k_build_grid     file-layout 3-D arrays -> z-fastest total extinction + collision records
//   k_layer_uniform  per 3-D layer: is the total extinction horizontally uniform? (such layers are
//                    flown through like 1-D layers: no voxel walk, no extinction reads)
//   k_build_column   per-column optical depth from every 3-D level up to the top of atmosphere
//   k_transport      persistent photon loop (see the comment on the kernel)
//   k_stats_*        per-run g-sum and sum / sum of squares over runs of the result fields
//   k_philox         test hook
//
// Random-number protocol, geometry and estimator are specified in DESIGN.md §3 and restated
// independently (double precision) in oracle/mi3d_oracle.c.
#include "mi3d_device.h"
#include "mi3d_diag.h"

namespace mi3d {

// ---------------------------------------------------------------------------------------------
// scene builders
// ---------------------------------------------------------------------------------------------
// One thread per voxel, x fastest on the read side (coalesced reads of the file-layout arrays).
__global__ void __launch_bounds__(256)
k_build_grid(int nx, int ny, int nz3, int k3lo, int np3d, const float *bt1d, const float *abst,
             const float *extp, const float *omgp, const float *apfp, float4 *vrec, float2 *csca, unsigned vcol_f4, unsigned vrow_f4, float *bext3) {
    const long nvox = (long)nx * ny * nz3;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x; // file index: (k3*ny + iy)*nx + ix
    if (v >= nvox) return;
    const int ix = (int)(v % nx);
    const int iy = (int)((v / nx) % ny);
    const int k3 = (int)(v / ((long)nx * ny));
    float bt = bt1d[k3lo + k3];
    if (abst) bt += abst[v];
    const long o = ((long)iy * nx + ix) * nz3 + k3;
    float ks0 = 0.0f, apf0 = 0.0f;
    for (int ip = 0; ip < np3d; ++ip) {
        const float e = extp[ip * nvox + v];
        bt += e;
        float2 c = make_float2(omgp[ip * nvox + v] * e, apfp[ip * nvox + v]);
        if (!(c.x > 0.0f)) c.y = 0.0f;   // (a constituent that never scatters here: its selector is never used, and 0 is harmless where
                                         //  the lean kernels evaluate Henyey-Greenstein without looking -- er3t writes -1 into clear voxels)
        if (ip == 0) { ks0 = c.x; apf0 = c.y; }
        if (np3d > 1) csca[o * np3d + ip] = c;
    }
    vrec[(size_t)iy * vrow_f4 + (size_t)ix * vcol_f4 + k3] = make_float4(fmaxf(bt, 0.0f), 0.0f, ks0, apf0); // .y is filled by k_build_column
    bext3[o] = fmaxf(bt, 0.0f);
}

// One block per 3-D layer: min and max of the total extinction over the layer (same expression
// and operation order as k_build_grid, so "min == max" means bext is constant there).
__global__ void __launch_bounds__(256)
k_layer_uniform(int nx, int ny, int nz3, int k3lo, int np3d, const float *bt1d, const float *abst,
                const float *extp, float *bmin, float *bmax) {
    const int k3 = blockIdx.x;
    const long ncol = (long)nx * ny, nvox = ncol * nz3;
    float lo = 3.0e38f, hi = 0.0f;
    for (long c = threadIdx.x; c < ncol; c += blockDim.x) {
        const long v = (long)k3 * ncol + c;
        float bt = bt1d[k3lo + k3];
        if (abst) bt += abst[v];
        for (int ip = 0; ip < np3d; ++ip) bt += extp[ip * nvox + v];
        bt = fmaxf(bt, 0.0f);
        lo = fminf(lo, bt); hi = fmaxf(hi, bt);
    }
    __shared__ float slo[256], shi[256];
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fminf(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmaxf(shi[threadIdx.x], shi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { bmin[k3] = slo[0]; bmax[k3] = shi[0]; }
}

// One thread per column: vertical optical depth from the top face of every voxel (and from the bottom of the
// 3-D region) up to TOA.
__global__ void __launch_bounds__(256)
k_build_column(int ncol, int nz3, int k3lo, int nz, const float *bt1d, const float *dz, float4 *vrec,
               float *tcol0, int nx, unsigned vcol_f4, unsigned vrow_f4) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int k3hi = k3lo + nz3;
    float tau = 0.0f;
    for (int k = nz - 1; k >= k3hi; --k) tau += bt1d[k] * dz[k];
    for (int k3 = nz3 - 1; k3 >= 0; --k3) {
        float4 *r = vrec + (size_t)(c / nx) * vrow_f4 + (size_t)(c % nx) * vcol_f4 + k3;
        r->y = tau;                       // optical depth above the top face of this voxel
        tau += r->x * dz[k3lo + k3];
    }
    tcol0[c] = tau;
}

// Range of tabulated phase functions the 3-D constituents refer to (apf >= 1 where there is extinction):
// out[0] = min table index, out[1] = max table index (0-based, fractional selectors count both neighbours);
// out[2] = 1 if some constituent with extinction is isotropic or Rayleigh (apf <= -1), i.e. not Henyey-Greenstein.
__global__ void __launch_bounds__(256)
k_apf_range(long n, const float *extp, const float *apfp, int *out) {
    int lo = 1 << 30, hi = -1, other = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = apfp[i];
        if (a >= 1.0f && extp[i] > 0.0f) {
            const float t = a - 1.0f;
            const int i0 = (int)floorf(t);
            lo = min(lo, i0);
            hi = max(hi, t > (float)i0 ? i0 + 1 : i0);
        }
        if (!(a > -1.0f) && extp[i] > 0.0f) other = 1;
    }
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); other |= __shfl_down(other, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (hi >= 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
        if (other) atomicOr(&out[2], 1);
    }
}

// Which XCDs does this device (or partition of one: CPX / QPX modes) run workgroups on?  Every block raises the flag of the
// XCD it finds itself on.  The event lists of the marched views are per XCD (k_transport_lean<.,.,2>): how many of the eight
// fill decides how many photons a launch may take.
__global__ void k_xcc_census(unsigned *flags) {
    if (threadIdx.x == 0) flags[__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u] = 1u;
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
// photon order: counting sort of a launch's photon indices by the tile of the domain they start in
// ---------------------------------------------------------------------------------------------
// A photon's start position is a function of its id alone (Philox block 0), so the ORDER in which a launch works through
// its ids is free.  Taken in id order, the 41 000 photons an XCD has in flight are spread over the whole domain and their
// voxel reads miss the XCD's 4 MiB L2 (92 MB of cloudy voxel records on the 480 x 480 x 100 scene): the walk then runs at
// the chip's random-gather rate from the Infinity Cache (profiles/r02/gather_ceiling.log).  Sorted by start tile and handed
// out in eight contiguous pieces, one per XCD (k_transport, block B4), an XCD works on one tile of a few dozen columns at a
// time, whose records stay in its L2.  Photon id -> history is untouched; only the order of the sums changes.
struct BinGeom {
    float Lx, Ly, inv_tx, inv_ty;   // domain size, 1 / tile edge [1/m]
    int nx, ny;
    int tcols, ntx, nty;   // tile edge in columns, tiles per row / column of tiles (ntx * nty <= kMaxTiles)
};
constexpr int kMaxTiles = 1024;

__device__ inline int launch_tile(const BinGeom G, uint64_t seed, uint64_t id) {
    uint32_t w[4];
    philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    float x = u01(w[0]) * G.Lx, y = u01(w[1]) * G.Ly;      // as the D_LAUNCH branch of k_transport
    if (x >= G.Lx) x = 0.0f;
    if (y >= G.Ly) y = 0.0f;
    // (the tile straight from the position, no integer division: a sort key for locality -- a photon that the rounding puts into
    //  the neighbouring tile is as well off there)
    const int tx = min((int)(x * G.inv_tx), G.ntx - 1), ty = min((int)(y * G.inv_ty), G.nty - 1);
    return ty * G.ntx + ((ty & 1) ? G.ntx - 1 - tx : tx);   // boustrophedon: consecutive tiles are neighbours
}

// pass 1: tile of every photon index (kept, 2 bytes each) and the histogram over tiles
__global__ void __launch_bounds__(256)
k_bin_count(const BinGeom G, uint64_t seed, uint64_t offset, uint32_t n, uint16_t *tile, uint32_t *hist) {
    __shared__ uint32_t lh[kMaxTiles];
    const int nt = G.ntx * G.nty;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) lh[i] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = launch_tile(G, seed, offset + i);
        tile[i] = (uint16_t)t;
        atomicAdd(&lh[t], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// pass 2 (one block): exclusive scan of the histogram -> where each tile's piece of the order begins.  256 threads, four tiles
// each: one wave per SIMD (a block of 1024 threads finds no room on a compute unit while workgroups of a photon loop are resident).
__global__ void __launch_bounds__(256)
k_bin_scan(int nt, const uint32_t *hist, uint32_t *cursor) {
    static_assert(kMaxTiles == 4 * 256, "four tiles per thread");
    __shared__ uint32_t s[256];
    const int t = threadIdx.x;
    uint32_t h[4], sum = 0u;
    for (int j = 0; j < 4; ++j) { h[j] = 4 * t + j < nt ? hist[4 * t + j] : 0u; sum += h[j]; }
    s[t] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    uint32_t run = s[t] - sum;
    for (int j = 0; j < 4; ++j) { if (4 * t + j < nt) cursor[4 * t + j] = run; run += h[j]; }
}

// pass 3: every block takes a contiguous slab of indices and reserves room for it in each tile's piece (one returning atomic per
// tile and block); then, 4096 indices at a time, a counting sort INSIDE LDS -- counts per tile, scan, places -- and the sorted chunk is
// copied out, the indices of a tile as one contiguous run.  (Written straight to their places the indices went out as 4-byte stores
// to as many regions as there are tiles: 1.2 ms per 5e8 indices over 64 tiles, 3.5 ms over 100.)  The cursors end up as the tiles' ENDS
// in the order: the lean loop's tally window reads them (DevCold::tile_end).
constexpr int kScatterR = 16;   // indices per thread and chunk
__global__ void __launch_bounds__(256)
k_bin_scatter(int nt, uint32_t n, uint32_t slab, const uint16_t *tile, uint32_t *cursor, uint32_t *order) {
    constexpr int NT = 256, CH = NT * kScatterR;
    __shared__ uint32_t lcount[kMaxTiles], lstart[kMaxTiles], gbase[kMaxTiles], part[NT / 64];
    __shared__ uint32_t sidx[CH];
    __shared__ uint16_t stile[CH];
    const unsigned tid = threadIdx.x;
    for (int i = tid; i < nt; i += NT) lcount[i] = 0u;
    __syncthreads();
    const uint32_t lo = blockIdx.x * slab, hi = min(n, lo + slab);
    for (uint32_t i = lo + tid; i < hi; i += NT) atomicAdd(&lcount[tile[i]], 1u);
    __syncthreads();
    for (int i = tid; i < nt; i += NT) gbase[i] = lcount[i] ? atomicAdd(&cursor[i], lcount[i]) : 0u;   // the slab's first slot in that tile
    // (every thread owns `per` consecutive tiles: it zeroes, scans and moves on the counters of those)
    const int per = (nt + NT - 1) / NT;
    const int tlo = min((int)tid * per, nt), thi = min(tlo + per, nt);
    __syncthreads();
    for (uint32_t c0 = lo; c0 < hi; c0 += CH) {
        for (int i = tlo; i < thi; ++i) lcount[i] = 0u;
        __syncthreads();
        uint32_t t[kScatterR];
#pragma unroll
        for (int r = 0; r < kScatterR; ++r) {
            const uint32_t i = c0 + (uint32_t)(r * NT) + tid;
            t[r] = i < hi ? (uint32_t)tile[i] : 0xffffffffu;
            if (t[r] != 0xffffffffu) atomicAdd(&lcount[t[r]], 1u);
        }
        __syncthreads();
        // exclusive scan of the counts: inside the wave by shuffles, across the waves through LDS
        uint32_t sum = 0;
        for (int i = tlo; i < thi; ++i) sum += lcount[i];
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
        for (int i = tlo; i < thi; ++i) {
            const uint32_t c = lcount[i];
            lstart[i] = run; lcount[i] = run;
            run += c;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kScatterR; ++r)
            if (t[r] != 0xffffffffu) {
                const uint32_t p = atomicAdd(&lcount[t[r]], 1u);
                sidx[p] = c0 + (uint32_t)(r * NT) + tid;
                stile[p] = (uint16_t)t[r];
            }
        __syncthreads();
        for (uint32_t j = tid; j < total; j += NT) {
            const uint32_t tt = stile[j];
            order[gbase[tt] + (j - lstart[tt])] = sidx[j];
        }
        __syncthreads();
        // (lcount[i] is where tile i ends in the sorted chunk by now)
        for (int i = tlo; i < thi; ++i) gbase[i] += lcount[i] - lstart[i];
    }
}

// End of a run of launches: add the accumulation image (one pixel per 128-byte line, rows of `rad_row` pixels of which the first nxr
// are used) into the tally and clear it.
__global__ void __launch_bounds__(256)
k_fold_rad(tally_t *__restrict__ acc, tally_t *__restrict__ tally, int stride, int n, int nxr, int rad_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t a = ((size_t)(i / nxr) * rad_row + (size_t)(i % nxr)) * stride;
    const tally_t v = acc[a];
    if (v != 0.0) { tally[i] += v; acc[a] = 0.0; }
}

// ---------------------------------------------------------------------------------------------
// run statistics (the reduction the reference's reader does on the host: mca_out.py:313-352, 438-500)
// ---------------------------------------------------------------------------------------------
// run_acc += factor[level] * (float)(tally * norm): float32 product and sum, in that order, like
// `raw.data*factors` followed by `+=` on float32 arrays.  `plane` elements share one factor;
// `nlevel` factors repeat (flux: 3 variables x (nz+1) levels).  down_lo >= 0 marks a flux tally: elements
// [down_lo, 2 down_lo) hold the diffuse downward flux, to which the direct beam [0, down_lo) is added.
__global__ void __launch_bounds__(256)
k_stats_add(const tally_t *__restrict__ tally, float *__restrict__ run_acc, const float *__restrict__ factor,
            double norm, int plane, int nlevel, int down_lo, const double *__restrict__ dir_level, int n) {
#pragma clang fp contract(off) // a fused multiply-add would round once where numpy rounds twice
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = tally[i];
    if (down_lo >= 0 && i >= down_lo && i < 2 * down_lo) t += tally[i - down_lo]; // total-down = direct + diffuse
    // (flux: the analytic direct beam of the levels above the 3-D region joins the direct and the total downward flux)
    const double a = (dir_level && i < 2 * down_lo) ? dir_level[(i / plane) % nlevel] : 0.0;
    const float v = (float)(t * norm + a);
    const float f = factor[(i / plane) % nlevel];
    const float prod = v * f;
    run_acc[i] = run_acc[i] + prod;
}

// mi3d_get_flux / mi3d_get_heating on the device: the raw float64 tallies normalised to the float32 fields of the output file, so that 4
// bytes per cell cross to the host instead of 8 and no host loop runs over millions of cells (a flux job's read-back took longer than its
// photons: 9.5 ms per job of 6e6 photons on 128 x 128 x 69).  The operations of the host code they replace, in its order, unfused.
//   flux (down_lo = cells per plane-block = n / 3): planes direct-down, DIFFUSE-down, up -> direct-down, TOTAL-down, up; `add`[level] is the
//   analytic direct beam of the levels above the 3-D region (or NULL);   heating (down_lo < 0): `add`[layer] is the layer's thickness, divided by;
//   radiance (down_lo < 0, no `add`): scaled.
__global__ void __launch_bounds__(256)
k_get_field(const tally_t *__restrict__ tally, float *__restrict__ out, double norm, unsigned plane, unsigned nlevel, long down_lo,
            const double *__restrict__ add, size_t n) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = tally[i];
    const unsigned lev = (unsigned)((i / plane) % nlevel);
    if (down_lo >= 0) {
        if (i >= (size_t)down_lo && i < 2 * (size_t)down_lo) t += tally[i - (size_t)down_lo];
        const double a = (add && i < 2 * (size_t)down_lo) ? add[lev] : 0.0;
        out[i] = (float)(t * norm + a);
    } else if (add) out[i] = (float)(t * norm / add[lev]);
    else out[i] = (float)(t * norm);      // (radiance: a scale)
}

// End of a run: fold the run's field into the sum and the sum of squares over runs (float64).
__global__ void __launch_bounds__(256)
k_stats_fold(float *__restrict__ run_acc, double *__restrict__ sum, double *__restrict__ sumsq, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = (double)run_acc[i];
    sum[i] += x;
    sumsq[i] += x * x;
    run_acc[i] = 0.0f;
}

// mean and population standard deviation over runs (numpy.std, ddof = 0)
__global__ void __launch_bounds__(256)
k_stats_final(const double *__restrict__ sum, const double *__restrict__ sumsq, float *__restrict__ mean,
              float *__restrict__ sdev, double inv_nrun, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double m = sum[i] * inv_nrun;
    const double v = sumsq[i] * inv_nrun - m * m;
    mean[i] = (float)m;
    sdev[i] = (float)sqrt(v > 0.0 ? v : 0.0);
}

// ---------------------------------------------------------------------------------------------
// transport
// ---------------------------------------------------------------------------------------------
// Lane modes.  A lane either walks a ray through voxels (phase A) or waits for phase B to serve it.
enum : int {
    M_FLY = 0,     // photon walking voxels of a horizontally varying layer
    M_LE = 1,      // local-estimate ray walking voxels towards the sensor of view `iv`
    M_UNIF = 2,    // photon inside a run of horizontally uniform layers: crossed in one go by phase B
    M_LEUNIF = 3,  // local-estimate ray inside such a run
    M_COLL = 4,    // photon stopped at a collision
    M_SURF = 5,    // photon arrived at the surface
    M_LEEND = 6,   // local-estimate ray finished: tally, then next view
    M_VIEWS = 7,   // event has views left whose rays must be marched
    M_FINISH = 8,  // all views served: new direction and weight
    M_NEED = 9,    // history over: take the next photon id
    M_DRAW = 10,   // needs its next Philox block (what for: `kind`)
    M_DONE = 11    // no photons left
};
enum : int { E_SCATTER = 0, E_SURFACE = 1, E_LAUNCH = 2,      // kind of event being finished
             D_FLIGHT = 3, D_ROULETTE = 4, D_LAUNCH = 5 };    // what the pending draw is for

#ifndef MI3D_THRESH
#define MI3D_THRESH 16   // phase A keeps stepping while at least this many lanes of the wave are in flight
#endif
#ifndef MI3D_LEAN
#define MI3D_LEAN 3      // every third pass of phase B is a full one (1: every pass), see the comment at the top of phase B
#endif
#ifndef MI3D_LEAN_MARCH
#define MI3D_LEAN_MARCH 2 // builds with marched views: every second pass serves the photons' events, every pass the rays
#endif
#ifndef MI3D_WAVES
// __launch_bounds__ second argument: minimum waves per SIMD the register budget must allow.  Five waves (<= 96 VGPRs)
// is what the builds without marched views need anyway give or take a register; the marched builds carry more state
// and would spill, so they stay at four.
#define MI3D_WAVES(MARCH, COUNT) (((MARCH) || (COUNT)) ? 4 : 5)   // (the instrumented builds carry their counters)
#endif
#ifdef MI3D_ABL_NOTALLY
#define RAD_ADD(ptr, val) asm volatile("" ::"v"(val), "v"(ptr))
#else
#define RAD_ADD(ptr, val) atomicAdd(ptr, (tally_t)(val))
#endif
#ifndef MI3D_CHUNK
#define MI3D_CHUNK 64    // (256 before: the last chunk of a wave is four photons per lane of tail; 64 costs nothing on long launches, profiles/r02/small_launches.log)
#endif
constexpr unsigned kChunk = MI3D_CHUNK; // photon ids a wave takes from the global counter at a time

struct Counters {
    uint32_t steps, steps3d, scatter, surface, le_rays, le_steps, le_steps3d, le_column, flux_tally,
        roulette, killed, escaped, absorbed, photons;
    // scheduler diagnostics (instrumented build): lane-iterations spent stepping / serving events
    uint32_t a_lanes, a_slots, b_lanes, b_slots;
    // wave clock ticks (s_memtime / 64) spent in: phase A, B0, B1+B2, B3+B4, B5, B6
    uint32_t cyc[6];
};

// i mod n for |i| < 2^23 without an integer division (tens of VALU instructions on this hardware)
__device__ inline int wrapi(int i, int n, float inv_n) {
    i -= (int)floorf((float)i * inv_n) * n;
    if (i < 0) i += n;
    if (i >= n) i -= n;
    return i;
}

// Fold an unbounded local position back into its cell, moving the column index with it.
__device__ inline void fold_xy_raw(float dx, float dy, int nx, int ny, float inv_dx, float inv_dy, float inv_nx, float inv_ny,
                                   float &px, float &py, int &ix, int &iy, bool ipa) {
    const float fx = floorf(px * inv_dx), fy = floorf(py * inv_dy);
    px = fminf(fmaxf(px - fx * dx, 0.0f), dx);
    py = fminf(fmaxf(py - fy * dy, 0.0f), dy);
    if (!ipa) {
        if (fx != 0.0f) ix = wrapi(ix + (int)fx, nx, inv_nx);
        if (fy != 0.0f) iy = wrapi(iy + (int)fy, ny, inv_ny);
    }
}
__device__ inline void fold_xy(const DevScene &S, const DevCold *C, float &px, float &py, int &ix, int &iy, bool ipa) {
    fold_xy_raw(S.dx, S.dy, S.nx, S.ny, C->inv_dx, C->inv_dy, C->inv_nx, C->inv_ny, px, py, ix, iy, ipa);
}

__device__ inline Sfc load_sfc(const DevScene &S, const DevCold *C, int ix, int iy, float px, float py) {
    Sfc sf;
    sf.type = C->sfc_mtype; sf.p0 = C->sfc_p0; sf.p1 = C->sfc_p1; sf.p2 = C->sfc_p2; sf.p3 = C->sfc_p3; sf.p4 = C->sfc_p4;
    const float *map = C->sfc2d;
    if (map) {
        const float xa = (float)ix * S.dx + px, ya = (float)iy * S.dy + py;
        const int ib = min(max((int)(xa * C->sfc_sx), 0), C->nxb - 1);
        const int jb = min(max((int)(ya * C->sfc_sy), 0), C->nyb - 1);
        const float4 q = *reinterpret_cast<const float4 *>(map + (unsigned)((jb * C->nxb + ib) * 8));
        sf.type = (int)(q.x + 0.5f); sf.p0 = q.y; sf.p1 = q.z; sf.p2 = q.w;
        sf.p3 = 0.0f; sf.p4 = 0.0f;
        if (sf.type == MI3D_SFC_DSM) {   // the only model with more than three parameters
            const float2 r = *reinterpret_cast<const float2 *>(map + (unsigned)((jb * C->nxb + ib) * 8 + 4));
            sf.p3 = r.x; sf.p4 = r.y;
        }
    }
    return sf;
}

#ifdef MI3D_CENSUS
// Census build (make EXTRA=-DMI3D_CENSUS; tools/census.py): per tally instruction, how many lanes take part and how many DISTINCT
// addresses they add to -- what a wave-level combine (shuffle / ballot, north_star) could save.  The instrumented build's counters
// ticks_b34 / ticks_b5 / ticks_b6 carry the number of tally instructions and the two sums instead of clock ticks.
__device__ inline void tally_census(const void *addr, Counters &cnt) {
    const unsigned long long m = __ballot(1);
    const unsigned long long a = (unsigned long long)addr;
    unsigned nd = 0;
    unsigned long long left = m;
    while (left) {
        const int l = __ffsll((long long)left) - 1;
        const unsigned long long a0 = __shfl(a, l, 64);
        left &= ~__ballot(a == a0);
        nd++;
    }
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) { cnt.cyc[3] += 1u; cnt.cyc[4] += (uint32_t)__popcll(m); cnt.cyc[5] += nd; }
}
#define MI3D_TALLY_CENSUS(ptr) do { if (COUNT) tally_census(ptr, cnt); } while (0)
#else
#define MI3D_TALLY_CENSUS(ptr)
#endif

template <bool COUNT>
__device__ inline void flux_add(const DevScene &S, int ix, int iy, float w, bool direct, int level, bool up,
                                Counters &cnt) {
    const unsigned plane = (unsigned)(S.nx * S.ny), nlev = (unsigned)(S.nz + 1);
    const unsigned i = (unsigned)((level * S.ny + iy) * S.nx + ix);
    // raw tally planes: 0 direct-down, 1 diffuse-down, 2 up -- one atomic per crossing; total-down = 0 + 1 is formed when
    // the result is read (mi3d_get_flux, mi3d_stats_add)
    MI3D_TALLY_CENSUS(&S.flux[(up ? 2u : (direct ? 0u : 1u)) * nlev * plane + i]);
#ifdef MI3D_ABL_NOFLUXATOMIC   // ablation (no result): the flux kernel without its tallies
    asm volatile("" :: "v"((up ? 2u : (direct ? 0u : 1u)) * nlev * plane + i), "v"(w));
#else
    atomicAdd(&S.flux[(up ? 2u : (direct ? 0u : 1u)) * nlev * plane + i], (tally_t)w);
#endif
    if (COUNT) cnt.flux_tally++;
}

// Persistent photon loop.  256-thread workgroups, grid = a few workgroups per CU; lanes pull photon
// ids from a wave-local pool refilled from one global counter, so wavefronts stay full to the end.
//
// Histories differ wildly in length and so do the flights between two events, so the loop is a
// two-phase state machine instead of nested per-photon loops:
//   phase A  every lane whose ray (photon or local-estimate ray: same code) is inside a horizontally
//            varying layer advances ONE voxel; repeated while at least MI3D_THRESH lanes are walking.
//            Nothing else lives in this loop: one extinction read, three face distances, one move.
//   phase B  everything rarer, each block executed once per pass for the lanes that need it: runs of
//            horizontally uniform layers (crossed in one go from LDS prefix sums), collisions and
//            surface hits (weight, local estimates), new photons, and ONE shared finish block (one
//            direction rotation) followed by ONE shared Philox block, so the rare kinds of event
//            carry no private copies of the expensive code.
// The layer table, the views and a per-lane stash for the event state live in LDS.
// Compile-time specialisations: COUNT (instrumented build), MARCH (some view needs its local-estimate ray marched
// cell by cell; without it the LE-ray modes, their state and the stash vanish), FLUX (flux tallies on).
// P3D (partial 3-D solver): the direct beam travels in 3-D up to its first event, everything after it (scattered photons
// and every local-estimate ray) stays in the column of that event like under the independent-pixel approximation.
template <bool COUNT, bool MARCH, bool FLUX, bool P3D>
__global__ void __launch_bounds__(256, MI3D_WAVES(MARCH, COUNT))
k_transport(const DevScene S, const uint64_t nphoton, const uint64_t seed, const uint64_t offset) {
    extern __shared__ float4 smem[];
    const LayerRec *lay = reinterpret_cast<const LayerRec *>(smem);
    const float4 *lay4 = smem;
    const ViewRec *views = reinterpret_cast<const ViewRec *>(smem + S.nz * (kLayStride / 4));
    const DevCold *cold = reinterpret_cast<const DevCold *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2);
    float *stash = reinterpret_cast<float *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + kColdF4) + threadIdx.x;
    {
        const float4 *src = reinterpret_cast<const float4 *>(S.cold->lay);
        for (int i = threadIdx.x; i < S.nz * (kLayStride / 4); i += blockDim.x) smem[i] = src[i];
        const float4 *vsrc = reinterpret_cast<const float4 *>(S.cold->views);
        for (int i = threadIdx.x; i < S.nview * 2; i += blockDim.x) smem[S.nz * (kLayStride / 4) + i] = vsrc[i];
        const float4 *csrc = reinterpret_cast<const float4 *>(S.cold);
        if (threadIdx.x < kColdF4) smem[S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + threadIdx.x] = csrc[threadIdx.x];
    }
    // phase-function tables the scene uses (mu grid, P and its CDF): staged behind the stash when they fit
    const float *ltab = nullptr;
    {
        const DevCold *C = S.cold;
        if (C->tab_n > 0) {
            float *dst = reinterpret_cast<float *>(smem + S.nz * (kLayStride / 4) + MI3D_MAX_VIEW * 2 + kColdF4) + 9 * blockDim.x;
            const int nang = C->nang, nt = C->tab_n * nang;
            for (int i = threadIdx.x; i < nang; i += blockDim.x) dst[i] = C->tmu[i];
            for (int i = threadIdx.x; i < nt; i += blockDim.x) {
                dst[nang + i] = C->tp[(long)C->tab_lo * nang + i];
                dst[nang + nt + i] = C->tcdf[(long)C->tab_lo * nang + i];
            }
            ltab = dst;
        }
    }
    __syncthreads();
    const int sstr = blockDim.x; // stash word w of this lane: stash[w * sstr]

    const bool ipa_all = (S.solver == MI3D_SOLVER_IPA);
#define IPA_NOW(is_le_) (ipa_all || (P3D && ((is_le_) || !direct)))
    const bool do_flux = FLUX;
#ifdef MI3D_ABL_NOLE       // ablation (no radiance at all): what the local estimates cost
    const bool do_rad = false;
#else
    const bool do_rad = (S.target & MI3D_TARGET_RADIANCE) != 0 && S.nview > 0;
#endif
    const bool jump = !FLUX; // flux needs every level crossing
    Counters cnt = {};

    // ---- lane state
    float px = 0, py = 0, pz = 0, ux = 0, uy = 0, uz = 1, iux = 1, iuy = 1, iuz = 1;
    int ix = 0, iy = 0, k = 0;
    float rem = 0.0f;  // photon: optical path left before the collision
    float acc = 0.0f;  // local-estimate ray: optical depth accumulated so far
    float w = 0.0f, bt_ev = 0.0f, contrib = 0.0f, zstop = 0.0f;
    float tkill = kTauCut; // local-estimate ray: optical depth at which it is given up (the fixed cut-off, or its roulette's verdict)
    float u1 = 0, u2 = 0, u3 = 0;
    uint64_t id = 0;
    uint32_t draw = 0;
    int mode = M_NEED, iv = 0, kind = E_LAUNCH, dkind = D_LAUNCH;
    bool direct = false;
    unsigned long long pool_next = 0, pool_end = 0; // wave-uniform: positions of the photon order this wave may still hand out
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID: the XCD this workgroup runs on (speed only)
    unsigned victim = 0;                              // pieces of the order found empty so far, counted from this XCD's own
    // column-table tallies of one history often hit the same pixel several times in a row (a photon moves about
    // a voxel per collision): they are summed in a register and flushed when the pixel changes or the history ends
    int pend_pix = -1;
    float pend_val = 0.0f;
    // rest of the voxel record of the pending event (optical depth above, first 3-D constituent): it arrives with the last
    // voxel step of phase A (or is read in B2 for an event inside a horizontally uniform layer of the 3-D region) and is used
    // again by the marched views and the finish block: another read would queue behind the tally atomic, vmcnt being in order
    float ev_ks0 = 0.0f, ev_apf0 = 0.0f, ev_tab = 0.0f;
    float &ev_sfc = ev_tab; // third surface parameter of a surface event (ev_tab has been consumed when it is written)

#ifdef MI3D_CENSUS
#define MI3D_TICK(slot) do { if (COUNT && (slot) < 3) { const long long t_ = clock64(); cnt.cyc[slot] += (uint32_t)((t_ - tick) >> 6); tick = t_; } } while (0)
#else
#define MI3D_TICK(slot) do { if (COUNT) { const long long t_ = clock64(); cnt.cyc[slot] += (uint32_t)((t_ - tick) >> 6); tick = t_; } } while (0)
#endif
    long long tick = COUNT ? clock64() : 0;
    unsigned pass_ctr = 0;
    for (;;) {
        // =================================== phase A: voxel steps ===================================
        MI3D_MARK("A");
        for (;;) {
            const bool flying = (mode <= M_LE);
            const int nfly = __popcll(__ballot(flying));
            if (nfly == 0) break;
            if (nfly < MI3D_THRESH && __ballot(mode > M_LE && mode != M_DONE) != 0ull) break;
            if (COUNT) { cnt.a_slots++; if (flying) cnt.a_lanes++; }
            if (flying) {
                const float4 L = lay4[k * (kLayStride / 4)];
                const float dz = L.x;
                const bool is_le = MARCH && (mode == M_LE);
                // the whole 16-byte record: a ray that ends in this voxel (collision, surface below it) hands the rest of the
                // record to phase B in registers instead of reading the voxel a second time
                const float4 r4 = S.vrec[(unsigned)iy * S.vrow_f4 + (unsigned)ix * S.vcol_f4 + (unsigned)(k - S.k3lo)];
                const float bt = r4.x;
                // distance to the nearest face of the voxel
                float s = (uz > 0.0f ? dz - pz : pz) * iuz;
                int axis = 2;
                const float sx = (ux > 0.0f ? S.dx - px : px) * iux;
                const float sy = (uy > 0.0f ? S.dy - py : py) * iuy;
                if (sx < s) { s = sx; axis = 0; }
                if (sy < s) { s = sy; axis = 1; }
                s = fmaxf(s, 0.0f);
                if (COUNT) {
                    if (is_le) { cnt.le_steps++; cnt.le_steps3d++; }
                    else { cnt.steps++; cnt.steps3d++; }
                }
                const float dtau = bt * s;
                if (!is_le && dtau >= rem) {
                    // ---- the collision lies inside this voxel
                    const float sc = rem * frcp(bt);
                    // (no clamping of x and y: a position a rounding error outside its voxel gives a negative face distance,
                    //  which the max(s, 0) above turns into a zero-length step across that face)
                    px += ux * sc; py += uy * sc;
                    pz = fminf(fmaxf(pz + uz * sc, 0.0f), dz);
                    bt_ev = bt; ev_tab = r4.y; ev_ks0 = r4.z; ev_apf0 = r4.w;
                    mode = M_COLL;
                } else if (is_le && (uz > 0.0f ? L.z + pz + uz * s >= zstop : L.z + pz + uz * s <= zstop)) {
                    // ---- sensor plane inside the atmosphere: the ray ends inside this voxel
                    acc += bt * fabsf(zstop - (L.z + pz)) * iuz;
                    mode = M_LEEND;
                } else {
                    if (is_le) acc += dtau; else rem -= dtau;
                    // ---- move onto the face and into the neighbour voxel
                    px += ux * s; py += uy * s;
                    pz = fminf(fmaxf(pz + uz * s, 0.0f), dz);
                    if (axis != 2) {
                        // x or y face: land exactly on it and step the column index with wrap-around
                        const bool xface = (axis == 0);
                        const bool fwd = xface ? (ux > 0.0f) : (uy > 0.0f);
                        const float edge = fwd ? 0.0f : (xface ? S.dx : S.dy);
                        px = xface ? edge : px;
                        py = xface ? py : edge;
                        if (!IPA_NOW(is_le)) {
                            const int n = xface ? S.nx : S.ny;
                            int c = (xface ? ix : iy) + (fwd ? 1 : -1);
                            c = c >= n ? 0 : (c < 0 ? n - 1 : c);
                            ix = xface ? c : ix;
                            iy = xface ? iy : c;
                        }
                    } else {
                        const bool up = uz > 0.0f;
                        int knew = up ? k + 1 : k - 1;
                        if (do_flux && !is_le) flux_add<COUNT>(S, ix, iy, w, direct, up ? knew : k, up, cnt);
                        if (knew >= S.nz) {
                            if (is_le) mode = M_LEEND;
                            else { if (COUNT) cnt.escaped++; mode = M_NEED; }
                        } else if (knew < 0) {
                            knew = 0;
                            if (is_le) mode = M_LEEND;   // (a ray towards an up-looking sensor on the ground, ended by rounding)
                            else { pz = 0.0f; mode = M_SURF; bt_ev = bt; ev_tab = r4.y; ev_ks0 = r4.z; ev_apf0 = r4.w; }
                        } else {
                            const float4 Ln = lay4[knew * (kLayStride / 4)];
                            pz = up ? 0.0f : Ln.x;
                            if (!(__float_as_int(Ln.w) & kLayStep3d)) mode = is_le ? M_LEUNIF : M_UNIF;
                        }
                        k = knew;
                    }
                    if (is_le && acc > tkill) mode = M_LEEND;
                }
            }
        }

        // =================================== phase B: everything else ===================================
        MI3D_TICK(0);
        MI3D_MARK("B0");
        if (COUNT) { cnt.b_slots++; if (mode > M_LE && mode != M_DONE) cnt.b_lanes++; }
        // Only every MI3D_LEAN-th pass is a full one.  The passes between serve collisions only (event, finish, flight draw);
        // the rarer kinds of work -- uniform-layer runs, surface hits, new photons, roulette -- wait for the next full pass, so
        // that their code is not executed by the whole wave for one lane in most passes (each of them is needed by SOME lane
        // in 60-100 % of the passes).  A pass with no collision pending is always full, so nothing can wait for ever.
        // With marched views it is the photons' own events that are the rarer kind (one event starts up to 16 rays): there
        // only every MI3D_LEAN_MARCH-th pass serves them (`evt`), every pass serves the rays.
        bool evt_m = true;
        if (MARCH) evt_m = MI3D_LEAN_MARCH <= 1 || ((pass_ctr++ % (unsigned)(MI3D_LEAN_MARCH)) == 0u) ||
                           __ballot(mode == M_LEEND || mode == M_VIEWS || mode == M_LEUNIF) == 0ull;
        const bool full = MARCH ? evt_m : (MI3D_LEAN <= 1 || ((pass_ctr++ % (unsigned)(MI3D_LEAN)) == 0u) ||
                          __ballot(mode == M_COLL || (mode == M_FINISH && (kind & 15) != E_SURFACE) || (mode == M_DRAW && dkind == D_FLIGHT)) == 0ull);
#define EVT (!MARCH || evt_m)

        // ---- B0: rays inside runs of horizontally uniform layers
        if (MARCH ? ((full && mode == M_UNIF) || mode == M_LEUNIF) : (full && (mode == M_UNIF || mode == M_LEUNIF))) {
            const bool is_le = MARCH && (mode == M_LEUNIF);
            const bool up = uz > 0.0f;
            bool done = false;
            // The whole rest of the run at once where nothing has to be done level by level: always without flux tallies; with
            // them for a local-estimate ray, and for the DIRECT beam above the 3-D region, whose flux is known analytically
            // and added when the result is read (S.kdir, mi3d_get_flux).  A sensor plane inside the atmosphere ends the ray
            // somewhere in the run: layer by layer.
            const LayerRec &Lk = lay[k];
            const bool can_jump = is_le ? !(zstop < INFINITY) : (jump || (direct && !up && Lk.run_lo >= S.kdir));
            if (can_jump) {
                // from the prefix sums of the layer table
                const int kend = up ? Lk.run_hi : Lk.run_lo;
                const LayerRec &Le = lay[kend];
                const float tv = up ? (Le.tauz + Le.bt * Le.dz - Lk.tauz) - Lk.bt * pz
                                    : (Lk.tauz - Le.tauz) + Lk.bt * pz;          // vertical optical depth
                const float hv = up ? (Le.zlo + Le.dz) - (Lk.zlo + pz) : (Lk.zlo + pz) - Le.zlo;
                const float tpath = tv * iuz;
                if (is_le || tpath < rem) {
                    if (is_le) acc += tpath; else rem -= tpath;
                    const float s = hv * iuz;
                    px += ux * s; py += uy * s;
                    if (COUNT) { if (is_le) cnt.le_steps++; else cnt.steps++; }
                    done = true;
                    if (up) {
                        k = kend + 1; pz = 0.0f;
                        if (k >= S.nz) {
                            if (is_le) mode = M_LEEND;
                            else { if (COUNT) cnt.escaped++; mode = M_NEED; }
                        } else { fold_xy(S, cold, px, py, ix, iy, IPA_NOW(is_le)); mode = is_le ? M_LE : M_FLY; }
                    } else {
                        k = kend - 1;
                        if (k < 0) { k = 0; pz = 0.0f; mode = is_le ? M_LEEND : M_SURF; }
                        else { pz = lay[k].dz; fold_xy(S, cold, px, py, ix, iy, IPA_NOW(is_le)); mode = M_FLY; }
                    }
                    if (is_le && acc > tkill) mode = M_LEEND;
                } else {
                    // the collision lies inside the run: the layer table holds the vertical optical depth below every layer,
                    // so the layer that contains it is found by bisection instead of walking the run layer by layer
                    const float T = Lk.tauz + Lk.bt * pz + (up ? rem : -rem) * fabsf(uz);
                    int lo = up ? k : kend, hi = up ? kend : k;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (lay[mid].tauz <= T) lo = mid; else hi = mid - 1;
                    }
                    const float4 Lj = lay4[lo * (kLayStride / 4)];     // {dz, bt, zlo, flags}
                    const float pzn = fminf(fmaxf((T - lay[lo].tauz) * frcp(fmaxf(Lj.y, 1e-30f)), 0.0f), Lj.x);
                    const float s = fabsf((Lj.z + pzn) - (Lk.zlo + pz)) * iuz;
                    px += ux * s; py += uy * s;
                    k = lo; pz = pzn;
                    bt_ev = Lj.y;
                    if (COUNT) cnt.steps++;
                    mode = M_COLL;
                    done = true;
                }
            }
            if (!done) {
                // layer by layer: the collision (or the sensor) lies inside the run, or flux is tallied per level
                for (int guard = 0; guard < kMaxLayers + 2; ++guard) {
                    const float4 L = lay4[k * (kLayStride / 4)];
                    if (__float_as_int(L.w) & kLayStep3d) { fold_xy(S, cold, px, py, ix, iy, IPA_NOW(is_le)); mode = is_le ? M_LE : M_FLY; break; }
                    const float dz = L.x, bt = L.y;
                    const float s = fmaxf((up ? dz - pz : pz) * iuz, 0.0f);
                    const float dtau = bt * s;
                    if (COUNT) { if (is_le) cnt.le_steps++; else cnt.steps++; }
                    if (!is_le && dtau >= rem) {
                        const float sc = rem * frcp(bt);
                        px += ux * sc; py += uy * sc;
                        pz = fminf(fmaxf(pz + uz * sc, 0.0f), dz);
                        bt_ev = bt;
                        mode = M_COLL;
                        break;
                    }
                    if (is_le && (up ? L.z + pz + uz * s >= zstop : L.z + pz + uz * s <= zstop)) {
                        acc += bt * fabsf(zstop - (L.z + pz)) * iuz;
                        mode = M_LEEND;
                        break;
                    }
                    if (is_le) acc += dtau; else rem -= dtau;
                    px += ux * s; py += uy * s;
                    const int knew = up ? k + 1 : k - 1;
                    if (do_flux && !is_le && !(direct && k >= S.kdir)) {   // (the direct beam moves down: the level crossed is k)
                        fold_xy(S, cold, px, py, ix, iy, IPA_NOW(false));
                        flux_add<COUNT>(S, ix, iy, w, direct, up ? knew : k, up, cnt);
                    }
                    if (knew >= S.nz) {
                        if (is_le) mode = M_LEEND;
                        else { if (COUNT) cnt.escaped++; mode = M_NEED; }
                        break;
                    }
                    if (knew < 0) { k = 0; if (is_le) mode = M_LEEND; else { pz = 0.0f; mode = M_SURF; } break; }
                    k = knew;
                    pz = up ? 0.0f : lay4[k * (kLayStride / 4)].x;
                    if (is_le && acc > tkill) { mode = M_LEEND; break; }
                }
            }
        }

        MI3D_TICK(1);
        MI3D_MARK("B1");
        // ---- B1: a local-estimate ray has arrived: tally it
        if (MARCH && mode == M_LEEND) {
            if (acc <= tkill && views[iv].point) {
                // camera: the ray has reached the point sensor; its pixel is where the direction the camera looks in to see the
                // event falls in the polar map U = theta cos(phi), V = theta sin(phi) (theta from the camera's axis); the value
                // already holds 1 / r^2, the patch dU dV has the solid angle (sin theta / theta) dU dV
                const CamRec Cm = cold->cams[iv];
                const float dxc = -(ux * Cm.xx + uy * Cm.xy + uz * Cm.xz), dyc = -(ux * Cm.yx + uy * Cm.yy + uz * Cm.yz);
                const float dzc = fminf(-(ux * Cm.zx + uy * Cm.zy + uz * Cm.zz), 1.0f);
                const float theta = acosf(dzc), rho2 = dxc * dxc + dyc * dyc;
                const float sc = rho2 > 1e-24f ? theta * frsq(rho2) : 0.0f;
                const int ir = (int)floorf(dxc * sc * Cm.inv_du + 0.5f * (float)S.nxr), jr = (int)floorf(dyc * sc * Cm.inv_dv + 0.5f * (float)S.nyr);
                if (ir >= 0 && ir < S.nxr && jr >= 0 && jr < S.nyr) {
                    const float sinc = theta > 1e-6f ? sinf(theta) / theta : 1.0f;
                    const ViewRec V = views[iv];
                    RAD_ADD(&S.rad[(unsigned)((iv * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride],
                            contrib * fexp_neg((V.roulette & 1) ? fminf(acc, cold->le_tau1) : acc) * Cm.inv_du * Cm.inv_dv / sinc);
                }
            } else if (acc <= tkill) {
                const ViewRec V = views[iv];
                // event position and height from the stash; pixel = where the line of sight meets zref
                const float epx = stash[0 * sstr], epy = stash[1 * sstr], epz = stash[2 * sstr];
                const int eix = __float_as_int(stash[3 * sstr]), eiy = __float_as_int(stash[4 * sstr]),
                          ek = __float_as_int(stash[5 * sstr]);
                float xr = (float)eix * S.dx + epx, yr = (float)eiy * S.dy + epy;
                if (!IPA_NOW(true)) {
                    const float ivz = frcp(V.vz);
                    const float t = (lay[ek].zlo + epz - V.zreg) * ivz;
                    xr -= V.vx * t; yr -= V.vy * t;
                    xr -= floorf(xr * cold->inv_Lx) * cold->Lx; yr -= floorf(yr * cold->inv_Ly) * cold->Ly;
                }
                const int ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                const int jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                RAD_ADD(&S.rad[(unsigned)((iv * S.nyr + jr) * S.rad_row + ir) * (unsigned)S.rad_stride], contrib * fexp_neg((V.roulette & 1) ? fminf(acc, cold->le_tau1) : acc) * frcp(fabsf(V.vz)));
            }
            iv += 1;
            mode = M_VIEWS;
        }

        MI3D_MARK("B2");
        // ---- B2: a new event: weight update, column-table views, stash for marched views
        if (EVT && (mode == M_COLL || (full && mode == M_SURF))) {
            kind = (mode == M_SURF) ? E_SURFACE : E_SCATTER;
            const LayerRec &Lk = lay[k];
            const bool in3d = (Lk.flags & kLayIn3d) != 0;
            if (!(Lk.flags & kLayStep3d)) fold_xy(S, cold, px, py, ix, iy, IPA_NOW(false));
            const unsigned col = (unsigned)(iy * S.nx + ix);
            const unsigned vox = col * (unsigned)S.nz3 + (unsigned)(k - S.k3lo);
            // one 16-byte read brings everything this voxel contributes: extinction, optical depth above, first constituent
            if (!(Lk.flags & kLayStep3d)) {
                // the event was found by the uniform-layer code: no voxel step has brought the record
                float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (in3d) rec = S.vrec[(unsigned)iy * S.vrow_f4 + (unsigned)ix * S.vcol_f4 + (unsigned)(k - S.k3lo)];
                ev_tab = rec.y; ev_ks0 = rec.z; ev_apf0 = rec.w;
            }
            const float4 rec = make_float4(bt_ev, ev_tab, ev_ks0, ev_apf0);
            #ifdef MI3D_ABL_NOTCOL0   // ablation (wrong physics): what the column-table read of events below the 3-D region costs
            const float tcol_here = in3d ? rec.y : Lk.tabove;
#else
            const float tcol_here = in3d ? rec.y : Lk.tabove + ((k < S.k3lo && S.nz3 > 0) ? cold->tcol0[col] : 0.0f);
#endif
            Sfc sf = {0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            float kstot = 0.0f;
            bool dead = false;
            if (mode == M_SURF) {
                if (COUNT) cnt.surface++;
                sf = load_sfc(S, cold, ix, iy, px, py);
                if (!(Lk.flags & kLayStep3d)) bt_ev = Lk.bt;
                // the surface record travels to the later blocks in the event registers a scattering event uses for its voxel
                ev_ks0 = sf.p0; ev_apf0 = sf.p1; ev_sfc = sf.p2; kind = E_SURFACE | (sf.type << 4);
            } else {
                if (COUNT) cnt.scatter++;
                for (int ip = 0; ip < S.np1d; ++ip) kstot += Lk.ks1d[ip];
                if (in3d) {
                    kstot += rec.z;
                    for (int ip = 1; ip < S.np3d; ++ip) kstot += cold->csca[vox * (unsigned)S.np3d + (unsigned)ip].x;
                }
                // (exactly 1 for conservative scattering: the approximate reciprocal must not nudge a weight that sits on
                //  the roulette threshold below it)
                const float w_in = w;
                w *= (kstot >= bt_ev) ? 1.0f : kstot * frcp(bt_ev);
                // heating rates (Flx_mhrt = 1): what the collision takes from the weight stays in this cell
                if (FLUX && cold->heat && kstot < bt_ev)
                    atomicAdd(&cold->heat[(unsigned)(k * S.ny + iy) * (unsigned)S.nx + (unsigned)ix], (double)(w_in * (bt_ev - kstot) * frcp(bt_ev)));
                if (!(w > 0.0f)) { if (COUNT) cnt.absorbed++; dead = true; }
            }
            if (dead) {
                mode = M_NEED;
            } else if (!do_rad) {
                mode = M_FINISH;
            } else {
                // views answered from the column table are tallied here and now
                const float zev = Lk.zlo + pz;
                if (S.nmarch < S.nview) {
                    for (int jv = 0; jv < S.nview; ++jv) {
                        const ViewRec V = views[jv];
                        if (!V.column || zev >= V.zs) continue;
                        float c;
                        if ((kind & 15) == E_SURFACE) {
                            c = w * surface_R(sf, ux, uy, uz, V.vx, V.vy, V.vz) * V.vz * (1.0f / kPi);
                        } else {
                            const float mu = ux * V.vx + uy * V.vy + uz * V.vz;
                            float P = 0.0f;
                            for (int ip = 0; ip < S.np1d; ++ip)
                                if (Lk.ks1d[ip] > 0.0f) P += Lk.ks1d[ip] * phase_eval(cold, ltab, Lk.apf1d[ip], mu);
                            if (in3d) {
                                if (rec.z > 0.0f) P += rec.z * phase_eval(cold, ltab, rec.w, mu);
                                for (int ip = 1; ip < S.np3d; ++ip) {
                                    const float2 cs = cold->csca[vox * (unsigned)S.np3d + (unsigned)ip];
                                    if (cs.x > 0.0f) P += cs.x * phase_eval(cold, ltab, cs.y, mu);
                                }
                            }
                            c = w * P * frcp(kstot) * (0.25f / kPi);
                        }
                        if (COUNT) { cnt.le_rays++; cnt.le_column++; }
                        if (c > 0.0f) {
                            const float tau = bt_ev * (Lk.dz - pz) + tcol_here;
                            const float xr = (float)ix * S.dx + px, yr = (float)iy * S.dy + py;
                            const int ir = min(max((int)(xr * S.pix_sx), 0), S.nxr - 1);
                            const int jr = min(max((int)(yr * S.pix_sy), 0), S.nyr - 1);
                            const int pix = (jv * S.nyr + jr) * S.rad_row + ir;
                            const float val = c * fexp_neg(tau) * frcp(V.vz);
                            if (pix == pend_pix) pend_val += val;
                            else {
                                if (pend_pix >= 0) RAD_ADD(&S.rad[(unsigned)pend_pix * (unsigned)S.rad_stride], pend_val);
                                pend_pix = pix; pend_val = val;
                            }
                        }
                    }
                }
                if (MARCH && S.nmarch > 0) {
                    stash[0 * sstr] = px; stash[1 * sstr] = py; stash[2 * sstr] = pz;
                    stash[3 * sstr] = __int_as_float(ix); stash[4 * sstr] = __int_as_float(iy);
                    stash[5 * sstr] = __int_as_float(k);
                    stash[6 * sstr] = ux; stash[7 * sstr] = uy; stash[8 * sstr] = uz;
                    iv = 0;
                    mode = M_VIEWS;
                } else {
                    mode = M_FINISH;
                }
            }
        }

        MI3D_TICK(2);
        MI3D_MARK("B3");
        // ---- B3: start the local-estimate ray of the next marched view, if any is left
        if (MARCH && mode == M_VIEWS) {
            // restore the event state (position and incoming direction)
            px = stash[0 * sstr]; py = stash[1 * sstr]; pz = stash[2 * sstr];
            ix = __float_as_int(stash[3 * sstr]); iy = __float_as_int(stash[4 * sstr]); k = __float_as_int(stash[5 * sstr]);
            ux = stash[6 * sstr]; uy = stash[7 * sstr]; uz = stash[8 * sstr];
            const LayerRec &Lk = lay[k];
            const float zev = Lk.zlo + pz;
            // skip the views answered from the column table, the sensors on the wrong side of the event, and -- for a surface
            // event -- the up-looking ones
            while (iv < S.nview && !views[iv].point &&
                   (views[iv].column || (views[iv].vz > 0.0f ? zev >= views[iv].zs : (zev <= views[iv].zs || (kind & 15) == E_SURFACE)))) ++iv;
            if (iv >= S.nview) {
                mode = M_FINISH;
            } else {
                ViewRec V = views[iv];
                float inv_r2 = 1.0f;
                bool visible = true;
                if (V.point) {
                    // camera: the ray goes to the nearest periodic image of the point sensor, a distance r away
                    const CamRec Cm = cold->cams[iv];
                    float rx = Cm.cx - ((float)ix * S.dx + px), ry = Cm.cy - ((float)iy * S.dy + py);
                    const float rz = Cm.cz - zev;
                    rx -= cold->Lx * floorf(rx * cold->inv_Lx + 0.5f); ry -= cold->Ly * floorf(ry * cold->inv_Ly + 0.5f);
                    const float r2 = rx * rx + ry * ry + rz * rz, ir = frsq(fmaxf(r2, 1e-30f));
                    V.vx = rx * ir; V.vy = ry * ir; V.vz = rz * ir;
                    V.zs = Cm.cz;
                    inv_r2 = frcp(fmaxf(r2, Cm.r2min));
                    // outside the cone of view, a line of sight within 0.06 degrees of the horizontal (its optical depth is found by
                    // marching to the camera's height), the surface seen from below: nothing to carry
                    visible = r2 > 0.0f && fabsf(V.vz) >= 1e-3f && -(V.vx * Cm.zx + V.vy * Cm.zy + V.vz * Cm.zz) >= Cm.cos_half &&
                              !((kind & 15) == E_SURFACE && V.vz <= 0.0f);
                }
                const bool in3d = (Lk.flags & kLayIn3d) != 0;
                const unsigned vox = (unsigned)((iy * S.nx + ix) * S.nz3 + (k - S.k3lo));
                float c;
                if (!visible) {
                    c = 0.0f;
                } else if ((kind & 15) == E_SURFACE) {
                    // (three parameters travel in the event registers; the five of the diffuse-specular mixture are read again)
                    const Sfc sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ix, iy, px, py) : Sfc{kind >> 4, ev_ks0, ev_apf0, ev_sfc, 0.0f, 0.0f};
                    c = w * surface_R(sf, ux, uy, uz, V.vx, V.vy, V.vz) * V.vz * (1.0f / kPi);
                } else {
                    const float mu = ux * V.vx + uy * V.vy + uz * V.vz;
                    float P = 0.0f, kstot = 0.0f;
                    for (int ip = 0; ip < S.np1d; ++ip) {
                        const float ks = Lk.ks1d[ip];
                        kstot += ks;
                        if (ks > 0.0f) P += ks * phase_eval(cold, ltab, Lk.apf1d[ip], mu);
                    }
                    if (in3d)
                        for (int ip = 0; ip < S.np3d; ++ip) {
                            float2 cs;
                            if (ip == 0) cs = make_float2(ev_ks0, ev_apf0);
                            else cs = cold->csca[vox * (unsigned)S.np3d + (unsigned)ip];
                            kstot += cs.x;
                            if (cs.x > 0.0f) P += cs.x * phase_eval(cold, ltab, cs.y, mu);
                        }
                    c = w * P * frcp(kstot) * (0.25f / kPi);
                }
                if (COUNT && visible) cnt.le_rays++;
                if (V.roulette & 2) c = le_weight_roulette(c, cold->le_cmin, seed, id, draw, iv);
                if (c > 0.0f) {
                    // (a camera's value carries 1 / r^2 and, being a radiance at a point, not the 1 / |vz| of a pixel's column
                    //  cross-section that block B1 applies to the satellite views)
                    contrib = V.point ? c * inv_r2 : c;
                    ux = V.vx; uy = V.vy; uz = V.vz;
                    iux = frcp(fmaxf(fabsf(ux), 1e-20f)); iuy = frcp(fmaxf(fabsf(uy), 1e-20f)); iuz = frcp(fabsf(uz));
                    acc = 0.0f; zstop = (uz < 0.0f || V.zs < cold->ztoa) ? V.zs : INFINITY; // a sensor above the atmosphere is never reached
                    // roulette: the ray survives to optical depth tau with probability min(1, exp(-(tau - tau1))) and then carries
                    // exp(-min(tau, tau1)); one hashed uniform number per ray fixes where it ends
                    tkill = (V.roulette & 1) ? cold->le_tau1 - 0.69314718f * __builtin_amdgcn_logf(le_roulette_u(seed, id, draw, iv)) : kTauCut;
                    mode = (Lk.flags & kLayStep3d) ? M_LE : M_LEUNIF;
                } else {
                    iv += 1; // nothing to carry: look at the next view on the next pass
                }
            }
        }

        MI3D_MARK("B4");
        // ---- B4: next photon.  Ids come from a wave-local pool refilled kChunk at a time by ONE lane
        // (a single global counter word saturates near 9e7 returning atomics per second chip-wide).
        if (full && mode == M_NEED && (id != 0 || draw != 0)) { // a history just ended
            cnt.photons++; id = 0; draw = 0;
            if (pend_pix >= 0) { RAD_ADD(&S.rad[(unsigned)pend_pix * (unsigned)S.rad_stride], pend_val); pend_pix = -1; }
        }
        for (;;) {
            const unsigned long long need = __ballot(full && mode == M_NEED);
            if (need == 0ull) break;
            if (pool_next >= pool_end) {
                // The launch's photon order (sorted by start tile, k_bin_*) is cut into eight contiguous pieces, one per XCD,
                // each with its own cursor; a wave takes kChunk positions from its XCD's piece and, once that is used up,
                // from the next XCD's that still has some (`victim` only grows: a piece found empty stays empty).
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
                if (!got) { // the launch has no photons left
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
            Sfc sf = {0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if ((kind & 15) == E_SURFACE) {
                sf = (kind >> 4) == MI3D_SFC_DSM ? load_sfc(S, cold, ix, iy, px, py) : Sfc{kind >> 4, ev_ks0, ev_apf0, ev_sfc, 0.0f, 0.0f};
                bx = 0.0f; by = 0.0f; bz = 1.0f;
                mu_rot = fsqrt(u2);
            } else if ((kind & 15) == E_SCATTER) {
                const LayerRec &Lk = lay[k];
                const bool in3d = (Lk.flags & kLayIn3d) != 0;
                const unsigned vox = (unsigned)((iy * S.nx + ix) * S.nz3 + (k - S.k3lo));
                float kstot = 0.0f;
                float2 c0 = make_float2(0.0f, 0.0f);
                for (int ip = 0; ip < S.np1d; ++ip) kstot += Lk.ks1d[ip];
                if (in3d) {
                    c0 = make_float2(ev_ks0, ev_apf0);
                    kstot += c0.x;
                    for (int ip = 1; ip < S.np3d; ++ip) kstot += cold->csca[vox * (unsigned)S.np3d + (unsigned)ip].x;
                }
                // choose the constituent that scatters: 1-D constituents first, then the 3-D ones
                const float target = u1 * kstot;
                float cum = 0.0f, usel = 0.0f, apf_sel = -2.0f;
                bool found = false;
                const int ncomp = S.np1d + (in3d ? S.np3d : 0);
                for (int q = 0; q < ncomp; ++q) {
                    float ks, apf;
                    if (q < S.np1d) { ks = Lk.ks1d[q]; apf = Lk.apf1d[q]; }
                    else if (q == S.np1d) { ks = c0.x; apf = c0.y; }
                    else { const float2 cs = cold->csca[vox * (unsigned)S.np3d + (unsigned)(q - S.np1d)]; ks = cs.x; apf = cs.y; }
                    if (!found && (target < cum + ks || q == ncomp - 1)) {
                        found = true;
                        apf_sel = apf;
                        usel = ks > 0.0f ? (target - cum) * frcp(ks) : 0.0f;
                    }
                    cum += ks;
                }
                usel = fminf(fmaxf(usel, 0.0f), 1.0f);
                mu_rot = phase_sample(cold, ltab, apf_sel, u2, usel);
            }
            if (!(kind == E_LAUNCH && cold->cos_cone >= 1.0f)) rotate_dir(bx, by, bz, mu_rot, u3);
            if ((kind & 15) == E_SURFACE) {
                bz = fmaxf(bz, 1e-9f);
                w *= surface_R(sf, ux, uy, uz, bx, by, bz);
                if (w > 0.0f && do_flux) flux_add<COUNT>(S, ix, iy, w, false, 0, true, cnt);
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
        // ---- B6: the one Philox block.  Most lanes arrive from B5 and leave flying; a roulette survivor and
        // a freshly launched photon come back for their flight draw on the next pass.
        if (EVT && mode == M_DRAW && (full || dkind == D_FLIGHT)) {
            float r0, r1, r2, r3;
            draw4(seed, id, draw++, r0, r1, r2, r3);
            if (dkind == D_FLIGHT) {
                rem = -0.69314718f * __builtin_amdgcn_logf(r0); // u >= 2^-24: never denormal, the bare v_log_f32 will do
                u1 = r1; u2 = r2; u3 = r3;
                iux = frcp(fmaxf(fabsf(ux), 1e-20f)); iuy = frcp(fmaxf(fabsf(uy), 1e-20f));
                iuz = frcp(fmaxf(fabsf(uz), 1e-20f));
                mode = (lay[k].flags & kLayStep3d) ? M_FLY : M_UNIF;
            } else if (dkind == D_ROULETTE) {
                if (r0 * S.wfac < w) { w = S.wfac; dkind = D_FLIGHT; }
                else { if (COUNT) cnt.killed++; mode = M_NEED; }
            } else { // D_LAUNCH: position at the top of the atmosphere, solar direction; jitter + free path follow
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
                u2 = 1.0f - r2 * (1.0f - cold->cos_cone); // polar cosine of the jitter inside the solar cone
                u3 = r3;
                asm volatile("" : "+v"(u3)); // hipcc 7.2 drops this store in the MARCH builds without the barrier (guarded by
                                                // tests/test_gpu_parity.py::test_single_histories_follow_the_oracle)
                w = 1.0f;
                direct = true;
                if (do_flux && S.nz < S.kdir) flux_add<COUNT>(S, ix, iy, w, true, S.nz, false, cnt);   // (never: the top level is analytic)
                kind = E_LAUNCH;
                mode = M_FINISH;
            }
        }

        MI3D_TICK(5);
        MI3D_MARK("END");
        if (__ballot(mode != M_DONE) == 0ull) break;
    }
#undef MI3D_TICK
#undef EVT

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
}

#undef IPA_NOW
#define MI3D_INST(C, M, F) template __global__ void k_transport<C, M, F, false>(const DevScene, const uint64_t, const uint64_t, const uint64_t); \
                           template __global__ void k_transport<C, M, F, true>(const DevScene, const uint64_t, const uint64_t, const uint64_t);
MI3D_INST(false, false, false) MI3D_INST(false, false, true) MI3D_INST(false, true, false) MI3D_INST(false, true, true)
MI3D_INST(true, false, false) MI3D_INST(true, false, true) MI3D_INST(true, true, false) MI3D_INST(true, true, true)
#undef MI3D_INST

} // namespace mi3d
