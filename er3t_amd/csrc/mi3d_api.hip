// mi3d_api.hip — host side of libmi3drt.so: the C-ABI declared in include/mi3d.h.
//
// Plays the role of the reference solver's start-up and I/O phases (reading the namelist and the
// three side files, er3t/rtm/mca/mca_inp.py:636-697, mca_atm.py:373-389, mca_sca.py:82-92,
// mca_sfc.py:136-146; writing out.bin, mca_out.py:94-103) with in-memory hand-off instead of
// files.  There is NO CPU fallback: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "mi3d_kernels.hip"
#include "mi3d_kernel_lean.hip"
#include "mi3d_kernel_rays.hip"
#include "mi3d_kernel_flux.hip"

using namespace mi3d;

#include <algorithm>

static constexpr size_t kTabLdsBudget = 24 * 1024; // LDS bytes a workgroup may spend on phase tables

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MI3D_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                        __FILE__, __LINE__);                                                       \
    } while (0)

template <typename T>
int dev_realloc(T *&p, size_t &cap, size_t n) {
    if (n <= cap && p) return MI3D_OK;
    if (p) { HIPCHK(hipFree(p)); p = nullptr; cap = 0; }
    if (n == 0) return MI3D_OK;
    HIPCHK(hipMalloc((void **)&p, n * sizeof(T)));
    cap = n;
    return MI3D_OK;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int alloc(size_t n) { return dev_realloc(p, cap, n); }
    int upload(const T *src, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        if (n) HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
        return MI3D_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};

} // namespace

struct mi3d_solver {
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;      // the stream every launch and asynchronous copy of this handle goes to: the caller's (NULL: the null stream) or own_stream
    hipStream_t own_stream = nullptr;  // mi3d_set_tuning "own_stream": non-blocking, so that two handles on one device do not serialise through the null stream
    bool use_own_stream = false;

    // ---- host copies of the small inputs
    int nz = 0, np1d = 0;
    std::vector<double> zgrd;
    std::vector<float> ext1d, omg1d, apf1d, abs1d;
    int nx = 1, ny = 1, nz3 = 0, iz3l = 1, np3d = 0;
    double dx = 1.0e4, dy = 1.0e4;
    bool has_abst = false;
    int nang = 0, npf = 0;
    std::vector<float> ang, pha;
    int sfc_mtype = MI3D_SFC_LAMBERT, nxb = 0, nyb = 0;
    bool sfc_lambert_only = true;   // no LSRT or DSM surface anywhere: the ray kernel's light build serves
    float sfc_param[5] = {0, 0, 0, 0, 0};
    std::vector<float> sfc2d_host;
    double src_flx = 1.0, src_qmax = 0.0, src_the = 180.0, src_phi = 0.0;
    int nview = 0, nxr = 1, nyr = 1;
    double view_the[MI3D_MAX_VIEW], view_phi[MI3D_MAX_VIEW], view_zloc[MI3D_MAX_VIEW], zref = 0.0;
    // cameras (mi3d_set_cameras): rad_kind 1, the views are point sensors
    int rad_kind = 2;
    double cam_psi[MI3D_MAX_VIEW], cam_xpos[MI3D_MAX_VIEW], cam_ypos[MI3D_MAX_VIEW], cam_qmax[MI3D_MAX_VIEW], cam_umax[MI3D_MAX_VIEW],
           cam_vmax[MI3D_MAX_VIEW], cam_apsize[MI3D_MAX_VIEW];
    DevBuf<CamRec> d_cams;
    int target = MI3D_TARGET_FLUX, solver = MI3D_SOLVER_3D, column_le = 1, counting = 0;
    double wmin = 0.2, wfac = 1.0, le_tau1 = 0.0, le_cmin = 0.0;
    std::vector<LayerRec> lay_host;  // the layer table as uploaded (mi3d_prepare)
    std::vector<double> dir_level;   // [nz+1] analytic direct-beam flux per unit Src_flx*mu0 at the levels >= kdir, 0 below
    int kdir = 0;
    DevBuf<double> d_dir_level;

    // ---- device data
    DevBuf<float> d_abst, d_extp, d_omgp, d_apfp;        // file-layout inputs
    DevBuf<LayerRec> d_lay;
    DevBuf<ViewRec> d_views;
    DevBuf<DevCold> d_cold;
    DevBuf<float> d_bt1d, d_dz, d_bmin, d_bmax;
    DevBuf<float4> d_vrec;
    DevBuf<float> d_bext3;           // the total extinction alone, for the ray kernel's walk (DevScene::bext3)
    double z_cloud = -1.0;               // height [m] around which the 3-D layers' extinction varies most (the tally window is centred where the direct beam gets there), -1: none
    int tally_window = 1;                // mi3d_set_tuning "tally_window": 0: every radiance tally of the lean loop is an atomic on the image
    unsigned vcol_f4 = 0, vrow_f4 = 0;   // strides of the voxel records (DevScene), set by mi3d_prepare
    int vpad_col = 0, vpad_row = 0;      // padding of a column / a row in records (MI3D_VPAD_COL, MI3D_VPAD_ROW)
    DevBuf<float> d_tcol0, d_tmu, d_tp, d_tcdf, d_sfc2d;
    DevBuf<uint16_t> d_tmuidx, d_tcdfidx;   // bucket indices into the tables (DevCold::tmu_idx, tcdf_idx)
    int nmarch = 0, n_step3d = 0, col0 = 0;
    int tab3d_lo = 1 << 30, tab3d_hi = -1; // table range referenced by the 3-D constituents
    bool hg3d = true;                      // every 3-D constituent with extinction is Henyey-Greenstein (-1 < apf < 1)
    int tab_lo = 0, tab_n = 0;             // tables staged in LDS
    DevBuf<int> d_tabrange;
    DevBuf<float2> d_csca;
    DevBuf<tally_t> d_rad_own, d_flux_own;
    DevBuf<tally_t> d_rad_acc;       // accumulation image of the radiance tally: one pixel per 128-byte line (kRadLine), see mi3d_run
    int rad_row_pad = -1;            // pixels of padding per row of the accumulation image (-1: chosen from its row length; MI3D_RAD_ROW_PAD)
    int rad_spread = -1;             // -1: spread the image when it stays below 1 GB, 0: never (MI3D_RAD_SPREAD overrides)
    tally_t *rad_ext = nullptr, *flux_ext = nullptr;
    DevBuf<double> d_heat_own;       // heating rates (MI3D_TARGET_HEAT): weight absorbed per cell [nz][ny][nx]
    double *heat_ext = nullptr;
    size_t heat_elems() const { return (size_t)nz * nx * ny; }
    double *heat_ptr() { return heat_ext ? heat_ext : d_heat_own.p; }
    DevBuf<unsigned long long> d_counters, d_next;
    // photon order of a launch (k_bin_*): indices sorted by start tile, the tile of every index, histogram and cursors
    DevBuf<uint32_t> d_order, d_hist, d_cursor;
    DevBuf<uint16_t> d_tile;
    DevBuf<float4> d_entry;          // entry records of the launch in flight (k_entry -> k_transport_lean), 48 bytes per photon
    // a second set of what the pre-pass kernels of a launch write (photon order, tiles' ends, entry records): the pre-pass of launch i + 1
    // runs on a stream of its own beside the photon loop of launch i (28 registers against the loop's 80 x 6: one more wave per SIMD fits)
    DevBuf<uint32_t> d_order2, d_cursor2;
    DevBuf<float4> d_entry2;
    hipStream_t pre_stream = nullptr;
    hipEvent_t pre_done[2] = {nullptr, nullptr}, pre_loop[2] = {nullptr, nullptr};   // pre-pass written / photon loop through with the set
    bool pre_used[2] = {false, false};
    hipEvent_t pre_main = nullptr;   // the last pre-pass that ran on the MAIN stream (a one-stream launch): the pre-pass stream waits for it (shared d_hist / d_tile)
    bool pre_main_used = false;
    uint64_t pre_no = 0;             // launches so far: launch n uses set n & 1
    int pre_last = 0;                // the set of the last launch (mi3d_debug_order)
    int overlap_pre = 1;             // mi3d_set_tuning "overlap_pre" (MI3D_OVERLAP_PRE): 1 two sets, the pre-pass beside the previous photon loop; 0 one stream
    int cam_images = -1;             // mi3d_set_tuning "cam_images": periodic images of a camera an event contributes to, in domain lengths around the nearest one;
                                     // -1 (default): 2 where the ray kernel serves the job, the nearest image alone (with a warning) where it cannot
    bool cam_warned = false;         // the warning of that fall-back has been printed for this handle
    bool general_warned = false;     // ... and the one that says a job has landed on the general photon loop without having asked for it
    int entry_records = 1;           // mi3d_set_tuning "entry_records": 0: new photons are launched inside the photon loop
    // marched views served by k_rays: event lists (one per XCD) and their counters; events per photon seen so far
    DevBuf<float4> d_events;
    DevBuf<unsigned long long> d_evctr, d_hvlist;
    // a second set of lists (round 5): the ray kernel of launch i works through set i & 1 on a stream of its own while the photon loop of
    // launch i + 1 fills the other set -- the tail of either kernel (a tenth of a launch) no longer leaves the chip half empty
    DevBuf<float4> d_events2;
    DevBuf<unsigned long long> d_evctr2, d_hvlist2;
    DevCold cold_host2;              // the second set's DevCold (d_cold.p + 1): cold_host with the other lists
    hipStream_t rays_stream = nullptr;
    hipEvent_t set_emit[2] = {nullptr, nullptr}, set_rays[2] = {nullptr, nullptr};   // photon loop / ray kernels of the launch that used the set last
    bool set_used[2] = {false, false};
    int overlap_rays = 0;            // mi3d_set_tuning "overlap_rays" (MI3D_OVERLAP_RAYS): 0 (default): one stream, one set; 1: measured 2.4 % SLOWER on the nine-view
                                     // workload (profiles/r05/ab_overlap_rays.log): both kernels wait on memory, side by side they wait more
    int rays_wg = 0, emit_wg = 0;    // mi3d_set_tuning "rays_wg" / "emit_wg": workgroups per CU of the ray kernel's light build / of the event-writing
                                     // photon loop (0: the builds' own figures) -- what share of a CU each takes while the two run side by side
    double ev_per_photon = 0.0;      // 0: nothing known, the next run with marched views starts with a pilot launch
    int n_xcd = 8;                   // XCDs workgroups of this device land on (k_xcc_census): that many event lists fill
    unsigned long long *h_evctr = nullptr;   // pinned: [kEvSlots][9 * kCtrStride] fill counters of the last launches, copied out in stream order
    hipEvent_t ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t ev_nb[4] = {0, 0, 0, 0};
    bool ev_busy[4] = {false, false, false, false};
    uint64_t ev_capn[4] = {0, 0, 0, 0};   // capacity of the lists the launch of each slot wrote to
    unsigned ev_epoch = 0, ev_epochn[4] = {0, 0, 0, 0};   // what ev_per_photon is an estimate FOR changes with the scene: a launch of an earlier epoch says nothing about it (ev_forget)
    bool ev_void[4] = {false, false, false, false};   // the launch's tallies have been cleared since (mi3d_reset): a full list no longer matters
    int ev_cap_log2 = 28;            // records per XCD list at most, log2 (and never more than a quarter of the free memory in all: 1.5e8 per list on an otherwise empty
                                     // MI355X): 2^26 / 2^27 / 2^28 -> 3.53 / 3.73 / 3.76e8 photons/s with nine views (launch tails; profiles/r05/ab_evcap.log)
    // flux jobs served by k_transport_flux: tally records, sorted into bins and summed after every launch (mi3d_kernel_flux.hip)
    DevBuf<uint2> d_tl_rec, d_tl_binned;
    DevBuf<uint32_t> d_tl_words;     // chunk fills, histogram, bin starts, placement cursors
    DevBuf<unsigned long long> d_tl_cursor;
    DevBuf<unsigned long long> d_tl_stats;   // [64][4] the counters of each launch in flight, copied there by its k_tl_prefix (tl_note reads them; slots as d_tldesc)
    // a second set of record lists: the sort and the sums of launch i run on a stream of their own (memory-bound, a few waves per CU) while
    // the photon loop of launch i + 1 (issue-bound) fills the other set; the sorted copy (d_tl_binned) is the sort stream's alone
    DevBuf<uint2> d_tl_rec2;
    DevBuf<uint2> d_tl_binned2;             // the second set's sorted copy (round 6: the sums of a launch wait on the main stream for the next launch of their set)
    // The sums of a launch on two streams (k_tl_sum: 128 KB of LDS, no room on a CU the photon loop holds) are not queued behind its sort: on the
    // sort stream they crept along in the loop's tails and kept the NEXT launch's sort waiting for as long as the loop beside them ran.  They
    // are launched on the MAIN stream, between two photon loops: in front of the next launch of their set, or by whoever joins (tl_join).
    struct PendingSum { bool on = false; TallyList TL; double *flux = nullptr; unsigned nflux = 0; double *heat = nullptr; unsigned nheat = 0; int split = 1; };
    PendingSum tl_pend[2];
    DevBuf<float4> d_tl_runs, d_tl_runs2;   // run records (round 6: one 32-byte record per flight through uniform layers, expanded by k_tl_runs), one buffer per set
    DevBuf<uint32_t> d_tl_words2;
    DevBuf<unsigned long long> d_tl_cursor2;
    hipStream_t tl_stream = nullptr;
    hipEvent_t tl_filled[2] = {nullptr, nullptr}, tl_sorted[2] = {nullptr, nullptr};   // photon loop / sums of the launch that used the set last
    bool tl_set_used[2] = {false, false};
    unsigned runs_unread = 0;        // runs since a call last looked at the tallies (sync_main): > 0 -- runs are queued back to back
    bool tl_unjoined = false;        // the main stream has not been made to wait for the last sorts yet (tl_join)
    hipEvent_t tl_scattered[2] = {nullptr, nullptr};   // the sort of the launch that used the set last is through its lists (its sums may still run)
    uint64_t tl_launch_no = 0;       // launches with record lists so far: launch n uses set n & 1
    int overlap_sort = 1;            // mi3d_set_tuning "overlap_sort" (MI3D_OVERLAP_SORT): 1 two sets, the sort beside the next photon loop, where a run is long enough or
                                     // queued behind another (mi3d_run); 2: always; 0 one stream
    int tl_split = 4;                // mi3d_set_tuning "tl_split": a run with overlap_sort is worked off in at least this many launches (the last sort is not hidden)
    double tl_per_photon = 0.0;      // tally records per photon the photon loop reserved in its lists, seen so far (0: nothing known)
    double tl_total_pp = 0.0, tl_runs_pp = 0.0;   // ... records per photon in all (expanded runs included: what the sorted copy holds), run records per photon
    int tally_runs = 1;              // mi3d_set_tuning "tally_runs" (MI3D_TALLY_RUNS): 1 flights through uniform layers leave run records (k_tl_runs expands them); 0 a record per level
    int tl_cap_log2 = 31;            // most records the lists may hold, log2 (mi3d_set_tuning "tlcap_log2": tests of the full-list path)
    int tally_lists = 1;             // mi3d_set_tuning "tally_lists": 0: every flux tally is an atomic (MI3D_TALLY_LISTS overrides)
    int lds_max = 65536;             // bytes of LDS a workgroup may ask for
    unsigned long long *h_tlctr = nullptr;   // pinned: [kEvSlots] records reserved by the last launches
    DevBuf<char> d_tldesc;           // [64] TallyList: the description the photon loop of each launch in flight reads
    char *h_tldesc = nullptr;        // pinned source of those copies
    hipEvent_t tldesc_ev[64] = {};   // the photon loop that read slot i is through: the host waits for it before it fills the slot again (it may be dozens of small launches ahead)
    bool tldesc_used[64] = {};
    hipEvent_t tl_done[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t tl_nb[4] = {0, 0, 0, 0}, tl_cap[4] = {0, 0, 0, 0}, tl_bcap[4] = {0, 0, 0, 0}, tl_rcap[4] = {0, 0, 0, 0};
    bool tl_busy[4] = {false, false, false, false};
    int kernel_choice = 0;           // 0: the lean kernels where they apply (marched views through k_rays), 1: always k_transport
                                     // (MI3D_KERNEL=generic); A/B and tests
    int tile_cols = -1;              // tile edge in columns: -1 choose from the scene, 0 no sorting (MI3D_TILE_COLS overrides)
    uint64_t batch = (uint64_t)1 << 30; // most photons per kernel launch (order and tile buffers hold one launch: 4 GB + 2 GB; entry records 48 B per photon, where half
                                        // of the free memory holds them).  Every launch ends with a tail in which the chip runs empty: 2^27 -> 2^29 is worth 2.8 %
                                        // (profiles/r02/launch_batch_size.log), 2^29 -> 2^30 another 1.4 % (profiles/r05/ab_batch_2p30.log)
    DevCold cold_host;               // source of the asynchronous upload in fill_scene: must outlive the call

    bool dirty_grid = true, dirty_phase = true, dirty_sfc = true, dirty_tally = true, dirty_views = true;
    bool have_1d = false;

    // ---- run statistics (mi3d_stats_*): index 0 radiance, 1 flux
    DevBuf<float> d_run_own[2], d_stat_out;
    DevBuf<float> d_get_out;         // mi3d_get_flux / mi3d_get_heating: the normalised float32 field on its way to the host
    DevBuf<double> d_get_add;        // ... and its per-level term (the analytic direct beam, the layers' thickness)
    float *run_ext[2] = {nullptr, nullptr};
    DevBuf<double> d_sum[2], d_sumsq[2];
    DevBuf<float> d_factor[2];
    bool stats_on = false, stats_joined = false;   // joined: this handle adds into another handle's run fields (mi3d_stats_join)
    hipEvent_t stats_ev = nullptr;   // recorded after every statistics kernel of this handle (mi3d_stats_chain)
    int stats_nrun = 0;
    double analytic_share = 1.0;     // mi3d_stats_set_analytic_share
    float *run_ptr(int w) { return run_ext[w] ? run_ext[w] : d_run_own[w].p; }
    size_t stat_elems(int w) const { return w == 0 ? (size_t)nview * nxr * nyr : flux_elems(); }

    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double kernel_ms = 0.0;
    uint64_t launches = 0;
    std::string last_kernel;

    size_t rad_elems() const { return (size_t)(nview > 0 ? nview : 1) * nxr * nyr; }
    size_t flux_elems() const { return (size_t)3 * (nz + 1) * nx * ny; }
    tally_t *rad_ptr() { return rad_ext ? rad_ext : d_rad_own.p; }
    tally_t *flux_ptr() { return flux_ext ? flux_ext : d_flux_own.p; }
};

namespace {

int drain_events(mi3d_solver *h) {
    for (auto &pr : h->pending) {
        HIPCHK(hipEventSynchronize(pr.second));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        h->kernel_ms += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    h->pending.clear();
    return MI3D_OK;
}

// Host part of the layer table: 1-D optical properties, optical depth above, uniform-layer runs.
// `uniform3d[k3]` / `bt3d[k3]` come from k_layer_uniform (total extinction of a 3-D layer that does
// not vary horizontally).
int build_layers(mi3d_solver *h, const std::vector<int> &uniform3d, const std::vector<float> &bt3d,
                 std::vector<LayerRec> &lay) {
    const int nz = h->nz;
    const int k3lo = h->nz3 > 0 ? h->iz3l - 1 : 0;
    const int k3hi = h->nz3 > 0 ? k3lo + h->nz3 : 0;
    lay.assign(nz, LayerRec{});
    std::vector<double> bt(nz);
    h->n_step3d = 0;
    for (int k = 0; k < nz; ++k) {
        LayerRec &L = lay[k];
        double b = h->abs1d[k];
        for (int ip = 0; ip < h->np1d; ++ip) {
            const double e = h->ext1d[(size_t)ip * nz + k];
            b += e;
            L.ks1d[ip] = (float)(e * (double)h->omg1d[(size_t)ip * nz + k]);
            L.apf1d[ip] = h->apf1d[(size_t)ip * nz + k];
        }
        bt[k] = b > 0.0 ? b : 0.0;
        L.zlo = (float)h->zgrd[k];
        L.dz = (float)(h->zgrd[k + 1] - h->zgrd[k]);
        L.bt = (float)bt[k];
        L.flags = 0;
        if (k >= k3lo && k < k3hi) {
            L.flags |= kLayIn3d;
            if (uniform3d[k - k3lo]) L.bt = bt3d[k - k3lo];
            else { L.flags |= kLayStep3d; L.bt = 0.0f; h->n_step3d++; }
        }
    }
    // vertical optical depth above each 1-D layer (to TOA, or to the bottom of the 3-D region)
    double acc = 0.0;
    for (int k = nz - 1; k >= k3hi; --k) { lay[k].tabove = (float)acc; acc += bt[k] * (h->zgrd[k + 1] - h->zgrd[k]); }
    acc = 0.0;
    for (int k = k3lo - 1; k >= 0; --k) { lay[k].tabove = (float)acc; acc += bt[k] * (h->zgrd[k + 1] - h->zgrd[k]); }
    // runs of consecutive uniform layers and the vertical optical depth below each layer
    acc = 0.0;
    for (int k = 0; k < nz; ++k) {
        lay[k].tauz = (float)acc;
        if (!(lay[k].flags & kLayStep3d)) acc += (double)lay[k].bt * (double)lay[k].dz;
    }
    for (int k = 0; k < nz;) {
        if (lay[k].flags & kLayStep3d) { lay[k].run_lo = k + 1; lay[k].run_hi = k; ++k; continue; }
        int e = k;
        while (e + 1 < nz && !(lay[e + 1].flags & kLayStep3d)) ++e;
        for (int j = k; j <= e; ++j) { lay[j].run_lo = k; lay[j].run_hi = e; }
        k = e + 1;
    }
    return MI3D_OK;
}

int build_tables(mi3d_solver *h) {
    if (h->npf <= 0) return MI3D_OK;
    const int n = h->nang;
    std::vector<double> mu(n), p(n), cdf(n);
    std::vector<float> fmu(n), fp((size_t)n * h->npf), fcdf((size_t)n * h->npf);
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < n; ++j) mu[j] = std::cos((double)h->ang[n - 1 - j] * pi / 180.0);
    mu[0] = -1.0; mu[n - 1] = 1.0;
    for (int j = 1; j < n; ++j)
        if (!(mu[j] > mu[j - 1])) return fail(MI3D_EINVAL, "phase-function angles must ascend strictly from 0 to 180");
    for (int j = 0; j < n; ++j) fmu[j] = (float)mu[j];
    for (int t = 0; t < h->npf; ++t) {
        for (int j = 0; j < n; ++j) p[j] = h->pha[(size_t)t * n + (n - 1 - j)];
        double tot = 0.0;
        for (int j = 1; j < n; ++j) tot += 0.25 * (p[j] + p[j - 1]) * (mu[j] - mu[j - 1]);
        if (!(tot > 0.0)) return fail(MI3D_EINVAL, "phase function %d integrates to %g", t + 1, tot);
        cdf[0] = 0.0;
        for (int j = 0; j < n; ++j) p[j] /= tot;
        for (int j = 1; j < n; ++j) cdf[j] = cdf[j - 1] + 0.25 * (p[j] + p[j - 1]) * (mu[j] - mu[j - 1]);
        cdf[n - 1] = 1.0;
        for (int j = 0; j < n; ++j) { fp[(size_t)t * n + j] = (float)p[j]; fcdf[(size_t)t * n + j] = (float)cdf[j]; }
    }
    int rc;
    if ((rc = h->d_tmu.upload(fmu.data(), n))) return rc;
    if ((rc = h->d_tp.upload(fp.data(), fp.size()))) return rc;
    if ((rc = h->d_tcdf.upload(fcdf.data(), fcdf.size()))) return rc;
    // bucket indices for the lean kernels' look-ups (lean_tab_find): per bucket edge the largest node whose (float) value does not exceed it
    if (n > 65535) return fail(MI3D_EINVAL, "phase tables of more than 65 535 angles");
    // (entry b: the last node whose bucket -- tab_bucket_mu / tab_bucket_u of its float value, the kernels' own arithmetic -- is below b)
    auto build_idx = [&](const float *a, bool is_mu, uint16_t *out) {
        int i = -1;
        for (int b = 0; b <= kTabNB; ++b) {
            while (i + 1 < n && (is_mu ? tab_bucket_mu(a[i + 1]) : tab_bucket_u(a[i + 1])) < b) ++i;
            out[b] = (uint16_t)std::min(std::max(i, 0), n - 2);
        }
        for (int b = kTabNB + 1; b < kTabIdxN; ++b) out[b] = out[kTabNB];
    };
    std::vector<uint16_t> mi(kTabIdxN), ci((size_t)h->npf * kTabIdxN);
    build_idx(fmu.data(), true, mi.data());
    for (int t = 0; t < h->npf; ++t) build_idx(fcdf.data() + (size_t)t * n, false, ci.data() + (size_t)t * kTabIdxN);
    if ((rc = h->d_tmuidx.upload(mi.data(), mi.size()))) return rc;
    if ((rc = h->d_tcdfidx.upload(ci.data(), ci.size()))) return rc;
    return MI3D_OK;
}

int needs_tables(const mi3d_solver *h) {
    for (float a : h->apf1d)
        if (a >= 1.0f) return 1;
    return 0; // 3-D apf values are not scanned on the host; the kernel falls back to isotropic if npf == 0
}

int build_views(mi3d_solver *h) {
    std::vector<ViewRec> v(h->nview > 0 ? h->nview : 1);
    const double pi = 3.14159265358979323846;
    const double ztoa = h->zgrd[h->nz];
    h->nmarch = 0;
    h->col0 = -1;
    if (h->rad_kind == 1) {
        // cameras: axes = world axes turned by Rz(phi) Ry(the) Rz(psi) (er3t/rtm/mca/mca_inp.py:324-330)
        std::vector<CamRec> cams(h->nview > 0 ? h->nview : 1);
        const double Lx = h->dx * h->nx, Ly = h->dy * h->ny;
        for (int iv = 0; iv < h->nview; ++iv) {
            const double t = h->view_the[iv] * pi / 180.0, p = h->view_phi[iv] * pi / 180.0, q = h->cam_psi[iv] * pi / 180.0;
            const double ct = std::cos(t), st = std::sin(t), cp = std::cos(p), sp = std::sin(p), cq = std::cos(q), sq = std::sin(q);
            const double Z[3] = {st * cp, st * sp, ct};
            const double X[3] = {cp * ct * cq - sp * sq, sp * ct * cq + cp * sq, -st * cq};
            const double Y[3] = {Z[1] * X[2] - Z[2] * X[1], Z[2] * X[0] - Z[0] * X[2], Z[0] * X[1] - Z[1] * X[0]};
            CamRec &C = cams[iv];
            C.cx = (float)(h->cam_xpos[iv] * Lx); C.cy = (float)(h->cam_ypos[iv] * Ly); C.cz = (float)h->view_zloc[iv];
            C.r2min = (float)(h->cam_apsize[iv] * h->cam_apsize[iv]);
            C.zx = (float)Z[0]; C.zy = (float)Z[1]; C.zz = (float)Z[2];
            C.xx = (float)X[0]; C.xy = (float)X[1]; C.xz = (float)X[2];
            C.yx = (float)Y[0]; C.yy = (float)Y[1]; C.yz = (float)Y[2];
            C.cos_half = (float)std::cos(0.5 * h->cam_qmax[iv] * pi / 180.0);
            C.inv_du = (float)(h->nxr / (h->cam_umax[iv] * pi / 180.0));
            C.inv_dv = (float)(h->nyr / (h->cam_vmax[iv] * pi / 180.0));
            ViewRec &V = v[iv];
            std::memset(&V, 0, sizeof(V));
            V.vx = C.zx; V.vy = C.zy; V.vz = C.zz; V.zs = C.cz; V.zreg = C.cz;
            V.column = 0; V.point = 1;
            V.roulette = h->le_tau1 > 0.0 ? 1 : 0;
            h->nmarch++;
        }
        int rc = h->d_cams.upload(cams.data(), cams.size());
        if (rc) return rc;
        return h->d_views.upload(v.data(), v.size());
    }
    for (int iv = 0; iv < h->nview; ++iv) {
        const double t = h->view_the[iv] * pi / 180.0, p = h->view_phi[iv] * pi / 180.0;
        const double vx = -std::sin(t) * std::cos(p), vy = -std::sin(t) * std::sin(p), vz = -std::cos(t);
        const bool vertical = std::fabs(vx) < 1e-7 && std::fabs(vy) < 1e-7;
        ViewRec &V = v[iv];
        std::memset(&V, 0, sizeof(V));
        const bool down = vz > 0.0;   // down-looking sensor: the light travels up to it
        V.vx = vertical ? 0.0f : (float)vx; V.vy = vertical ? 0.0f : (float)vy; V.vz = vertical ? (down ? 1.0f : -1.0f) : (float)vz;
        double zs = h->view_zloc[iv] < ztoa ? h->view_zloc[iv] : ztoa;
        if (!down && zs < h->zgrd[0]) zs = h->zgrd[0];
        V.zs = (float)zs;
        V.zreg = (float)(down ? h->zref : zs);
        V.column = (h->column_le && down && vertical && h->view_zloc[iv] >= ztoa) ? 1 : 0;
        const bool free_of_charge = down && vertical && h->view_zloc[iv] >= ztoa;   // answered from the column table (or could be): no roulette
        V.roulette = ((h->le_tau1 > 0.0 && !free_of_charge) ? 1 : 0) | ((h->le_cmin > 0.0 && !free_of_charge) ? 2 : 0);
        if (!V.column) h->nmarch++;
        else if (h->col0 < 0) h->col0 = iv;
    }
    return h->d_views.upload(v.data(), v.size());
}

int fill_scene(mi3d_solver *h, DevScene &S) {
    std::memset(&S, 0, sizeof(S));
    DevCold &C = h->cold_host;
    std::memset(&C, 0, sizeof(C));
    const double Lx = h->dx * h->nx, Ly = h->dy * h->ny;
    S.nz = h->nz;
    S.k3lo = h->nz3 > 0 ? h->iz3l - 1 : 0;
    S.nx = h->nx; S.ny = h->ny; S.nz3 = h->nz3; S.np1d = h->np1d; S.np3d = h->np3d;
    S.kdir = h->kdir;
    S.dx = (float)h->dx; S.dy = (float)h->dy;
    C.Lx = (float)Lx; C.Ly = (float)Ly;
    C.inv_dx = (float)(1.0 / h->dx); C.inv_dy = (float)(1.0 / h->dy);
    C.inv_nx = (float)(1.0 / h->nx); C.inv_ny = (float)(1.0 / h->ny);
    S.pix_sx = (float)(h->nxr / Lx); S.pix_sy = (float)(h->nyr / Ly);
    S.bext3 = h->d_bext3.p; S.vrec = h->d_vrec.p; S.vcol_f4 = h->vcol_f4; S.vrow_f4 = h->vrow_f4; C.csca = h->d_csca.p; C.tcol0 = h->d_tcol0.p;
    const double pi = 3.14159265358979323846;
    const double th = h->src_the * pi / 180.0, ph = h->src_phi * pi / 180.0;
    C.sdx = (float)(std::sin(th) * std::cos(ph));
    C.sdy = (float)(std::sin(th) * std::sin(ph));
    C.sdz = (float)std::cos(th);
    C.cos_cone = (float)std::cos(0.5 * h->src_qmax * pi / 180.0);
    if (h->src_qmax <= 0.0) C.cos_cone = 1.0f;
    S.nview = h->nview; S.nmarch = h->nmarch; S.nxr = h->nxr; S.nyr = h->nyr; S.rad_row = h->nxr; S.col0 = h->col0 > 0 ? h->col0 : 0;
    S.target = h->target; S.solver = h->solver; S.wmin = (float)h->wmin; S.wfac = (float)h->wfac;
    {   // er3t's default mixture (mca_atm.py:95-102,299-303): Rayleigh as the one 1-D constituent, Henyey-Greenstein in every voxel:
        // the lean kernels then evaluate the two phase functions without looking at their selectors (bit 8 of the target word)
        bool plain = h->np1d == 1 && h->hg3d;
        for (float a : h->apf1d) plain = plain && (a > -1.5f && a <= -1.0f);
        if (plain) S.target |= kTargetPlainPhase;
        bool ray1 = h->np1d == 1;
        for (float a : h->apf1d) ray1 = ray1 && (a > -1.5f && a <= -1.0f);
        if (ray1) S.target |= kTargetRayleigh1d;
    }
    S.rad = h->rad_ptr(); S.flux = h->flux_ptr(); S.rad_stride = 1;
    C.next_photon = h->d_next.p;
    C.le_tau1 = (float)h->le_tau1;
    C.le_cmin = (float)h->le_cmin;

    {   // direct beam above the 3-D region (everything, without one): horizontally uniform, known analytically
        const int nz = h->nz;
        h->kdir = h->nz3 > 0 ? (h->iz3l - 1) + h->nz3 : 0;
        // ... and so it is in the horizontally uniform layers at the top of the 3-D region (clear air above the highest cloud
        // top: 15 of the 50 layers of BASELINE config 3): nothing the beam has met so far varies from column to column
        while (h->kdir > 0 && h->kdir <= nz && (int)h->lay_host.size() == nz && (h->lay_host[h->kdir - 1].flags & kLayIn3d) &&
               !(h->lay_host[h->kdir - 1].flags & kLayStep3d))
            h->kdir--;
        // (a source cone much wider than the solar disc -- er3t hard-wires 0.533 deg, mcarats.py:378 -- spreads the path
        //  lengths of the direct beam: then every crossing is tallied like anywhere else)
        if (h->src_qmax > 1.0) h->kdir = nz + 1;
        S.kdir = h->kdir;
        h->dir_level.assign(nz + 1, 0.0);
        const double mu0 = std::fabs(std::cos(th));
        double tau = 0.0;
        for (int L = nz; L >= h->kdir; --L) {
            if (L < nz) {
                double b = h->abs1d[L];
                for (int ip = 0; ip < h->np1d; ++ip) b += h->ext1d[(size_t)ip * nz + L];
                if ((int)h->lay_host.size() == nz && (h->lay_host[L].flags & kLayIn3d)) b = h->lay_host[L].bt;   // uniform layer of the 3-D region
                tau += (b > 0.0 ? b : 0.0) * (h->zgrd[L + 1] - h->zgrd[L]);
            }
            h->dir_level[L] = mu0 > 0.0 ? std::exp(-tau / mu0) : 0.0;
        }
    }
    C.ztoa = (float)h->zgrd[h->nz]; C.zref = (float)h->zref;
    C.inv_Lx = (float)(1.0 / Lx); C.inv_Ly = (float)(1.0 / Ly);
    C.nang = h->nang; C.npf = h->npf; C.tmu = h->d_tmu.p; C.tp = h->d_tp.p; C.tcdf = h->d_tcdf.p;
    C.tmu_idx = h->d_tmuidx.p; C.tcdf_idx = h->d_tcdfidx.p;
    {   // which tables does the scene refer to?  (1-D selectors scanned here, 3-D ones by k_apf_range)
        int lo = h->tab3d_lo, hi = h->tab3d_hi;
        for (float a : h->apf1d)
            if (a >= 1.0f) {
                const float t = a - 1.0f;
                const int i0 = (int)std::floor(t);
                lo = std::min(lo, i0);
                hi = std::max(hi, t > (float)i0 ? i0 + 1 : i0);
            }
        h->tab_lo = 0; h->tab_n = 0;
        if (h->npf > 0 && hi >= 0) {
            lo = std::max(lo, 0); hi = std::min(hi, h->npf - 1);
            // what staging costs the largest consumer: mu, p, cdf AND the bucket indices of the lean kernels (many short tables: the indices
            // dominate -- ADVICE r5), on top of the largest fixed part (the ray kernels' pools, the general kernel's per-lane stash)
            const size_t bytes = std::max((size_t)(1 + 2 * (hi - lo + 1)) * h->nang * sizeof(float), lean_tab_floats(h->nang, hi - lo + 1) * sizeof(float));
            const size_t fixed = std::max((size_t)h->nz * sizeof(LayerRec) + MI3D_MAX_VIEW * sizeof(ViewRec) + sizeof(DevCold) + (size_t)9 * 256 * sizeof(float),
                                          (size_t)(h->nz + 2) * sizeof(LayerRec) + MI3D_MAX_VIEW * sizeof(ViewRec) + sizeof(DevCold) + rays_lds_extra(h->nz, true));
            const size_t room = fixed < 64 * 1024 ? 64 * 1024 - fixed : 0; // default dynamic-LDS limit of a launch
            if (hi >= lo && bytes <= kTabLdsBudget + (size_t)(1 + (hi - lo + 1)) * kTabIdxN * sizeof(uint16_t) && bytes <= room) { h->tab_lo = lo; h->tab_n = hi - lo + 1; }
        }
        C.tab_lo = h->tab_lo; C.tab_n = h->tab_n;
    }
    C.sfc_mtype = h->sfc_mtype; C.nxb = h->nxb; C.nyb = h->nyb;
    C.sfc_p0 = h->sfc_param[0]; C.sfc_p1 = h->sfc_param[1]; C.sfc_p2 = h->sfc_param[2];
    C.sfc_p3 = h->sfc_param[3]; C.sfc_p4 = h->sfc_param[4];
    C.sfc_sx = (float)(h->nxb / Lx); C.sfc_sy = (float)(h->nyb / Ly);
    C.sfc2d = h->sfc2d_host.empty() ? nullptr : h->d_sfc2d.p;
    C.lay = h->d_lay.p; C.views = h->d_views.p; C.counters = h->d_counters.p;
    C.cams = h->rad_kind == 1 ? h->d_cams.p : nullptr;
    C.heat = (h->target & MI3D_TARGET_HEAT) ? h->heat_ptr() : nullptr;
    int rc = h->d_cold.alloc(2);     // ([1]: the second set of event lists)
    if (rc) return rc;
    S.cold = h->d_cold.p;     // (uploaded by mi3d_run from h->cold_host, which outlives the asynchronous copy)
    return MI3D_OK;
}

// The sorts and sums of a flux run's last launches may still be on their way on the sort stream when mi3d_run returns (the next run's
// photon loops then start beside them): whoever reads the tallies, clears them or changes what the kernels work on makes the main
// stream wait for them first.
// the sums of the launch that used set q last (mi3d_solver::tl_pend), on stream st
static hipError_t launch_sum(mi3d_solver *h, int q, hipStream_t st) {
    mi3d_solver::PendingSum &P = h->tl_pend[q];
    P.on = false;
    if ((sizeof(double) << P.TL.shift) > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tl_sum), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) << P.TL.shift));
    hipLaunchKernelGGL(k_tl_sum, dim3((unsigned)(P.TL.nbins * P.split)), dim3(1024), sizeof(double) << P.TL.shift, st, P.TL, P.flux, P.nflux, P.heat, P.nheat, P.split);
    return hipGetLastError();
}

static hipError_t tl_join(mi3d_solver *h) {
    if (!h->tl_unjoined) return hipSuccess;
    for (int q = 0; q < 2; ++q) {
        // (the main stream behind the sort of the set's last launch; that launch's sums, if they are still to come, on the main stream now)
        if (h->tl_set_used[q] && h->tl_scattered[q]) { const hipError_t e = hipStreamWaitEvent(h->stream, h->tl_scattered[q], 0); if (e != hipSuccess) return e; }
        if (h->tl_pend[q].on) { const hipError_t e = launch_sum(h, q, h->stream); if (e != hipSuccess) return e; }
    }
    h->tl_unjoined = false;
    return hipSuccess;
}
static hipError_t sync_main(mi3d_solver *h) {
    h->runs_unread = 0;
    const hipError_t e = tl_join(h);
    return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
}

// both streams of a handle (the ray kernels of launches with two sets of event lists run on a stream of their own)
static hipError_t sync_streams(mi3d_solver *h) {
    hipError_t e = tl_join(h);     // (sums that are still to come: on the main stream, before it is waited for)
    h->tl_unjoined = false;
    { const hipError_t e2 = hipStreamSynchronize(h->stream); if (e == hipSuccess) e = e2; }
    if (h->rays_stream) { const hipError_t e2 = hipStreamSynchronize(h->rays_stream); if (e == hipSuccess) e = e2; }
    if (h->tl_stream) { const hipError_t e2 = hipStreamSynchronize(h->tl_stream); if (e == hipSuccess) e = e2; }
    if (h->pre_stream) { const hipError_t e2 = hipStreamSynchronize(h->pre_stream); if (e == hipSuccess) e = e2; }
    return e;
}

// the launch's DevCold goes to the device: [0] with the first set of event lists, [1] the same with the second set
static hipError_t upload_cold(mi3d_solver *h, bool two_sets, bool pre_two = false) {
    hipError_t e = hipMemcpyAsync(h->d_cold.p, &h->cold_host, sizeof(DevCold), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess || !(two_sets || pre_two)) return e;
    h->cold_host2 = h->cold_host;
    if (two_sets) {
        h->cold_host2.ev_list = h->d_events2.p; h->cold_host2.ev_ctr = h->d_evctr2.p;
        h->cold_host2.hv_list = h->cold_host.hv_list ? h->d_hvlist2.p : nullptr;
    }
    if (pre_two) {   // (the second set of what the pre-pass writes)
        if (h->cold_host.order) h->cold_host2.order = h->d_order2.p;
        if (h->cold_host.entry) h->cold_host2.entry = h->d_entry2.p;
        if (h->cold_host.tile_end) h->cold_host2.tile_end = h->d_cursor2.p;
    }
    return hipMemcpyAsync(h->d_cold.p + 1, &h->cold_host2, sizeof(DevCold), hipMemcpyHostToDevice, h->stream);
}

// Nothing is known about the events per photon any more (another scene, surface, source, solver): the next run with marched views
// starts with a pilot launch, and launches still on their way no longer count (mi3d_run does not wait for its last ones).
static inline void ev_forget(mi3d_solver *h) { h->ev_per_photon = 0.0; h->ev_epoch++; }

int check_handle(mi3d_solver *h) {
    if (!h) return fail(MI3D_EINVAL, "null solver handle");
    HIPCHK(hipSetDevice(h->device));
    return MI3D_OK;
}

} // namespace

// =================================================================================================
extern "C" {

int mi3d_version(void) { return MI3D_VERSION; }

const char *mi3d_last_error(void) { return g_err.c_str(); }

int mi3d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mi3d_create(int device, mi3d_solver **out) {
    if (!out) return fail(MI3D_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MI3D_EDEVICE, "no HIP device available (%s); this solver has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(MI3D_EINVAL, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    mi3d_solver *h = new mi3d_solver();
    h->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        h->num_cu = prop.multiProcessorCount;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v >= 32768) h->lds_max = v; else (void)hipGetLastError(); }
    int rc;
    if ((rc = h->d_counters.alloc(MI3D_NCOUNTER)) || (rc = h->d_next.alloc(8 * kCtrStride)) ||
        (rc = h->d_hist.alloc(kMaxTiles)) || (rc = h->d_cursor.alloc(kMaxTiles))) { delete h; return rc; }
    HIPCHK(hipMemset(h->d_counters.p, 0, MI3D_NCOUNTER * sizeof(unsigned long long)));
    HIPCHK(hipMemset(h->d_next.p, 0, 8 * kCtrStride * sizeof(unsigned long long)));
    {   // XCDs in use (eight on an MI355X in SPX mode; fewer in CPX / QPX partitions)
        DevBuf<unsigned> fl;
        unsigned host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fl.alloc(8) == MI3D_OK && hipMemset(fl.p, 0, sizeof(host)) == hipSuccess) {
            hipLaunchKernelGGL(k_xcc_census, dim3(4096), dim3(64), 0, nullptr, fl.p);
            if (hipMemcpy(host, fl.p, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess) {
                int n = 0;
                for (unsigned f : host) n += f ? 1 : 0;
                if (n >= 1) h->n_xcd = n;
            }
        }
        (void)hipGetLastError();
        fl.release();
    }
    if (const char *e = getenv("MI3D_TILE_COLS")) h->tile_cols = atoi(e);          // tuning knobs, not part of the C-ABI
    if (const char *e = getenv("MI3D_RAD_SPREAD")) h->rad_spread = atoi(e);
    if (const char *e = getenv("MI3D_RAD_ROW_PAD")) h->rad_row_pad = atoi(e);
    if (const char *e = getenv("MI3D_TALLY_WINDOW")) h->tally_window = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MI3D_TALLY_LISTS")) h->tally_lists = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MI3D_TALLY_RUNS")) h->tally_runs = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MI3D_ENTRY_RECORDS")) h->entry_records = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MI3D_OVERLAP_RAYS")) h->overlap_rays = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MI3D_OVERLAP_SORT")) h->overlap_sort = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("MI3D_OVERLAP_PRE")) h->overlap_pre = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("MI3D_TL_SPLIT")) h->tl_split = std::max(1, std::min(64, atoi(e)));
    if (const char *e = getenv("MI3D_RAYS_WG")) h->rays_wg = std::max(0, std::min(8, atoi(e)));
    if (const char *e = getenv("MI3D_EMIT_WG")) h->emit_wg = std::max(0, std::min(8, atoi(e)));
    if (const char *e = getenv("MI3D_VPAD_COL")) h->vpad_col = std::max(0, atoi(e));
    if (const char *e = getenv("MI3D_VPAD_ROW")) h->vpad_row = std::max(0, atoi(e));
    if (const char *e = getenv("MI3D_KERNEL")) h->kernel_choice = std::strcmp(e, "generic") == 0 ? 1 : 0;
    if (const char *e = getenv("MI3D_EVCAP_LOG2")) { const int b = atoi(e); if (b >= 12 && b <= 28) h->ev_cap_log2 = b; }
    if (const char *e = getenv("MI3D_BATCH_LOG2")) { const int b = atoi(e); if (b >= 8 && b <= 30) h->batch = (uint64_t)1 << b; }
    *out = h;
    return MI3D_OK;
}

int mi3d_destroy(mi3d_solver *h) {
    if (!h) return MI3D_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto &pr : h->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (h->stats_ev) (void)hipEventDestroy(h->stats_ev);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    for (hipEvent_t &e : h->ev_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->set_emit) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->set_rays) if (e) (void)hipEventDestroy(e);
    if (h->rays_stream) (void)hipStreamDestroy(h->rays_stream);
    for (hipEvent_t &e : h->tldesc_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->tl_filled) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->tl_sorted) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->tl_scattered) if (e) (void)hipEventDestroy(e);
    if (h->tl_stream) (void)hipStreamDestroy(h->tl_stream);
    for (hipEvent_t &e : h->pre_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->pre_loop) if (e) (void)hipEventDestroy(e);
    if (h->pre_main) (void)hipEventDestroy(h->pre_main);
    if (h->pre_stream) (void)hipStreamDestroy(h->pre_stream);
    h->d_order2.release(); h->d_cursor2.release(); h->d_entry2.release();
    h->d_tl_rec2.release(); h->d_tl_words2.release(); h->d_tl_cursor2.release(); h->d_tl_runs.release(); h->d_tl_runs2.release(); h->d_tl_binned2.release();
    h->d_events2.release(); h->d_evctr2.release(); h->d_hvlist2.release();
    if (h->h_evctr) (void)hipHostFree(h->h_evctr);
    h->d_abst.release(); h->d_extp.release(); h->d_omgp.release(); h->d_apfp.release();
    h->d_lay.release(); h->d_vrec.release(); h->d_bext3.release(); h->d_tcol0.release(); h->d_tmu.release(); h->d_tp.release();
    h->d_tcdf.release(); h->d_tmuidx.release(); h->d_tcdfidx.release(); h->d_sfc2d.release(); h->d_csca.release(); h->d_rad_own.release();
    h->d_flux_own.release(); h->d_heat_own.release(); h->d_counters.release(); h->d_next.release();
    h->d_rad_acc.release(); h->d_cams.release();
    h->d_order.release(); h->d_hist.release(); h->d_cursor.release(); h->d_tile.release(); h->d_entry.release();
    h->d_events.release(); h->d_evctr.release(); h->d_hvlist.release();
    for (hipEvent_t &e : h->tl_done) if (e) (void)hipEventDestroy(e);
    if (h->h_tlctr) (void)hipHostFree(h->h_tlctr);
    if (h->h_tldesc) (void)hipHostFree(h->h_tldesc);
    h->d_tldesc.release();
    h->d_tl_rec.release(); h->d_tl_binned.release(); h->d_tl_words.release(); h->d_tl_cursor.release(); h->d_tl_stats.release();
    for (int w = 0; w < 2; ++w) { h->d_run_own[w].release(); h->d_sum[w].release(); h->d_sumsq[w].release(); h->d_factor[w].release(); }
    h->d_stat_out.release(); h->d_dir_level.release(); h->d_get_out.release(); h->d_get_add.release();
    h->d_views.release(); h->d_cold.release(); h->d_tabrange.release(); h->d_bt1d.release(); h->d_dz.release(); h->d_bmin.release(); h->d_bmax.release();
    delete h;
    return MI3D_OK;
}

int mi3d_set_atm1d(mi3d_solver *h, int nz, const double *zgrd, int np1d, const float *ext, const float *omg,
                   const float *apf, const float *abs) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (nz < 1 || nz > kMaxLayers) return fail(MI3D_EINVAL, "Atm_nz=%d outside [1,%d]", nz, kMaxLayers);
    if (np1d < 1 || np1d > MI3D_MAX_NP1D) return fail(MI3D_EINVAL, "Atm_np1d=%d outside [1,%d]", np1d, MI3D_MAX_NP1D);
    if (!zgrd || !ext || !omg || !apf) return fail(MI3D_EINVAL, "NULL 1-D profile");
    for (int k = 0; k < nz; ++k)
        if (!(zgrd[k + 1] > zgrd[k])) return fail(MI3D_EINVAL, "Atm_zgrd0 must ascend strictly (level %d)", k + 1);
    const size_t n = (size_t)nz * np1d;
    for (size_t i = 0; i < n; ++i)
        if (!(ext[i] >= 0.0f) || !(omg[i] >= 0.0f)) return fail(MI3D_EINVAL, "negative or NaN 1-D extinction / albedo");
    if (h->nz != nz) h->dirty_tally = true;
    h->nz = nz; h->np1d = np1d;
    h->zgrd.assign(zgrd, zgrd + nz + 1);
    h->ext1d.assign(ext, ext + n); h->omg1d.assign(omg, omg + n); h->apf1d.assign(apf, apf + n);
    if (abs) h->abs1d.assign(abs, abs + nz); else h->abs1d.assign(nz, 0.0f);
    h->have_1d = true;
    h->dirty_grid = true;
    return MI3D_OK;
}

int mi3d_set_atm3d(mi3d_solver *h, int nx, int ny, int nz3, int iz3l, int np3d, double dx, double dy,
                   const float *abst, const float *extp, const float *omgp, const float *apfp) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (nx < 1 || ny < 1 || nz3 < 0) return fail(MI3D_EINVAL, "bad grid %d x %d x %d", nx, ny, nz3);
    if (!(dx > 0.0) || !(dy > 0.0)) return fail(MI3D_EINVAL, "Atm_dx / Atm_dy must be positive");
    if (nz3 > 0) {
        if (np3d < 1 || np3d > MI3D_MAX_NP3D) return fail(MI3D_EINVAL, "Atm_np3d=%d outside [1,%d]", np3d, MI3D_MAX_NP3D);
        if (!extp || !omgp || !apfp) return fail(MI3D_EINVAL, "NULL 3-D array");
        if (iz3l < 1) return fail(MI3D_EINVAL, "Atm_iz3l=%d must be >= 1", iz3l);
    }
    if (h->nx != nx || h->ny != ny) h->dirty_tally = true;
    h->nx = nx; h->ny = ny; h->nz3 = nz3; h->iz3l = nz3 > 0 ? iz3l : 1; h->np3d = nz3 > 0 ? np3d : 0;
    h->dx = dx; h->dy = dy;
    h->has_abst = false;
    if (nz3 > 0) {
        const size_t nvox = (size_t)nx * ny * nz3;
        if (abst) { if ((rc = h->d_abst.upload(abst, nvox))) return rc; h->has_abst = true; }
        if ((rc = h->d_extp.upload(extp, nvox * np3d))) return rc;
        if ((rc = h->d_omgp.upload(omgp, nvox * np3d))) return rc;
        if ((rc = h->d_apfp.upload(apfp, nvox * np3d))) return rc;
    }
    h->dirty_grid = true;
    return MI3D_OK;
}

int mi3d_set_phase(mi3d_solver *h, int nang, int npf, const float *ang, const float *pha) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (npf <= 0) { h->npf = 0; h->nang = 0; h->ang.clear(); h->pha.clear(); h->dirty_phase = true; return MI3D_OK; }
    if (nang < 2 || !ang || !pha) return fail(MI3D_EINVAL, "bad phase table (Sca_nangi=%d)", nang);
    for (size_t i = 0; i < (size_t)nang * npf; ++i)
        if (!(pha[i] >= 0.0f)) return fail(MI3D_EINVAL, "negative or NaN phase-function value");
    h->nang = nang; h->npf = npf;
    h->ang.assign(ang, ang + nang);
    h->pha.assign(pha, pha + (size_t)nang * npf);
    h->dirty_phase = true;
    return MI3D_OK;
}

int mi3d_set_surface(mi3d_solver *h, int mtype, const float param[5]) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (mtype != MI3D_SFC_LAMBERT && mtype != MI3D_SFC_LSRT && mtype != MI3D_SFC_DSM) return fail(MI3D_EINVAL, "unknown Sfc_mtype=%d", mtype);
    if (!param) return fail(MI3D_EINVAL, "NULL Sfc_param");
    h->sfc_mtype = mtype;
    h->sfc_lambert_only = (mtype == MI3D_SFC_LAMBERT);
    for (int i = 0; i < 5; ++i) h->sfc_param[i] = param[i];
    h->sfc2d_host.clear(); h->nxb = h->nyb = 0;
    h->dirty_sfc = true;
    ev_forget(h);   // (a brighter surface: more events per photon; the next run with marched views starts with a pilot launch)
    return MI3D_OK;
}

int mi3d_set_surface2d(mi3d_solver *h, int nxb, int nyb, const float *tmps, const float *jsfc, const float *psfc) {
    (void)tmps;
    int rc = check_handle(h);
    if (rc) return rc;
    if (nxb < 1 || nyb < 1 || !jsfc || !psfc) return fail(MI3D_EINVAL, "bad 2-D surface (%d x %d)", nxb, nyb);
    const size_t n = (size_t)nxb * nyb;
    std::vector<float> packed(n * 8, 0.0f);
    bool lambert_only = true;
    for (size_t i = 0; i < n; ++i) {
        const int t = (int)std::lround(jsfc[i]);
        lambert_only = lambert_only && t == MI3D_SFC_LAMBERT;
        if (t != MI3D_SFC_LAMBERT && t != MI3D_SFC_LSRT && t != MI3D_SFC_DSM) return fail(MI3D_EINVAL, "unknown surface model id %d in jsfc2d", t);
        packed[i * 8 + 0] = (float)t;
        for (int q = 0; q < 5; ++q) packed[i * 8 + 1 + q] = psfc[q * n + i];
    }
    h->sfc2d_host.swap(packed);
    h->sfc_lambert_only = lambert_only;
    h->nxb = nxb; h->nyb = nyb;
    h->dirty_sfc = true;
    ev_forget(h);
    return MI3D_OK;
}

int mi3d_set_source(mi3d_solver *h, double flx, double qmax_deg, double the_deg, double phi_deg) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!(the_deg > 90.0 && the_deg <= 180.0)) return fail(MI3D_EINVAL, "Src_the=%g: the sun must shine downwards (90 < the <= 180)", the_deg);
    if (!(qmax_deg >= 0.0 && qmax_deg < 90.0)) return fail(MI3D_EINVAL, "Src_qmax=%g out of range", qmax_deg);
    if (h->src_the != the_deg || h->src_phi != phi_deg || h->src_qmax != qmax_deg) ev_forget(h);
    h->src_flx = flx; h->src_qmax = qmax_deg; h->src_the = the_deg; h->src_phi = phi_deg;
    return MI3D_OK;
}

int mi3d_set_views(mi3d_solver *h, int nview, const double *the_deg, const double *phi_deg, const double *zloc,
                   double zref, int nxr, int nyr) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (nview < 0 || nview > MI3D_MAX_VIEW) return fail(MI3D_EINVAL, "Rad_nrad=%d outside [0,%d]", nview, MI3D_MAX_VIEW);
    if (nxr < 1 || nyr < 1) return fail(MI3D_EINVAL, "bad Rad_nxr/Rad_nyr");
    for (int i = 0; i < nview; ++i) {
        if (!(the_deg[i] >= 0.0 && the_deg[i] <= 180.0) || std::fabs(std::cos(the_deg[i] * 3.14159265358979323846 / 180.0)) <= 1e-6)
            return fail(MI3D_EINVAL, "Rad_the=%g: the line of sight must not be horizontal (0 <= the <= 180, the != 90)", the_deg[i]);
        h->view_the[i] = the_deg[i]; h->view_phi[i] = phi_deg[i]; h->view_zloc[i] = zloc[i];
    }
    if (h->nview != nview || h->nxr != nxr || h->nyr != nyr) h->dirty_tally = true;
    h->nview = nview; h->zref = zref; h->nxr = nxr; h->nyr = nyr;
    h->rad_kind = 2;
    h->dirty_views = true;
    return MI3D_OK;
}

int mi3d_set_cameras(mi3d_solver *h, int ncam, const double *the_deg, const double *phi_deg, const double *psi_deg,
                     const double *xpos, const double *ypos, const double *zloc, const double *qmax_deg,
                     const double *umax_deg, const double *vmax_deg, const double *apsize, int nxr, int nyr) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (ncam < 1 || ncam > MI3D_MAX_VIEW) return fail(MI3D_EINVAL, "Rad_nrad=%d outside [1,%d]", ncam, MI3D_MAX_VIEW);
    if (nxr < 1 || nyr < 1) return fail(MI3D_EINVAL, "bad Rad_nxr/Rad_nyr");
    if (!the_deg || !phi_deg || !psi_deg || !xpos || !ypos || !zloc || !qmax_deg || !umax_deg || !vmax_deg || !apsize)
        return fail(MI3D_EINVAL, "NULL camera array");
    for (int i = 0; i < ncam; ++i) {
        if (!(qmax_deg[i] > 0.0 && qmax_deg[i] <= 360.0) || !(umax_deg[i] > 0.0) || !(vmax_deg[i] > 0.0) || !(apsize[i] >= 0.0))
            return fail(MI3D_EINVAL, "camera %d: Rad_qmax=%g, Rad_umax=%g, Rad_vmax=%g, Rad_apsize=%g", i + 1, qmax_deg[i], umax_deg[i], vmax_deg[i], apsize[i]);
        h->view_the[i] = the_deg[i]; h->view_phi[i] = phi_deg[i]; h->view_zloc[i] = zloc[i];
        h->cam_psi[i] = psi_deg[i]; h->cam_xpos[i] = xpos[i]; h->cam_ypos[i] = ypos[i];
        h->cam_qmax[i] = qmax_deg[i]; h->cam_umax[i] = umax_deg[i]; h->cam_vmax[i] = vmax_deg[i]; h->cam_apsize[i] = apsize[i];
    }
    if (h->nview != ncam || h->nxr != nxr || h->nyr != nyr) h->dirty_tally = true;
    h->nview = ncam; h->nxr = nxr; h->nyr = nyr; h->zref = 0.0;
    h->rad_kind = 1;
    h->dirty_views = true;
    return MI3D_OK;
}

int mi3d_set_options(mi3d_solver *h, int target, int solver, double wmin, double wfac, int column_le) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (target < 1 || target > 7) return fail(MI3D_EINVAL, "target=%d", target);
    if ((target & MI3D_TARGET_HEAT) && !(target & MI3D_TARGET_FLUX))
        return fail(MI3D_EINVAL, "target=%d: heating rates (Flx_mhrt=1) come with the fluxes (MI3D_TARGET_FLUX | MI3D_TARGET_HEAT)", target);
    if ((target & MI3D_TARGET_HEAT) != (h->target & MI3D_TARGET_HEAT)) h->dirty_tally = true;
    if (solver != MI3D_SOLVER_3D && solver != MI3D_SOLVER_P3D && solver != MI3D_SOLVER_IPA) return fail(MI3D_EINVAL, "solver=%d", solver);
    if (!(wmin >= 0.0 && wmin <= 1.0)) return fail(MI3D_EINVAL, "Pho_wmin=%g outside [0,1]", wmin);
    if (!(wfac >= wmin && wfac > 0.0)) return fail(MI3D_EINVAL, "Pho_wfac=%g must be positive and not below Pho_wmin=%g", wfac, wmin);
    if (h->solver != solver || h->wmin != wmin || h->wfac != wfac) ev_forget(h);
    h->target = target; h->solver = solver; h->wmin = wmin; h->wfac = wfac; h->column_le = column_le ? 1 : 0;
    h->dirty_views = true;
    return MI3D_OK;
}

int mi3d_set_le_roulette(mi3d_solver *h, double tau1) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!(tau1 >= 0.0) || tau1 > 16.0) return fail(MI3D_EINVAL, "le roulette threshold %g outside [0, 16]", tau1);
    h->le_tau1 = tau1;
    h->dirty_views = true;
    return MI3D_OK;
}

int mi3d_set_le_weight_roulette(mi3d_solver *h, double cmin) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!(cmin >= 0.0) || cmin > 1.0) return fail(MI3D_EINVAL, "le weight roulette threshold %g outside [0, 1]", cmin);
    h->le_cmin = cmin;
    h->dirty_views = true;
    return MI3D_OK;
}

int mi3d_set_counting(mi3d_solver *h, int on) {
    int rc = check_handle(h);
    if (rc) return rc;
    h->counting = on ? 1 : 0;
    return MI3D_OK;
}

int mi3d_bind_device_buffers(mi3d_solver *h, void *rad_sum, void *flux_sum, void *stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    h->rad_ext = (tally_t *)rad_sum;
    h->flux_ext = (tally_t *)flux_sum;
    const hipStream_t st = stream ? (hipStream_t)stream : (h->use_own_stream ? h->own_stream : nullptr);
    if (st != h->stream) {
        (void)sync_streams(h);                            // nothing of this handle is left behind on the stream it leaves (a caller's stream
                                                          // that no longer exists is the caller's business: not an error here)
        (void)hipGetLastError();
        h->stream = st;
    }
    if (!rad_sum || !flux_sum) h->dirty_tally = true;     // own buffers are (re)created on demand by mi3d_prepare
    return MI3D_OK;
}

int mi3d_bind_heating_buffer(mi3d_solver *h, void *heat_sum) {
    int rc = check_handle(h);
    if (rc) return rc;
    h->heat_ext = (double *)heat_sum;
    if (!heat_sum && (h->target & MI3D_TARGET_HEAT)) h->dirty_tally = true;
    return MI3D_OK;
}

int mi3d_prepare(mi3d_solver *h) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!h->have_1d) return fail(MI3D_ESTATE, "mi3d_set_atm1d has not been called");
    // the uploads below are synchronous copies that do not wait for this handle's (non-blocking) stream: nothing still running
    // there may read what they overwrite
    if (h->dirty_grid || h->dirty_views || h->dirty_phase || h->dirty_sfc || h->dirty_tally) HIPCHK(sync_streams(h));
    if (h->dirty_grid) {
        const int nz = h->nz;
        const int k3lo = h->nz3 > 0 ? h->iz3l - 1 : 0, k3hi = h->nz3 > 0 ? k3lo + h->nz3 : 0;
        if (h->nz3 > 0 && (k3lo < 0 || k3hi > nz))
            return fail(MI3D_EINVAL, "3-D layers %d..%d (Atm_iz3l=%d, Atm_nz3=%d) do not fit Atm_nz=%d", k3lo + 1,
                        k3hi, h->iz3l, h->nz3, nz);
        std::vector<float> bt1d(nz), dz(nz);
        for (int k = 0; k < nz; ++k) {
            double b = h->abs1d[k];
            for (int ip = 0; ip < h->np1d; ++ip) b += h->ext1d[(size_t)ip * nz + k];
            bt1d[k] = (float)(b > 0.0 ? b : 0.0);
            dz[k] = (float)(h->zgrd[k + 1] - h->zgrd[k]);
        }
        h->tab3d_lo = 1 << 30; h->tab3d_hi = -1; h->hg3d = true;
        std::vector<int> uniform3d(h->nz3, 0);
        std::vector<float> bt3d(h->nz3, 0.0f);
        if (h->nz3 > 0) {
            const size_t nvox = (size_t)h->nx * h->ny * h->nz3, ncol = (size_t)h->nx * h->ny;
            if ((rc = h->d_bt1d.upload(bt1d.data(), nz)) || (rc = h->d_dz.upload(dz.data(), nz))) return rc;
            h->vcol_f4 = (unsigned)(h->nz3 + h->vpad_col); h->vrow_f4 = (unsigned)h->nx * h->vcol_f4 + (unsigned)h->vpad_row;
            if ((rc = h->d_bext3.alloc(nvox)) || (rc = h->d_vrec.alloc((size_t)h->ny * h->vrow_f4)) || (rc = h->d_csca.alloc(h->np3d > 1 ? nvox * h->np3d : 1)) ||
                (rc = h->d_tcol0.alloc(ncol)) || (rc = h->d_bmin.alloc(h->nz3)) ||
                (rc = h->d_bmax.alloc(h->nz3)))
                return rc;
            const int tb = 256;
            const float *abst = h->has_abst ? h->d_abst.p : nullptr;
            hipLaunchKernelGGL(k_build_grid, dim3((unsigned)((nvox + tb - 1) / tb)), dim3(tb), 0, h->stream, h->nx,
                               h->ny, h->nz3, k3lo, h->np3d, h->d_bt1d.p, abst, h->d_extp.p, h->d_omgp.p,
                               h->d_apfp.p, h->d_vrec.p, h->d_csca.p, h->vcol_f4, h->vrow_f4, h->d_bext3.p);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_layer_uniform, dim3(h->nz3), dim3(tb), 0, h->stream, h->nx, h->ny, h->nz3, k3lo,
                               h->np3d, h->d_bt1d.p, abst, h->d_extp.p, h->d_bmin.p, h->d_bmax.p);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_build_column, dim3((unsigned)((ncol + tb - 1) / tb)), dim3(tb), 0, h->stream,
                               (int)ncol, h->nz3, k3lo, nz, h->d_bt1d.p, h->d_dz.p, h->d_vrec.p, h->d_tcol0.p, h->nx, h->vcol_f4, h->vrow_f4);
            HIPCHK(hipGetLastError());
            int init[3] = {1 << 30, -1, 0};
            if ((rc = h->d_tabrange.upload(init, 3))) return rc;
            hipLaunchKernelGGL(k_apf_range, dim3(1024), dim3(tb), 0, h->stream, (long)(nvox * h->np3d), h->d_extp.p,
                               h->d_apfp.p, h->d_tabrange.p);
            HIPCHK(hipGetLastError());
            std::vector<float> bmin(h->nz3), bmax(h->nz3);
            HIPCHK(sync_main(h));
            HIPCHK(hipMemcpy(init, h->d_tabrange.p, sizeof(init), hipMemcpyDeviceToHost));
            h->tab3d_lo = init[0]; h->tab3d_hi = init[1]; h->hg3d = (init[2] == 0) && (init[1] < 0);
            HIPCHK(hipMemcpy(bmin.data(), h->d_bmin.p, h->nz3 * sizeof(float), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(bmax.data(), h->d_bmax.p, h->nz3 * sizeof(float), hipMemcpyDeviceToHost));
            double wsum = 0.0, wz = 0.0;
            for (int k3 = 0; k3 < h->nz3; ++k3) {
                uniform3d[k3] = (bmin[k3] == bmax[k3]) ? 1 : 0;
                bt3d[k3] = bmin[k3];
                const double wgt = (double)bmax[k3] - (double)bmin[k3];
                wsum += wgt; wz += wgt * 0.5 * (h->zgrd[k3lo + k3] + h->zgrd[k3lo + k3 + 1]);
            }
            h->z_cloud = wsum > 0.0 ? wz / wsum : -1.0;
        }
        std::vector<LayerRec> lay;
        if ((rc = build_layers(h, uniform3d, bt3d, lay))) return rc;
        if ((rc = h->d_lay.upload(lay.data(), lay.size()))) return rc;
        h->lay_host = lay;
        h->dirty_grid = false;
        h->dirty_views = true;
        // another scene (or another g of it: the gas absorption moves the events per photon too): the next run with marched views
        // starts with a pilot launch again -- an event list that runs full fails the run.  The tally-record lists of flux jobs KEEP what
        // earlier launches reported (round 6): a list of theirs that runs full loses nothing but speed, the next launch is sized by what this
        // one needed, and er3t's jobs -- sixteen g of one scene, a few million photons each -- are not cut in two by a pilot launch and a
        // wait each
        ev_forget(h);
        for (bool &b : h->tl_busy) b = false;
    }
    if (h->dirty_views) {
        if ((rc = build_views(h))) return rc;
        h->dirty_views = false;
    }
    if (h->dirty_phase) {
        if ((rc = build_tables(h))) return rc;
        h->dirty_phase = false;
    }
    if (h->npf <= 0 && needs_tables(h))
        return fail(MI3D_ESTATE, "a 1-D component selects a tabulated phase function (apf >= 1) but no table is loaded");
    if (h->dirty_sfc) {
        if (!h->sfc2d_host.empty())
            if ((rc = h->d_sfc2d.upload(h->sfc2d_host.data(), h->sfc2d_host.size()))) return rc;
        h->dirty_sfc = false;
    }
    if (h->dirty_tally) {
        if (!h->rad_ext) {
            if ((rc = h->d_rad_own.alloc(h->rad_elems()))) return rc;
            HIPCHK(hipMemsetAsync(h->d_rad_own.p, 0, h->rad_elems() * sizeof(tally_t), h->stream));
        }
        if (!h->flux_ext) {
            if ((rc = h->d_flux_own.alloc(h->flux_elems()))) return rc;
            HIPCHK(hipMemsetAsync(h->d_flux_own.p, 0, h->flux_elems() * sizeof(tally_t), h->stream));
        }
        if ((h->target & MI3D_TARGET_HEAT) && !h->heat_ext) {
            if ((rc = h->d_heat_own.alloc(h->heat_elems()))) return rc;
            HIPCHK(hipMemsetAsync(h->d_heat_own.p, 0, h->heat_elems() * sizeof(double), h->stream));
        }
        h->dirty_tally = false;
    }
    return MI3D_OK;
}

int mi3d_reset(mi3d_solver *h) {
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = mi3d_prepare(h))) return rc;
    // (a run that failed half way may have left ray kernels on their own stream: the tallies are cleared after them)
    for (int q = 0; q < 2; ++q) if (h->set_used[q] && h->set_rays[q]) HIPCHK(hipStreamWaitEvent(h->stream, h->set_rays[q], 0));
    HIPCHK(tl_join(h));
    HIPCHK(hipMemsetAsync(h->rad_ptr(), 0, h->rad_elems() * sizeof(tally_t), h->stream));
    HIPCHK(hipMemsetAsync(h->flux_ptr(), 0, h->flux_elems() * sizeof(tally_t), h->stream));
    if ((h->target & MI3D_TARGET_HEAT) && h->heat_ptr()) HIPCHK(hipMemsetAsync(h->heat_ptr(), 0, h->heat_elems() * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->d_counters.p, 0, MI3D_NCOUNTER * sizeof(unsigned long long), h->stream));
    // (the accumulation image is folded and zeroed at the end of a successful mi3d_run; a run that failed half way -- an event
    //  list ran full, a launch failed -- leaves its partial tallies there)
    if (h->d_rad_acc.p) HIPCHK(hipMemsetAsync(h->d_rad_acc.p, 0, h->d_rad_acc.cap * sizeof(tally_t), h->stream));
    if ((rc = drain_events(h))) return rc;
    for (int s = 0; s < 4; ++s) if (h->ev_busy[s]) h->ev_void[s] = true;   // (launches still on their way: their tallies are gone)
    h->kernel_ms = 0.0;
    h->launches = 0;
    return MI3D_OK;
}

// ---- launchers shared by mi3d_run and its pipelined form -------------------------------------------------------------------
static hipError_t launch_bins(mi3d_solver *h, hipStream_t st, const BinGeom &G, int ntile, uint64_t seed, uint64_t off, uint64_t nb, uint32_t *order, uint32_t *cursor) {
    hipError_t err = hipMemsetAsync(h->d_hist.p, 0, kMaxTiles * sizeof(uint32_t), st);
    if (err != hipSuccess) return err;
    const unsigned nblk = (unsigned)std::min<uint64_t>((nb + 4095) / 4096, 4096);
    hipLaunchKernelGGL(k_bin_count, dim3(nblk), dim3(256), 0, st, G, seed, off, (uint32_t)nb, h->d_tile.p, h->d_hist.p);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(256), 0, st, ntile, h->d_hist.p, cursor);
    const uint32_t slab = (uint32_t)((nb + nblk - 1) / nblk);
    hipLaunchKernelGGL(k_bin_scatter, dim3(nblk), dim3(256), 0, st, ntile, (uint32_t)nb, slab, h->d_tile.p, cursor, order);
    return hipGetLastError();
}

// The photons of a launch up to their first voxel walk (k_entry, mi3d_kernel_lean.hip); one photon per thread: the stream of records
// leaves at the rate a plain copy reaches.
static hipError_t launch_entry(mi3d_solver *h, hipStream_t st, const DevScene &S, uint64_t nb, uint64_t seed, uint64_t off, const uint32_t *order, float4 *entry) {
    const DevCold &C = h->cold_host;
    EntryArgs A;
    A.lay = C.lay;
    A.Lx = C.Lx; A.Ly = C.Ly; A.dx = S.dx; A.dy = S.dy; A.inv_dx = C.inv_dx; A.inv_dy = C.inv_dy; A.inv_nx = C.inv_nx; A.inv_ny = C.inv_ny;
    A.sdx = C.sdx; A.sdy = C.sdy; A.sdz = C.sdz; A.cos_cone = C.cos_cone;
    A.nx = S.nx; A.ny = S.ny; A.nz = S.nz; A.solver = S.solver; A.target = S.target; A.kdir = S.kdir;
    hipLaunchKernelGGL(k_entry, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, A, nb, seed, off, order, entry);
    return hipGetLastError();
}

static hipError_t launch_lean(mi3d_solver *h, hipStream_t st, const DevScene &S, bool emit, int mix, int nt, unsigned grid, size_t lds, uint64_t nb, uint64_t seed, uint64_t off) {
    // mix: 0 er3t's default scene, 1 a second 3-D constituent, 2 the general mixture (several 1-D constituents, tables); nt: threads per workgroup
    const int v = (h->counting ? 4 : 0) + (h->solver == MI3D_SOLVER_P3D ? 2 : 0) + (emit ? 1 : 0);
#define MI3D_LEAN_LAUNCH(C, P, M)                                                                                                        \
    do {                                                                                                                                 \
        if (mix == 3 && nt == 512) {                                                                                                     \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_transport_lean<C, P, 0, 3, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_transport_lean<C, P, 0, 3, 512>), dim3(grid), dim3(512), lds, st, S, nb, seed, off);                    \
        } else if (mix == 3) {                                                                                                           \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_transport_lean<C, P, 0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_transport_lean<C, P, 0, 3>), dim3(grid), dim3(256), lds, st, S, nb, seed, off);                         \
        } else if (mix == 2 && nt == 512 && (M) == 0) {                                                                                  \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_transport_lean<C, P, 0, 2, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_transport_lean<C, P, 0, 2, 512>), dim3(grid), dim3(512), lds, st, S, nb, seed, off);                    \
        } else if (mix == 2) {                                                                                                           \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_transport_lean<C, P, M, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_transport_lean<C, P, M, 2>), dim3(grid), dim3(256), lds, st, S, nb, seed, off);                         \
        } else if (mix == 1) hipLaunchKernelGGL((k_transport_lean<C, P, M, 1>), dim3(grid), dim3(256), lds, st, S, nb, seed, off);        \
        else hipLaunchKernelGGL((k_transport_lean<C, P, M, 0>), dim3(grid), dim3(256), lds, st, S, nb, seed, off);                        \
    } while (0)
    switch (v) {
        case 0: MI3D_LEAN_LAUNCH(false, false, 0); break;
        case 1: MI3D_LEAN_LAUNCH(false, false, 2); break;
        case 2: MI3D_LEAN_LAUNCH(false, true, 0); break;
        case 3: MI3D_LEAN_LAUNCH(false, true, 2); break;
        case 4: MI3D_LEAN_LAUNCH(true, false, 0); break;
        case 5: MI3D_LEAN_LAUNCH(true, false, 2); break;
        case 6: MI3D_LEAN_LAUNCH(true, true, 0); break;
        default: MI3D_LEAN_LAUNCH(true, true, 2); break;
    }
#undef MI3D_LEAN_LAUNCH
    return hipGetLastError();
}

static hipError_t launch_rays(mi3d_solver *h, hipStream_t st, const DevScene &S, bool heavy, size_t lds, uint64_t seed) {
    if (h->rad_kind == 1) {   // cameras: the build whose rays carry their own direction (3-D solver: mi3d_run has checked)
        const unsigned gridc = (unsigned)h->num_cu * 4u;
        if (lds > 65536) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rays<true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rays<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (h->counting) hipLaunchKernelGGL((k_rays<true, false, false, true>), dim3(gridc), dim3(256), lds, st, S, seed);
        else hipLaunchKernelGGL((k_rays<false, false, false, true>), dim3(gridc), dim3(256), lds, st, S, seed);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)h->num_cu * (unsigned)((h->rays_wg > 0 && !heavy) ? std::min(h->rays_wg, MI3D_RAYS_WAVES(h->counting != 0, heavy)) : MI3D_RAYS_WAVES(h->counting != 0, heavy));
    const bool p3d = h->solver == MI3D_SOLVER_P3D;
    const bool plain = (S.target & kTargetPlainPhase) != 0 && !heavy;     // (the heavy build evaluates surface models only)
    // (more than the default 64 KB of dynamic LDS -- staged tables behind the pools --: say so, as launch_lean and launch_flux do)
#define MI3D_LAUNCH_RAYS(C, P, X)                                                                                                    \
    do {                                                                                                                             \
        if (plain) {                                                                                                                 \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rays<C, P, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_rays<C, P, false, false, true>), dim3(grid), dim3(256), lds, st, S, seed);                          \
        } else {                                                                                                                     \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rays<C, P, X, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_rays<C, P, X, false, false>), dim3(grid), dim3(256), lds, st, S, seed);                             \
        }                                                                                                                            \
    } while (0)
    switch ((h->counting ? 4 : 0) | (p3d ? 2 : 0) | (heavy ? 1 : 0)) {
        case 0: MI3D_LAUNCH_RAYS(false, false, false); break;
        case 1: MI3D_LAUNCH_RAYS(false, false, true); break;
        case 2: MI3D_LAUNCH_RAYS(false, true, false); break;
        case 3: MI3D_LAUNCH_RAYS(false, true, true); break;
        case 4: MI3D_LAUNCH_RAYS(true, false, false); break;
        case 5: MI3D_LAUNCH_RAYS(true, false, true); break;
        case 6: MI3D_LAUNCH_RAYS(true, true, false); break;
        default: MI3D_LAUNCH_RAYS(true, true, true); break;
    }
#undef MI3D_LAUNCH_RAYS
    return hipGetLastError();
}

// Photons a launch may have so that its events fit the lists, at `per_photon` events per photon (0: nothing known yet, a
// pilot).  A launch of a few million photons fills the lists of the XCDs in use evenly (twice the room asked for); a smaller one is
// taken by whichever workgroups start first, all on one XCD in the worst case: then ONE list must hold it.
static uint64_t photons_that_fit(uint64_t ev_cap, double per_photon, int n_xcd) {
    if (!(per_photon > 0.0)) return std::min<uint64_t>(16384, std::max<uint64_t>(ev_cap / 256, 16));   // pilot: room for 256 events per photon on one list
    const double even = 0.5 * n_xcd * (double)ev_cap / per_photon;
    return (uint64_t)(even >= 2.0e6 ? even : std::max(16.0, (double)ev_cap / (1.5 * per_photon)));
}

constexpr int kEvSlots = 4;   // launches whose fill counters may be on their way to the host at once

// The fill counters of a launch that has ended, as copied out in stream order (ev_note): how full the lists got sizes the
// launches still to come; a list that ran full has dropped events.  wait: also for launches still running.
static int ev_collect(mi3d_solver *h, uint64_t ev_cap, bool wait) {
    for (int s = 0; s < kEvSlots; ++s) {
        if (!h->ev_busy[s]) continue;
        if (wait) HIPCHK(hipEventSynchronize(h->ev_done[s]));
        else {
            const hipError_t q = hipEventQuery(h->ev_done[s]);
            if (q == hipErrorNotReady) continue;
            HIPCHK(q);
        }
        h->ev_busy[s] = false;
        const unsigned long long *c = h->h_evctr + (size_t)s * 9 * kCtrStride;
        unsigned long long mx = 0, sum = 0;
        for (int x = 0; x < 8; ++x) { mx = std::max(mx, c[x * kCtrStride]); sum += c[x * kCtrStride]; }
        ev_cap = h->ev_capn[s];
        if ((c[8 * kCtrStride] != 0ull || mx > ev_cap) && !h->ev_void[s]) {
            (void)sync_streams(h);   // what is still queued of this run ends; its counters are of no interest any more
            for (bool &b : h->ev_busy) b = false;
            return fail(MI3D_ESTATE, "an event list of the marched views ran full (%llu events on one XCD from %llu photons, room for %llu): "
                                     "the tallies of this run are incomplete; call mi3d_reset and run it again (MI3D_KERNEL=generic needs no lists)",
                        mx, (unsigned long long)h->ev_nb[s], (unsigned long long)ev_cap);
        }
        // (records reserved, unused ones included: what the lists must hold)
        if (h->ev_epochn[s] == h->ev_epoch) h->ev_per_photon = std::max(0.5 * h->ev_per_photon, (double)sum / (double)h->ev_nb[s]);
    }
    return MI3D_OK;
}

// The launches of a run that has returned may still be on their way (mi3d_run does not wait for its last ones): whoever is about to
// look at the tallies -- mi3d_sync, the read-outs, the end of a statistics run -- settles them first.  A list that ran full fails
// THAT call: never silently short.
static int ev_settle(mi3d_solver *h) {
    for (bool b : h->ev_busy) if (b) return ev_collect(h, 0, true);
    return MI3D_OK;
}

// after the kernels of a launch: its fill counters go to a pinned slot (the next launch zeroes them on the device)
static int ev_note(mi3d_solver *h, uint64_t ev_cap, uint64_t nb, hipStream_t st = nullptr, const unsigned long long *ctr = nullptr) {
    if (!ctr) { st = h->stream; ctr = h->d_evctr.p; }
    int s = -1;
    for (int i = 0; i < kEvSlots; ++i) if (!h->ev_busy[i]) { s = i; break; }
    if (s < 0) {   // every slot is on its way: wait for them
        int rc = ev_collect(h, ev_cap, true);
        if (rc) return rc;
        s = 0;
    }
    if (!h->ev_done[s]) HIPCHK(hipEventCreateWithFlags(&h->ev_done[s], hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(h->h_evctr + (size_t)s * 9 * kCtrStride, ctr, 9 * kCtrStride * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(h->ev_done[s], st));
    h->ev_busy[s] = true;
    h->ev_nb[s] = nb;
    h->ev_capn[s] = ev_cap;
    h->ev_void[s] = false;
    h->ev_epochn[s] = h->ev_epoch;
    return MI3D_OK;
}

static hipError_t launch_flux(mi3d_solver *h, int tset, hipStream_t st, hipStream_t sort_st, hipEvent_t filled, hipEvent_t scattered, bool wait_scattered, const DevScene &S, const TallyList &TL0, int mix, unsigned grid, size_t lds, uint64_t nb, uint64_t seed, uint64_t off) {
    TallyList TL = TL0;
    TL.nwave = (int)grid * 4;   // (256-thread workgroups)
    TL.stats = (TL.cap && h->d_tl_stats.p) ? h->d_tl_stats.p + 4 * (h->launches % 64) : nullptr;
    // the photon loop reads the description from memory (one slot per launch in flight: the copy is asynchronous)
    static_assert(sizeof(TallyList) % 8 == 0, "TallyList slots");
    const int slot = (int)(h->launches % 64);
    // (the pinned copy of a slot is read by an ASYNCHRONOUS copy: a host that is 64 launches ahead of the device -- runs of dozens of small launches
    //  queued back to back -- must not overwrite it before that copy and the loop that reads the slot are through.  Until round 6 the host happened
    //  to wait every fourth launch for its record counters; without that wait launches read the description of launch + 64: another set of lists
    //  where one-stream and two-stream runs follow each other, test_record_sort_beside_the_next_photon_loop_changes_no_result[b2b_auto])
    if (h->tldesc_used[slot]) { const hipError_t ew = hipEventSynchronize(h->tldesc_ev[slot]); if (ew != hipSuccess) return ew; }
    std::memcpy(h->h_tldesc + (size_t)slot * sizeof(TallyList), &TL, sizeof(TallyList));
    hipError_t e0 = hipMemcpyAsync(h->d_tldesc.p + (size_t)slot * sizeof(TallyList), h->h_tldesc + (size_t)slot * sizeof(TallyList), sizeof(TallyList), hipMemcpyHostToDevice, st);
    if (e0 != hipSuccess) return e0;
    const TallyList *TLd = reinterpret_cast<const TallyList *>(h->d_tldesc.p + (size_t)slot * sizeof(TallyList));
    // (two sets of lists: the set is free once the sort of the launch that used it last is through it -- that launch's sums read the sorted
    //  copy and the bins' starts, which this photon loop does not touch: they may still run beside it)
    if (wait_scattered && (e0 = hipStreamWaitEvent(st, scattered, 0)) != hipSuccess) return e0;
    // (the sums of that launch, if they are still to come: here, on the main stream, between two photon loops -- they read the set's sorted copy
    //  and bins' starts, which this launch's sort writes again)
    if (h->tl_pend[tset].on && (e0 = launch_sum(h, tset, st)) != hipSuccess) return e0;
    // (the set's cursors: behind the wait -- the sort of the launch that used the set last writes one of them and copies them out on its stream)
    if (TL.cap && (e0 = hipMemsetAsync(TL.cursor, 0, 3 * sizeof(unsigned long long), st)) != hipSuccess) return e0;
#define MI3D_FLUX_LAUNCH(C, P)                                                                                              \
    do {                                                                                                                    \
        if (mix == 2) {                                                                                                     \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_transport_flux<C, P, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_transport_flux<C, P, 2>), dim3(grid), dim3(256), lds, st, S, TLd, nb, seed, off);          \
        } else if (mix == 1) hipLaunchKernelGGL((k_transport_flux<C, P, 1>), dim3(grid), dim3(256), lds, st, S, TLd, nb, seed, off);  \
        else hipLaunchKernelGGL((k_transport_flux<C, P, 0>), dim3(grid), dim3(256), lds, st, S, TLd, nb, seed, off);         \
    } while (0)
    switch ((h->counting ? 2 : 0) | (h->solver == MI3D_SOLVER_P3D ? 1 : 0)) {
        case 0: MI3D_FLUX_LAUNCH(false, false); break;
        case 1: MI3D_FLUX_LAUNCH(false, true); break;
        case 2: MI3D_FLUX_LAUNCH(true, false); break;
        default: MI3D_FLUX_LAUNCH(true, true); break;
    }
#undef MI3D_FLUX_LAUNCH
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
        if (!h->tldesc_ev[slot]) err = hipEventCreateWithFlags(&h->tldesc_ev[slot], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventRecord(h->tldesc_ev[slot], st);
        h->tldesc_used[slot] = (err == hipSuccess);
    }
    if (err != hipSuccess || TL.cap == 0) return err;
    const bool two_streams = sort_st != st;
    if (two_streams) {   // (two sets of lists: the sort on its own stream, behind this photon loop)
        // The sums of the launch BEFORE this one (the other set), whose sort has run beside this photon loop: on the main stream straight behind
        // the loop, alone on the chip for the millisecond or two they take -- this launch's sort waits for them too (`filled` below), so it does
        // not take their CUs, and starts together with the next launch's photon loop, beside which it runs.  (Queued behind their own sort they
        // crept along in the loop's tails -- 13 ms for 1.6 ms of work -- and held up every sort behind them.)
        const int other = tset ^ 1;
        if (h->tl_pend[other].on) {
            if (h->tl_set_used[other] && h->tl_scattered[other]) err = hipStreamWaitEvent(st, h->tl_scattered[other], 0);
            if (err == hipSuccess) err = launch_sum(h, other, st);
            if (err != hipSuccess) return err;
        }
        err = hipEventRecord(filled, st);
        if (err == hipSuccess) err = hipStreamWaitEvent(sort_st, filled, 0);
        if (err != hipSuccess) return err;
        st = sort_st;
    }
    // the records of this launch: where every wave's share of every bin goes, counting sort, one LDS sum per bin
    double *heat = (h->target & MI3D_TARGET_HEAT) ? h->heat_ptr() : nullptr;
    // (round 6) the run records first: how many records of every bin each row's runs expand to -- the runs' rows of the histogram
    RunGeom Gm;
    Gm.lay = h->d_lay.p; Gm.ztoa = h->cold_host.ztoa; Gm.inv_dx = h->cold_host.inv_dx; Gm.inv_dy = h->cold_host.inv_dy;
    Gm.inv_nx = h->cold_host.inv_nx; Gm.inv_ny = h->cold_host.inv_ny; Gm.nz = h->nz; Gm.nx = h->nx; Gm.ny = h->ny; Gm.diag = h->d_counters.p;
    const size_t lds_runs = tl_runs_lds(TL.nbins, h->nz);
    // (large tables of bins: workgroups of 512 threads -- twice the waves on the same LDS)
    const bool runs_wide = lds_runs > 40960;
#define MI3D_RUNS_LAUNCH(W)                                                                                                          \
    do {                                                                                                                             \
        if (TL.hist_wg && runs_wide) {                                                                                               \
            if (lds_runs > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tl_runs<W, 4, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_runs); \
            hipLaunchKernelGGL((k_tl_runs<W, 4, 512>), dim3((unsigned)TL.nwave / 4u), dim3(512), lds_runs, st, TL, Gm, S.flux);        \
        } else if (TL.hist_wg) {                                                                                                     \
            if (lds_runs > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tl_runs<W, 4, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_runs); \
            hipLaunchKernelGGL((k_tl_runs<W, 4, 256>), dim3((unsigned)TL.nwave / 4u), dim3(256), lds_runs, st, TL, Gm, S.flux);        \
        } else {                                                                                                                     \
            if (lds_runs > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tl_runs<W, 1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_runs); \
            hipLaunchKernelGGL((k_tl_runs<W, 1, 256>), dim3((unsigned)TL.nwave), dim3(256), lds_runs, st, TL, Gm, S.flux);             \
        }                                                                                                                            \
    } while (0)
    if (TL.run_cap) MI3D_RUNS_LAUNCH(false);
    hipLaunchKernelGGL(k_tl_wavescan<256>, dim3((unsigned)TL.nbins), dim3(256), 0, st, TL);
    hipLaunchKernelGGL(k_tl_prefix<256>, dim3(1), dim3(256), 0, st, TL);
    // (workgroups of 256 threads with 16-KB tiles -- eight records per thread --, up to eight to a CU: 8.9e8 photons/s on the 128 x 128 flux
    //  scene; 16 / 32 records per thread 8.7e8 / 6.7e8, 4: 8.5e8; 128 / 512 / 1024 threads 8.7e8 / 7.5e8 / 7.9e8: profiles/r03/flux_records_experiments.log)
    const size_t lds_sc = ((size_t)4 * TL.nbins + 16 + 2 * kTlIds) * sizeof(uint32_t) + (size_t)MI3D_TLS_R * MI3D_TLS_NT * sizeof(uint2);
    if (TL.hist_wg) {
        // tallies of more than 1024 bins: one workgroup of the sort per workgroup of the photon loop (four waves' chunk lists, one
        // histogram), 1024 threads and 64-KB tiles -- the tables over the bins (60 KB at 5000 bins) leave room for one workgroup per CU,
        // and every tile pays for a pass over them: 5.4e8 photons/s on 480 x 480 x 100 against 4.0e8 with 256 threads (atomics: 3.35e8)
        const size_t lds_b = ((size_t)3 * TL.nbins + 16 + 2 * kTlIds) * sizeof(uint32_t) + (size_t)8 * 1024 * sizeof(uint2);
        if (lds_b > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tl_scatter<1024, 8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL((k_tl_scatter<1024, 8, 4>), dim3((unsigned)TL.nwave / 4u), dim3(1024), lds_b, st, TL, S.flux, (unsigned)h->flux_elems(), heat, (unsigned)(heat ? h->heat_elems() : 0));
    } else {
        if (lds_sc > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tl_scatter<MI3D_TLS_NT, MI3D_TLS_R, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        hipLaunchKernelGGL((k_tl_scatter<MI3D_TLS_NT, MI3D_TLS_R, 1>), dim3((unsigned)TL.nwave), dim3(MI3D_TLS_NT), lds_sc, st, TL, S.flux, (unsigned)h->flux_elems(), heat, (unsigned)(heat ? h->heat_elems() : 0));
    }
    // ... and the runs' records, level by level, behind each row's share of every bin
    if (TL.run_cap) MI3D_RUNS_LAUNCH(true);
#undef MI3D_RUNS_LAUNCH
    if (two_streams && (err = hipEventRecord(scattered, st)) != hipSuccess) return err;
    const int split = std::max(1, std::min(64, (h->num_cu * 8) / TL.nbins));
    mi3d_solver::PendingSum &P = h->tl_pend[tset];
    P.on = true; P.TL = TL; P.flux = S.flux; P.nflux = (unsigned)h->flux_elems(); P.heat = heat; P.nheat = (unsigned)(heat ? h->heat_elems() : 0); P.split = split;
    // one stream: the sums straight behind the sort.  Two: they wait for the main stream's next pause -- the next launch of this set, or a join
    static const bool defer = !(getenv("MI3D_SUM_DEFER") && atoi(getenv("MI3D_SUM_DEFER")) == 0);   // (measurements: 0 -- the sums behind their own sort, as in round 5)
    if (!two_streams || !defer) return launch_sum(h, tset, st);
    return hipGetLastError();
}

// Records reserved by the launches that have ended, as copied out in stream order (tl_note): they size the launches to come.
// A list that ran full has lost nothing (the tallies went out as atomics from there on), only speed.
static int tl_collect(mi3d_solver *h, bool wait) {
    for (int s = 0; s < kEvSlots; ++s) {
        if (!h->tl_busy[s]) continue;
        if (wait) HIPCHK(hipEventSynchronize(h->tl_done[s]));
        else {
            const hipError_t q = hipEventQuery(h->tl_done[s]);
            if (q == hipErrorNotReady) continue;
            HIPCHK(q);
        }
        h->tl_busy[s] = false;
        // ([0] records the loop reserved, [1] records in all with the runs expanded (k_tl_prefix), [2] run records reserved)
        auto upd = [&](double &pp, unsigned long long got, uint64_t cap) {
            const double seen = (double)got / (double)h->tl_nb[s];
            if (got > cap) pp = std::max(1.5 * pp, seen);   // ran full: what it needed is not known
            else pp = pp > 0.0 ? std::max(0.7 * pp, seen) : seen;
        };
        upd(h->tl_per_photon, h->h_tlctr[4 * s], h->tl_cap[s]);
        upd(h->tl_total_pp, h->h_tlctr[4 * s + 1], h->tl_bcap[s]);
        if (h->tl_rcap[s]) upd(h->tl_runs_pp, h->h_tlctr[4 * s + 2], h->tl_rcap[s]);
    }
    return MI3D_OK;
}

// (st: the stream behind whose work the counters are complete -- the sort stream where the sort runs on one of its own)
static int tl_note(mi3d_solver *h, uint64_t cap, uint64_t bcap, uint64_t rcap, uint64_t nb, const unsigned long long *cursor, hipStream_t st) {
    int s = -1;
    for (int i = 0; i < kEvSlots; ++i) if (!h->tl_busy[i]) { s = i; break; }
    if (s < 0) {
        // (every slot on its way: this launch's counters are not looked at -- the host does not wait here: the copies follow the SORT of their
        //  launch, and a host that waited for them could queue the next photon loop only when the sort before it was through, beside nothing)
        int rc = tl_collect(h, false);
        if (rc) return rc;
        for (int i = 0; i < kEvSlots; ++i) if (!h->tl_busy[i]) { s = i; break; }
        if (s < 0) return MI3D_OK;
    }
    if (!h->tl_done[s]) HIPCHK(hipEventCreateWithFlags(&h->tl_done[s], hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(h->h_tlctr + 4 * s, cursor, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(h->tl_done[s], st));
    h->tl_busy[s] = true; h->tl_nb[s] = nb; h->tl_cap[s] = cap; h->tl_bcap[s] = bcap; h->tl_rcap[s] = rcap;
    return MI3D_OK;
}

// Tile edge (in columns) of the photon order.  Two things pull in opposite directions (profiles/r02/tile_sweep_les480.log):
// the voxel records of a tile plus a margin of ten columns on every side (a photon wanders about a kilometre from where
// it enters the cloud) must fit an XCD's 4 MiB L2 with room to spare, counting the layers that are walked voxel by voxel;
// but the smaller the tile, the fewer cache lines of the radiance image take the tally atomics of the 41 000 photons an
// XCD has in flight, and float atomics to one line are served one after the other at the memory side (16-column tiles ran
// 17 % slower than 64-column ones at the same 84 % L2 hit rate).  So: as large as 3 MB of records allow, at most 64.
// Does the job keep a tally window (DevCold::tile_end ...)?  One predicate for the tile edge chosen here and for mi3d_run, which gives the
// kernel the LDS for it: the lean loop's column view of a satellite image with one pixel per column -- the build without marched views
// (k_transport_lean<.,.,0,.>), or the event-writing one when it is built with MI3D_LEAN_WIN_EMIT.  (mi3d_run adds what only it knows: the
// launch is sorted by tile, the lean loop serves the job.)
static bool window_wanted(const mi3d_solver *h) {
    return h->tally_window && !(h->target & MI3D_TARGET_FLUX) && h->rad_kind == 2 && h->kernel_choice == 0 && h->nview > 0 && h->nxr == h->nx && h->nyr == h->ny &&
           (h->nmarch == 0 || (MI3D_LEAN_WIN_EMIT && h->nmarch < h->nview));
}

static int choose_tile_cols(const mi3d_solver *h) {
    if (h->tile_cols >= 0) return h->tile_cols;
    if (h->nz3 <= 0 || h->n_step3d <= 0 || (long)h->nx * h->ny < 4096) return 0;   // nothing to gain: one tile
    // a tile's voxel records (16 bytes x the layers walked voxel by voxel) in three quarters of an XCD's 4 MiB L2, less a margin
    // of 20 columns.  Swept again in round 4 (profiles/r04/ab_tile_cols*.log): on the 480 x 480 scene 16 / 32 / 48 columns lose
    // 20 / 8 / 3 %, 240 lose 4 %, and between 60 and 120 the rates lie within what two processes with the same tile differ by (1.7 %).
    const double cols = std::sqrt(3.0e6 / (16.0 * h->n_step3d)) - 20.0;
    // (where the lean loop keeps its tally window -- a column view, one pixel per column -- a tile is narrower than the window by a
    //  margin: 40-56 columns 2.755e9 photons/s on the 480 x 480 scene against 2.725e9 at 64; ab_window_tile_cols.log)
    return (int)std::min(window_wanted(h) ? 48.0 : 64.0, std::max(24.0, cols));
}

int mi3d_run(mi3d_solver *h, uint64_t nphoton, uint64_t seed, uint64_t photon_offset) {
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = mi3d_prepare(h))) return rc;
    if ((h->target & MI3D_TARGET_RADIANCE) && h->nview == 0)
        return fail(MI3D_ESTATE, "radiance requested but no view is set (mi3d_set_views)");
    if ((h->target & MI3D_TARGET_RADIANCE) && h->rad_kind == 1 && h->solver != MI3D_SOLVER_3D)
        return fail(MI3D_EUNSUP, "cameras (Rad_mrkind=1) need the 3-D solver");
    if (nphoton == 0) return MI3D_OK;
    {   // the kernels index every table with 32-bit arithmetic
        const double lim = 2147483647.0;
        const double nvox = (double)h->nx * h->ny * (h->nz3 + 1) * (h->np3d > 0 ? h->np3d : 1);
        if (nvox > lim || (double)h->flux_elems() > lim || (double)h->rad_elems() > lim || (double)h->heat_elems() > lim)
            return fail(MI3D_EUNSUP, "grid too large for the 32-bit table indices of the transport kernel");
    }
    // ---- photon order: tiles of the domain (none: the launch runs in id order)
    BinGeom G;
    std::memset(&G, 0, sizeof(G));
    int ntile = 1;
    {
        int tc = choose_tile_cols(h);
        if (tc > 0) {
            while ((long)((h->nx + tc - 1) / tc) * ((h->ny + tc - 1) / tc) > kMaxTiles) tc *= 2;
            G.Lx = (float)(h->dx * h->nx); G.Ly = (float)(h->dy * h->ny);
            G.inv_tx = (float)(1.0 / (h->dx * tc)); G.inv_ty = (float)(1.0 / (h->dy * tc));
            G.nx = h->nx; G.ny = h->ny; G.tcols = tc; G.ntx = (h->nx + tc - 1) / tc; G.nty = (h->ny + tc - 1) / tc;
            ntile = G.ntx * G.nty;
        }
    }
    const bool sorted = ntile > 1 && nphoton >= 4096;
    if (sorted) {
        const size_t cap = (size_t)std::min<uint64_t>(nphoton, h->batch);
        if ((rc = h->d_order.alloc(cap)) || (rc = h->d_tile.alloc(cap))) return rc;
    }
    DevScene S;
    if ((rc = fill_scene(h, S))) return rc;
    h->cold_host.order = sorted ? h->d_order.p : nullptr;

    const int tb = 256;
    const size_t lds = (size_t)h->nz * sizeof(LayerRec) + MI3D_MAX_VIEW * sizeof(ViewRec) + sizeof(DevCold) + (size_t)9 * tb * sizeof(float) +
                       (size_t)(h->tab_n > 0 ? (1 + 2 * h->tab_n) * h->nang * sizeof(float) : 0);
    const bool march = (h->target & MI3D_TARGET_RADIANCE) && h->nmarch > 0;
    const bool flux = (h->target & MI3D_TARGET_FLUX) != 0;
    const int variant = (h->counting ? 4 : 0) | (march ? 2 : 0) | (flux ? 1 : 0);
    // as many workgroups as are resident at once (a workgroup that starts after the pool is empty only stages LDS and leaves)
#ifndef MI3D_BLOCKS_PER_CU
#define MI3D_BLOCKS_PER_CU(MARCH, COUNT) MI3D_WAVES(MARCH, COUNT)
#endif
    // the lean kernel (mi3d_kernel_lean.hip): radiance only, satellite views (column table or marched), at most two 3-D constituents,
    // byte offsets of the voxel records within 32 bits (cameras, Rad_mrkind = 1, 3-D solver: through the event lists and the ray
    // kernel's camera build).  Its builds (`mix`): 0 er3t's default scene -- one 1-D and one 3-D constituent, analytic phase functions --,
    // 1 a second 3-D constituent, 2 (round 5) the general mixture: several 1-D constituents and / or TABULATED phase functions, in 1-D
    // layers (func_ref_vs_cot's cloud slab, rtm/mca/util.py:153) or in voxels (the Mie branch as mca_atm.py:275-277 would write it)
    const bool cam_ok = h->rad_kind == 1 && h->solver == MI3D_SOLVER_3D && h->kernel_choice == 0;
    bool gen = h->np1d > 1 || h->tab3d_hi >= 0, tabs = h->tab3d_hi >= 0;
    for (float a : h->apf1d) if (a >= 1.0f) { gen = true; tabs = true; }
    static const bool force_gen = getenv("MI3D_FORCE_GEN") && atoi(getenv("MI3D_FORCE_GEN")) != 0;   // (measurements: the general-mixture builds on any scene)
    if (force_gen) gen = true;
    const int mix = gen ? 2 : (h->np3d > 1 ? 1 : 0);
    // (tables that do not fit the LDS budget of fill_scene, or none loaded where a selector asks for one: the general kernel, which reads them from global memory)
    const bool tabs_ok = !tabs || h->tab_n > 0;
    // (tables the scene refers to, staged in LDS by the lean kernels when they fit the budget of fill_scene: mu, p, cdf and the bucket indices)
    const size_t lds_tab = (gen && h->tab_n > 0) ? lean_tab_floats(h->nang, h->tab_n) * sizeof(float) : 0;
    bool use_col = !flux && h->nview > 0 && (h->rad_kind == 2 || cam_ok) && h->np3d <= 2 && tabs_ok &&
                   (double)h->ny * h->vrow_f4 * 16.0 < 4.0e9 && h->kernel_choice != 1;
    const size_t lds_col = (size_t)(h->nz + 2) * sizeof(LayerRec) + MI3D_MAX_VIEW * sizeof(ViewRec) + sizeof(DevCold);   // (+2: the lean loop's end records)
    // the lean flux kernel (mi3d_kernel_flux.hip): flux / heating rates without radiance, the same scenes as the lean radiance kernel
    bool use_fl = flux && !((h->target & MI3D_TARGET_RADIANCE) && h->nview > 0) && h->np3d <= 2 && tabs_ok &&
                  (double)h->ny * h->vrow_f4 * 16.0 < 4.0e9 && h->kernel_choice != 1 && h->nx < 65536 && h->ny < 65536 && h->nz < 65535;
    TallyList TL, TL2;
    std::memset(&TL, 0, sizeof(TL));
    bool tl_two = h->overlap_sort != 0;
    size_t lds_fl = (size_t)(h->nz + 2) * sizeof(LayerRec) + sizeof(DevCold);
    // Tally records instead of atomics: bins of 2^shift tally cells, as many as one workgroup can sum in LDS in float64; the
    // record lists take what the launch needs at the records per photon seen so far, at most 2^31 records and a quarter of the
    // memory that is free (two lists of 8 bytes per record).  Not for short runs (the sort has a fixed cost) or tallies of more
    // than 8192 bins (134 million tally cells).
    auto size_tally_lists = [&](uint64_t nb_max) -> uint64_t {
        if (!use_fl || !h->tally_lists || nphoton < 4096) return 0;
        int shift = 10;
        while (shift < 14 && ((size_t)16 << shift) <= (size_t)h->lds_max) ++shift;
        const size_t ncell = h->flux_elems() + ((h->target & MI3D_TARGET_HEAT) ? h->heat_elems() : 0);   // (heating-rate cells follow the flux cells)
        if ((double)ncell > 4.0e9) return 0;
        const int nbins = (int)((ncell + ((size_t)1 << shift) - 1) >> shift);
        if (nbins > 8192 || ((size_t)3 * nbins + 16 + 2 * kTlIds) * sizeof(uint32_t) + (size_t)8 * 1024 * sizeof(uint2) > (size_t)h->lds_max) return 0;
        const int hist_wg = nbins > 1024 ? 1 : 0;   // (four histograms per workgroup of the photon loop would not fit its LDS any more)
        const bool runs = h->tally_runs != 0 && h->nz <= 1022;   // (a run record holds its first level and its count in ten bits each)
        const int nset = tl_two ? 2 : 1;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
        free_b += (h->d_tl_rec.cap + h->d_tl_binned.cap + h->d_tl_rec2.cap + h->d_tl_binned2.cap) * sizeof(uint2) + (h->d_tl_runs.cap + h->d_tl_runs2.cap) * sizeof(float4);
        // per photon, with a margin: records in all (what the sorted copy holds), records the loop writes one by one, run records.  Nothing
        // known yet: a record per level and a half, every one of them written by the loop, a run for every kRunMin of them
        const double tpp = h->tl_total_pp > 0.0 ? 1.15 * h->tl_total_pp : (h->tl_per_photon > 0.0 && !runs ? 1.15 * h->tl_per_photon : 1.5 * (h->nz + 1));
        const double dpp = h->tl_per_photon > 0.0 ? 1.15 * h->tl_per_photon : tpp;
        const double rpp = !runs ? 0.0 : h->tl_runs_pp > 0.0 ? 1.15 * h->tl_runs_pp + 0.05 : tpp / kRunMin;
        const uint64_t waves = (uint64_t)h->num_cu * MI3D_FLUX_WAVES(h->counting != 0) * 4;
        const uint64_t lim = ((uint64_t)1 << h->tl_cap_log2) - kTlChunk;
        // the lists in all: a quarter of the memory that is free (the sorted copy once, records and runs once per set)
        double nb = (double)nb_max;
        const double bytes_pp = nset * (8.0 * tpp + 8.0 * dpp + 32.0 * rpp);
        nb = std::min(nb, 0.25 * (double)free_b / bytes_pp);
        auto cap_of = [&](double pp, uint64_t chunk, uint64_t have, double item_bytes) -> uint64_t {
            uint64_t c = (uint64_t)(pp * nb) + (waves + 1) * chunk;
            // (lists only grow, and by a quarter at least: the rates per photon creep upwards by a per cent from launch to launch, and every
            //  step of a list that is not at its limit would free and allocate gigabytes -- seen as one run in three taking seconds.
            //  A LARGE list that is near enough stays where it is: a launch more per run costs less than moving sixteen gigabytes.  A small one
            //  moves: er3t's jobs of a few million photons come by the dozen, one size, and a list 30 % short made two launches of every one
            //  of them -- 7.3 instead of 5.6 ms per job of 6e6 photons, profiles/r06/small_flux_jobs_one_launch.log)
            if (have > 0 && (double)have >= (double)c) c = have;
            else if (have > 0 && (double)have >= 0.7 * (double)c && (double)have * item_bytes >= 1.6e10) c = have;
            else if (c > have && have > 0) c = std::max<uint64_t>(c, have + have / 4);
            c = std::min<uint64_t>(c, lim);
            c = std::max<uint64_t>(c, std::min<uint64_t>(have, lim));
            return c / chunk * chunk;
        };
        uint64_t want_cap = cap_of(dpp, kTlChunk, h->d_tl_rec.cap, 8.0 * nset);
        const uint64_t want_bcap = cap_of(tpp, kTlChunk, h->d_tl_binned.cap, 8.0 * nset);
        const uint64_t want_rcap = runs ? std::max<uint64_t>(cap_of(rpp, kRunChunk, h->d_tl_runs.cap / 2, 32.0 * nset), 64 * kRunChunk) : 0;
        if (want_cap < 64 * kTlChunk) return 0;
        const size_t nwave_max = (size_t)waves;
        const size_t wcap = std::max<size_t>(64, 4 * (size_t)(want_cap / kTlChunk) / nwave_max);
        const size_t rwcap = runs ? std::max<size_t>(64, 4 * (size_t)(want_rcap / kRunChunk) / nwave_max) : 0;
        const size_t nrow_max = (hist_wg ? nwave_max / 4 : nwave_max) * (runs ? 2 : 1);
        const size_t nwords = (size_t)(want_cap / kTlChunk) + nwave_max * wcap + nwave_max + 2 * nrow_max * nbins + 2 * (size_t)nbins + 1 +
                              (runs ? (size_t)(want_rcap / kRunChunk) + nwave_max * rwcap + nwave_max : 0);
        static const bool tl_verbose = getenv("MI3D_TL_VERBOSE") != nullptr;
        if (tl_verbose && (want_cap > h->d_tl_rec.cap || want_bcap > h->d_tl_binned.cap || 2 * want_rcap > h->d_tl_runs.cap || (tl_two && (want_cap > h->d_tl_rec2.cap || 2 * want_rcap > h->d_tl_runs2.cap))))
            fprintf(stderr, "[mi3d tally lists] records %.3g (had %.3g), sorted copy %.3g (%.3g), runs %.3g (%.3g), sets %d; per photon: %.2f / %.2f / %.3f known %d\n", (double)want_cap, (double)h->d_tl_rec.cap,
                    (double)want_bcap, (double)h->d_tl_binned.cap, (double)want_rcap, (double)h->d_tl_runs.cap / 2, nset, dpp, tpp, rpp, h->tl_total_pp > 0.0 ? 1 : 0);
        // (a list that moves: nothing may still be on its way that reads the old one -- launches, sorts, and the sums that are kept back for the main
        //  stream's next pause, whose description holds the old pointers: everything joined and waited for first)
        if (want_cap > h->d_tl_rec.cap || want_bcap > h->d_tl_binned.cap || nwords > h->d_tl_words.cap || 2 * want_rcap > h->d_tl_runs.cap ||
            (tl_two && (want_cap > h->d_tl_rec2.cap || want_bcap > h->d_tl_binned2.cap || nwords > h->d_tl_words2.cap || 2 * want_rcap > h->d_tl_runs2.cap)))
            if (sync_streams(h) != hipSuccess) { (void)hipGetLastError(); return 0; }
        int r = h->d_tl_rec.alloc(want_cap);
        if (!r) r = h->d_tl_binned.alloc(want_bcap);
        if (!r) r = h->d_tl_words.alloc(nwords);
        if (!r) r = h->d_tl_cursor.alloc(kCtrStride);
        if (!r) r = h->d_tl_stats.alloc(64 * 4);
        if (!r && runs) r = h->d_tl_runs.alloc(2 * want_rcap);
        if (!r && tl_two) r = h->d_tl_rec2.alloc(want_cap);
        if (!r && tl_two) r = h->d_tl_binned2.alloc(want_bcap);
        if (!r && tl_two) r = h->d_tl_words2.alloc(nwords);
        if (!r && tl_two) r = h->d_tl_cursor2.alloc(kCtrStride);
        if (!r && tl_two && runs) r = h->d_tl_runs2.alloc(2 * want_rcap);
        if (!r && !h->h_tlctr && hipHostMalloc((void **)&h->h_tlctr, (size_t)4 * kEvSlots * sizeof(unsigned long long)) != hipSuccess) r = MI3D_EDEVICE;
        if (r) {
            (void)hipGetLastError(); h->d_tl_rec.release(); h->d_tl_binned.release(); h->d_tl_rec2.release(); h->d_tl_runs.release(); h->d_tl_runs2.release(); h->d_tl_binned2.release();
            fprintf(stderr, "Warning [mi3d_run]: no device memory for the tally-record lists of this flux job (%.1f GB free); an atomic per level crossing instead (same results, slower).\n", (double)free_b / 1.0e9);
            return 0;
        }
        auto lay_out = [&](TallyList &T, uint32_t *words, uint2 *rec, uint2 *binned, float4 *runs_p, unsigned long long *cursor) {
            T.rec = rec; T.binned = binned;
            T.chunk_fill = words;
            T.wave_chunks = T.chunk_fill + want_cap / kTlChunk; T.wave_nchunk = T.wave_chunks + nwave_max * wcap;
            T.whist = T.wave_nchunk + nwave_max; T.wbase = T.whist + nrow_max * nbins;
            T.hist = T.wbase + nrow_max * nbins; T.bin_start = T.hist + nbins;
            T.run_fill = T.bin_start + nbins + 1; T.run_chunks = T.run_fill + (runs ? want_rcap / kRunChunk : 0); T.run_nchunk = T.run_chunks + nwave_max * rwcap;
            T.runs = runs ? runs_p : nullptr; T.run_cap = (unsigned)want_rcap; T.run_wcap = (int)rwcap;
            T.hist_wg = hist_wg;
            T.wcap = (int)wcap; T.nwave = (int)nwave_max;
            T.cursor = cursor;
            T.cap = (unsigned)want_cap; T.bcap = (unsigned)want_bcap; T.shift = shift; T.nbins = nbins;
            // the compact histogram of a workgroup of the photon loop (hist_wg, with run records): the bins of the levels around the layers that
            // are walked voxel by voxel, for the three planes and the heating cells -- merged where they touch; without run records: all bins
            T.run_min = kRunMin;
            for (int r = 0; r < 4; ++r) { T.cb_lo[r] = 1; T.cb_hi[r] = 0; T.cb_off[r] = 0; }
            T.cb_lo[0] = 0; T.cb_hi[0] = nbins - 1; T.ncb = nbins;
            int ks_lo = h->nz, ks_hi = -1;
            for (int kq = 0; kq < h->nz && kq < (int)h->lay_host.size(); ++kq) if (h->lay_host[kq].flags & kLayStep3d) { ks_lo = std::min(ks_lo, kq); ks_hi = std::max(ks_hi, kq); }
            if (hist_wg && runs && ks_hi >= ks_lo) {
                const size_t ncol_ = (size_t)h->nx * h->ny, nlev_ = (size_t)h->nz + 1, nflux_ = 3 * nlev_ * ncol_;
                std::vector<std::pair<int, int>> rg;
                for (size_t pl = 0; pl < 3; ++pl) rg.emplace_back((int)(((pl * nlev_ + ks_lo) * ncol_) >> shift), (int)((((pl * nlev_ + ks_hi + 2) * ncol_) - 1) >> shift));
                if (h->target & MI3D_TARGET_HEAT) rg.emplace_back((int)((nflux_ + (size_t)ks_lo * ncol_) >> shift), (int)((nflux_ + ((size_t)ks_hi + 1) * ncol_ - 1) >> shift));
                std::sort(rg.begin(), rg.end());
                std::vector<std::pair<int, int>> mg;
                for (auto &q : rg) { if (!mg.empty() && q.first <= mg.back().second + 1) mg.back().second = std::max(mg.back().second, q.second); else mg.push_back(q); }
                int off = 0;
                for (size_t r = 0; r < 4; ++r) {
                    if (r < mg.size()) { T.cb_lo[r] = mg[r].first; T.cb_hi[r] = std::min(mg[r].second, nbins - 1); T.cb_off[r] = off; off += T.cb_hi[r] - T.cb_lo[r] + 1; }
                    else { T.cb_lo[r] = 1; T.cb_hi[r] = 0; T.cb_off[r] = 0; }
                }
                T.ncb = std::max(off, 1);
                T.run_min = 1;
            }
        };
        lay_out(TL, h->d_tl_words.p, h->d_tl_rec.p, h->d_tl_binned.p, h->d_tl_runs.p, h->d_tl_cursor.p);
        TL2 = TL;
        // the second set: lists, counters, cursor and sorted copy of its own
        if (tl_two) lay_out(TL2, h->d_tl_words2.p, h->d_tl_rec2.p, h->d_tl_binned2.p, h->d_tl_runs2.p, h->d_tl_cursor2.p);
        return want_cap;
    };
    if (use_fl) {
        use_col = false;
        if ((rc = h->d_tldesc.alloc((size_t)64 * sizeof(TallyList)))) return rc;
        if (!h->h_tldesc && hipHostMalloc((void **)&h->h_tldesc, (size_t)64 * sizeof(TallyList)) != hipSuccess) return fail(MI3D_EDEVICE, "no pinned memory for the tally-list descriptions");
        // Two streams pay where something runs beside the sort: a run long enough for four launches of eight million photons (a run of 4e7 alone: +7 %;
        // 1e7 / 2e7 in two or three launches: -2.5 %, tools/small_runs.py), or a run that follows another whose tallies nobody has looked at in between
        // (runs queued back to back: +19 % at 5e6 photons a run).  A small run whose tallies are read before the next one -- er3t's jobs of a few
        // million photons, one by one -- only pays for the launches' tails and the hops between the streams: one stream.
        if (tl_two && h->overlap_sort < 2 && nphoton < ((uint64_t)1 << 25) && h->runs_unread == 0) tl_two = false;   // (before the lists are sized: no second set for such a run; "overlap_sort" 2: two streams whatever the run)
        // (the stream and its events only where they are used: every stream more of a process shares the device's few hardware queues with the
        //  others -- two handles that each kept an idle sort stream ran their jobs 12 % slower side by side, tools/time_dropin.py)
        if (tl_two && !h->tl_stream && hipStreamCreateWithFlags(&h->tl_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); tl_two = false; }
        for (int q = 0; q < 2 && tl_two; ++q) {
            if (!h->tl_filled[q] && hipEventCreateWithFlags(&h->tl_filled[q], hipEventDisableTiming) != hipSuccess) tl_two = false;
            if (tl_two && !h->tl_sorted[q] && hipEventCreateWithFlags(&h->tl_sorted[q], hipEventDisableTiming) != hipSuccess) tl_two = false;
            if (tl_two && !h->tl_scattered[q] && hipEventCreateWithFlags(&h->tl_scattered[q], hipEventDisableTiming) != hipSuccess) tl_two = false;
        }
        // (nothing known about the scene's records per photon: lists for a pilot launch of two million photons -- sized by the guess "a record and a half
        //  per level, a run for every three" they would take tens of gigabytes -- and the right size after it, below)
        const bool tl_pilot = !(h->tl_total_pp > 0.0) && nphoton > ((uint64_t)1 << 21);
        if (size_tally_lists(tl_pilot ? (uint64_t)1 << 21 : std::min<uint64_t>(nphoton, h->batch))) lds_fl += TL.hist_wg ? ((size_t)TL.ncb + 3) / 4 * 16 : (size_t)TL.nbins * 16;   // (a histogram per wave, or one for the workgroup)
        lds_fl += (size_t)4 * 128 * sizeof(float4) + (size_t)4 * kTlStage * sizeof(uint2);   // the waves' run records and staged tallies
        lds_fl += lds_tab;                                                                   // ... and the phase tables behind them
        if (!TL.cap) tl_two = false;
        // (one stream after runs on two: the shared sorted copy and the first set of lists are free once the sorts before are through)
        if (!tl_two && TL.cap) HIPCHK(tl_join(h));
    }
    // marched views (and cameras) of the lean build: by k_rays from event lists; a scene whose cell numbers do not fit the records' 16 bits goes
    // to the general kernel
    const bool can_split = march && h->nx < 65536 && h->ny < 65536 && h->nz < 65536;
    if ((march || h->rad_kind == 1) && !can_split) use_col = false;
    bool split = use_col && march;
    // cameras in the cyclic domain: the periodic images of a camera are served by the ray kernel's camera build.  What keeps a camera job
    // from it: kernel choice 1 / 2, more than one 1-D or two 3-D constituents, voxel records beyond 4 GB.  Asked for explicitly
    // ("cam_images" > 0) such a job is refused; left to the default (-1) it runs with the nearest image alone and says so.
    auto cam_fallback = [&](const char *why) -> int {
        if (h->cam_images > 0)
            return fail(MI3D_EUNSUP, "cam_images=%d: the periodic images of a camera are served by the ray kernel only, and this job cannot use it (%s)", h->cam_images, why);
        if (h->cam_images < 0 && !h->cam_warned) {
            fprintf(stderr, "Warning [mi3d_run]: this camera job cannot use the ray kernel (%s): the NEAREST periodic image of the camera alone is served -- "
                            "lines of sight that leave the domain sideways miss what its continuation adds (set \"cam_images\" to 0 to say so yourself).\n", why);
            h->cam_warned = true;
        }
        return MI3D_OK;
    };
    if (h->rad_kind == 1 && !split) {
        // (the reason from the predicates that cleared use_col / split above)
        const char *why = h->kernel_choice != 0 ? "kernel choice 1" : h->solver != MI3D_SOLVER_3D ? "cameras are served under the 3-D solver only" : h->np3d > 2 ? "more than two 3-D constituents"
                          : flux ? "flux together with radiance" : !tabs_ok ? "phase tables too large for the LDS" : !can_split ? "cell numbers beyond the event records' 16 bits"
                          : !((double)h->ny * h->vrow_f4 * 16.0 < 4.0e9) ? "voxel records beyond 4 GB" : "a scene the lean photon loop does not serve";
        if ((rc = cam_fallback(why))) return rc;
    }
    uint64_t ev_cap = 0;
    // Capacity of each XCD's event list.  Nothing known about the scene yet: room for a pilot launch.  A short run: room for every
    // event of the run on ONE list (64 per photon: the workgroups that start first take most of its photons).  A long one: what
    // the run's events need at the number per photon seen so far, twice over (the lists fill unevenly), at most 2^ev_cap_log2
    // records -- and never more than a quarter of the memory that is free now (other handles, a host framework and smaller parts
    // share the device).  Lists only grow (mi3d_set_tuning "evcap_log2" and a job without marched views release them).
    // (two sets of lists and a stream of its own for the ray kernels, so that they run beside the next launch's photon loop)
    bool two_sets = h->overlap_rays && split;
    if (two_sets && !h->rays_stream && hipStreamCreateWithFlags(&h->rays_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); two_sets = false; }
    for (int q = 0; q < 2 && two_sets; ++q) {
        if (!h->set_emit[q] && hipEventCreateWithFlags(&h->set_emit[q], hipEventDisableTiming) != hipSuccess) two_sets = false;
        if (two_sets && !h->set_rays[q] && hipEventCreateWithFlags(&h->set_rays[q], hipEventDisableTiming) != hipSuccess) two_sets = false;
    }
    auto size_lists = [&]() -> int {
        const double per_rec = (double)kEvBlockF4 * 16.0 / 64.0 + (h->sfc_lambert_only ? 0.0 : 8.0);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
        free_b += h->d_events.cap * sizeof(float4) + h->d_hvlist.cap * sizeof(unsigned long long);   // (what this handle holds already is reused)
        free_b += h->d_events2.cap * sizeof(float4) + h->d_hvlist2.cap * sizeof(unsigned long long);
        const int nset = two_sets ? 2 : 1;
        const uint64_t cap_mem = (uint64_t)(0.25 * (double)free_b / (8.0 * per_rec * nset));
        const uint64_t cap_max = (uint64_t)1 << h->ev_cap_log2;
        uint64_t want_cap = std::min<uint64_t>(cap_max, 64 * nphoton + 65536);
        if (h->ev_per_photon > 0.0)
            want_cap = std::min<uint64_t>(want_cap, std::max<uint64_t>((uint64_t)(4.0 * h->ev_per_photon * (double)nphoton / h->n_xcd), 65536));
        else want_cap = std::min<uint64_t>(want_cap, (uint64_t)1 << 22);
        want_cap = std::max<uint64_t>(want_cap, std::min<uint64_t>(h->d_events.cap / ((size_t)8 * kEvBlockF4) * 64, cap_max));
        want_cap = std::min<uint64_t>(want_cap, cap_mem);
        want_cap &= ~(uint64_t)63;      // (the records stand in blocks of 64: ev_index)
        // (an allocation that fails although the device reported the room -- fragmentation, another process -- is tried again at half the
        //  size: launches are sized to the lists, so small lists cost launches, not results; below 65 536 records per list: the general kernel)
        int r = MI3D_OK;
        for (;;) {
            r = MI3D_OK;
            if (want_cap < 65536) r = fail(MI3D_EDEVICE, "no memory for event lists");
            if (!r) r = h->d_events.alloc((size_t)8 * ev_list_f4(want_cap));
            if (!r) r = h->d_evctr.alloc(kCtrWords * kCtrStride);
            if (!r && !h->sfc_lambert_only) r = h->d_hvlist.alloc((size_t)8 * want_cap);
            if (!r && two_sets) r = h->d_events2.alloc((size_t)8 * ev_list_f4(want_cap));
            if (!r && two_sets) r = h->d_evctr2.alloc(kCtrWords * kCtrStride);
            if (!r && two_sets && !h->sfc_lambert_only) r = h->d_hvlist2.alloc((size_t)8 * want_cap);
            if (!r && !h->h_evctr && hipHostMalloc((void **)&h->h_evctr, (size_t)kEvSlots * 9 * kCtrStride * sizeof(unsigned long long)) != hipSuccess) r = MI3D_EDEVICE;
            if (!r || want_cap < 65536) break;
            (void)hipGetLastError();
            h->d_events.release(); h->d_hvlist.release(); h->d_events2.release(); h->d_hvlist2.release();
            want_cap = (want_cap / 2) & ~(uint64_t)63;
        }
        if (r) {
            (void)hipGetLastError();
            fprintf(stderr, "Warning [mi3d_run]: no device memory for the event lists of the marched views (%.1f GB free); the general kernel marches them inside its photon loop (same results, slower).\n", (double)free_b / 1.0e9);
            return r;
        }
        ev_cap = want_cap;
        h->cold_host.ev_list = h->d_events.p; h->cold_host.ev_ctr = h->d_evctr.p; h->cold_host.ev_cap = (int)ev_cap;
        h->cold_host.hv_list = h->sfc_lambert_only ? nullptr : h->d_hvlist.p;
        return MI3D_OK;
    };
    if (split) {
        if (size_lists() != MI3D_OK) {
            // no room for the lists: the general kernel marches the rays inside its photon loop instead (same results, slower) -- except for
            // a camera that was to see its periodic images: the general kernel serves the nearest one only
            h->d_events.release(); h->d_hvlist.release(); h->d_events2.release(); h->d_hvlist2.release();
            split = false; use_col = false; two_sets = false;
            if (h->rad_kind == 1) {
                if (h->cam_images > 0) return fail(MI3D_EDEVICE, "no device memory for the event lists of a camera job with cam_images=%d (the general kernel serves the nearest image only)", h->cam_images);
                if ((rc = cam_fallback("no device memory for its event lists"))) return rc;
            }
        }
    } else if (h->d_events.p) {   // this job needs no lists: what an earlier one held goes back to the device
        HIPCHK(sync_streams(h));
        h->d_events.release(); h->d_hvlist.release(); h->d_events2.release(); h->d_hvlist2.release();
    }
    // (the general mixture's common scene -- one Rayleigh 1-D constituent, one 3-D constituent with tables -- has a build of its own where the column
    //  view serves: k_transport_lean<.,.,0,3>)
    static const bool no_mix3 = getenv("MI3D_NO_MIX3") && atoi(getenv("MI3D_NO_MIX3")) != 0;   // (measurements)
    const int mix_lean = (mix == 2 && !split && !no_mix3 && (S.target & kTargetRayleigh1d) != 0 && h->np3d == 1) ? 3 : mix;
    {
        char nm[96];
        if (use_fl) snprintf(nm, sizeof(nm), TL.cap ? "k_transport_flux<%d,%d,%d> + k_tl_scatter + k_tl_sum" : "k_transport_flux<%d,%d,%d>", h->counting ? 1 : 0, h->solver == MI3D_SOLVER_P3D ? 1 : 0, mix);
        else if (use_col) snprintf(nm, sizeof(nm), split ? "k_transport_lean<%d,%d,2,%d> + k_rays" : "k_transport_lean<%d,%d,0,%d>", h->counting ? 1 : 0,
                              h->solver == MI3D_SOLVER_P3D ? 1 : 0, mix_lean);
        else snprintf(nm, sizeof(nm), "k_transport<%d,%d,%d,%d>", h->counting ? 1 : 0, march ? 1 : 0, flux ? 1 : 0, h->solver == MI3D_SOLVER_P3D ? 1 : 0);
        h->last_kernel = nm;
        // A job that was not sent to the general loop by its caller (mi3d_set_kernel 1) but landed there says so, once per handle: the loop of
        // round 1 serves it correctly and at a fraction of the lean loops' speed (bench.py's `general_kernel` leg has the figure)
        if (!use_fl && !use_col && h->kernel_choice == 0 && !h->general_warned && nphoton >= 4096) {
            const bool both = flux && (h->target & MI3D_TARGET_RADIANCE) && h->nview > 0;
            const char *why = both ? "flux (or heating rates) TOGETHER with radiance: two jobs, one per target, as er3t's mcarats_ng submits them (mcarats.py:238-245), each take a lean loop"
                              : h->np3d > 2 ? "more than two 3-D constituents" : !tabs_ok ? "phase tables too large for the LDS (or none loaded where a selector asks for one)"
                              : !((double)h->ny * h->vrow_f4 * 16.0 < 4.0e9) ? "voxel records beyond 4 GB" : (march || h->rad_kind == 1) && !split ? "no room for the event lists of the marched views, or cell numbers beyond their 16 bits"
                              : "a scene the lean photon loops do not serve";
            fprintf(stderr, "Warning [mi3d_run]: this job runs on the general photon loop (k_transport), not on a lean one: %s.  Same results, a third to a half of the speed.\n", why);
            h->general_warned = true;
        }
    }
    static const int flux_wg_env = getenv("MI3D_FLUX_GRID_WG") ? atoi(getenv("MI3D_FLUX_GRID_WG")) : 0;   // (measurements: workgroups of the flux loop per CU, 1 ... 4)
    const uint64_t cap = (uint64_t)h->num_cu * (use_fl ? (flux_wg_env > 0 && flux_wg_env <= MI3D_FLUX_WAVES(false) ? flux_wg_env : MI3D_FLUX_WAVES(h->counting != 0)) : use_col ? MI3D_LEAN_WAVES(h->counting != 0, march) : MI3D_BLOCKS_PER_CU(march, h->counting != 0));

    // Entry records (k_entry, mi3d_kernel_lean.hip): the lean loop's builds without rays inside them take new photons where their
    // first voxel walk begins.  48 bytes per photon of a launch, never more than half of the memory that is free: without them
    // (no room, "entry_records" 0) the photons are launched inside the loop as before.
    bool use_entry = false;
#if MI3D_LEAN_FAST
    if ((use_col || use_fl) && h->entry_records && h->nx < 65536 && h->ny < 65536 && h->nz < 32768) {
        const size_t need = entry_f4((size_t)std::min<uint64_t>(nphoton, h->batch));
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
        free_b += h->d_entry.cap * sizeof(float4);
        if (need <= h->d_entry.cap || need * sizeof(float4) <= free_b / 2) {
            if (h->d_entry.alloc(need) == MI3D_OK) use_entry = true;
            else (void)hipGetLastError();
        }
    }
#endif
    h->cold_host.entry = use_entry ? h->d_entry.p : nullptr;
    // The pre-pass of a launch (photon order, entry records) on a stream of its own beside the photon loop of the launch before: a second set
    // of what it writes.  (Not together with the second set of event lists of "overlap_rays": one spare DevCold.)
    // Where the loop leaves room on a CU: the flux loop (four waves per SIMD) and the event-writing loop of jobs with marched views (+4 % for a flux
    // run alone, +2.6 % with nine views).  The column / tally-window loop holds six waves per SIMD and most of the LDS: there the pre-pass only finds
    // room in the loop's tail, and what runs beside it costs the loop more than it saves (-4 % on the 480 x 480 nadir bench, -3 % on 128 x 128:
    // profiles/r05/ab_overlap_pre.log).
    // A run of a few million photons that is read before the next one has no loop before it to hide behind and pays for the hop between the streams.
    static const int pre_nadir = getenv("MI3D_PRE_NADIR") ? atoi(getenv("MI3D_PRE_NADIR")) : 0;     // (measurements: the pre-pass beside the column-view loop too; 2: on a stream of the lowest priority)
    bool pre_two = h->overlap_pre && !two_sets && (sorted || use_entry) && (use_fl || (use_col && split) || (use_col && pre_nadir > 0)) && (h->overlap_pre > 1 || nphoton >= ((uint64_t)1 << 22) || h->runs_unread > 0);   // ("overlap_pre" 2: whatever the run)
    // (a launch of a kind that MAY be followed by a two-stream launch on this handle leaves the events the latter waits for, also when it runs on one stream itself)
    const bool pre_track = h->overlap_pre && (sorted || use_entry);
    h->runs_unread++;
    if (pre_two && !h->pre_stream) {
        int pr_least = 0, pr_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
        const hipError_t ec = pre_nadir > 1 ? hipStreamCreateWithPriority(&h->pre_stream, hipStreamNonBlocking, pr_least) : hipStreamCreateWithFlags(&h->pre_stream, hipStreamNonBlocking);
        if (ec != hipSuccess) { (void)hipGetLastError(); pre_two = false; }
    }
    for (int q = 0; q < 2 && pre_two; ++q) {
        if (!h->pre_done[q] && hipEventCreateWithFlags(&h->pre_done[q], hipEventDisableTiming) != hipSuccess) pre_two = false;
        if (pre_two && !h->pre_loop[q] && hipEventCreateWithFlags(&h->pre_loop[q], hipEventDisableTiming) != hipSuccess) pre_two = false;
    }
    if (pre_two && sorted && (h->d_order2.alloc(h->d_order.cap) != MI3D_OK || h->d_cursor2.alloc(kMaxTiles) != MI3D_OK)) { (void)hipGetLastError(); pre_two = false; }
    if (pre_two && use_entry && h->d_entry2.cap < h->d_entry.cap) {   // (a second set of the same size, where half of the free memory holds it)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        free_b += h->d_entry2.cap * sizeof(float4);
        if (h->d_entry.cap * sizeof(float4) > free_b / 2 || h->d_entry2.alloc(h->d_entry.cap) != MI3D_OK) { (void)hipGetLastError(); pre_two = false; }
    }
    // (a job that does not use the second set gives back what is large: a hipFree waits for the device -- not for every small job)
    if (!pre_two && (h->d_entry2.cap * sizeof(float4) + h->d_order2.cap * sizeof(uint32_t)) > ((size_t)256 << 20)) { h->d_order2.release(); h->d_entry2.release(); }
    // The tally window of the lean loop (DevCold::tile_end ...): for the column view of a satellite image with one pixel per column,
    // when the launch is worked through tile by tile.  Its place relative to a tile: where the direct beam that enters the top of
    // the atmosphere above the tile reaches the height of the clouds, centred.
    h->cold_host.tile_end = nullptr; h->cold_host.win_tc = 0; h->cold_host.win_ntx = 0; h->cold_host.win_ntile = 0; h->cold_host.win_off = 0u;
    if (use_col && window_wanted(h) && sorted && h->nx < 32768 && h->ny < 32768 && h->nx >= kWin && h->ny >= kWin && h->z_cloud >= 0.0 && h->cold_host.sdz < 0.0f) {
        const double ztoa = h->zgrd[h->nz];
        // (under the independent-pixel approximation a photon never leaves the column it was launched above)
        const double way = h->solver == MI3D_SOLVER_IPA ? 0.0 : (ztoa - h->z_cloud) / std::fabs((double)h->cold_host.sdz);
        const int margin = (kWin - std::min(G.tcols, kWin)) / 2 - std::max(0, G.tcols - kWin) / 2;   // (a tile wider than the window: its middle)
        auto wrap = [](long v, int n) { v %= n; if (v < 0) v += n; return (unsigned)v; };
        // (the centring was checked by moving the window about: ±12 columns change the share of the tallies it catches by less than a
        //  point, ±24 cost ten -- profiles/r04/win_offset_probe.log)
        const unsigned ox = wrap(std::lround(h->cold_host.sdx * way / h->dx) - margin, h->nx);
        const unsigned oy = wrap(std::lround(h->cold_host.sdy * way / h->dy) - margin, h->ny);
        h->cold_host.tile_end = h->d_cursor.p;
        h->cold_host.win_tc = G.tcols; h->cold_host.win_ntx = G.ntx; h->cold_host.win_ntile = ntile;
        h->cold_host.win_off = ox | (oy << 16);
    }
    // Radiance tallies go to an accumulation image with one pixel per 128-byte line and are folded into the tally buffer (the
    // library's own or the caller's) after the last launch of this call -- where the tallies are atomics from the photon loop itself
    // ("rad_spread" -1, the default).  Not where the ray kernel makes them (marched views, cameras): its tallies spread over the images
    // of several views or a wide camera image, and the compact image (8 bytes per pixel: 1.8 MB per view on the 480 x 480 scene, at home
    // in the L2s) serves them better than 29 MB per view -- nine views 2.83 -> 3.11e8 photons/s, the camera +1 %
    // (profiles/r04/ab_rad_line_density.log).  With the tally window either image does (2.79 / 2.78e9): the accumulation image stays.
    // Rows of the accumulation image an ODD number of 4 KiB pages apart (32 pixels of kRadLine * 8 bytes a page).  The atomics of an
    // XCD go to the pixels of one tile of columns, a few rows of which are busy at any time; with rows a multiple of four pages
    // apart they meet in a few of the L2's sixteen channels: 128 pixels per row (4 pages) 1.93e9 photons/s against 2.47e9 at 5 pages
    // and 2.52 at 4.5; 256 (8 pages) 1.63 against 2.25 at 9; 64 pixels padded to 4 pages 1.19 against 2.5; 512 (16) 2.11 against 2.25
    // at 17.  Odd page counts came within 2 % of the best stride on every size tried (192 ... 512 pixels: 7, 9, 11, 13, 15, 17
    // pages); half pages are good on some sizes and the worst choice on others (15.5 and 16.5 pages: 1.96 and 2.00 against 2.28).
    // profiles/r04/stride_probe4.log ... stride_probe6.log, ab_rad_row_pad*.log; tools/stride_probe*.py
    int rad_row = h->nxr + (h->rad_row_pad >= 0 ? h->rad_row_pad : 0);
    if (h->rad_row_pad < 0) rad_row = 32 * (((h->nxr + 31) / 32) | 1);
    const size_t acc_elems = (size_t)h->nview * h->nyr * rad_row;
    const bool spread = (h->target & MI3D_TARGET_RADIANCE) && h->nview > 0 && h->rad_spread != 0 && !(h->rad_spread < 0 && use_col && split) &&
                        (double)acc_elems * kRadLine * sizeof(tally_t) <= 1.0e9 && (double)acc_elems * kRadLine < 2147483647.0;
    if (spread) {
        const size_t need = acc_elems * kRadLine;
        if (h->d_rad_acc.cap < need || !h->d_rad_acc.p) {
            if ((rc = h->d_rad_acc.alloc(need))) return rc;
            HIPCHK(hipMemsetAsync(h->d_rad_acc.p, 0, need * sizeof(tally_t), h->stream));
        }
        S.rad = h->d_rad_acc.p; S.rad_stride = kRadLine; S.rad_row = rad_row;
    }
    h->cold_host.cam_images = (unsigned)(split ? (h->cam_images < 0 ? 2 : h->cam_images) : 0);

    HIPCHK(upload_cold(h, two_sets, pre_two));
    // With two sets the time of a run is the span from its first launch to the end of its last ray kernel (one pair of events on the
    // main stream, which joins the rays' stream at the end); launches that overlap cannot be timed one by one
    hipEvent_t run_e0 = nullptr;
    // (every way out of this function that still owns run_e0 is an error exit of a run on two streams: the event goes, and what is queued on
    //  the sort / pre-pass / rays streams is waited for -- ADVICE r5: several early returns left both behind)
    struct RunGuard { hipEvent_t &e; mi3d_solver *h; ~RunGuard() { if (e) { (void)hipEventDestroy(e); e = nullptr; (void)sync_streams(h); } } } run_guard{run_e0, h};
    if (!use_fl) tl_two = false;
    const bool run_timed = two_sets || tl_two;   // (kernels on two streams: the run is timed as a whole, not launch by launch)
    if (run_timed) { HIPCHK(hipEventCreate(&run_e0)); HIPCHK(hipEventRecord(run_e0, h->stream)); }
    uint64_t ilaunch = 0;

    // equal launches (a short last one would be mostly tail).  With k_rays a launch is as many photons as the event lists hold
    // at the number of events per photon seen so far (twice the room: the lists fill unevenly); the first launch of a handle is a
    // pilot of 65 536 photons.
    const uint64_t nlaunch = (nphoton + h->batch - 1) / h->batch;
    uint64_t per = (nphoton + nlaunch - 1) / nlaunch;
    for (uint64_t done = 0; done < nphoton; done += per) {
        if (split) {
            if ((rc = ev_collect(h, ev_cap, false))) { return rc; }
            const uint64_t room = photons_that_fit(ev_cap, h->ev_per_photon, h->n_xcd);
            const uint64_t left = nphoton - done, want_n = std::max<uint64_t>(std::min<uint64_t>(room, h->batch), 64);
            const uint64_t nl = (left + want_n - 1) / want_n;
            per = (left + nl - 1) / nl;
        }
        if (TL.cap) {
            // as many photons as the record list holds at the records per photon seen so far
            if ((rc = tl_collect(h, false))) return rc;
            const uint64_t waves = (uint64_t)h->num_cu * MI3D_FLUX_WAVES(h->counting != 0) * 4;
            // (what each of the three lists holds at its rate per photon seen so far: the loop's records, the sorted copy, the run records)
            auto room_of = [&](uint64_t cap_, uint64_t chunk, double pp_) -> uint64_t {
                return (uint64_t)((double)(cap_ > 2 * waves * chunk ? cap_ - waves * chunk : cap_ / 2) / pp_);
            };
            const double tpp = h->tl_total_pp > 0.0 ? 1.15 * h->tl_total_pp : (h->tl_per_photon > 0.0 && !TL.run_cap ? 1.15 * h->tl_per_photon : 1.5 * (h->nz + 1));
            uint64_t room = room_of(TL.cap, kTlChunk, h->tl_per_photon > 0.0 ? 1.15 * h->tl_per_photon : tpp);
            room = std::min(room, room_of(TL.bcap, kTlChunk, tpp));
            if (TL.run_cap) room = std::min(room, room_of(TL.run_cap, kRunChunk, h->tl_runs_pp > 0.0 ? 1.15 * h->tl_runs_pp + 0.05 : tpp / kRunMin));
            const uint64_t left = nphoton - done, want_n = std::max<uint64_t>(std::min<uint64_t>(room, h->batch), 4096);
            uint64_t nl = (left + want_n - 1) / want_n;
            // (the sort beside the next photon loop: a run in tl_split launches at least, of eight million photons or more -- all but the last sort are hidden)
            if (tl_two && done == 0) nl = std::max<uint64_t>(nl, std::min<uint64_t>((uint64_t)h->tl_split, nphoton >> 23));
            else if (tl_two) nl = std::max<uint64_t>(nl, std::min<uint64_t>((left + per - 1) / std::max<uint64_t>(per, 1), (uint64_t)h->tl_split));
            nl = std::max<uint64_t>(nl, 1);
            per = (left + nl - 1) / nl;
            static const bool tl_verbose2 = getenv("MI3D_TL_VERBOSE") != nullptr;
            if (tl_verbose2) fprintf(stderr, "[mi3d launch] left %.4g room %.4g (rec %.4g sorted %.4g runs %.4g) caps %u %u %u pp %.3f %.3f %.4f two %d -> %llu launches\n", (double)left, (double)room,
                                     (double)room_of(TL.cap, kTlChunk, h->tl_per_photon > 0.0 ? 1.15 * h->tl_per_photon : tpp), (double)room_of(TL.bcap, kTlChunk, tpp),
                                     TL.run_cap ? (double)room_of(TL.run_cap, kRunChunk, h->tl_runs_pp > 0.0 ? 1.15 * h->tl_runs_pp + 0.05 : tpp / kRunMin) : 0.0, TL.cap, TL.bcap, TL.run_cap,
                                     h->tl_per_photon, h->tl_total_pp, h->tl_runs_pp, (int)tl_two, (unsigned long long)nl);
        }
        const uint64_t nb = std::min<uint64_t>(per, nphoton - done), off = photon_offset + done;
        if (h->pending.size() >= 64 && (rc = drain_events(h))) return rc;
        HIPCHK(hipMemsetAsync(h->d_next.p, 0, 8 * kCtrStride * sizeof(unsigned long long), h->stream));
        // ---- the pre-pass: photon order and entry records of this launch (set pset; with two sets on their own stream, behind the photon loop
        //      that read the set last and beside the one that is running)
        const int pset = pre_two ? (int)(h->pre_no & 1) : 0;
        hipStream_t const ps = pre_two ? h->pre_stream : h->stream;
        uint32_t *const ord = pset ? h->d_order2.p : h->d_order.p;
        float4 *const ent = pset ? h->d_entry2.p : h->d_entry.p;
        {
            hipError_t eb = hipSuccess;
            // (set pset is free once the photon loop of the launch that read it last is through -- whichever stream arrangement that launch
            //  had: a one-stream launch records pre_loop[0] too, below -- and the histogram / tile scratch the two sets share once the last
            //  pre-pass on the main stream is: ADVICE r5, a two-stream run queued straight behind a one-stream run)
            if (pre_two && h->pre_used[pset]) eb = hipStreamWaitEvent(ps, h->pre_loop[pset], 0);
            if (eb == hipSuccess && pre_two && h->pre_main_used) { eb = hipStreamWaitEvent(ps, h->pre_main, 0); h->pre_main_used = false; }
            if (eb == hipSuccess && sorted) eb = launch_bins(h, ps, G, ntile, seed, off, nb, ord, pset ? h->d_cursor2.p : h->d_cursor.p);
            if (eb == hipSuccess && use_entry)   // the photons of this launch up to their first voxel walk
                eb = launch_entry(h, ps, S, nb, seed, off, sorted ? (const uint32_t *)ord : (const uint32_t *)nullptr, ent);
            if (eb == hipSuccess && pre_two) eb = hipEventRecord(h->pre_done[pset], ps);
            if (eb == hipSuccess && pre_track && !pre_two) {
                if (!h->pre_main) eb = hipEventCreateWithFlags(&h->pre_main, hipEventDisableTiming);
                if (eb == hipSuccess) { eb = hipEventRecord(h->pre_main, h->stream); h->pre_main_used = true; }
            }
            if (eb == hipSuccess && pre_two) eb = hipStreamWaitEvent(h->stream, h->pre_done[pset], 0);
            if (eb != hipSuccess) { (void)sync_streams(h); return fail(MI3D_EDEVICE, "photon order / entry records: %s", hipGetErrorString(eb)); }
        }
        DevScene Sl = S;
        if (pset) Sl.cold = h->d_cold.p + 1;   // (this launch's set of pre-pass buffers)
        const uint64_t want = (nb + tb - 1) / tb;
        const unsigned grid = (unsigned)(want < cap ? want : cap);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipError_t err = hipSuccess;
        if (!run_timed) {
            err = hipEventCreate(&e0);
            if (err == hipSuccess) err = hipEventCreate(&e1);
            if (err == hipSuccess) err = hipEventRecord(e0, h->stream);
        }
        const int set = two_sets ? (int)(ilaunch & 1) : 0;
        const int tset = tl_two ? (int)(h->tl_launch_no & 1) : 0;
        const TallyList &TLs = tset ? TL2 : TL;
        unsigned long long *const set_ctr = set ? h->d_evctr2.p : h->d_evctr.p;
        hipStream_t const rs = two_sets ? h->rays_stream : h->stream;
        if (err == hipSuccess && use_fl) {
            if (err == hipSuccess) err = launch_flux(h, tset, h->stream, tl_two ? h->tl_stream : h->stream, h->tl_filled[tset], h->tl_scattered[tset], tl_two && h->tl_set_used[tset], Sl, TLs, mix, grid, lds_fl, nb, seed, off);
            if (err == hipSuccess && tl_two) { err = hipEventRecord(h->tl_sorted[tset], h->tl_stream); h->tl_set_used[tset] = true; h->tl_unjoined = true; }
        } else if (err == hipSuccess && use_col) {
            const int emit_wg = h->counting ? 4 : (h->emit_wg > 0 ? std::min(h->emit_wg, MI3D_LEAN_EMIT_GRID) : MI3D_LEAN_EMIT_GRID);
            const unsigned gridp = split ? (unsigned)std::min<uint64_t>(want, (uint64_t)h->num_cu * emit_wg) : grid;
            DevScene Sx = Sl;
            if (set) Sx.cold = h->d_cold.p + 1;       // (this launch's set of event lists)
            // (the set is free once the ray kernels of the launch that used it last are through it)
            if (two_sets && h->set_used[set]) err = hipStreamWaitEvent(h->stream, h->set_rays[set], 0);
            if (err == hipSuccess && split) err = hipMemsetAsync(set_ctr, 0, kCtrWords * kCtrStride * sizeof(unsigned long long), h->stream);
            if (err == hipSuccess) {
                // (the general mixture with staged tables AND a tally window: three workgroups of 512 threads keep six waves per SIMD where
                //  six of 256 would not find the LDS)
                const size_t lds_lean = lds_col + (h->cold_host.tile_end ? kWinLds : 0) + lds_tab;
                static const int wide_env = getenv("MI3D_LEAN_WIDE") ? atoi(getenv("MI3D_LEAN_WIDE")) : -1;     // (measurements: 0 never, 1 whenever three fit)
                const bool wide = mix >= 2 && !split && !h->counting && lds_lean * 3 <= (size_t)160 * 1024 && (wide_env < 0 ? lds_lean * 6 > (size_t)160 * 1024 : wide_env != 0);
                unsigned gridw = wide ? (unsigned)std::min<uint64_t>((nb + 511) / 512, (uint64_t)h->num_cu * 3) : gridp;
                // (the general mixture in workgroups of 256: as many as its registers -- five waves per SIMD -- and its LDS let a CU hold)
                if (mix >= 2 && !wide && !split && !h->counting) gridw = std::min<unsigned>(gridw, (unsigned)h->num_cu * (unsigned)std::max<size_t>(1, std::min<size_t>(MI3D_GEN_NARROW_WAVES, ((size_t)160 * 1024) / lds_lean)));
                err = launch_lean(h, h->stream, Sx, split, mix_lean, wide ? 512 : 256, gridw, lds_lean, nb, seed, off);
            }
            if (err == hipSuccess && split) {  // the rays of the events just written: on their own stream beside the next launch's photon loop
                if (two_sets) {
                    err = hipEventRecord(h->set_emit[set], h->stream);
                    if (err == hipSuccess) err = hipStreamWaitEvent(rs, h->set_emit[set], 0);
                }
                const bool plain_scene = (S.target & kTargetPlainPhase) != 0;
                if (err == hipSuccess) err = launch_rays(h, rs, Sx, false, lds_col + rays_lds_extra(h->nz, h->rad_kind == 1) + (plain_scene && h->rad_kind != 1 ? 0 : lds_tab), seed);
                if (err == hipSuccess && !h->sfc_lambert_only && h->rad_kind != 1)   // the reflections off LSRT / DSM surfaces it left aside
                    err = launch_rays(h, rs, Sx, true, lds_col + rays_lds_extra(h->nz) + lds_tab, seed);
            }
        } else if (err == hipSuccess) {
#define MI3D_LAUNCH(C, M, F)                                                                                              \
    do {                                                                                                                 \
        if (h->solver == MI3D_SOLVER_P3D)                                                                                \
            hipLaunchKernelGGL((k_transport<C, M, F, true>), dim3(grid), dim3(tb), lds, h->stream, S, nb, seed, off);    \
        else                                                                                                             \
            hipLaunchKernelGGL((k_transport<C, M, F, false>), dim3(grid), dim3(tb), lds, h->stream, S, nb, seed, off);   \
    } while (0)
            switch (variant) {
                case 0: MI3D_LAUNCH(false, false, false); break;
                case 1: MI3D_LAUNCH(false, false, true); break;
                case 2: MI3D_LAUNCH(false, true, false); break;
                case 3: MI3D_LAUNCH(false, true, true); break;
                case 4: MI3D_LAUNCH(true, false, false); break;
                case 5: MI3D_LAUNCH(true, false, true); break;
                case 6: MI3D_LAUNCH(true, true, false); break;
                default: MI3D_LAUNCH(true, true, true); break;
            }
#undef MI3D_LAUNCH
            err = hipGetLastError();
        }
        if (err == hipSuccess && !run_timed) err = hipEventRecord(e1, h->stream);
        h->pre_last = pset;
        if (err == hipSuccess && (pre_two || pre_track)) {   // (the set may be written again once this launch's photon loop is through it)
            if (!h->pre_loop[pset]) err = hipEventCreateWithFlags(&h->pre_loop[pset], hipEventDisableTiming);
            if (err == hipSuccess) err = hipEventRecord(h->pre_loop[pset], h->stream);
            h->pre_used[pset] = true;
            if (pre_two) h->pre_no++;
        }
        if (err != hipSuccess) {   // (no event is left behind on the error path)
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            (void)sync_streams(h);
            return fail(MI3D_EDEVICE, "transport launch failed: %s", hipGetErrorString(err));
        }
        if (!run_timed) h->pending.emplace_back(e0, e1);
        h->launches++; ilaunch++;
        if (TL.cap) {
            const bool first = !(h->tl_total_pp > 0.0);
            if ((rc = tl_note(h, TL.cap, TL.bcap, TL.run_cap, nb, h->d_tl_stats.p + 4 * ((h->launches - 1) % 64), tl_two ? h->tl_stream : h->stream))) return rc;
            h->tl_launch_no++;
            // (nothing known about the scene's records per photon yet: the first launch is waited for -- the ones to come are then
            //  sized by what it needed instead of by a guess twice too large)
            if (first && done + nb < nphoton) {
                if ((rc = tl_collect(h, true))) return rc;
                // the pilot is through: lists for the rest of the run at the rates it has shown (every stream joined: the lists may move)
                HIPCHK(sync_streams(h));
                const unsigned cap_before = TL.cap;
                if (!size_tally_lists(std::min<uint64_t>(nphoton - done - nb, h->batch)) && cap_before) return fail(MI3D_EDEVICE, "no device memory to grow the tally-record lists after the pilot launch");
            }
        }
        if (split) {
            // how full the lists got sizes the launches to come; read while they run (only a pilot is waited for)
            rc = ev_note(h, ev_cap, nb, rs, set_ctr);
            if (!rc && two_sets) {            // (after the copy of its fill counters: the set may be zeroed and filled again)
                if (hipEventRecord(h->set_rays[set], rs) != hipSuccess) rc = fail(MI3D_EDEVICE, "event record failed");
                h->set_used[set] = true;
            }
            if (rc) { (void)sync_streams(h); return rc; }
            if (!(h->ev_per_photon > 0.0)) {
                // a pilot: wait for it, then give the lists the size the rest of the run needs
                if ((rc = ev_collect(h, ev_cap, true))) { return rc; }
                if (done + nb < nphoton) {
                    HIPCHK(sync_streams(h));
                    if (size_lists() != MI3D_OK) return fail(MI3D_EDEVICE, "no device memory to grow the event lists of the marched views after the pilot launch");
                    HIPCHK(upload_cold(h, two_sets, pre_two));
                }
            }
        }
    }
    // (the last launches' fill counters are looked at by whoever reads the tallies next -- ev_settle -- or by the next run: the
    //  host does not wait here, so that the next run's launches queue up behind this one's)
    // (tallies in the handle's own buffers are read through calls that join the sort stream first -- tl_join --: the next run's photon
    //  loops may start beside this run's last sort.  Buffers or a stream of the caller's: work the caller queues after this call must
    //  find the tallies complete, the main stream waits here)
    // ONE completion rule (round 6): when mi3d_run returns, everything it has started is queued on the handle's main stream or joined to it --
    // whatever the caller queues on that stream next, and every mi3d call, finds the tallies complete in stream order.  (Round 5 returned before
    // the last sort had joined where the tallies lived in the handle's own buffers, so that the next run's loops started beside it: worth 1-2 %
    // for runs queued back to back since the sums moved between the loops, and a rule a ctypes caller had to know.  "overlap_sort" 2 still asks
    // for it, for such runs on the handle's own buffers and stream: every reading call joins.)
    const bool tl_lazy = h->overlap_sort == 2 && tl_two && !two_sets && !h->flux_ext && !h->heat_ext && !h->rad_ext && (h->stream == nullptr || h->stream == h->own_stream);
    if (run_timed) {
        // (tl_unjoined has been set with the first sort of this run: a run that fails half way leaves it set, and the next call that looks at
        //  the tallies joins)
        if (tl_two && !tl_lazy) HIPCHK(tl_join(h));   // (... or the sort's)
        // the main stream joins the rays' stream: whatever follows on it -- the fold below, the next run, a read-out after mi3d_sync -- comes
        // after the last ray kernel; the run's time is the span up to here
        for (int q = 0; q < 2 && two_sets; ++q) if (h->set_used[q]) HIPCHK(hipStreamWaitEvent(h->stream, h->set_rays[q], 0));
        hipEvent_t run_e1 = nullptr;
        HIPCHK(hipEventCreate(&run_e1)); HIPCHK(hipEventRecord(run_e1, tl_two && tl_lazy ? h->tl_stream : h->stream));
        h->pending.emplace_back(run_e0, run_e1);
        run_e0 = nullptr;   // (owned by the pending list from here on)
    }
    if (split && (rc = ev_collect(h, ev_cap, false))) { (void)sync_streams(h); return rc; }
    if (TL.cap && (rc = tl_collect(h, false))) return rc;
    if (spread) {
        const int n = (int)h->rad_elems();
        hipLaunchKernelGGL(k_fold_rad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_rad_acc.p, h->rad_ptr(), kRadLine, n, h->nxr, rad_row);
        HIPCHK(hipGetLastError());
    }
    return MI3D_OK;
}

int mi3d_sync(mi3d_solver *h) {
    int rc = check_handle(h);
    if (rc) return rc;
    HIPCHK(sync_main(h));
    return ev_settle(h);
}

int mi3d_set_kernel(mi3d_solver *h, int choice) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (choice < 0 || choice > 1) return fail(MI3D_EINVAL, "kernel choice %d (0: the lean kernels where they apply, 1: always the general one; choice 2, the lean loop with the rays of marched views inside it, was retired in round 5)", choice);
    HIPCHK(tl_join(h));   // (as every call that changes what the kernels work on: sorts still on their way join the main stream first)
    h->kernel_choice = choice;
    return MI3D_OK;
}

int mi3d_set_tuning(mi3d_solver *h, const char *key, int value) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!key) return fail(MI3D_EINVAL, "NULL key");
    const std::string k(key);
    if (k == "tile_cols") { if (value < -1 || value > 4096) return fail(MI3D_EINVAL, "tile_cols=%d", value); h->tile_cols = value; }
    else if (k == "batch_log2") { if (value < 8 || value > 30) return fail(MI3D_EINVAL, "batch_log2=%d outside [8,30]", value); h->batch = (uint64_t)1 << value; }
    else if (k == "evcap_log2") {
        if (value < 10 || value > 28) return fail(MI3D_EINVAL, "evcap_log2=%d outside [10,28]", value);
        HIPCHK(sync_streams(h));
        h->ev_cap_log2 = value; ev_forget(h);
        h->d_events.release(); h->d_hvlist.release(); h->d_events2.release(); h->d_hvlist2.release();   // (lists only grow otherwise)
    }
    else if (k == "rad_spread") h->rad_spread = value < 0 ? -1 : (value ? 1 : 0);   // (-1: the default choice by route, mi3d_run)
    else if (k == "tally_window") h->tally_window = value ? 1 : 0;
    else if (k == "rad_row_pad") { if (value < -1 || value > 4096) return fail(MI3D_EINVAL, "rad_row_pad=%d outside [-1,4096]", value); h->rad_row_pad = value; }
    else if (k == "tlcap_log2") {
        if (value < 16 || value > 31) return fail(MI3D_EINVAL, "tlcap_log2=%d outside [16,31]", value);
        HIPCHK(sync_main(h));
        h->tl_cap_log2 = value; h->tl_per_photon = 0.0; h->tl_total_pp = 0.0; h->tl_runs_pp = 0.0;
        for (bool &b : h->tl_busy) b = false;
        h->d_tl_rec.release(); h->d_tl_binned.release(); h->d_tl_rec2.release(); h->d_tl_runs.release(); h->d_tl_runs2.release(); h->d_tl_binned2.release();
    }
    else if (k == "vpad_col" || k == "vpad_row") {
        // padding of the voxel records' strides, in records of 16 bytes (DevScene::vcol_f4, vrow_f4): where a grid's strides alias in
        // the L2 -- the sorted photon order then gains little -- another stride may bring some of it back (profiles/r04/stride_probe*.log)
        if (value < 0 || value > 4096) return fail(MI3D_EINVAL, "%s=%d outside [0,4096]", key, value);
        (k == "vpad_col" ? h->vpad_col : h->vpad_row) = value;
        h->dirty_grid = true;
    }
    else if (k == "overlap_rays") { HIPCHK(sync_streams(h)); h->overlap_rays = value ? 1 : 0; }
    else if (k == "overlap_pre") { if (value < 0 || value > 2) return fail(MI3D_EINVAL, "overlap_pre=%d outside [0,2]", value); HIPCHK(sync_streams(h)); h->overlap_pre = value; if (!value) { h->d_order2.release(); h->d_entry2.release(); h->pre_last = 0; } }
    else if (k == "overlap_sort") { if (value < 0 || value > 2) return fail(MI3D_EINVAL, "overlap_sort=%d outside [0,2]", value); HIPCHK(sync_streams(h)); h->overlap_sort = value; if (!value) { h->d_tl_rec2.release(); h->d_tl_words2.release(); h->d_tl_binned2.release(); h->d_tl_runs2.release(); } }
    else if (k == "tl_split") { if (value < 1 || value > 64) return fail(MI3D_EINVAL, "tl_split=%d outside [1,64]", value); h->tl_split = value; }
    else if (k == "rays_wg" || k == "emit_wg") { if (value < 0 || value > 8) return fail(MI3D_EINVAL, "%s=%d outside [0,8]", key, value); (k == "rays_wg" ? h->rays_wg : h->emit_wg) = value; }
    else if (k == "cam_images") { if (value < -1 || value > 8) return fail(MI3D_EINVAL, "cam_images=%d outside [-1,8]", value); h->cam_images = value; }
    else if (k == "entry_records") {
        HIPCHK(sync_main(h));
        h->entry_records = value ? 1 : 0;
        if (!value) h->d_entry.release();
    }
    else if (k == "tally_runs") { HIPCHK(sync_main(h)); h->tally_runs = value ? 1 : 0; h->tl_per_photon = 0.0; h->tl_total_pp = 0.0; h->tl_runs_pp = 0.0; if (!value) { h->d_tl_runs.release(); h->d_tl_runs2.release(); } }
    else if (k == "tally_lists") {
        HIPCHK(sync_main(h));
        h->tally_lists = value ? 1 : 0;
        if (!value) { h->d_tl_rec.release(); h->d_tl_binned.release(); h->d_tl_rec2.release(); h->d_tl_binned2.release(); h->d_tl_runs.release(); h->d_tl_runs2.release(); }
    }
    else if (k == "own_stream") {
        // a stream of the handle's own (non-blocking) wherever the caller binds none; the caller orders its own work with mi3d_sync
        HIPCHK(sync_main(h));
        const bool was_default = (h->stream == nullptr) || (h->stream == h->own_stream);
        if (value && !h->own_stream && hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
            return fail(MI3D_EDEVICE, "cannot create a stream");
        h->use_own_stream = value != 0;
        if (was_default) h->stream = h->use_own_stream ? h->own_stream : nullptr;
    }
    else return fail(MI3D_EINVAL, "unknown tuning key '%s'", key);
    return MI3D_OK;
}

const char *mi3d_last_kernel(mi3d_solver *h) { return h ? h->last_kernel.c_str() : ""; }

int mi3d_get_timing(mi3d_solver *h, double *kernel_ms, uint64_t *launches) {
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = drain_events(h))) return rc;
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    if (launches) *launches = h->launches;
    return MI3D_OK;
}

int mi3d_get_radiance(mi3d_solver *h, uint64_t nphoton_total, float *out) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out || nphoton_total == 0) return fail(MI3D_EINVAL, "bad arguments to mi3d_get_radiance");
    if (!h->rad_ptr()) return fail(MI3D_ESTATE, "no radiance tally (nothing has run)");
    HIPCHK(sync_main(h));
    if ((rc = ev_settle(h))) return rc;
    const size_t n = (size_t)h->nview * h->nxr * h->nyr;
    const double pi = 3.14159265358979323846;
    const double mu0 = std::fabs(std::cos(h->src_the * pi / 180.0));
    // satellite: radiance averaged over the pixel's share of the domain area; camera: the tallies hold 1 / (r^2 dOmega) already and
    // a photon stands for Src_flx mu0 Lx Ly / N of power
    const double fac = h->rad_kind == 1 ? h->src_flx * mu0 * (h->dx * h->nx) * (h->dy * h->ny) / (double)nphoton_total
                                        : h->src_flx * mu0 * (double)h->nxr * (double)h->nyr / (double)nphoton_total;
    // (scaled on the device, k_get_field: nine views of 480 x 480 pixels were 16 MB of float64 to the host and a loop over them)
    if (n == 0) return MI3D_OK;
    if ((rc = h->d_get_out.alloc(n))) return rc;
    hipLaunchKernelGGL(k_get_field, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const tally_t *)h->rad_ptr(), h->d_get_out.p, fac,
                       1u, 1u, -1L, (const double *)nullptr, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->d_get_out.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI3D_OK;
}

int mi3d_get_flux(mi3d_solver *h, uint64_t nphoton_total, float *out) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out || nphoton_total == 0) return fail(MI3D_EINVAL, "bad arguments to mi3d_get_flux");
    if (!h->flux_ptr()) return fail(MI3D_ESTATE, "no flux tally (nothing has run)");
    HIPCHK(sync_main(h));
    const size_t n = h->flux_elems();
    const double pi = 3.14159265358979323846;
    const double mu0 = std::fabs(std::cos(h->src_the * pi / 180.0));
    const double fac = h->src_flx * mu0 * (double)h->nx * (double)h->ny / (double)nphoton_total;
    // the raw planes are direct-down, diffuse-down, up (one atomic per crossing); the result planes direct-down, total-down, up.
    // At the levels above the 3-D region the direct beam is not tallied but known: Src_flx*mu0*exp(-tau/mu0).
    // (normalised on the device, k_get_field: float32 crosses to the host)
    const double amp = h->src_flx * mu0;
    const double *add = nullptr;
    if (!h->dir_level.empty()) {
        std::vector<double> a(h->dir_level);
        for (double &x : a) x = amp * x;
        if ((rc = h->d_get_add.upload(a.data(), a.size()))) return rc;
        add = h->d_get_add.p;
    }
    if (n == 0) return MI3D_OK;
    if ((rc = h->d_get_out.alloc(n))) return rc;
    hipLaunchKernelGGL(k_get_field, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->flux_ptr(), h->d_get_out.p, fac,
                       (unsigned)((size_t)h->nx * h->ny), (unsigned)(h->nz + 1), (long)(n / 3), add, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->d_get_out.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI3D_OK;
}

int mi3d_get_direct_levels(mi3d_solver *h, double *out) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(MI3D_EINVAL, "out is NULL");
    if ((int)h->dir_level.size() != h->nz + 1) return fail(MI3D_ESTATE, "no job has run on this handle since its grid was set");
    const double pi = 3.14159265358979323846;
    const double amp = h->src_flx * std::fabs(std::cos(h->src_the * pi / 180.0));
    for (int L = 0; L <= h->nz; ++L) out[L] = amp * h->dir_level[L];
    return MI3D_OK;
}

int mi3d_get_heating(mi3d_solver *h, uint64_t nphoton_total, float *out) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out || nphoton_total == 0) return fail(MI3D_EINVAL, "bad arguments to mi3d_get_heating");
    if (!(h->target & MI3D_TARGET_HEAT) || !h->heat_ptr()) return fail(MI3D_ESTATE, "no heating-rate tally (the job's target does not include MI3D_TARGET_HEAT, or nothing has run)");
    HIPCHK(sync_main(h));
    const size_t n = h->heat_elems();
    const double pi = 3.14159265358979323846;
    const double mu0 = std::fabs(std::cos(h->src_the * pi / 180.0));
    const double fac = h->src_flx * mu0 * (double)h->nx * (double)h->ny / (double)nphoton_total;
    // (on the device as the flux: tally * fac / layer thickness, float32 to the host)
    std::vector<double> dz((size_t)h->nz);
    for (int k = 0; k < h->nz; ++k) dz[k] = h->zgrd[k + 1] - h->zgrd[k];
    if ((rc = h->d_get_add.upload(dz.data(), dz.size()))) return rc;
    if (n == 0) return MI3D_OK;
    if ((rc = h->d_get_out.alloc(n))) return rc;
    hipLaunchKernelGGL(k_get_field, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const tally_t *)h->heat_ptr(), h->d_get_out.p, fac,
                       (unsigned)((size_t)h->nx * h->ny), (unsigned)h->nz, -1L, (const double *)h->d_get_add.p, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->d_get_out.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI3D_OK;
}

int mi3d_get_counters(mi3d_solver *h, uint64_t out[MI3D_NCOUNTER]) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(MI3D_EINVAL, "out is NULL");
    HIPCHK(sync_main(h));
    if ((rc = ev_settle(h))) return rc;
    unsigned long long tmp[MI3D_NCOUNTER];
    HIPCHK(hipMemcpy(tmp, h->d_counters.p, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int i = 0; i < MI3D_NCOUNTER; ++i) out[i] = tmp[i];
    return MI3D_OK;
}

// ---- run statistics --------------------------------------------------------------------------
static int stats_mark(mi3d_solver *h) {
    if (!h->stats_ev) HIPCHK(hipEventCreateWithFlags(&h->stats_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->stats_ev, h->stream));
    return MI3D_OK;
}

int mi3d_stats_begin(mi3d_solver *h, void *rad_run, void *flux_run) {
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = mi3d_prepare(h))) return rc;
    h->run_ext[0] = (float *)rad_run;
    h->run_ext[1] = (float *)flux_run;
    for (int w = 0; w < 2; ++w) {
        const bool used = (h->target & (w == 0 ? MI3D_TARGET_RADIANCE : MI3D_TARGET_FLUX)) != 0;
        const size_t n = used ? h->stat_elems(w) : 0;
        if (!used || n == 0) continue;
        if (!h->run_ext[w] && (rc = h->d_run_own[w].alloc(n))) return rc;
        if ((rc = h->d_sum[w].alloc(n)) || (rc = h->d_sumsq[w].alloc(n))) return rc;
        HIPCHK(hipMemsetAsync(h->run_ptr(w), 0, n * sizeof(float), h->stream));
        HIPCHK(hipMemsetAsync(h->d_sum[w].p, 0, n * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(h->d_sumsq[w].p, 0, n * sizeof(double), h->stream));
    }
    h->stats_on = true;
    h->stats_joined = false;
    h->stats_nrun = 0;
    return MI3D_OK;
}

int mi3d_stats_set_analytic_share(mi3d_solver *h, double share) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!(share >= 0.0 && share <= 1.0)) return fail(MI3D_EINVAL, "analytic share %g outside [0,1]", share);
    h->analytic_share = share;
    return MI3D_OK;
}

int mi3d_stats_add(mi3d_solver *h, uint64_t nphoton_total, const float *factor_rad, const float *factor_flux) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!h->stats_on) return fail(MI3D_ESTATE, "mi3d_stats_begin has not been called");
    if (nphoton_total == 0) return fail(MI3D_EINVAL, "nphoton_total is 0");
    const double pi = 3.14159265358979323846;
    const double mu0 = std::fabs(std::cos(h->src_the * pi / 180.0));
    for (int w = 0; w < 2; ++w) {
        if (!(h->target & (w == 0 ? MI3D_TARGET_RADIANCE : MI3D_TARGET_FLUX))) continue;
        const size_t n = h->stat_elems(w);
        if (n == 0) continue;
        if (!h->stats_joined && (!h->d_sum[w].p || h->d_sum[w].cap < n)) return fail(MI3D_ESTATE, "the scene changed shape since mi3d_stats_begin");
        const int nlevel = w == 0 ? h->nview : h->nz + 1;
        const int plane = w == 0 ? h->nxr * h->nyr : h->nx * h->ny;
        const float *fsrc = w == 0 ? factor_rad : factor_flux;
        std::vector<float> ones;
        if (!fsrc) { ones.assign(nlevel, 1.0f); fsrc = ones.data(); }
        // the factors are consumed by a kernel that may still be queued when the caller's buffer goes away
        if ((rc = h->d_factor[w].alloc(nlevel))) return rc;
        HIPCHK(hipMemcpyAsync(h->d_factor[w].p, fsrc, nlevel * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(sync_main(h));
        // (this call reads the tallies of the job that has just run: a launch of it whose event list ran full fails the call HERE -- the next
        //  mi3d_reset would forget it, and the short tallies would be part of the run's mean and standard deviation for good)
        if ((rc = ev_settle(h))) return rc;
        const double norm = w == 0 ? (h->rad_kind == 1 ? h->src_flx * mu0 * (h->dx * h->nx) * (h->dy * h->ny) / (double)nphoton_total
                                                       : h->src_flx * mu0 * (double)h->nxr * (double)h->nyr / (double)nphoton_total)
                                   : h->src_flx * mu0 * (double)h->nx * (double)h->ny / (double)nphoton_total; // as mi3d_get_*
        const tally_t *tally = w == 0 ? h->rad_ptr() : h->flux_ptr();
        const double *dir_dev = nullptr;
        if (w == 1 && !h->dir_level.empty()) {
            std::vector<double> a(h->dir_level);
            for (double &x : a) x *= h->src_flx * mu0 * h->analytic_share;
            if ((rc = h->d_dir_level.upload(a.data(), a.size()))) return rc;
            dir_dev = h->d_dir_level.p;
        }
        hipLaunchKernelGGL(k_stats_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, tally, h->run_ptr(w),
                           h->d_factor[w].p, norm, plane, nlevel, w == 0 ? -1 : (int)(n / 3), dir_dev, (int)n);
        HIPCHK(hipGetLastError());
    }
    return stats_mark(h);
}

int mi3d_stats_end_run(mi3d_solver *h, float *rad_run_out, float *flux_run_out) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!h->stats_on) return fail(MI3D_ESTATE, "mi3d_stats_begin has not been called");
    if (h->stats_joined) return fail(MI3D_ESTATE, "this handle adds into another handle's run (mi3d_stats_join): close the run on that one");
    for (int w = 0; w < 2; ++w) {
        if (!(h->target & (w == 0 ? MI3D_TARGET_RADIANCE : MI3D_TARGET_FLUX))) continue;
        const size_t n = h->stat_elems(w);
        if (n == 0) continue;
        float *out = w == 0 ? rad_run_out : flux_run_out;
        if ((rc = ev_settle(h))) return rc;   // (nothing busy after mi3d_stats_add; a handle that has only joined another's run may still have launches on their way)
        if (out) {
            HIPCHK(sync_main(h));
            HIPCHK(hipMemcpy(out, h->run_ptr(w), n * sizeof(float), hipMemcpyDeviceToHost));
        }
        hipLaunchKernelGGL(k_stats_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->run_ptr(w),
                           h->d_sum[w].p, h->d_sumsq[w].p, (int)n);
        HIPCHK(hipGetLastError());
    }
    h->stats_nrun++;
    return stats_mark(h);
}

// Two handles on one device may gather the jobs of ONE run between them (each transports every other job, so that the tail of
// one job's launch runs beside the next job's): the second one joins the first one's run fields (mi3d_stats_join), and every mi3d_stats_add / mi3d_stats_end_run is preceded by mi3d_stats_chain(h,
// other), which makes h's next statistics kernel wait for the other's last one: the run field is then summed in job order,
// exactly as one handle would have summed it.
int mi3d_stats_join(mi3d_solver *h, mi3d_solver *owner) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!owner || owner == h || !owner->stats_on || owner->stats_joined) return fail(MI3D_ESTATE, "mi3d_stats_join: the owner has not begun statistics of its own");
    if (owner->device != h->device) return fail(MI3D_EINVAL, "mi3d_stats_join: the two handles are on different devices");
    if ((rc = mi3d_prepare(h))) return rc;
    if (h->target != owner->target) return fail(MI3D_EINVAL, "mi3d_stats_join: the two handles have different targets");
    for (int w = 0; w < 2; ++w) {
        h->run_ext[w] = nullptr;
        if (!(h->target & (w == 0 ? MI3D_TARGET_RADIANCE : MI3D_TARGET_FLUX))) continue;
        if (h->stat_elems(w) != owner->stat_elems(w)) return fail(MI3D_EINVAL, "mi3d_stats_join: the two scenes have different result shapes");
        h->run_ext[w] = owner->run_ptr(w);
    }
    h->stats_on = true;
    h->stats_joined = true;
    h->stats_nrun = 0;
    return MI3D_OK;
}

int mi3d_stats_chain(mi3d_solver *h, mi3d_solver *after) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!after || after == h) return MI3D_OK;
    if (after->device != h->device) return fail(MI3D_EINVAL, "mi3d_stats_chain: the two handles are on different devices");
    if (after->stats_ev) HIPCHK(hipStreamWaitEvent(h->stream, after->stats_ev, 0));
    return MI3D_OK;
}

int mi3d_stats_get(mi3d_solver *h, int which, float *mean, float *sdev, int *nrun) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!h->stats_on) return fail(MI3D_ESTATE, "mi3d_stats_begin has not been called");
    if (which != MI3D_TARGET_RADIANCE && which != MI3D_TARGET_FLUX) return fail(MI3D_EINVAL, "which=%d", which);
    if (nrun) *nrun = h->stats_nrun;
    if (!mean && !sdev) return MI3D_OK;
    if (!(h->target & which)) return fail(MI3D_ESTATE, "no statistics of that kind were gathered");
    if (h->stats_nrun == 0) return fail(MI3D_ESTATE, "no run has been closed (mi3d_stats_end_run)");
    const int w = which == MI3D_TARGET_RADIANCE ? 0 : 1;
    const size_t n = h->stat_elems(w);
    if ((rc = h->d_stat_out.alloc(2 * n))) return rc;
    hipLaunchKernelGGL(k_stats_final, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_sum[w].p,
                       h->d_sumsq[w].p, h->d_stat_out.p, h->d_stat_out.p + n, 1.0 / (double)h->stats_nrun, (int)n);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_main(h));
    if ((rc = ev_settle(h))) return rc;   // (a run whose event lists ran full has gone into these statistics: the call fails)
    if (mean) HIPCHK(hipMemcpy(mean, h->d_stat_out.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (sdev) HIPCHK(hipMemcpy(sdev, h->d_stat_out.p + n, n * sizeof(float), hipMemcpyDeviceToHost));
    return MI3D_OK;
}

int mi3d_debug_philox(mi3d_solver *h, uint64_t seed, uint64_t id0, uint32_t draw, int n, uint32_t *out) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (n <= 0 || !out) return fail(MI3D_EINVAL, "bad arguments to mi3d_debug_philox");
    DevBuf<uint32_t> d;
    if ((rc = d.alloc((size_t)4 * n))) return rc;
    hipLaunchKernelGGL(k_philox, dim3((n + 255) / 256), dim3(256), 0, h->stream, seed, id0, draw, n, d.p);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_main(h));
    HIPCHK(hipMemcpy(out, d.p, (size_t)4 * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    d.release();
    return MI3D_OK;
}

int mi3d_debug_order(mi3d_solver *h, uint64_t n, uint32_t *order_out, uint32_t *tile_end_out, int ntile_max) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (n == 0 || !order_out) return fail(MI3D_EINVAL, "bad arguments to mi3d_debug_order");
    if (!h->d_order.p || h->d_order.cap < n) return fail(MI3D_ESTATE, "no photon order of that length (the last launch ran in id order, or was shorter)");
    HIPCHK(sync_main(h));
    HIPCHK(hipMemcpy(order_out, h->pre_last ? h->d_order2.p : h->d_order.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (tile_end_out && ntile_max > 0)
        HIPCHK(hipMemcpy(tile_end_out, h->pre_last ? h->d_cursor2.p : h->d_cursor.p, (size_t)std::min(ntile_max, kMaxTiles) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MI3D_OK;
}

} // extern "C"
