// What does a wave64 vector instruction cost a SIMD of gfx950 -- by WALL time, at a known number of resident waves, for the opcode mix
// of the photon loop?  (profiles/r02/valu_rates.log prices v_xor_b32 at 1.65-2.2 cycles per SIMD from in-kernel
// clocks, valu_rates2.log the same instruction at 4.04 by wall time; bench.py's issue_frac came out above 1.)
//
// What this tool does differently from the two round-2 tools:
//   * occupancy is FORCED, not hoped for: every 256-thread workgroup asks for 160 KiB / W of LDS, so exactly W fit a CU (one wave of
//     each per SIMD: W waves per SIMD), and the grid holds 8 x CUs x W workgroups: the chip is in steady state for seven eighths of
//     the run and the wall time of the launch prices the instructions;
//   * the clock is measured (s_memtime against the 100 MHz s_memrealtime, median over the waves), and so is the residency (every wave
//     notes when it began and ended: waves alive at the launch's mid time / SIMDs);
//   * three figures side by side per case: cycles per wave-instruction per SIMD by wall time, the same from the median wave's own
//     clock / W (what valu_rates.hip printed), and wave-instructions per second for the whole chip;
//   * `mix`: a loop whose instruction classes come from the command line in the proportions measured on the real kernel
//     (SQ_INSTS_VALU_* counters, tools/mix_rates.sh), all lanes active, no memory access: the ceiling of THAT mix.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mix_rates.hip -o tools/microbench/mix_rates
//   tools/microbench/mix_rates ops                  one opcode at a time, W = 4, 6, 8
//   tools/microbench/mix_rates mix fma=30 mul=25 add=12 trans=15 int=40 mullo=8 mad64=20 cmp=12 cnd=15 pk=4 mov=14 [iters]
// Under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU the same runs calibrate what the hardware
// counters read on a loop that does nothing but issue (bench.py: `sq_active_inst_valu`).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { OP_FMA, OP_MUL, OP_ADD, OP_XOR, OP_BITOP3, OP_CND64, OP_CMP64, OP_CMPCND, OP_MULLO, OP_MAD64, OP_MIN3, OP_MED3, OP_RCP, OP_RSQ, OP_EXP, OP_LOG, OP_SIN,
       OP_PKFMA, OP_PKMUL, OP_CVT, OP_ALIGN, OP_MOV, OP_FMA_S, OP_LSHLADD,
       OP_MUL_S, OP_ADD_K, OP_ADD_LIT, OP_CND32, OP_CMP32, OP_AND, OP_LSHL, OP_ADDU, OP_ADDU_S, OP_MAX32, OP_MAX64ABS, OP_BITOP3V, OP_MUL24, OP_MAD24, OP_ADD3, OP_FMAC,
       OP_PKADD, OP_CVTI, OP_FLOOR, OP_READLANE, OP_MOV_S, OP_SUB, OP_MULHI, OP_N };
static const char *kOpName[OP_N] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_xor_b32", "v_bitop3_b32 (sgpr)", "v_cndmask_b32_e64 (sgpr mask)", "v_cmp_gt_f32_e64 -> sgpr",
                                   "v_cmp -> vcc + v_cndmask (pair)", "v_mul_lo_u32", "v_mad_u64_u32", "v_min3_f32", "v_med3_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32",
                                   "v_log_f32", "v_sin_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cvt_f32_u32", "v_alignbit_b32", "v_mov_b32", "v_fma_f32 (sgpr operand)",
                                   "v_lshl_add_u32", "v_mul_f32_e32 v,s,v", "v_add_f32_e32 v,1.0,v", "v_add_f32_e32 v,literal,v", "v_cndmask_b32_e32 (vcc)", "v_cmp_gt_f32_e32 -> vcc",
                                   "v_and_b32_e32", "v_lshlrev_b32_e32 v,3,v", "v_add_u32_e32", "v_add_u32_e32 v,s,v", "v_max_f32_e32", "v_max_f32_e64 |v|", "v_bitop3_b32 v,v,v",
                                   "v_mul_u32_u24_e32", "v_mad_u32_u24", "v_add3_u32", "v_fmac_f32_e32", "v_pk_add_f32", "v_cvt_i32_f32", "v_floor_f32", "v_readlane_b32",
                                   "v_mov_b32 v,s", "v_sub_f32_e32", "v_mul_hi_u32"};

struct Mix { int n[16]; };   // groups of 8 instructions per class and iteration: fma mul add trans int mullo mad64 cmp cnd pk mov

typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__device__ __forceinline__ void op8(uint32_t (&a)[8], float (&f)[8], unsigned long long (&q)[8], float2v (&p)[8], float sval, unsigned long long smask) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"((uint32_t)smask));
        if (OP == OP_CND64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(smask));
        if (OP == OP_CMP64) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(q[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
        if (OP == OP_CMPCND) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc");
        if (OP == OP_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
        if (OP == OP_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        if (OP == OP_RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
        if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
        if (OP == OP_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
        if (OP == OP_SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(f[i]));
        if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == OP_CVT) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        if (OP == OP_ALIGN) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_FMA_S) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(f[i]) : "s"(sval));
        if (OP == OP_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_MUL_S) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(f[i]) : "s"(sval));
        if (OP == OP_ADD_K) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(f[i]));
        if (OP == OP_ADD_LIT) asm volatile("v_add_f32_e32 %0, 0x3f9d70a4, %0" : "+v"(f[i]));
        if (OP == OP_CND32) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_CMP32) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc");
        if (OP == OP_AND) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_LSHL) asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(a[i]));
        if (OP == OP_ADDU) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_ADDU_S) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)smask));
        if (OP == OP_MAX32) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == OP_MAX64ABS) asm volatile("v_max_f32_e64 %0, |%0|, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == OP_BITOP3V) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
        if (OP == OP_MUL24) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
        if (OP == OP_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == OP_CVTI) asm volatile("v_cvt_i32_f32_e32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
        if (OP == OP_FLOOR) asm volatile("v_floor_f32_e32 %0, %0" : "+v"(f[i]));
        if (OP == OP_READLANE) { uint32_t rl_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(rl_) : "v"(a[i])); q[i] = rl_; }
        if (OP == OP_MOV_S) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "s"((uint32_t)smask));
        if (OP == OP_SUB) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == OP_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
    }
}

struct Stamp { unsigned long long c0, c1, r0, r1; };

#define PROLOGUE                                                                                                         \
    extern __shared__ float lds_[];                                                                                      \
    uint32_t a[8]; float f[8]; unsigned long long q[8]; float2v p[8];                                                    \
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + 1u; f[i] = 1.0f + (float)(a[i] >> 9) * 1e-8f; q[i] = a[i]; p[i] = float2v{f[i], 0.5f * f[i]}; } \
    if (iters < 0) lds_[threadIdx.x] = f[0];                                                                             \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define EPILOGUE                                                                                                         \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                   \
    uint32_t s = 0;                                                                                                      \
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i] + (uint32_t)q[i] + (uint32_t)p[i].x + (uint32_t)p[i].y;       \
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;                                                              \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};

template <int OP>
__global__ void __launch_bounds__(256) k_op(uint32_t *out, Stamp *st, int iters, float sval, unsigned long long smask) {
    PROLOGUE
    for (int it = 0; it < iters; ++it) op8<OP>(a, f, q, p, sval, smask);
    EPILOGUE
}

__global__ void __launch_bounds__(256) k_mix(uint32_t *out, Stamp *st, int iters, float sval, unsigned long long smask, const Mix M) {
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        // (the classes in turn, eight independent instructions at a time; between waves the classes interleave as they do in the loop)
        for (int r = 0; r < M.n[0]; ++r) op8<OP_FMA>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[1]; ++r) op8<OP_MUL>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[2]; ++r) op8<OP_ADD>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[3]; ++r) { op8<OP_RCP>(a, f, q, p, sval, smask); for (int i = 0; i < 8; ++i) f[i] = f[i] * 0.5f + 1.0f; }   // (kept finite)
        for (int r = 0; r < M.n[4]; ++r) op8<OP_XOR>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[5]; ++r) op8<OP_MULLO>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[6]; ++r) op8<OP_MAD64>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[7]; ++r) op8<OP_CMP64>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[8]; ++r) op8<OP_CND64>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[9]; ++r) op8<OP_PKFMA>(a, f, q, p, sval, smask);
        for (int r = 0; r < M.n[10]; ++r) op8<OP_MOV>(a, f, q, p, sval, smask);
    }
    EPILOGUE
}

struct Res { double ms, ghz, cyc_wall, cyc_wave, rate, resident; };

static int analyse(const std::vector<Stamp> &st, int nw, float ms, double winstr_total, int ncu, int W, Res &R) {
    std::vector<double> ghz(nw), cyc(nw);
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int w = 0; w < nw; ++w) {
        ghz[w] = (double)(st[w].c1 - st[w].c0) / (double)(st[w].r1 - st[w].r0) * 0.1; cyc[w] = (double)(st[w].c1 - st[w].c0);
        rmin = std::min(rmin, st[w].r0); rmax = std::max(rmax, st[w].r1);
    }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const unsigned long long mid = rmin + (rmax - rmin) / 2;
    long alive = 0;
    for (int w = 0; w < nw; ++w) alive += (st[w].r0 <= mid && st[w].r1 > mid) ? 1 : 0;
    R.ms = ms; R.ghz = ghz[nw / 2];
    const double winstr_wave = winstr_total / nw;
    R.cyc_wall = (double)ms * 1e-3 * R.ghz * 1e9 * (ncu * 4.0) / winstr_total;
    R.cyc_wave = cyc[nw / 2] / winstr_wave / W;
    R.rate = winstr_total / ((double)ms * 1e-3);
    R.resident = (double)alive / (ncu * 4.0);
    return 0;
}

template <int OP>
static int run_op(uint32_t *d, Stamp *dst, int ncu, int ninstr) {
    for (int W : {4, 6, 8}) {
        const int blocks = 8 * ncu * W, nw = blocks * 4;
        const size_t lds = (size_t)(160 * 1024 / W) / 1024 * 1024 - (W == 8 ? 512 : 0);
        CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_op<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int iters = 10000;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), lds, 0, d, dst, 50, 1.5f, 0x5555aaaa5555aaaaull);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), lds, 0, d, dst, iters, 1.5f, 0x5555aaaa5555aaaaull);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> st(nw);
        CHK(hipMemcpy(st.data(), dst, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
        Res R; analyse(st, nw, ms, (double)nw * iters * 8.0 * ninstr, ncu, W, R);
        printf("%-34s W=%d  %7.3f ms  clock %.2f GHz  resident %.2f waves/SIMD at mid-run  by WALL %5.2f cycles per wave-instr per SIMD  by the median wave's clock / W %5.2f  chip %.3e wave-instr/s\n",
               kOpName[OP], W, R.ms, R.ghz, R.resident, R.cyc_wall, R.cyc_wave, R.rate);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return 0;
}

int main(int argc, char **argv) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, ncu);
    uint32_t *d; Stamp *dst;
    CHK(hipMalloc(&d, (size_t)8 * ncu * 8 * 256 * 4)); CHK(hipMalloc(&dst, (size_t)8 * ncu * 8 * 4 * sizeof(Stamp)));
    const std::string mode = argc > 1 ? argv[1] : "ops";
    if (mode == "ops2") {     // (the operand kinds and encodings the first table left open)
        if (run_op<OP_MUL_S>(d, dst, ncu, 1) || run_op<OP_ADD_K>(d, dst, ncu, 1) || run_op<OP_ADD_LIT>(d, dst, ncu, 1) || run_op<OP_SUB>(d, dst, ncu, 1) || run_op<OP_FMAC>(d, dst, ncu, 1) ||
            run_op<OP_MAX32>(d, dst, ncu, 1) || run_op<OP_MAX64ABS>(d, dst, ncu, 1) || run_op<OP_CND32>(d, dst, ncu, 1) || run_op<OP_CMP32>(d, dst, ncu, 1) || run_op<OP_AND>(d, dst, ncu, 1) ||
            run_op<OP_LSHL>(d, dst, ncu, 1) || run_op<OP_ADDU>(d, dst, ncu, 1) || run_op<OP_ADDU_S>(d, dst, ncu, 1) || run_op<OP_BITOP3V>(d, dst, ncu, 1) || run_op<OP_MUL24>(d, dst, ncu, 1) ||
            run_op<OP_MAD24>(d, dst, ncu, 1) || run_op<OP_ADD3>(d, dst, ncu, 1) || run_op<OP_MULHI>(d, dst, ncu, 1) || run_op<OP_PKADD>(d, dst, ncu, 1) || run_op<OP_CVTI>(d, dst, ncu, 1) ||
            run_op<OP_FLOOR>(d, dst, ncu, 1) || run_op<OP_READLANE>(d, dst, ncu, 1) || run_op<OP_MOV_S>(d, dst, ncu, 1))
            return 1;
        return 0;
    }
    if (mode == "ops") {
        if (run_op<OP_FMA>(d, dst, ncu, 1) || run_op<OP_FMA_S>(d, dst, ncu, 1) || run_op<OP_MUL>(d, dst, ncu, 1) || run_op<OP_ADD>(d, dst, ncu, 1) || run_op<OP_XOR>(d, dst, ncu, 1) ||
            run_op<OP_BITOP3>(d, dst, ncu, 1) || run_op<OP_MOV>(d, dst, ncu, 1) || run_op<OP_LSHLADD>(d, dst, ncu, 1) || run_op<OP_ALIGN>(d, dst, ncu, 1) || run_op<OP_CVT>(d, dst, ncu, 1) ||
            run_op<OP_CND64>(d, dst, ncu, 1) || run_op<OP_CMP64>(d, dst, ncu, 1) || run_op<OP_CMPCND>(d, dst, ncu, 2) || run_op<OP_MIN3>(d, dst, ncu, 1) || run_op<OP_MED3>(d, dst, ncu, 1) ||
            run_op<OP_MULLO>(d, dst, ncu, 1) || run_op<OP_MAD64>(d, dst, ncu, 1) || run_op<OP_PKFMA>(d, dst, ncu, 1) || run_op<OP_PKMUL>(d, dst, ncu, 1) ||
            run_op<OP_RCP>(d, dst, ncu, 1) || run_op<OP_RSQ>(d, dst, ncu, 1) || run_op<OP_EXP>(d, dst, ncu, 1) || run_op<OP_LOG>(d, dst, ncu, 1) || run_op<OP_SIN>(d, dst, ncu, 1))
            return 1;
        return 0;
    }
    // mix: class=count pairs (instructions of that class per `unit` of the real loop, e.g. per photon); scaled to groups of 8
    static const char *cls[11] = {"fma", "mul", "add", "trans", "int", "mullo", "mad64", "cmp", "cnd", "pk", "mov"};
    double cnt[11] = {0}; int iters = 0;
    for (int i = 2; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { iters = atoi(argv[i]); continue; }
        for (int c = 0; c < 11; ++c) if (strncmp(argv[i], cls[c], eq - argv[i]) == 0 && strlen(cls[c]) == (size_t)(eq - argv[i])) cnt[c] = atof(eq + 1);
    }
    double tot = 0; for (double v : cnt) tot += v;
    if (!(tot > 0)) { printf("mix: no class counts given\n"); return 1; }
    Mix M; memset(&M, 0, sizeof(M));
    int groups = 0;
    for (int c = 0; c < 11; ++c) { M.n[c] = (int)(cnt[c] / tot * 64.0 + 0.5); groups += M.n[c]; }   // ~64 groups of 8 per iteration
    printf("mix (groups of 8 per iteration):"); for (int c = 0; c < 11; ++c) printf(" %s=%d", cls[c], M.n[c]); printf("\n");
    // (trans: v_rcp_f32 + the v_fma_f32 that keeps its argument finite: counted as 2 instructions per pair)
    const double per_iter = 8.0 * (groups + M.n[3]);
    if (iters <= 0) iters = 40;
    for (int W : {4, 5, 6, 7, 8}) {
        const int blocks = 8 * ncu * W, nw = blocks * 4;
        const size_t lds = (size_t)(160 * 1024 / W) / 1024 * 1024 - (W == 8 ? 512 : 0);
        CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mix), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), lds, 0, d, dst, 2, 1.5f, 0x5555aaaa5555aaaaull, M);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), lds, 0, d, dst, iters, 1.5f, 0x5555aaaa5555aaaaull, M);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> st(nw);
        CHK(hipMemcpy(st.data(), dst, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
        Res R; analyse(st, nw, ms, (double)nw * iters * per_iter, ncu, W, R);
        printf("mix                                W=%d  %7.3f ms  clock %.2f GHz  resident %.2f waves/SIMD at mid-run  by WALL %5.2f cycles per wave-instr per SIMD  by the median wave's clock / W %5.2f  chip %.3e wave-instr/s\n",
               W, R.ms, R.ghz, R.resident, R.cyc_wall, R.cyc_wave, R.rate);
    }
    return 0;
}
