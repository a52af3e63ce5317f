// Second issue-rate table (round 2): what the operand KIND costs.  v_cndmask_b32 reading VCC measured 12-19 cycles
// per wave-instruction in valu_rates.hip where v_xor_b32 measured 2: this file separates the candidates -- a VCC / SGPR
// mask operand, an SGPR data operand, a compare that writes VCC, the 64-bit multiply-add Philox could use -- and adds a
// residency census (how many waves of a plain 256-thread kernel are co-resident per SIMD, and on which XCD a block runs).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates2.hip -o tools/microbench/valu_rates2 && tools/microbench/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <map>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, unsigned long long *clk, int iters, float sval, unsigned long long smask) {
    uint32_t a[8];
    float f[8];
    unsigned long long q[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + 1u; f[i] = (float)a[i] * 1e-9f; q[i] = a[i];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)smask));                 // SGPR data operand
            if (OP == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "s"(sval));                  // SGPR data operand
            if (OP == 3) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(f[i]));                              // inline constant
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(0x1234567u));    // mask in VCC
            if (OP == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(0x1234567u), "s"(smask)); // mask in an SGPR pair
            if (OP == 6) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc");  // compare -> VCC
            if (OP == 7) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc"); // the pair (2 instr)
            if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(0xD2511F53u) : "vcc");
            if (OP == 9) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
            if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (OP == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (OP == 12) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(0x1234u));
            if (OP == 14) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(f[i]) : "s"(sval));             // SGPR operand in an FMA
            if (OP == 15) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(q[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));   // compare -> SGPR pair
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i] + (uint32_t)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
int run(const char *name, int ninstr, uint32_t *d, unsigned long long *dclk, int ncu) {
    const int iters = 10000;
    for (int wps : {1, 2, 4}) {
        const int blocks = ncu * wps, nw = blocks * 4;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        k<OP><<<blocks, 256>>>(d, dclk, 100, 1.5f, 0x5555aaaa5555aaaaull);
        CHK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(d, dclk, iters, 1.5f, 0x5555aaaa5555aaaaull); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(nw);
        CHK(hipMemcpy(c.data(), dclk, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end());
        const double per_wave = (double)c[nw / 2] / ((double)iters * 8 * ninstr);
        printf("%-34s %d waves/SIMD  %8.3f ms  %6.2f cycles per instr per wave (in-kernel clock)  -> %5.2f per SIMD;  by wall time at 2.3 GHz: %5.2f per SIMD\n",
               name, wps, ms, per_wave, per_wave / wps, ms * 1e-3 * 2.3e9 / ((double)iters * 8 * ninstr * wps));
    }
    return 0;
}

// residency census: every wave records where it ran and when (100 MHz real-time clock), then spins ~100 us so that all co-resident waves overlap
__global__ void __launch_bounds__(256) k_census(unsigned *info, unsigned long long *t) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);        // HW_REG_XCC_ID bits [3:0]
    while (__builtin_amdgcn_s_memrealtime() - r0 < 10000ull) __builtin_amdgcn_s_sleep(8);
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        info[2 * w] = hwid; info[2 * w + 1] = xcc; t[2 * w] = r0; t[2 * w + 1] = r1;
    }
}

int census(int ncu, int bpc) {
    const int blocks = ncu * bpc, nw = blocks * 4;
    unsigned *dinfo; unsigned long long *dt;
    CHK(hipMalloc(&dinfo, nw * 8)); CHK(hipMalloc(&dt, nw * 16));
    k_census<<<blocks, 256>>>(dinfo, dt);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned> info(2 * nw); std::vector<unsigned long long> t(2 * nw);
    CHK(hipMemcpy(info.data(), dinfo, nw * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(t.data(), dt, nw * 16, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < nw; ++w) { tmin = std::min(tmin, t[2 * w]); tmax = std::max(tmax, t[2 * w + 1]); }
    // waves that started within the first 20 us ran together
    std::map<unsigned, int> per_simd; int early = 0; std::map<unsigned, int> per_xcc; int xcc_rr_ok = 0;
    for (int w = 0; w < nw; ++w) {
        const unsigned hw = info[2 * w], xcc = info[2 * w + 1];
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (CDNA: se_id wider)
        const unsigned key = (xcc << 24) | (hw & 0xfffffff0u & 0x00ffffffu);
        if (t[2 * w] - tmin < 2000ull) { early++; per_simd[key]++; }
        per_xcc[xcc]++;
        if (xcc == (unsigned)((w / 4) % 8 + info[1]) % 8) xcc_rr_ok++;
    }
    int mx = 0; std::map<int, int> hist;
    for (auto &p : per_simd) { mx = std::max(mx, p.second); hist[p.second]++; }
    printf("census %d blocks/CU requested: %d waves, %d started within 20 us of the first, kernel span %.1f us; distinct (xcc,se,sh,cu,simd) keys %zu, max waves per key %d; histogram of waves per key:",
           bpc, nw, early, (tmax - tmin) * 0.01, per_simd.size(), mx);
    for (auto &h : hist) printf(" %d:%d", h.first, h.second);
    printf("\n   waves per XCC_ID:");
    for (auto &p : per_xcc) printf(" %u:%d", p.first, p.second);
    printf("   blocks with xcc == (block + xcc(block 0)) %% 8: %d of %d waves; hw_id of wave 0 = 0x%08x\n", xcc_rr_ok, nw, info[0]);
    (void)hipFree(dinfo); (void)hipFree(dt);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, ncu);
    uint32_t *d; unsigned long long *dclk;
    CHK(hipMalloc(&d, (size_t)ncu * 8 * 256 * 4)); CHK(hipMalloc(&dclk, (size_t)ncu * 8 * 4 * 8));
    if (run<0>("v_xor_b32 v,v,v", 1, d, dclk, ncu)) return 1;
    if (run<1>("v_xor_b32 v,s,v", 1, d, dclk, ncu)) return 1;
    if (run<2>("v_add_f32 v,s,v", 1, d, dclk, ncu)) return 1;
    if (run<3>("v_add_f32 v,1.0,v", 1, d, dclk, ncu)) return 1;
    if (run<14>("v_fma_f32 v,s,v,v", 1, d, dclk, ncu)) return 1;
    if (run<4>("v_cndmask_b32 (vcc)", 1, d, dclk, ncu)) return 1;
    if (run<5>("v_cndmask_b32_e64 (sgpr pair)", 1, d, dclk, ncu)) return 1;
    if (run<6>("v_cmp_gt_f32 -> vcc", 1, d, dclk, ncu)) return 1;
    if (run<15>("v_cmp_gt_f32_e64 -> sgpr pair", 1, d, dclk, ncu)) return 1;
    if (run<7>("v_cmp + v_cndmask (per instr)", 2, d, dclk, ncu)) return 1;
    if (run<8>("v_mad_u64_u32", 1, d, dclk, ncu)) return 1;
    if (run<9>("v_min3_f32", 1, d, dclk, ncu)) return 1;
    if (run<11>("v_max_f32", 1, d, dclk, ncu)) return 1;
    if (run<10>("v_rcp_f32", 1, d, dclk, ncu)) return 1;
    if (run<12>("v_cvt_f32_u32", 1, d, dclk, ncu)) return 1;
    if (run<13>("v_mad_u32_u24", 1, d, dclk, ncu)) return 1;
    for (int bpc : {2, 4, 5, 8}) if (census(ncu, bpc)) return 1;
    return 0;
}
