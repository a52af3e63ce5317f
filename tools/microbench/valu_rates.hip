// Issue cost of wave64 vector instructions on gfx950 as a function of the waves resident per SIMD, in REAL shader
// cycles (s_memtime deltas of the timed loop, not wall time x a nominal clock), next to the clock the chip held
// (s_memtime / s_memrealtime, 100 MHz reference).  Settles what one wave-instruction costs a SIMD for the
// instruction mix of k_transport: v_fma_f32, the v_mul_lo/hi_u32 Philox is made of, v_xor_b32, transcendentals,
// and the packed v_pk_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o tools/microbench/valu_rates && tools/microbench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, unsigned long long *clk, int iters) {
    uint32_t a[8];
    float f[8];
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + 1u; f[i] = (float)a[i] * 1e-9f;
        p[i] = float2v{f[i], f[i] * 0.5f};
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // eight independent chains: issue-bound, not dependency-bound
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 4) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i]));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(0x1234567u));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i] + (uint32_t)p[i].x + (uint32_t)p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0;
    }
}

template <int OP>
int run(const char *name, uint32_t *d, unsigned long long *dclk, int ncu) {
    const int iters = 20000;
    for (int wps : {1, 2, 4, 5, 8}) {                 // 256-thread blocks per CU = waves per SIMD
        const int blocks = ncu * wps, nw = blocks * 4;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        k<OP><<<blocks, 256>>>(d, dclk, 100);
        CHK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(d, dclk, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(2 * nw);
        CHK(hipMemcpy(c.data(), dclk, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
        std::vector<double> cyc(nw), ghz(nw);
        for (int w = 0; w < nw; ++w) { cyc[w] = (double)c[2 * w]; ghz[w] = (double)c[2 * w] / (double)c[2 * w + 1] * 0.1; }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double per_wave = cyc[nw / 2] / ((double)iters * 8);   // cycles one wave needs per instruction of its own
        printf("%-14s %d waves/SIMD  %8.3f ms  clock %.2f GHz  %.2f cycles per instr per wave  -> %.2f cycles per wave-instr per SIMD\n",
               name, wps, ms, ghz[nw / 2], per_wave, per_wave / wps);
    }
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, ncu);
    uint32_t *d; unsigned long long *dclk;
    CHK(hipMalloc(&d, (size_t)ncu * 8 * 256 * 4)); CHK(hipMalloc(&dclk, (size_t)ncu * 8 * 4 * 2 * 8));
    if (run<0>("v_fma_f32", d, dclk, ncu)) return 1;
    if (run<6>("v_add_f32", d, dclk, ncu)) return 1;
    if (run<5>("v_pk_fma_f32", d, dclk, ncu)) return 1;
    if (run<1>("v_mul_lo_u32", d, dclk, ncu)) return 1;
    if (run<2>("v_mul_hi_u32", d, dclk, ncu)) return 1;
    if (run<3>("v_xor_b32", d, dclk, ncu)) return 1;
    if (run<7>("v_cndmask_b32", d, dclk, ncu)) return 1;
    if (run<4>("v_log_f32", d, dclk, ncu)) return 1;
    return 0;
}
