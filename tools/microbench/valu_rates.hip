// Issue cost of the integer multiplies Philox is made of, against v_fma_f32, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o tools/microbench/valu_rates && tools/microbench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters) {
    uint32_t a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + 1u; f[i] = (float)a[i] * 1e-9f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 3) { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a[i]), "v"(0xD2511F53u) : "vcc"); a[i] = (uint32_t)r ^ (uint32_t)(r >> 32); }
            if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x511F53u));
            if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
            if (OP == 6) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, uint32_t *d, int extra) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 100);
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 /*waves*/ * iters * 8 * (1 + extra);
    // 1024 SIMDs at ~2.4 GHz
    printf("%-16s %8.3f ms  %.3g wave-instr/s  -> %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, ms, instr / (ms * 1e-3), 1024.0 * 2.4e9 / (instr / (ms * 1e-3)));
}

int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d, 0); run<1>("v_mul_lo_u32", d, 0); run<2>("v_mul_hi_u32", d, 0); run<3>("v_mad_u64_u32(+xor)", d, 0);
    run<4>("v_mul_u32_u24", d, 0); run<5>("v_xor_b32", d, 0); run<6>("v_log_f32", d, 0);
    return 0;
}
