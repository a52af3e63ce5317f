// Does a wave64 vector instruction cost less when a whole quarter (16 lanes) or half of the wave is masked off?  v_fma_f32 chains at six
// waves per SIMD under different EXEC masks, by wall time.  (If empty quarters were skipped, keeping the active lanes of the photon loop
// together would pay; they are not: see profiles/r04/exec_mask_rates.log.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/exec_mask_rates.hip -o tools/microbench/exec_mask_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k(float *out, unsigned long long mask, int iters) {
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(threadIdx.x * 8 + i) * 1e-6f;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, blocks = ncu * 6, iters = 20000;
    float *d; CHK(hipMalloc(&d, sizeof(float) * blocks * 256));
    struct { const char *name; unsigned long long mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lower 32 lanes", 0xffffffffull}, {"upper 32 lanes", 0xffffffff00000000ull},
        {"lower 16 lanes", 0xffffull}, {"lanes 16-31", 0xffff0000ull}, {"even lanes", 0x5555555555555555ull},
        {"every fourth lane", 0x1111111111111111ull}, {"one lane", 1ull}};
    for (auto &c : cases) {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        k<<<blocks, 256>>>(d, c.mask, 100);
        CHK(hipEventRecord(e0)); k<<<blocks, 256>>>(d, c.mask, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double insts = (double)blocks * 4 * iters * 8;               // wave-instructions
        printf("%-20s %8.3f ms   %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", c.name, ms, ms * 1e-3 * 2.4e9 / (insts / (ncu * 4.0)));
    }
    return 0;
}
