// LDS atomics: lane-operations per clock and CU by type, by whether the old value is returned, and by how many distinct
// addresses the 64 lanes of an instruction hit.  Question behind it: the tally-record sort and sum of the flux path
// (er3t_amd/csrc/mi3d_kernel_flux.hip) spend one LDS atomic per record and pass -- what does one cost?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench/lds_atomic_rates.hip -o tools/microbench/lds_atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// OP 0: u32 add no return, 1: u32 add returning, 2: f64 add no return, 3: f32 add no return, 4: plain ds_write_b64 (reference)
template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned *out, int niter, unsigned distinct, unsigned region) {
    extern __shared__ double lds[];
    for (unsigned i = threadIdx.x; i < region; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const unsigned lane = threadIdx.x & 63u;
    for (int i = 0; i < niter; ++i) {
        s = s * 1664525u + 1013904223u;
        // `distinct` addresses per instruction: lanes l and l + distinct share one; the group of addresses moves at random
        const unsigned wave_rand = __builtin_amdgcn_readfirstlane(s);
        const unsigned idx = distinct >= 64u ? (s >> 8) % region : ((wave_rand >> 8) + (lane % distinct) * 97u) % region;
        if (OP == 0) atomicAdd(reinterpret_cast<unsigned *>(lds) + idx, 1u);
        if (OP == 1) acc += atomicAdd(reinterpret_cast<unsigned *>(lds) + idx, 1u);
        if (OP == 2) atomicAdd(lds + idx, 1.0);
        if (OP == 3) atomicAdd(reinterpret_cast<float *>(lds) + idx, 1.0f);
        if (OP == 4) reinterpret_cast<unsigned long long *>(lds)[idx] = s;
    }
    __syncthreads();
    unsigned long long sum = 0;
    for (unsigned i = threadIdx.x; i < region; i += blockDim.x) sum += reinterpret_cast<unsigned *>(lds)[2 * i] + (unsigned long long)(lds[i] != 0.0);
    if (sum == 0x12345ull || acc == 0x7777777u) out[0] = 1;
}

template <int OP>
int run(const char *name, unsigned *d, unsigned distinct, int ncu, double clk_ghz) {
    const int niter = 4000, blocks = ncu;
    const unsigned region = 8192;
    k<OP><<<blocks, 1024, region * 8>>>(d, 10, distinct, region);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0)); k<OP><<<blocks, 1024, region * 8>>>(d, niter, distinct, region); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 1024 * niter;
    printf("%-22s %2u distinct addresses per instruction  %8.3f ms  %.3g lane-ops/s chip-wide  %.2f per clock and CU (at %.1f GHz)\n", name, distinct, ms,
           ops / (ms * 1e-3), ops / (ms * 1e-3) / ncu / (clk_ghz * 1e9), clk_ghz);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    int lds_max = 0; CHK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, 0));
    printf("%s: %d CUs, %.2f GHz, LDS per workgroup %d bytes; one workgroup of 1024 threads per CU\n", p.name, ncu, ghz, lds_max);
    unsigned *d; CHK(hipMalloc(&d, 64));
    for (unsigned distinct : {64u, 16u, 4u, 1u}) {
        if (run<0>("u32 add", d, distinct, ncu, ghz)) return 1;
        if (run<1>("u32 add returning", d, distinct, ncu, ghz)) return 1;
        if (run<2>("f64 add", d, distinct, ncu, ghz)) return 1;
        if (run<3>("f32 add", d, distinct, ncu, ghz)) return 1;
        if (run<4>("ds_write_b64 (plain)", d, distinct, ncu, ghz)) return 1;
    }
    return 0;
}
