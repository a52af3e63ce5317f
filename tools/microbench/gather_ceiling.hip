// gather_ceiling.hip — what the chip sustains for the access pattern of the transport kernel: every lane follows a
// chain of DEPENDENT random 16-byte reads in a table far larger than L2 (the next index comes out of the record just
// read), a few VALU instructions between reads, W waves per SIMD.  Prints reads/s for several occupancies.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_ceiling.hip -o gpurun_out/gather_ceiling && gpurun_out/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VGPR_PAD>
__global__ void __launch_bounds__(256) k_chase(const uint4 *tab, unsigned mask, int nstep, unsigned *sink) {
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
    unsigned acc = 0;
    for (int i = 0; i < nstep; ++i) {
        const uint4 r = tab[idx];
        acc += r.y;
        idx = (r.x ^ (acc * 40503u) ^ (unsigned)i) & mask; // next voxel depends on what was read
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const unsigned n = 1u << 24;                       // 16 M records x 16 B = 256 MiB (bench scene: 23 M x 16 B)
    std::vector<uint4> h(n);
    unsigned s = 12345u;
    for (unsigned i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = make_uint4(s >> 4, s, 0, 0); }
    uint4 *d; unsigned *sink;
    CHK(hipMalloc(&d, (size_t)n * sizeof(uint4))); CHK(hipMalloc(&sink, 4));
    CHK(hipMemcpy(d, h.data(), (size_t)n * sizeof(uint4), hipMemcpyHostToDevice));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int nstep = 2000;
    for (int wps = 1; wps <= 8; ++wps) {               // waves per SIMD = 256-thread blocks per CU
        const int grid = p.multiProcessorCount * wps;
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_chase<0>, dim3(grid), dim3(256), 0, 0, d, n - 1, 100, sink);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(k_chase<0>, dim3(grid), dim3(256), 0, 0, d, n - 1, nstep, sink);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        const double reads = (double)grid * 256 * nstep;
        printf("%d waves/SIMD: %.3g dependent 16-B gathers/s  (%.0f ns per gather per lane)\n", wps, reads / (ms * 1e-3),
               ms * 1e6 / nstep);
    }
    return 0;
}
