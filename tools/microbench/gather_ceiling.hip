// gather_ceiling.hip — what the chip sustains for the access pattern of the transport kernel: every lane follows a
// chain of DEPENDENT random 16-byte reads (the next index comes out of the record just read), a few VALU instructions
// between reads, W waves per SIMD.  The table size is swept from "fits every XCD's L2" over "Infinity Cache" to "HBM":
// the rates say what a voxel walk confined to an L2-sized tile can gain over one spread over the whole domain, and
// (run under rocprofv3 --pmc FETCH_SIZE / TCC_HIT_sum TCC_MISS_sum with ONE configuration) calibrate the fabric-traffic
// counter on a known number of 16-byte per-lane gathers.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_ceiling.hip -o tools/microbench/gather_ceiling
//   tools/microbench/gather_ceiling                 # sweep
//   tools/microbench/gather_ceiling 24 5 2000       # one configuration: 2^24 records, 5 waves/SIMD, 2000 steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

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

static double one(const uint4 *d, unsigned n, int ncu, int wps, int nstep, unsigned *sink) {
    const int grid = ncu * wps;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, d, n - 1, 100, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, d, n - 1, nstep, sink);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    const double reads = (double)grid * 256 * nstep;
    printf("table %7.1f MiB  %d waves/SIMD: %.3g dependent 16-B gathers/s  (%.0f ns per gather per lane; %.0f reads, %.4g B in this launch)\n",
           n * 16.0 / 1048576.0, wps, reads / (ms * 1e-3), ms * 1e6 / nstep, reads, reads * 16.0);
    return reads / (ms * 1e-3);
}

int main(int argc, char **argv) {
    const unsigned nmax = 1u << 25;                    // 32 M records x 16 B = 512 MiB (bench scene: 23 M x 16 B, 5.8 M of them cloudy)
    std::vector<uint4> h(nmax);
    unsigned s = 12345u;
    for (unsigned i = 0; i < nmax; ++i) { s = s * 1664525u + 1013904223u; h[i] = make_uint4(s >> 4, s, 0, 0); }
    uint4 *d; unsigned *sink;
    CHK(hipMalloc(&d, (size_t)nmax * sizeof(uint4))); CHK(hipMalloc(&sink, 4));
    CHK(hipMemcpy(d, h.data(), (size_t)nmax * sizeof(uint4), hipMemcpyHostToDevice));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    if (argc >= 4) { one(d, 1u << atoi(argv[1]), ncu, atoi(argv[2]), atoi(argv[3]), sink); return 0; }
    for (int lg : {17, 20, 22, 24, 25})                // 2 MiB (every L2), 16 MiB (the eight L2s together), 64 and 256 MiB (Infinity Cache), 512 MiB (HBM)
        for (int wps : {2, 4, 5, 8}) one(d, 1u << lg, ncu, wps, 2000, sink);
    return 0;
}
