// stream_copy.hip -- the HBM rate a plain streaming kernel reaches on this device: float4 copy of a buffer far larger than the
// 256 MiB Infinity Cache (read N bytes + write N bytes per pass).  bench.py runs the binary after its timed region and puts the
// figure next to the 8 TB/s specification as the second denominator of `roofline` ("record the measured stream
// peak as the denominator too").
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_copy.hip -o tools/microbench/stream_copy
//   tools/microbench/stream_copy [GiB per buffer, default 4] [passes, default 20]   ->   one line: "stream_copy <GB/s> GB/s ..."
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float vf4 __attribute__((ext_vector_type(4)));   // (the non-temporal builtins take native vector types)

template <bool NT>
__global__ void __launch_bounds__(256) k_copy(const vf4 *__restrict__ a, vf4 *__restrict__ b, size_t n) {
    // grid-stride, four independent 16-byte loads in flight per thread; NT: non-temporal loads and stores
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        vf4 v0, v1, v2, v3;
        if (NT) { v0 = __builtin_nontemporal_load(a + i); v1 = __builtin_nontemporal_load(a + i + stride); v2 = __builtin_nontemporal_load(a + i + 2 * stride); v3 = __builtin_nontemporal_load(a + i + 3 * stride); }
        else { v0 = a[i]; v1 = a[i + stride]; v2 = a[i + 2 * stride]; v3 = a[i + 3 * stride]; }
        if (NT) { __builtin_nontemporal_store(v0, b + i); __builtin_nontemporal_store(v1, b + i + stride); __builtin_nontemporal_store(v2, b + i + 2 * stride); __builtin_nontemporal_store(v3, b + i + 3 * stride); }
        else { b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3; }
    }
    for (; i < n; i += stride) b[i] = a[i];
}

// one float4 per thread, as many workgroups as it takes
template <bool NT>
__global__ void __launch_bounds__(256) k_copy1(const vf4 *__restrict__ a, vf4 *__restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i); else b[i] = a[i]; }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "stream_copy: %s\n", hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int passes = argc > 2 ? atoi(argv[2]) : 20;
    const size_t n = (size_t)(gib * 1073741824.0 / 16.0);
    vf4 *a = nullptr, *b = nullptr;
    CHK(hipMalloc((void **)&a, n * 16)); CHK(hipMalloc((void **)&b, n * 16));
    CHK(hipMemset(a, 1, n * 16)); CHK(hipMemset(b, 0, n * 16));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    double best = 0.0;
    char how[64] = "";
    for (int variant = 0; variant < 4; ++variant)
    for (int blocks_per_cu : {4, 8, 16, 32, 64}) {
        const bool nt = (variant & 1) != 0, one = (variant & 2) != 0;
        if (one && blocks_per_cu != 4) continue;
        const unsigned grid = one ? (unsigned)((n + 255) / 256) : (unsigned)p.multiProcessorCount * blocks_per_cu;
        auto launch = [&](const vf4 *src, vf4 *dst) {
            if (one) { if (nt) hipLaunchKernelGGL(k_copy1<true>, dim3(grid), dim3(256), 0, nullptr, src, dst, n); else hipLaunchKernelGGL(k_copy1<false>, dim3(grid), dim3(256), 0, nullptr, src, dst, n); }
            else { if (nt) hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, nullptr, src, dst, n); else hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, nullptr, src, dst, n); }
        };
        launch(a, b);   // warm-up
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < passes; ++r) launch((r & 1) ? b : a, (r & 1) ? a : b);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        float ms = 0.0f; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double gbs = 2.0 * (double)n * 16.0 * passes / (ms * 1.0e-3) / 1.0e9;
        fprintf(stderr, "  %s %s grid %u: %.1f GB/s\n", one ? "one float4 per thread" : "grid-stride", nt ? "nt" : "plain", grid, gbs);
        if (gbs > best) { best = gbs; snprintf(how, sizeof(how), "%s, %s, %u workgroups", one ? "one float4 per thread" : "grid-stride x4", nt ? "non-temporal" : "plain", grid); }
    }
    printf("stream_copy %.1f GB/s (best of the variants tried: %s; float4 copy, %.1f GiB read + %.1f GiB written per pass, %d passes, %s)\n", best, how, gib, gib, passes, p.gcnArchName);
    return 0;
}
