// Scattered atomic adds: rate by data type and SCOPE, shared image against one private image per XCD.
// Question: the tallies of the transport kernel are scattered global_atomic_add_f64 (agent scope), which execute at the
// memory side (MI355X_MICROARCH.md, Global float atomics).  Do integer atomics, or atomics of workgroup scope into an image
// that only ONE XCD touches (chosen by HW_REG_XCC_ID, so its lines live in that XCD's L2), run faster -- and is no add lost?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench/atomic_rates.hip -o tools/microbench/atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// OP: 0 f64 agent, 1 u64 agent, 2 u64 workgroup, 3 f64 workgroup, 4 u32 agent, 5 u32 workgroup, 6 f32 agent, 7 f32 workgroup
template <int OP>
__global__ void __launch_bounds__(256) k(void *img, unsigned region, unsigned stride_elems, int per_xcd, int nadd, unsigned span) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned base = per_xcd ? xcc * stride_elems : 0u;
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    // `span`: the lanes of a wave scatter over a window of that many elements which moves through the region (1 << 30: anywhere)
    for (int i = 0; i < nadd; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned win = (span >= region) ? 0u : (((unsigned)i * 977u + blockIdx.x * 131u) % (region - span));
        const unsigned idx = base + win + (s >> 8) % (span >= region ? region : span);
        if (OP == 0) __hip_atomic_fetch_add((double *)img + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (OP == 1) __hip_atomic_fetch_add((unsigned long long *)img + idx, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (OP == 2) __hip_atomic_fetch_add((unsigned long long *)img + idx, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 3) __hip_atomic_fetch_add((double *)img + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 4) __hip_atomic_fetch_add((unsigned *)img + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (OP == 5) __hip_atomic_fetch_add((unsigned *)img + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 6) __hip_atomic_fetch_add((float *)img + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (OP == 7) __hip_atomic_fetch_add((float *)img + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <int OP>
int run(const char *name, int esize, bool is_float, void *d, unsigned region, int per_xcd, unsigned span, int ncu) {
    const unsigned stride = region;
    const size_t bytes = (size_t)region * esize * (per_xcd ? 8 : 1);
    const int blocks = ncu * 5, nadd = 2000;
    CHK(hipMemset(d, 0, bytes));
    k<OP><<<blocks, 256>>>(d, region, stride, per_xcd, 10, span);
    CHK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(d, region, stride, per_xcd, nadd, span); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<char> h(bytes);
    CHK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
    double sum = 0.0;
    const size_t n = bytes / esize;
    for (size_t i = 0; i < n; ++i) {
        if (esize == 8) sum += is_float ? ((double *)h.data())[i] : (double)((unsigned long long *)h.data())[i];
        else sum += is_float ? (double)((float *)h.data())[i] : (double)((unsigned *)h.data())[i];
    }
    const double want = (double)blocks * 256 * nadd;
    printf("%-16s region %8u elems %-8s span %-10u %8.3f ms  %.3g adds/s   sum %.0f of %.0f %s\n", name, region, per_xcd ? "x8 (XCD)" : "shared", span,
           ms, want / (ms * 1e-3), sum, want, sum == want ? "exact" : "LOST/ROUNDED");
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    void *d; CHK(hipMalloc(&d, (size_t)8 * 230400 * 8 * 2));
    for (unsigned region : {230400u, 6400u}) {          // the 480 x 480 image; one 64-column tile with its margin
        for (unsigned span : {1u << 30, 1024u}) {
            if (span < (1u << 30) && region <= 6400u) continue;
            for (int px = 0; px < 2; ++px) {
                if (run<0>("f64 agent", 8, true, d, region, px, span, ncu)) return 1;
                if (run<3>("f64 workgroup", 8, true, d, region, px, span, ncu)) return 1;
                if (run<1>("u64 agent", 8, false, d, region, px, span, ncu)) return 1;
                if (run<2>("u64 workgroup", 8, false, d, region, px, span, ncu)) return 1;
                if (run<6>("f32 agent", 4, true, d, region, px, span, ncu)) return 1;
                if (run<7>("f32 workgroup", 4, true, d, region, px, span, ncu)) return 1;
                if (run<4>("u32 agent", 4, false, d, region, px, span, ncu)) return 1;
                if (run<5>("u32 workgroup", 4, false, d, region, px, span, ncu)) return 1;
            }
        }
    }
    return 0;
}
