// Microbenchmark: throughput of scattered 16-byte-record accesses (the shape of the per-arc
// recombination table): every lane touches a pseudo-random record of a table of `n` records.
//   mode 0: returning 64-bit atomicMax        mode 1: non-returning 64-bit atomicMax
//   mode 2: plain 8-byte load                 mode 3: plain 16-byte load
//   mode 4: 8-byte load, then atomicMax only if it would raise the key (about half do)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct __align__(16) Rec { unsigned long long key; int slot; int pad; };
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(256) void probe(Rec *tab, unsigned n, int per_lane, unsigned salt, unsigned long long *sink)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (int k = 0; k < per_lane; ++k) {
        const unsigned h = hash32(gid * 977u + k * 0x9e3779b9u + salt);
        Rec *r = tab + (h % n);
        const unsigned long long key = ((unsigned long long)hash32(h) << 32) | gid;
        if (MODE == 0) acc += atomicMax(&r->key, key);
        else if (MODE == 1) (void)__hip_atomic_fetch_max(&r->key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) acc += __hip_atomic_load(&r->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 3) { const int4 v = *(const int4 *)r; acc += (unsigned)v.x + (unsigned)v.z; }
        else { const unsigned long long o = __hip_atomic_load(&r->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               if (key > o) acc += atomicMax(&r->key, key); }
    }
    if (acc == 0x123456789abcdefULL) sink[0] = acc;
}
template <int MODE> float run(Rec *tab, unsigned n, int grid, int per_lane, unsigned long long *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, tab, n, per_lane, 1u, sink);
    hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, tab, n, per_lane, 7u + i, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / 5.0f;
}
int main(int argc, char **argv)
{
    unsigned long long *sink; hipMalloc(&sink, 64);
    const int grid = 4096, per_lane = 4;                  // 4096 x 256 x 4 = 4.2M accesses per launch
    const double n_acc = (double)grid * 256 * per_lane;
    printf("%.1fM scattered accesses per launch; G accesses/s by table size\n", n_acc / 1e6);
    printf("%10s %10s %10s %10s %10s %10s\n", "table MB", "max64ret", "max64nr", "load8", "load16", "ld+max");
    for (unsigned mb : {4u, 32u, 128u, 512u, 2048u}) {
        const unsigned n = mb * (1u << 20) / sizeof(Rec);
        Rec *tab; if (hipMalloc(&tab, (size_t)n * sizeof(Rec)) != hipSuccess) break;
        hipMemset(tab, 0, (size_t)n * sizeof(Rec));
        const float t0 = run<0>(tab, n, grid, per_lane, sink), t1 = run<1>(tab, n, grid, per_lane, sink);
        const float t2 = run<2>(tab, n, grid, per_lane, sink), t3 = run<3>(tab, n, grid, per_lane, sink);
        hipMemset(tab, 0, (size_t)n * sizeof(Rec));
        const float t4 = run<4>(tab, n, grid, per_lane, sink);
        printf("%10u %10.2f %10.2f %10.2f %10.2f %10.2f\n", mb, n_acc / t0 / 1e3, n_acc / t1 / 1e3, n_acc / t2 / 1e3,
               n_acc / t3 / 1e3, n_acc / t4 / 1e3);
        hipFree(tab);
    }
    return 0;
}
