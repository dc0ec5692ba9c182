// Dependent-access latency on MI355X under the access flavours k_search uses: plain load, agent-scope
// (sc1) load, returning 64-bit atomicMax, sc1 store followed by sc1 load - pointer chasing over a
// table of `mb` MiB, every CU busy with `waves` waves.  hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
template <int MODE>
__global__ void chase(unsigned long long *tab, int hops, unsigned long long *out, long long *clk)
{
    unsigned long long idx = (blockIdx.x * blockDim.x + threadIdx.x) * 977ull % 1000003ull;
    const long long t0 = wall_clock64();
    unsigned long long acc = 0;
    for (int h = 0; h < hops; ++h) {
        unsigned long long v;
        if (MODE == 0) v = tab[idx];
        else if (MODE == 1) v = __hip_atomic_load(tab + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) v = atomicMax(tab + idx, 1ull);   // never exceeds the stored next-index (>= 1)... keeps value
        else { __hip_atomic_store(tab + idx + 1, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               v = __hip_atomic_load(tab + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        acc += v;
        idx = v;
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main(int argc, char **argv)
{
    const int mb = argc > 1 ? atoi(argv[1]) : 64, waves = argc > 2 ? atoi(argv[2]) : 16, hops = 200;
    const size_t n = (size_t)mb * 1024 * 1024 / 16;       // 16-byte cells: [next, scratch]
    std::vector<unsigned long long> h(n * 2);
    std::vector<unsigned> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::shuffle(perm.begin(), perm.end(), std::mt19937(1));
    for (size_t i = 0; i < n; ++i) { h[2 * (size_t)perm[i]] = 2ull * perm[(i + 1) % n]; h[2 * (size_t)perm[i] + 1] = 0; }
    unsigned long long *d, *out; long long *clk;
    const int blocks = 256, threads = waves * 64;
    hipMalloc(&d, h.size() * 8); hipMalloc(&out, (size_t)blocks * threads * 8); hipMalloc(&clk, blocks * 8);
    const char *names[4] = {"plain load", "sc1 load", "atomicMax (returning)", "sc1 store + sc1 load"};
    for (int mode = 0; mode < 4; ++mode) {
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(threads), 0, 0, d, hops, out, clk);
            if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(threads), 0, 0, d, hops, out, clk);
            if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(threads), 0, 0, d, hops, out, clk);
            if (mode == 3) hipLaunchKernelGGL(chase<3>, dim3(blocks), dim3(threads), 0, 0, d, hops, out, clk);
            hipDeviceSynchronize();
        }
        std::vector<long long> c(blocks);
        hipMemcpy(c.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += v;
        printf("%-24s table %4d MiB, %2d waves/CU: %.3f us per dependent hop\n", names[mode], mb, waves, s / blocks / hops / 100.0);
    }
    return 0;
}
