// Microbenchmark: cost of same-address device-scope atomics issued by many blocks.
// Each block does ONE atomic (thread 0) on counter[blockIdx.x / share]; `share` blocks share a counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>   // 0: returning atomicAdd (int), 1: non-returning atomicAdd, 2: returning 64-bit add, 3: 64-bit atomicMax no return, 4: none
__global__ void probe(int *ctr, unsigned long long *ctr64, int share, int *sink)
{
    __shared__ int sh;
    const int c = (blockIdx.x / share) * 32;     // counters 128 B apart
    if (threadIdx.x == 0) {
        if (MODE == 0) sh = atomicAdd(&ctr[c], 1);
        else if (MODE == 1) { (void)__hip_atomic_fetch_add(&ctr[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh = 0; }
        else if (MODE == 2) sh = (int)atomicAdd(&ctr64[c / 2], 1ULL);
        else if (MODE == 3) { (void)__hip_atomic_fetch_max(&ctr64[c / 2], (unsigned long long)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh = 0; }
        else sh = 0;
    }
    __syncthreads();
    if (threadIdx.x == 1 && sh == 123456789) sink[0] = sh;
}
template <int MODE> float run(int grid, int share, int *ctr, unsigned long long *c64, int *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, ctr, c64, share, sink);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, ctr, c64, share, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / 20.0f;
}
int main()
{
    int *ctr, *sink; unsigned long long *c64;
    hipMalloc(&ctr, 1 << 22); hipMalloc(&c64, 1 << 22); hipMalloc(&sink, 64);
    hipMemset(ctr, 0, 1 << 22); hipMemset(c64, 0, 1 << 22);
    const int grid = 8192;
    printf("grid %d blocks x 256 threads, one atomic per block; us per launch\n", grid);
    printf("%8s %10s %10s %10s %10s %10s\n", "share", "ret32", "noret32", "ret64", "max64nr", "none");
    for (int share : {1, 8, 32, 128, 512, 8192})
        printf("%8d %10.1f %10.1f %10.1f %10.1f %10.1f\n", share, run<0>(grid, share, ctr, c64, sink), run<1>(grid, share, ctr, c64, sink),
               run<2>(grid, share, ctr, c64, sink), run<3>(grid, share, ctr, c64, sink), run<4>(grid, share, ctr, c64, sink));
    return 0;
}
