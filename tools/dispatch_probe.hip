// Microbenchmark: cost of launching many workgroups that exit at once (after one global load),
// the shape of the flattened search kernels' surplus blocks.  us per launch by grid size,
// for 0 / 18 KB of static LDS per block and 64 / 256 threads per block.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void probe(const int *flag, int *sink)
{
    __shared__ int sh[LDS > 0 ? LDS : 1];
    if (flag[blockIdx.x & 63] == 12345) { sh[threadIdx.x % (LDS > 0 ? LDS : 1)] = 1; sink[0] = sh[0]; }
}
template <int LDS> float run(int grid, int threads, const int *flag, int *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<LDS>, dim3(grid), dim3(threads), 0, 0, flag, sink);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<LDS>, dim3(grid), dim3(threads), 0, 0, flag, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / 20.0f;
}
int main()
{
    int *flag, *sink; hipMalloc(&flag, 64 * 4); hipMalloc(&sink, 64); hipMemset(flag, 0, 64 * 4);
    printf("%8s %12s %12s %12s %12s\n", "blocks", "256thr", "256thr+18KB", "64thr", "1024thr");
    for (int grid : {64, 1024, 2048, 4096, 8192, 16384, 32768})
        printf("%8d %12.2f %12.2f %12.2f %12.2f\n", grid, run<0>(grid, 256, flag, sink), run<4648>(grid, 256, flag, sink),
               run<0>(grid, 64, flag, sink), run<0>(grid, 1024, flag, sink));
    return 0;
}
