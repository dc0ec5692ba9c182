// tools/cu_census.hip - which (XCC, SE, CU) ids does a grid of one-workgroup-per-CU land on?  (hipcc --offload-arch=gfx950 -o tools/cu_census tools/cu_census.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(64) void census(unsigned *out)
{
    __shared__ unsigned char whole[163840];
    whole[threadIdx.x * 997 % 163840] = 1;
    __syncthreads();
    if (threadIdx.x) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    out[blockIdx.x] = (xcc << 28) | (hw & 0x0fffffff);
    const long long t = wall_clock64() + 2000000;    // stay 20 ms: everybody is resident at once
    while (wall_clock64() < t) __builtin_amdgcn_s_sleep(100);
    if (whole[5] == 77) out[0] = 0;
}
int main()
{
    int n = 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); n = p.multiProcessorCount;
    unsigned *d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(census, dim3(n), dim3(64), 0, 0, d);
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, std::vector<unsigned>>> m;
    for (int b = 0; b < n; ++b) { unsigned v = h[b]; m[v >> 28][(v >> 13) & 7].push_back((v >> 8) & 0xf); }
    for (auto &x : m) for (auto &s : x.second) { printf("xcc %u se %u:", x.first, s.first); for (unsigned c : s.second) printf(" %u", c); printf("\n"); }
    printf("block 0..15 -> xcc:"); for (int b = 0; b < 16; ++b) printf(" %u", h[b] >> 28); printf("\n");
    return 0;
}
