// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of k_search
// (MI355X_MICROARCH.md, HBM: "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Every kernel below performs a KNOWN number of accesses of one
// shape on a table far larger than L2 + Infinity Cache (default 8 GiB); run it under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- tools/traffic_probe
//     rocprofv3 --kernel-trace --pmc WRITE_SIZE  -- tools/traffic_probe
// (separate passes) and divide the counters by the access counts it prints: tools/traffic_calib.py.
//
//   p_stream_read16    coalesced 16 B / lane streaming read           (the guide's calibrated case: tally = bytes / 2)
//   p_stream_write16   coalesced 16 B / lane streaming write
//   p_chunk_read16     phase A's record reads: 1 KiB runs (64 lanes x 16 B), 5 fields 1 KiB apart, chunks at random
//   p_gather16         one random 16-byte record per lane, `sc1` load  (arc keys, state records, items)
//   p_gather16x2       both 16-byte halves of a random 32-byte record  (the per-state record)
//   p_gather4          one random 4-byte word per lane                  (likelihoods, list entries)
//   p_scatter16        one random 16-byte store per lane, write-through (`sc1`)
//   p_scatter16_plain  ... plain store (the XCD-local flavour)
//   p_atomic_agent     returning 64-bit atomic max, agent scope, random record
//   p_atomic_wg        ... workgroup scope (performed in the XCD's L2: the XCD-local flavour)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned long long rnd64(unsigned a, unsigned b) { return ((unsigned long long)hash32(a * 0x9e3779b9u + b) << 32) | hash32(b * 0x85ebca6bu + a + 17u); }

#define GRID 8192
#define BLOCK 256
#define PER_LANE 8

__global__ __launch_bounds__(BLOCK) void p_stream_read16(const v4i *tab, unsigned long long n16, unsigned long long *sink)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x, stride = (unsigned long long)GRID * BLOCK;
    int acc = 0;
    for (int k = 0; k < PER_LANE * 8; ++k) { const v4i v = tab[(gid + k * stride) % n16]; acc += v.x + v.w; }
    if (acc == 0x12345678) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void p_stream_write16(v4i *tab, unsigned long long n16)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x, stride = (unsigned long long)GRID * BLOCK;
    for (int k = 0; k < PER_LANE * 8; ++k) tab[(gid + k * stride) % n16] = (v4i){(int)gid, k, 0, 1};
}
__global__ __launch_bounds__(BLOCK) void p_chunk_read16(const v4i *tab, unsigned long long n16, unsigned long long *sink)
{
    const unsigned wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int acc = 0;
    for (int k = 0; k < PER_LANE; ++k) {
        const unsigned long long chunk = rnd64(wave, (unsigned)k) % (n16 / 320);       // a chunk = 5 fields x 64 records x 16 B
        const v4i *c = tab + chunk * 320;
#pragma unroll
        for (int f = 0; f < 5; ++f) { const v4i v = c[f * 64 + lane]; acc += v.x + v.w; }
    }
    if (acc == 0x12345678) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void p_gather16(const v4i *tab, unsigned long long n16, unsigned long long *sink)
{
    const unsigned gid = blockIdx.x * BLOCK + threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)0xffffffffu, 0x00020000);
    int acc = 0;
    for (int k = 0; k < PER_LANE; ++k) {
        const unsigned off = (unsigned)(rnd64(gid, (unsigned)k) % (0xf0000000ULL / 16)) * 16u;   // (32-bit offsets: the first 3.75 GiB)
        const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
        acc += v.x + v.w;
    }
    if (acc == 0x12345678) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void p_gather16x2(const v4i *tab, unsigned long long n16, unsigned long long *sink)
{
    const unsigned gid = blockIdx.x * BLOCK + threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)0xffffffffu, 0x00020000);
    int acc = 0;
    for (int k = 0; k < PER_LANE; ++k) {
        const unsigned off = (unsigned)(rnd64(gid, (unsigned)k + 100u) % (0xf0000000ULL / 32)) * 32u;
        const v4i a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16), b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off + 16, 0, 16);
        acc += a.x + b.w;
    }
    if (acc == 0x12345678) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void p_gather4(const int *tab, unsigned long long n4, unsigned long long *sink)
{
    const unsigned gid = blockIdx.x * BLOCK + threadIdx.x;
    int acc = 0;
    for (int k = 0; k < PER_LANE; ++k) acc += __hip_atomic_load(tab + rnd64(gid, (unsigned)k + 200u) % n4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 0x12345678) sink[0] = acc;
}
template <bool SC1>
__global__ __launch_bounds__(BLOCK) void p_scatter16(v4i *tab, unsigned long long n16)
{
    const unsigned gid = blockIdx.x * BLOCK + threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)0xffffffffu, 0x00020000);
    for (int k = 0; k < PER_LANE; ++k) {
        const unsigned off = (unsigned)(rnd64(gid, (unsigned)k + 300u) % (0xf0000000ULL / 16)) * 16u;
        __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)gid, k, 0, 1}, r, (int)off, 0, SC1 ? 16 : 0);
    }
}
template <bool AGENT>
__global__ __launch_bounds__(BLOCK) void p_atomic(unsigned long long *tab, unsigned long long n16, unsigned long long *sink)
{
    const unsigned gid = blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long acc = 0;
    for (int k = 0; k < PER_LANE; ++k) {
        unsigned long long *p = tab + 2 * (rnd64(gid, (unsigned)k + 400u) % n16);
        const unsigned long long key = rnd64(gid + 7u, (unsigned)k);
        acc += AGENT ? __hip_atomic_fetch_max(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                     : __hip_atomic_fetch_max(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (acc == 0x123456789abcdefULL) sink[0] = acc;
}

#define TIME(name, bytes_known, launch) do { hipEvent_t a_, b_; hipEventCreate(&a_); hipEventCreate(&b_); hipEventRecord(a_, 0); launch; \
    hipEventRecord(b_, 0); hipEventSynchronize(b_); float ms_; hipEventElapsedTime(&ms_, a_, b_); \
    printf("{\"kernel\": \"%s\", \"accesses\": %.0f, \"useful_bytes\": %.0f, \"ms\": %.4f}\n", name, n_acc, (double)(bytes_known), ms_); } while (0)

int main(int argc, char **argv)
{
    const unsigned long long gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8ULL;
    const unsigned long long bytes = gib << 30, n16 = bytes / 16;
    v4i *tab; unsigned long long *sink;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(tab, 0, bytes);
    hipDeviceSynchronize();
    double n_acc = (double)GRID * BLOCK * PER_LANE * 8;
    TIME("p_stream_read16", n_acc * 16, hipLaunchKernelGGL(p_stream_read16, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16, sink));
    TIME("p_stream_write16", n_acc * 16, hipLaunchKernelGGL(p_stream_write16, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16));
    n_acc = (double)GRID * BLOCK * PER_LANE;
    TIME("p_chunk_read16", n_acc * 80, hipLaunchKernelGGL(p_chunk_read16, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16, sink));
    TIME("p_gather16", n_acc * 16, hipLaunchKernelGGL(p_gather16, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16, sink));
    TIME("p_gather16x2", n_acc * 32, hipLaunchKernelGGL(p_gather16x2, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16, sink));
    TIME("p_gather4", n_acc * 4, hipLaunchKernelGGL(p_gather4, dim3(GRID), dim3(BLOCK), 0, 0, (const int *)tab, bytes / 4, sink));
    TIME("p_scatter16<true>", n_acc * 16, hipLaunchKernelGGL(p_scatter16<true>, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16));
    TIME("p_scatter16<false>", n_acc * 16, hipLaunchKernelGGL(p_scatter16<false>, dim3(GRID), dim3(BLOCK), 0, 0, tab, n16));
    TIME("p_atomic<true>", n_acc * 8, hipLaunchKernelGGL(p_atomic<true>, dim3(GRID), dim3(BLOCK), 0, 0, (unsigned long long *)tab, n16, sink));
    TIME("p_atomic<false>", n_acc * 8, hipLaunchKernelGGL(p_atomic<false>, dim3(GRID), dim3(BLOCK), 0, 0, (unsigned long long *)tab, n16, sink));
    hipDeviceSynchronize();
    return 0;
}
