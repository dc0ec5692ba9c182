// jd_gmm.h - the companion scoring kernels of juicer_amd (included by jd_device.hip; gfx950 only):
// HTKFlatModels::calcGMMOutput + logAdd (src/HTKFlatModels.cpp:226-293) for every tied state of every frame of a
// likelihood table - jd_gmm_kernel39 (D = 39: two frames per lane, packed fp32, the reference's roundings) and the
// generic jd_gmm_kernel - with the bit-exact replica of glibc's expf they need.
#pragma once

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, ARM optimized-routines
// algorithm): N=32 table + cubic in double, rounded once to float.  Replicated
// so that device logAdd equals the host libm result bit for bit (verified on
// the host for all 1.2e8 floats in [-18.5, -1e-3]: tests/test_expf.py, through jd_debug_expf).
#define JD_EXP2F_TAB                                                                              \
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,   \
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,   \
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,   \
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,   \
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,   \
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,   \
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,   \
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL
__device__ __constant__ unsigned long long jd_exp2f_tab[32] = {JD_EXP2F_TAB};
static const unsigned long long jd_exp2f_tab_host[32] = {JD_EXP2F_TAB};     // jd_debug_expf(device = -1)

// one source for the device function and its host twin (jd_debug_expf checks both against libm)
__host__ __device__ __forceinline__ float jd_expf_impl(float x, const unsigned long long *tab)
{
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = InvLn2N * (double)x;
    double kd = z + SHIFT;
    unsigned long long ki;
    memcpy(&ki, &kd, sizeof ki);
    kd -= SHIFT;
    double r = z - kd;
    unsigned long long t = tab[ki & 31];
    t += ki << 47;
    double s;
    memcpy(&s, &t, sizeof s);
    double p = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1.0;
    y = p * r2 + y;
    y = y * s;
    return (float)y;
}
__device__ __forceinline__ float jd_expf(float x) { return jd_expf_impl(x, jd_exp2f_tab); }

// HTKFlatModels::logAdd, HTKFlatModels.cpp:266-293
__device__ __forceinline__ float jd_log_add(float x, float y)
{
    if (x < y) { float t = x; x = y; y = t; }
    float diff = y - x;
    if (diff < -18.42) return x;
    return (float)((double)x + log(1.0 + (double)jd_expf(diff)));
}

// ------------------------------------------------------------------- GMM kernel

// par: [g][m][D][2] = (mean, ivar) interleaved; det: [g][m]; rows: row_src[r] is
// the frame index into feats (or -1); ll: [n_rows][G].
template <int DT>
__global__ __launch_bounds__(256) void jd_gmm_kernel(const float *__restrict__ feats,
                                                     const int *__restrict__ row_src, int n_rows,
                                                     const float *__restrict__ par,
                                                     const float *__restrict__ det,
                                                     const int *__restrict__ n_mix, int G, int M, int D,
                                                     float *__restrict__ ll, int skip_unused)
{
    constexpr int DP = (DT > 0) ? (DT | 1) : 0;      // odd row stride: conflict-free per-lane rows
    extern __shared__ __align__(16) char smem[];
    const int dp = (DT > 0) ? DP : (D | 1);
    float *sx = (float *)smem;                        // [64][dp]
    float *so = sx + GMM_ROWS * dp;                   // [64][GMM_GT+1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Dn = (DT > 0) ? DT : D;
    // tiles = (64-row tile, GMM_GT-state group); the grid may be smaller than the number of
    // tiles (launch_gmm bounds how many wave slots the scoring may hold next to the search)
    const int n_rt = (n_rows + GMM_ROWS - 1) / GMM_ROWS, n_gt = (G + GMM_GT - 1) / GMM_GT;
    for (int tile = blockIdx.x; tile < n_rt * n_gt; tile += gridDim.x) {
    // row tile skewed by the state group: a bounded grid whose size is a multiple of n_rt would
    // otherwise hand each workgroup the same row tile every time (and the skipped ones no work)
    const int gt = tile / n_rt;
    const int r0 = ((tile + gt) % n_rt) * GMM_ROWS;
    const int g0 = gt * GMM_GT;
    // a tile whose rows are all unused (stream finished / chunk shorter than its slot) is skipped:
    // the valid rows of a stream's slot are a prefix of it and slots are multiples of the tile
    // (rows_per_slot % GMM_ROWS == 0), so the tile's first row decides
    if (skip_unused && row_src[r0] < 0) continue;
    __syncthreads();                                  // previous tile's LDS reads are done

    // stage the 64 x D feature tile (coalesced along D)
    for (int e = tid; e < GMM_ROWS * Dn; e += 256) {
        int r = e / Dn, j = e - r * Dn;
        int src = (r0 + r < n_rows) ? row_src[r0 + r] : -1;
        sx[r * dp + j] = (src >= 0) ? feats[(size_t)src * Dn + j] : 0.0f;
    }
    __syncthreads();

    float x[(DT > 0) ? DT : 1];
    if (DT > 0) {
#pragma unroll
        for (int j = 0; j < DT; ++j) x[j] = sx[lane * dp + j];
    }

    constexpr int GPW = GMM_GT / 4;                   // tied states per wave
    for (int gi = 0; gi < GPW; ++gi) {
        const int gl = wid * GPW + gi;                // wave-uniform
        const int g = g0 + gl;
        float acc = LZ;
        if (g < G) {
            const int nm = n_mix[g];
            const float *pg = par + (size_t)g * M * Dn * 2;
            const float *dg = det + (size_t)g * M;
            for (int m = 0; m < nm; ++m) {
                const float *pm = pg + (size_t)m * Dn * 2;
                float sum = 0.0f;
                if (DT > 0) {
#pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        float xmu = x[j] - pm[2 * j];          // HTKFlatModels.cpp:249
                        sum += xmu * xmu * pm[2 * j + 1];      // :250  (no contraction)
                    }
                } else {
                    for (int j = 0; j < Dn; ++j) {
                        float xmu = sx[lane * dp + j] - pm[2 * j];
                        sum += xmu * xmu * pm[2 * j + 1];
                    }
                }
                float comp = (float)(-0.5 * (double)sum + (double)dg[m]);   // :254
                acc = jd_log_add(acc, comp);
            }
        }
        so[lane * (GMM_GT + 1) + gl] = acc;
    }
    __syncthreads();
    // coalesced store of the [64 rows][GMM_GT] tile
    for (int e = tid; e < GMM_ROWS * GMM_GT; e += 256) {
        int r = e / GMM_GT, c = e - r * GMM_GT;
        if (r0 + r < n_rows && g0 + c < G) ll[(size_t)(r0 + r) * G + g0 + c] = so[r * (GMM_GT + 1) + c];
    }
    }
}


// ---- the D = 39 kernel: TWO frames per lane, packed fp32 arithmetic.  GT tied states per tile (a quarter of them per
// wave): 64 for tables, 16 for the few rows of a streaming push - a tile is a chain of GT / 4 states x 16 mixtures per
// wave (0.3 ms at 64) whatever the number of rows, and a push of 64 frames is 47 such tiles on a chip of 1024 slots.
//
// A lane owns rows r and r + 64 of a 128-row tile; their vectors sit side by side in register
// pairs, so every VALU instruction of the distance loop is a packed one (v_pk_add_f32 /
// v_pk_mul_f32: two IEEE fp32 operations, no contraction - the same roundings as the reference's
// scalar code, HTKFlatModels.cpp:249-250) with the tied state's (mean, ivar) pairs arriving
// through the scalar cache.  logAdd (HTKFlatModels.cpp:266-293) evaluates log(1.0 + e), e in
// (0, 1], in double with a 128-interval table (c = 1 + k/128; log y = -log(invc) + log1p(y invc - 1),
// degree-7 polynomial: < 1 ulp in double, like the libm the reference links).
#define GMM_ROWS2 128
typedef float jd_f2 __attribute__((ext_vector_type(2)));
struct JdLogTab { double invc, logc; };

// One logAdd step for the two frames of a lane, straight-line (no branch: some lane of a wave always
// takes the long path) and written pairwise so that the two dependent chains interleave.  etab is
// the LDS copy of jd_exp2f_tab, tab the LDS copy of the log table.
__device__ __forceinline__ void jd_log_add2x2(float &a0, float &a1, float c0, float c1, const JdLogTab *tab,
                                              const unsigned long long *etab)
{
    const bool s0 = a0 < c0, s1 = a1 < c1;
    const float x0 = s0 ? c0 : a0, y0 = s0 ? a0 : c0;
    const float x1 = s1 ? c1 : a1, y1 = s1 ? a1 : c1;
    const float d0 = y0 - x0, d1 = y1 - x1;
    const bool keep0 = d0 < -18.42, keep1 = d1 < -18.42;               // HTKFlatModels.cpp:276 (double compare)
    // (the clamp keeps the table index in range for the lanes whose result is discarded)
    const double e0 = (double)jd_expf_impl(fmaxf(d0, -19.0f), etab), e1 = (double)jd_expf_impl(fmaxf(d1, -19.0f), etab);
    const double yy0 = 1.0 + e0, yy1 = 1.0 + e1;
    const JdLogTab t0 = tab[(int)(e0 * 128.0 + 0.5)], t1 = tab[(int)(e1 * 128.0 + 0.5)];
    const double r0 = __builtin_fma(yy0, t0.invc, -1.0), r1 = __builtin_fma(yy1, t1.invc, -1.0);
    double q0 = 1.0 / 7.0, q1 = 1.0 / 7.0;
    q0 = __builtin_fma(q0, r0, -1.0 / 6.0); q1 = __builtin_fma(q1, r1, -1.0 / 6.0);
    q0 = __builtin_fma(q0, r0, 1.0 / 5.0);  q1 = __builtin_fma(q1, r1, 1.0 / 5.0);
    q0 = __builtin_fma(q0, r0, -1.0 / 4.0); q1 = __builtin_fma(q1, r1, -1.0 / 4.0);
    q0 = __builtin_fma(q0, r0, 1.0 / 3.0);  q1 = __builtin_fma(q1, r1, 1.0 / 3.0);
    q0 = __builtin_fma(q0, r0, -1.0 / 2.0); q1 = __builtin_fma(q1, r1, -1.0 / 2.0);
    q0 = __builtin_fma(q0, r0, 1.0);        q1 = __builtin_fma(q1, r1, 1.0);
    const float n0 = (float)((double)x0 + (t0.logc + q0 * r0)), n1 = (float)((double)x1 + (t1.logc + q1 * r1));
    a0 = keep0 ? x0 : n0;
    a1 = keep1 ? x1 : n1;
}

template <int GT>
__global__ __launch_bounds__(256, 4) void jd_gmm_kernel39(const float *__restrict__ feats,
                                                       const int *__restrict__ row_src, int n_rows,
                                                       const float *__restrict__ par,
                                                       const float *__restrict__ det,
                                                       const int *__restrict__ n_mix, int G, int M,
                                                       float *__restrict__ ll, int skip_unused,
                                                       const JdLogTab *__restrict__ logtab,
                                                       const int *__restrict__ rt_base, int n_rt_list)
{
    constexpr int DT = 39, DP = 39;                   // odd row stride: conflict-free per-lane rows
    extern __shared__ __align__(16) char smem[];
    JdLogTab *stab = (JdLogTab *)smem;                // [129] (+ pad)
    unsigned long long *setab = (unsigned long long *)(smem + 130 * sizeof(JdLogTab));   // [32]
    float *sx = (float *)(smem + 130 * sizeof(JdLogTab) + 32 * sizeof(unsigned long long));   // [128][DP]
    float *so = sx;                                   // [128][GT+1]: the feature tile is in registers by then
                                                      // (36 KB per workgroup: four of them share a CU's LDS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 129; i += 256) stab[i] = logtab[i];
    if (tid < 32) setab[tid] = jd_exp2f_tab[tid];
    // rt_base (or null): the row tiles to score, by first row - the chunks of several streams, each in its own region of
    // the table (the resident kernel's side, jd_res_stage_many) - instead of every tile of rows [0, n_rows)
    const int n_rt = rt_base ? n_rt_list : (n_rows + GMM_ROWS2 - 1) / GMM_ROWS2, n_gt = (G + GT - 1) / GT;
    for (int tile = blockIdx.x; tile < n_rt * n_gt; tile += gridDim.x) {
        // row tile skewed by the state group (see jd_gmm_kernel)
        const int gt = tile / n_rt;
        const int r0 = rt_base ? rt_base[(tile + gt) % n_rt] : ((tile + gt) % n_rt) * GMM_ROWS2;
        const int g0 = gt * GT;
        if (skip_unused && row_src[r0] < 0) continue;
        __syncthreads();                              // previous tile's LDS reads are done
        for (int e = tid; e < GMM_ROWS2 * DT; e += 256) {
            const int r = e / DT, j = e - r * DT;
            const int src = (r0 + r < n_rows) ? row_src[r0 + r] : -1;
            sx[r * DP + j] = (src >= 0) ? feats[(size_t)src * DT + j] : 0.0f;
        }
        __syncthreads();
        jd_f2 x[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) { x[j].x = sx[lane * DP + j]; x[j].y = sx[(lane + 64) * DP + j]; }
        __syncthreads();                              // sx is re-used as the output tile
        constexpr int GPW = GT / 4;               // tied states per wave
        for (int gi = 0; gi < GPW; ++gi) {
            const int gl = wid * GPW + gi;            // wave-uniform
            const int g = g0 + gl;
            float acc0 = LZ, acc1 = LZ;
            if (g < G) {
                const int nm = n_mix[g];
                const float *pg = par + (size_t)g * M * DT * 2;
                const float *dg = det + (size_t)g * M;
                for (int m = 0; m < nm; ++m) {
                    const float *pm = pg + (size_t)m * DT * 2;
                    jd_f2 sum = {0.0f, 0.0f};
                    // three dimensions at a time: their squared distances are independent, only the
                    // running sum is a chain (added in the reference's order)
#pragma unroll
                    for (int j = 0; j < DT; j += 3) {
                        const jd_f2 mu0 = {pm[2 * j], pm[2 * j]}, iv0 = {pm[2 * j + 1], pm[2 * j + 1]};
                        const jd_f2 mu1 = {pm[2 * j + 2], pm[2 * j + 2]}, iv1 = {pm[2 * j + 3], pm[2 * j + 3]};
                        const jd_f2 mu2 = {pm[2 * j + 4], pm[2 * j + 4]}, iv2 = {pm[2 * j + 5], pm[2 * j + 5]};
                        const jd_f2 u0 = x[j] - mu0, u1 = x[j + 1] - mu1, u2 = x[j + 2] - mu2;   // HTKFlatModels.cpp:249
                        const jd_f2 w0 = u0 * u0, w1 = u1 * u1, w2 = u2 * u2;
                        const jd_f2 z0 = w0 * iv0, z1 = w1 * iv1, z2 = w2 * iv2;                 // :250  (no contraction)
                        if (j == 0) sum = z0; else sum += z0;             // (0.0f + z0 == z0: z0 >= +0)
                        sum += z1; sum += z2;
                    }
                    const double dm = (double)dg[m];
                    const float c0 = (float)(-0.5 * (double)sum.x + dm), c1 = (float)(-0.5 * (double)sum.y + dm);   // :254
                    // logAdd(LOG_ZERO, c) is c for every c > LOG_ZERO and LOG_ZERO else (the difference is below -18.42,
                    // or the sum rounds back): the first mixture needs no exponential and no logarithm
                    if (m == 0) { acc0 = LZ < c0 ? c0 : LZ; acc1 = LZ < c1 ? c1 : LZ; }
                    else jd_log_add2x2(acc0, acc1, c0, c1, stab, setab);
                }
            }
            so[lane * (GT + 1) + gl] = acc0;
            so[(lane + 64) * (GT + 1) + gl] = acc1;
        }
        __syncthreads();
        for (int e = tid; e < GMM_ROWS2 * GT; e += 256) {
            const int r = e / GT, c = e - r * GT;
            if (r0 + r < n_rows && g0 + c < G) ll[(size_t)(r0 + r) * G + g0 + c] = so[r * (GT + 1) + c];
        }
    }
}


// ---- the D = 39 kernel with the scoring OPTION of jd_dec_set_scoring(JD_SCORE_FAST): the same tiling (two frames per lane, a quarter of a
// tile's tied states per wave, parameters through the scalar cache) without the reference's roundings.
//   distance   s = sqrt(ivar), t = -mean s (prepared on the host: AmDevBuf::par_fast): u = fma(x, s, t), sum = fma(u, u, sum) - two packed
//              fused multiply-adds per dimension and frame pair where the exact kernel issues four packed operations (sub, mul, mul, add)
//   logAdd     max + log(1 + exp(min - max)) in fp32 on the hardware's exp2 / log2 (v_exp_f32 / v_log_f32), the reference's -18.42 cut kept:
//              ~12 instructions per frame where the bit-exact replica (glibc's expf + an fp64 log(1 + e)) takes ~58
// north_star asks for path / acoustic scores within 1e-4 relative of the reference and identical words and times; this kernel is held to that
// (tests/test_gpu_fastscore.py: every fixture), not to bit equality - the DEFAULT stays jd_gmm_kernel39, whose table equals the CPU oracle's
// bit for bit.  Error: a log-likelihood of magnitude ~50-100 moves by ~1e-5 (39 fused terms of relative error 2^-24 each and one logAdd per
// mixture of absolute error ~1e-7).
__device__ __forceinline__ float jd_log_add_fast(float a, float c)
{
    const float x = fmaxf(a, c), y = fminf(a, c);
    const float d = y - x;
    const float e = __builtin_amdgcn_exp2f(d * 1.44269504088896340736f);      // exp(d), d <= 0
    const float l = __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994530942f;
    return d < -18.42f ? x : x + l;                                           // HTKFlatModels.cpp:276 (LOG_ZERO - anything: -inf < cut)
}

template <int GT>
__global__ __launch_bounds__(256, 4) void jd_gmm_fast39(const float *__restrict__ feats, const int *__restrict__ row_src, int n_rows,
                                                      const float *__restrict__ par_fast, const float *__restrict__ det,
                                                      const int *__restrict__ n_mix, int G, int M, float *__restrict__ ll, int skip_unused,
                                                      const int *__restrict__ rt_base, int n_rt_list)
{
    constexpr int DT = 39, DP = 39;
    extern __shared__ __align__(16) char smem[];
    float *sx = (float *)smem;                        // [128][DP]
    float *so = sx;                                   // [128][GT+1]: the feature tile is in registers by then
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_rt = rt_base ? n_rt_list : (n_rows + GMM_ROWS2 - 1) / GMM_ROWS2, n_gt = (G + GT - 1) / GT;
    for (int tile = blockIdx.x; tile < n_rt * n_gt; tile += gridDim.x) {
        const int gt = tile / n_rt;                   // row tile skewed by the state group (see jd_gmm_kernel)
        const int r0 = rt_base ? rt_base[(tile + gt) % n_rt] : ((tile + gt) % n_rt) * GMM_ROWS2;
        const int g0 = gt * GT;
        if (skip_unused && row_src[r0] < 0) continue;
        __syncthreads();
        for (int e = tid; e < GMM_ROWS2 * DT; e += 256) {
            const int r = e / DT, j = e - r * DT;
            const int src = (r0 + r < n_rows) ? row_src[r0 + r] : -1;
            sx[r * DP + j] = (src >= 0) ? feats[(size_t)src * DT + j] : 0.0f;
        }
        __syncthreads();
        jd_f2 x[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) { x[j].x = sx[lane * DP + j]; x[j].y = sx[(lane + 64) * DP + j]; }
        __syncthreads();
        constexpr int GPW = GT / 4;
        for (int gi = 0; gi < GPW; ++gi) {
            const int gl = wid * GPW + gi;            // wave-uniform
            const int g = g0 + gl;
            float acc0 = LZ, acc1 = LZ;
            if (g < G) {
                const int nm = n_mix[g];
                const float *pg = par_fast + (size_t)g * M * DT * 2;
                const float *dg = det + (size_t)g * M;
                for (int m = 0; m < nm; ++m) {
                    const float *pm = pg + (size_t)m * DT * 2;
                    jd_f2 sum = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        const jd_f2 sj = {pm[2 * j], pm[2 * j]}, tj = {pm[2 * j + 1], pm[2 * j + 1]};
                        const jd_f2 u = __builtin_elementwise_fma(x[j], sj, tj);
                        sum = __builtin_elementwise_fma(u, u, sum);
                    }
                    const float dm = dg[m];
                    const float c0 = __builtin_fmaf(-0.5f, sum.x, dm), c1 = __builtin_fmaf(-0.5f, sum.y, dm);
                    if (m == 0) { acc0 = LZ < c0 ? c0 : LZ; acc1 = LZ < c1 ? c1 : LZ; }
                    else { acc0 = jd_log_add_fast(acc0, c0); acc1 = jd_log_add_fast(acc1, c1); }
                }
            }
            so[lane * (GT + 1) + gl] = acc0;
            so[(lane + 64) * (GT + 1) + gl] = acc1;
        }
        __syncthreads();
        for (int e = tid; e < GMM_ROWS2 * GT; e += 256) {
            const int r = e / GT, c = e - r * GT;
            if (r0 + r < n_rows && g0 + c < G) ll[(size_t)(r0 + r) * G + g0 + c] = so[r * (GT + 1) + c];
        }
    }
}
