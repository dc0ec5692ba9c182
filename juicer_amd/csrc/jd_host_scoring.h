// jd_host_scoring.h - the host side of the companion scoring kernels (included by jd_device.hip): the model parameters on the device,
// launch_gmm (jd_gmm_kernel39, the default; jd_gmm_fast39, jd_dec_set_scoring's option; the generic and the hybrid kernel),
// jd_am_score_frames.  Reference: HTKFlatModels::calcGMMOutput + logAdd, src/HTKFlatModels.cpp:190-293.
#pragma once

// hybrid scoring (HTKFlatModels.cpp:190-222): output = x[model] - log prior; rows as in the GMM kernels
__global__ void jd_hybrid_kernel(const float *__restrict__ feats, const int *__restrict__ row_src, int n_rows,
                                 const float *__restrict__ log_prior, int G, float *__restrict__ ll)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_rows * G) return;
    const int r = (int)(i / G), g = (int)(i - (long long)r * G);
    const int src = row_src[r];
    ll[i] = (src >= 0) ? feats[(size_t)src * G + g] - log_prior[g] : 0.0f;
}

struct AmDevBuf {
    float *par = nullptr, *det = nullptr; int *n_mix = nullptr;
    float *par_fast = nullptr;      // jd_dec_set_scoring(JD_SCORE_FAST): [g][m][D][2] = (sqrt(ivar), -mean sqrt(ivar)), made when first asked for
    int fast = 0;                   // launch_gmm scores with jd_gmm_fast39
    float *log_prior = nullptr;
    JdLogTab *logtab = nullptr;
    int device = -1;
};

static int upload_am_gmm(const jd_am *a, AmDevBuf &b)
{
    const size_t gm = (size_t)a->n_gmm * a->max_mix, D = (size_t)a->D;
    std::vector<float> par(gm * D * 2);
    for (size_t i = 0; i < gm; ++i)
        for (size_t j = 0; j < D; ++j) {
            par[(i * D + j) * 2] = a->mean[i * D + j];
            par[(i * D + j) * 2 + 1] = a->ivar[i * D + j];
        }
    HIPCHK(hipMalloc(&b.par, par.size() * sizeof(float)));
    HIPCHK(hipMalloc(&b.det, gm * sizeof(float)));
    HIPCHK(hipMalloc(&b.n_mix, (size_t)a->n_gmm * sizeof(int)));
    HIPCHK(hipMemcpy(b.par, par.data(), par.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b.det, a->det.data(), gm * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b.n_mix, a->n_mix.data(), (size_t)a->n_gmm * sizeof(int), hipMemcpyHostToDevice));
    {   // table of jd_log_add2: c_k = 1 + k/128, invc = fl(1/c_k), logc = -log(invc) (so that the identity
        // log y = logc + log1p(y invc - 1) holds for the ROUNDED invc)
        std::vector<JdLogTab> t(129);
        for (int k = 0; k <= 128; ++k) {
            const double c = 1.0 + k / 128.0;
            t[(size_t)k].invc = (k == 0) ? 1.0 : 1.0 / c;
            t[(size_t)k].logc = (k == 0) ? 0.0 : (double)(-logl((long double)t[(size_t)k].invc));
        }
        HIPCHK(hipMalloc(&b.logtab, t.size() * sizeof(JdLogTab)));
        HIPCHK(hipMemcpy(b.logtab, t.data(), t.size() * sizeof(JdLogTab), hipMemcpyHostToDevice));
    }
    if (a->hybrid) {
        HIPCHK(hipMalloc(&b.log_prior, a->log_prior.size() * sizeof(float)));
        HIPCHK(hipMemcpy(b.log_prior, a->log_prior.data(), a->log_prior.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return JD_OK;
}

// the parameters of the scoring option (jd_gmm.h: jd_gmm_fast39)
static int upload_am_fast(const jd_am *a, AmDevBuf &b)
{
    if (b.par_fast) return JD_OK;
    const size_t gm = (size_t)a->n_gmm * a->max_mix, D = (size_t)a->D;
    std::vector<float> par(gm * D * 2);
    for (size_t i = 0; i < gm; ++i)
        for (size_t j = 0; j < D; ++j) {
            const double s = sqrt((double)a->ivar[i * D + j]);
            par[(i * D + j) * 2] = (float)s;
            par[(i * D + j) * 2 + 1] = (float)(-(double)a->mean[i * D + j] * s);
        }
    HIPCHK(hipMalloc(&b.par_fast, par.size() * sizeof(float)));
    HIPCHK(hipMemcpy(b.par_fast, par.data(), par.size() * sizeof(float), hipMemcpyHostToDevice));
    return JD_OK;
}

static void free_am_gmm(AmDevBuf &b)
{
    if (b.par) (void)hipFree(b.par);
    if (b.par_fast) (void)hipFree(b.par_fast);
    if (b.det) (void)hipFree(b.det);
    if (b.n_mix) (void)hipFree(b.n_mix);
    if (b.logtab) (void)hipFree(b.logtab);
    if (b.log_prior) (void)hipFree(b.log_prior);
    b = AmDevBuf();
}

// max_blocks > 0 bounds the grid (the kernel strides over the tiles): next to the search, a
// chip-filling scoring launch holds every wave slot for milliseconds and the latency-bound search
// kernels, which need slots for microseconds at a time, all but stop (measured: 25 ms of
// scoring cost the search 20 ms).  A bounded grid scores in the background instead.
// skip_unused: row_src marks unused rows with -1 in whole-tile runs (decode_wave's stream slots).
// used_row_tiles >= 0: the row tiles that are not skipped (else: all of them)
// rt_base (device, or null) / n_rt_list: score these row tiles (first rows) only - D = 39
static int launch_gmm(const jd_am *a, const AmDevBuf &b, const float *d_feats, const int *d_row_src, int n_rows,
                      float *d_ll, hipStream_t st, int max_blocks = 0, int skip_unused = 0, int used_row_tiles = -1,
                      const int *rt_base = nullptr, int n_rt_list = 0)
{
    if (n_rows <= 0) return JD_OK;
    if (a->hybrid) {
        const long long n = (long long)n_rows * a->n_gmm;
        hipLaunchKernelGGL(jd_hybrid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_feats, d_row_src, n_rows, b.log_prior,
                           a->n_gmm, d_ll);
        HIPCHK(hipGetLastError());
        return JD_OK;
    }
    const int rows_per_tile = (a->D == 39) ? GMM_ROWS2 : GMM_ROWS;
    if (rt_base && a->D != 39) return jd_fail(JD_EINVAL, "launch_gmm: tile lists are the D = 39 kernel's");
    const long long row_tiles = rt_base ? n_rt_list : (n_rows + rows_per_tile - 1) / rows_per_tile;
    long long tiles = row_tiles * ((a->n_gmm + GMM_GT - 1) / GMM_GT);
    // few rows (a streaming push, a tick of the broker): tiles of 16 states, four times as many and a quarter as long
    const bool small_tiles = a->D == 39 && (used_row_tiles >= 0 ? (long long)used_row_tiles * ((a->n_gmm + GMM_GT - 1) / GMM_GT) : tiles) < 1024;
    if (small_tiles) tiles = row_tiles * ((a->n_gmm + GMM_GT_SMALL - 1) / GMM_GT_SMALL);
    dim3 grid((unsigned)((max_blocks > 0 && tiles > max_blocks) ? max_blocks : tiles));
    if (a->D == 39 && b.fast && b.par_fast) {
        const size_t sm = (size_t)GMM_ROWS2 * std::max(39, GMM_GT + 1) * sizeof(float);
        if (small_tiles)
            hipLaunchKernelGGL(jd_gmm_fast39<GMM_GT_SMALL>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par_fast, b.det,
                               b.n_mix, a->n_gmm, a->max_mix, d_ll, skip_unused, rt_base, n_rt_list);
        else
            hipLaunchKernelGGL(jd_gmm_fast39<GMM_GT>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par_fast, b.det,
                               b.n_mix, a->n_gmm, a->max_mix, d_ll, skip_unused, rt_base, n_rt_list);
    } else if (a->D == 39) {
        const size_t sm = 130 * sizeof(JdLogTab) + 32 * sizeof(unsigned long long) + (size_t)GMM_ROWS2 * std::max(39, GMM_GT + 1) * sizeof(float);
        if (small_tiles)
            hipLaunchKernelGGL(jd_gmm_kernel39<GMM_GT_SMALL>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                               b.n_mix, a->n_gmm, a->max_mix, d_ll, skip_unused, b.logtab, rt_base, n_rt_list);
        else
            hipLaunchKernelGGL(jd_gmm_kernel39<GMM_GT>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                               b.n_mix, a->n_gmm, a->max_mix, d_ll, skip_unused, b.logtab, rt_base, n_rt_list);
    } else {
        const int dp = a->D | 1;
        const size_t sm = (size_t)(GMM_ROWS * dp + GMM_ROWS * (GMM_GT + 1)) * sizeof(float);
        hipLaunchKernelGGL(jd_gmm_kernel<0>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                           b.n_mix, a->n_gmm, a->max_mix, a->D, d_ll, skip_unused);
    }
    HIPCHK(hipGetLastError());
    return JD_OK;
}

static int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return jd_fail(JD_ENODEV, "no HIP device available (%s); juicer_amd has no CPU fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return jd_fail(JD_ENODEV, "HIP device %d out of range (have %d)", device, n);
    HIPCHK(hipSetDevice(device));
    return JD_OK;
}

static int score_frames_mode(const jd_am *a, int32_t device, int32_t mode, const float *frames, int32_t n_frames, float *out);
extern "C" int jd_am_score_frames(const jd_am *a, int32_t device, const float *frames, int32_t n_frames,
                                  float *out)
{
    return score_frames_mode(a, device, JD_SCORE_EXACT, frames, n_frames, out);
}
extern "C" int jd_am_score_frames_mode(const jd_am *a, int32_t device, int32_t mode, const float *frames, int32_t n_frames,
                                       float *out)
{
    return score_frames_mode(a, device, mode, frames, n_frames, out);
}
static int score_frames_mode(const jd_am *a, int32_t device, int32_t mode, const float *frames, int32_t n_frames, float *out)
{
    if (!a || !frames || !out || n_frames < 0) return jd_fail(JD_EINVAL, "jd_am_score_frames: bad argument");
    if (mode != JD_SCORE_EXACT && mode != JD_SCORE_FAST) return jd_fail(JD_EINVAL, "jd_am_score_frames_mode: mode %d (JD_SCORE_EXACT or JD_SCORE_FAST)", mode);
    if (mode == JD_SCORE_FAST && (a->D != 39 || a->hybrid)) return jd_fail(JD_EINVAL, "JD_SCORE_FAST: 39-dimensional GMM models only");
    int rc = check_device(device);
    if (rc) return rc;
    if (n_frames == 0) return JD_OK;
    AmDevBuf b;
    rc = upload_am_gmm(a, b);
    if (rc) return rc;
    if (mode == JD_SCORE_FAST) { rc = upload_am_fast(a, b); if (rc) return rc; b.fast = 1; }
    float *d_x = nullptr, *d_ll = nullptr;
    int *d_src = nullptr;
    std::vector<int> src((size_t)n_frames);
    for (int i = 0; i < n_frames; ++i) src[i] = i;
    HIPCHK(hipMalloc(&d_x, (size_t)n_frames * a->D * sizeof(float)));
    HIPCHK(hipMalloc(&d_ll, (size_t)n_frames * a->n_gmm * sizeof(float)));
    HIPCHK(hipMalloc(&d_src, (size_t)n_frames * sizeof(int)));
    HIPCHK(hipMemcpy(d_x, frames, (size_t)n_frames * a->D * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_src, src.data(), (size_t)n_frames * sizeof(int), hipMemcpyHostToDevice));
    rc = launch_gmm(a, b, d_x, d_src, n_frames, d_ll, 0);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_ll, (size_t)n_frames * a->n_gmm * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_x); (void)hipFree(d_ll); (void)hipFree(d_src);
    free_am_gmm(b);
    return JD_OK;
}
