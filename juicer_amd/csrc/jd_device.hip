// jd_device.hip - gfx950 (MI355X) kernels + decoder runtime of juicer_amd.
//
// Two kernels make up the hot path (reference: WFSTDecoderLite::processFrame,
// src/WFSTDecoderLite.cpp:311-372, and HTKFlatModels::calcGMMOutput,
// src/HTKFlatModels.cpp:226-262):
//
//  jd_gmm_kernel     companion kernel.  Diagonal-GMM log-likelihood of every
//                    tied state for a tile of 64 stream-frames.  One lane owns
//                    one frame (its 39-dim vector staged through LDS into
//                    registers); the tied state is wave-uniform, so its
//                    mean/inverse-variance stream arrives through the scalar
//                    cache and the per-lane work is pure VALU in the
//                    reference's operation order (no FMA contraction, no MFMA:
//                    elementwise + reduction).  The log-sum-exp over mixtures is
//                    the reference's sequential logAdd chain, evaluated per lane.
//
//  jd_search_kernel  persistent token-passing search: ONE workgroup owns ONE
//                    utterance stream and runs all frames of a chunk without
//                    returning to the host.  Per frame: (A) HMM-internal
//                    propagation over the active arc instances with beam /
//                    histogram pruning and ballot+scan compaction of the
//                    active list, (B) frontier expansion over the CSR arc table
//                    iterated to epsilon/tee closure, with 64-bit atomic-max
//                    Viterbi recombination into entry tokens and word-boundary
//                    Path records appended to a device arena.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "jd_internal.h"

#define LZ (-3.402823466e+38f)
#define NT 1024                 // threads of the search workgroup (16 waves)
#define GMM_ROWS 64             // stream-frames per GMM tile (one per lane)
#define GMM_GT 64               // tied states per GMM workgroup (16 per wave)
#define HIST_MAX_BINS 2048

struct __align__(16) Tok { float score, ac, lm; int path; };
struct __align__(16) PathRec { int prev, frame, label, pad0; float score, ac, lm, pad1; };

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return jd_fail(JD_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                           __FILE__, __LINE__);                                              \
    } while (0)

// ------------------------------------------------------------------ device utils

__device__ __forceinline__ unsigned f2o(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o)
{
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int rank_in(unsigned long long bal)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
}

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, ARM optimized-routines
// algorithm): N=32 table + cubic in double, rounded once to float.  Replicated
// so that device logAdd equals the host libm result bit for bit (verified on
// the host for all 1.2e8 floats in [-18.5, -1e-3]; see tests/test_expf.py).
__device__ __constant__ unsigned long long jd_exp2f_tab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};

__device__ __forceinline__ float jd_expf(float x)
{
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = InvLn2N * (double)x;
    double kd = z + SHIFT;
    unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= SHIFT;
    double r = z - kd;
    unsigned long long t = jd_exp2f_tab[ki & 31];
    t += ki << 47;
    double s = __longlong_as_double((long long)t);
    double p = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1.0;
    y = p * r2 + y;
    y = y * s;
    return (float)y;
}

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
                                                     float *__restrict__ ll)
{
    constexpr int DP = (DT > 0) ? (DT | 1) : 0;      // odd row stride: conflict-free per-lane rows
    extern __shared__ __align__(16) char smem[];
    const int dp = (DT > 0) ? DP : (D | 1);
    float *sx = (float *)smem;                        // [64][dp]
    float *so = sx + GMM_ROWS * dp;                   // [64][GMM_GT+1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * GMM_ROWS;
    const int g0 = blockIdx.y * GMM_GT;
    const int Dn = (DT > 0) ? DT : D;

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

// ---------------------------------------------------------------- search kernel

struct DecConst {
    // network (CSR in HBM)
    const int *row_ptr; const JdArc *arcs; const float *fin_w; int init_state;
    // models
    int G, max_n;
    const int *hmm_n, *hmm_tm, *hmm_gmm; const float *hmm_tee; const float *trP; const int *se32;
    // pruning (WFSTDecoderLite ctor, WFSTDecoderLite.cpp:38-82)
    float start_win, emit_win, end_win, word_win;
    int max_hyps, hist_min, hist_max, hist_nbins;
    // arena capacities (per stream)
    int cap_slots, cap_items, cap_paths;
};

enum { ST_EMIT = 0, ST_END, ST_MODELS, ST_PEMIT, ST_PEND, ST_ARCS, ST_PATHS, ST_INSTS, ST_N };

struct StreamDev {
    // persistent scalars
    int par;            // token-half parity: tokens of slot s live at tok[(s*2+par)*max_n ..]
    int lst;            // which active list is current
    int n_act;          // entries in the current active list
    int hw;             // slot high-water mark
    int n_free;         // entries on the free-slot stack
    int n_paths;
    int frame;          // next frame to process
    int error;
    int needs_init;
    int hist_count_unused;
    float best_emit;    // bestEmitScore left by the previous frame
    float pad0;
    Tok best_final;     // bestFinalToken of the last processed frame
    long long st[ST_N];
    // arenas
    Tok *tok; int *slot_arc; int *slot_hmm; int *act[2]; int *free_stk; int *waste;
    unsigned long long *ekey; int *map;
    Tok *item_tok; int *item_arc; Tok *cand_tok; int *cand_arc;
    PathRec *paths; int *hist;
    // result of jd_finish_kernel
    int res_n; int *res_label; int *res_time; float *res_score, *res_ac, *res_lm; int res_cap;
};

__device__ __forceinline__ Tok null_tok() { Tok t; t.score = LZ; t.ac = LZ; t.lm = LZ; t.path = -1; return t; }

// exclusive block scan of a small per-thread count (sum over block < 2^31)
__device__ __forceinline__ int block_excl_scan(int v, int *sh_w, int &total)
{
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh_w[wid] = x;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int s = sh_w[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

struct FrameShared {
    int wsum[32];
    unsigned long long wsum3[NT / 64];
    unsigned best;                      // ordered-uint bestEmitScore of this frame
    int n_items, n_paths, n_alloc, n_waste, n_newact, err, new_live;
    unsigned long long final_key;
    float emit_th;
    int hist[HIST_MAX_BINS];
};

// Frontier expansion to epsilon/tee closure + entry-token resolve.
// propagateToken (WFSTDecoderLite.cpp:491-605) for every item in item[0, n0):
// an item is a token that has just traversed arc item_arc (or -1 = the NULL
// transition of recognitionStart, :226).
__device__ void expand_and_resolve(const DecConst &C, StreamDev &S, FrameShared &F, char *sh_raw, int frame,
                                   float endTh, float wordTh, int par_next, int *act_next, int nB, int n0,
                                   int nfree0, int hw0, int *cnt)
{
    const int tid = threadIdx.x, lane = lane_id();
    const int MN = C.max_n;
    Tok *sh_itok = (Tok *)sh_raw;                     // [NT]
    int *sh_off = (int *)(sh_raw + NT * sizeof(Tok)); // [NT+1]
    int *sh_rs = sh_off + NT + 8;                     // [NT]
    const float INF = __builtin_inff();

    int r0 = 0, r1 = n0;
    while (r1 > r0) {
        for (int cb = r0; cb < r1; cb += NT) {
            // ---- P1: arrive at the arc: word boundary (:497-509), final state (:513-520)
            const int i = cb + tid;
            int deg = 0, rs = 0;
            Tok t = null_tok();
            if (i < r1) {
                t = S.item_tok[i];
                const int a = S.item_arc[i];
                int state = C.init_state;
                if (a >= 0) {
                    const JdArc A = C.arcs[a];
                    if (A.out != 0) {
                        int p = atomicAdd(&F.n_paths, 1);
                        if (p < C.cap_paths) {
                            PathRec pr;
                            pr.prev = t.path; pr.frame = frame; pr.label = A.out; pr.pad0 = 0;
                            pr.score = t.score; pr.ac = t.ac; pr.lm = t.lm; pr.pad1 = 0.0f;
                            S.paths[p] = pr;
                            t.path = p;
                            S.item_tok[i].path = p;
                            cnt[ST_PATHS]++;
                        } else F.err = JD_ENOMEM;
                    }
                    const float fw = C.fin_w[A.to];
                    if (fw < INF) {
                        const float c = t.score + fw;
                        if (c > LZ) atomicMax(&F.final_key, ((unsigned long long)f2o(c) << 32) | (unsigned)i);
                    }
                    state = A.to;
                }
                rs = C.row_ptr[state];
                deg = C.row_ptr[state + 1] - rs;
            }
            sh_itok[tid] = t;
            sh_rs[tid] = rs;
            int total;
            const int off = block_excl_scan(deg, F.wsum, total);
            sh_off[tid] = off;
            if (tid == 0) sh_off[NT] = total;
            __syncthreads();
            // ---- P2: one (item, out-arc) pair per thread, load balanced over the flattened range
            for (int v0 = 0; v0 < total; v0 += NT) {
                const int v = v0 + tid;
                const bool act = v < total;
                bool mk = false;
                Tok u = null_tok();
                int ub = -1;
                if (act) {
                    int lo = 0, hi = NT;
                    while (hi - lo > 1) {
                        int mid = (lo + hi) >> 1;
                        if (sh_off[mid] <= v) lo = mid; else hi = mid;
                    }
                    const int j = lo;
                    const int b = sh_rs[j] + (v - sh_off[j]);
                    const Tok tj = sh_itok[j];
                    const int ii = cb + j;
                    const JdArc B = C.arcs[b];
                    cnt[ST_ARCS]++;
                    if (B.in == 0) {                                   // :533-540 epsilon input
                        u = tj;
                        u.score = tj.score + B.w;
                        u.lm = tj.lm + B.w;
                        mk = u.score > endTh;
                        ub = b;
                    } else {                                           // :544-582 entry-token recombination
                        const float ns = tj.score + B.w;
                        const int hm = B.in - 1;
                        int slot = __hip_atomic_load(&S.map[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (slot < 0) {
                            const int k = atomicAdd(&F.n_alloc, 1);
                            const int ns_ = (k < nfree0) ? S.free_stk[nfree0 - 1 - k] : hw0 + (k - nfree0);
                            if (ns_ >= C.cap_slots) {
                                F.err = JD_ENOMEM;
                            } else {
                                const int old = atomicCAS(&S.map[b], -1, ns_);
                                if (old == -1) {                       // attachNetInst :751-774
                                    slot = ns_;
                                    S.slot_arc[slot] = b;
                                    S.slot_hmm[slot] = hm;
                                    const int n = C.hmm_n[hm];
                                    Tok *tp = S.tok + ((size_t)slot * 2 + par_next) * MN;
                                    for (int q = 0; q < n; ++q) tp[q] = null_tok();
                                    const int pos = atomicAdd(&F.n_newact, 1);
                                    act_next[nB + pos] = slot;
                                } else {
                                    slot = old;
                                    const int wq = atomicAdd(&F.n_waste, 1);
                                    S.waste[wq] = ns_;
                                }
                            }
                        }
                        if (slot >= 0)
                            atomicMax(&S.ekey[slot], ((unsigned long long)f2o(ns) << 32) | (unsigned)ii);
                        const float tee = C.hmm_tee[hm];
                        if (tee > LZ) {                                // :584-600 tee model
                            const float ns2 = ns + tee;
                            u.score = ns2;
                            u.ac = tj.ac + tee;
                            u.lm = tj.lm + B.w;
                            u.path = tj.path;
                            mk = ns2 > ((B.out != 0) ? wordTh : endTh);
                            ub = b;
                        }
                    }
                }
                const unsigned long long bal = __ballot(mk);
                if (bal) {
                    int base = 0;
                    const int first = __ffsll((long long)bal) - 1;
                    if (lane == first) base = atomicAdd(&F.n_items, __popcll(bal));
                    base = __shfl(base, first);
                    if (mk) {
                        const int idx = base + rank_in(bal);
                        if (idx < C.cap_items) { S.item_tok[idx] = u; S.item_arc[idx] = ub; }
                        else F.err = JD_ENOMEM;
                    }
                }
            }
            __syncthreads();
        }
        r0 = r1;
        r1 = F.n_items < C.cap_items ? F.n_items : C.cap_items;
        __syncthreads();
    }

    // ---- resolve: winning candidate of every touched instance becomes its entry token
    const int n_act = nB + F.n_newact;
    for (int q = tid; q < n_act; q += NT) {
        const int slot = act_next[q];
        const unsigned long long key = atomicExch(&S.ekey[slot], 0ULL);
        if (key != 0ULL) {
            const float sc = o2f((unsigned)(key >> 32));
            if (sc > LZ) {
                const int ii = (int)(unsigned)(key & 0xffffffffULL);
                const Tok it = S.item_tok[ii];
                const JdArc B = C.arcs[S.slot_arc[slot]];
                Tok e;
                e.score = sc; e.ac = it.ac; e.lm = it.lm + B.w; e.path = it.path;
                S.tok[((size_t)slot * 2 + par_next) * MN] = e;
                atomicMax(&F.best, f2o(sc));                           // :572-573
                if (q >= nB) atomicAdd(&F.new_live, 1);
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void jd_search_kernel(DecConst C, StreamDev *streams, int s0,
                                                       const int *__restrict__ Tarr,
                                                       const float *__restrict__ ll, long long ll_stride,
                                                       int f0, int Fc)
{
    __shared__ __align__(16) char sh_raw[NT * JD_MAXN * 4];            // 32 KB, phase-aliased
    __shared__ FrameShared F;
    __shared__ long long sh_stats[ST_N];
    StreamDev &S = streams[s0 + blockIdx.x];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int MN = C.max_n;
    const int T = Tarr[s0 + blockIdx.x];
    int cnt[ST_N];
#pragma unroll
    for (int k = 0; k < ST_N; ++k) cnt[k] = 0;
    if (tid < ST_N) sh_stats[tid] = 0;

    // persistent scalars (uniform registers)
    int par = S.par, lst = S.lst, n_act = S.n_act, hw = S.hw, n_free = S.n_free;
    int frame = S.frame;
    float best_emit = S.best_emit;
    Tok best_final = S.best_final;
    const bool use_hist = C.max_hyps > 0;

    if (tid == 0) { F.err = 0; F.n_paths = S.n_paths; }
    for (int b = tid; b < C.hist_nbins; b += NT) F.hist[b] = use_hist ? S.hist[b] : 0;
    __syncthreads();

    // ------------------------------------------------ recognitionStart (:139-228)
    if (S.needs_init) {
        int *actc = S.act[lst];
        for (int q = tid; q < n_act; q += NT) S.map[S.slot_arc[actc[q]]] = -1;
        for (int b = tid; b < C.hist_nbins; b += NT) F.hist[b] = 0;
        if (tid == 0) {
            F.n_paths = 0; F.best = f2o(LZ); F.n_items = 1; F.n_alloc = 0; F.n_waste = 0; F.n_newact = 0;
            F.new_live = 0; F.final_key = 0ULL;
            Tok z; z.score = 0.0f; z.ac = 0.0f; z.lm = 0.0f; z.path = -1;
            S.item_tok[0] = z; S.item_arc[0] = -1;
        }
        __syncthreads();
        n_act = 0; hw = 0; n_free = 0; frame = 0;
        expand_and_resolve(C, S, F, sh_raw, 0, LZ, LZ, par ^ 1, S.act[lst ^ 1], 0, 1, 0, 0, cnt);
        n_act = F.n_newact;
        hw = F.n_alloc;
        // wasted slots go straight to the free stack
        for (int q = tid; q < F.n_waste; q += NT) S.free_stk[q] = S.waste[q];
        n_free = F.n_waste;
        best_emit = o2f(F.best);
        best_final = null_tok();
        par ^= 1; lst ^= 1;
        __syncthreads();
    }

    const int fend = (f0 + Fc < T) ? f0 + Fc : T;
    for (; frame < fend && F.err == 0; ++frame) {
        const float *llrow = ll + (size_t)blockIdx.x * ll_stride + (size_t)(frame - f0) * C.G;
        int *act_cur = S.act[lst], *act_next = S.act[lst ^ 1];
        // ---- thresholds (:318-339)
        const float normalise = (best_emit > LZ) ? best_emit : 0.0f;
        if (use_hist) {
            if (wid == 0) {                                            // Histogram::calcThresh, Histogram.cpp:134-158
                const int nb = C.hist_nbins, K = (nb + 63) >> 6;
                const int hi = nb - 1 - lane * K;
                int sum = 0;
                for (int k = 0; k < K; ++k) { int b = hi - k; if (b >= 0) sum += F.hist[b]; }
                int inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(inc, o); if (lane >= o) inc += y; }
                const int total = __shfl(inc, 63);
                float th;
                if (total <= C.max_hyps) th = (float)C.hist_min - 0.5f;
                else {
                    const unsigned long long m = __ballot(inc >= C.max_hyps);
                    const int L = __ffsll((long long)m) - 1;
                    int res = 0;
                    if (lane == L) {
                        int acc = inc - sum;
                        for (int k = 0; k < K; ++k) {
                            int b = hi - k;
                            if (b < 0) break;
                            acc += F.hist[b];
                            res = b;
                            if (acc >= C.max_hyps) break;
                        }
                    }
                    res = __shfl(res, L);
                    th = (float)(res + C.hist_min) - 0.5f;
                }
                th -= normalise;                                        // :325
                if (C.emit_win > 0.0f && th < -C.emit_win) th = -C.emit_win;   // :326-327
                if (lane == 0) F.emit_th = th;
            }
            __syncthreads();
        }
        const float emitTh = use_hist ? F.emit_th : (C.emit_win > 0.0f ? -C.emit_win : LZ);
        const float startTh = (C.start_win > 0.0f) ? (best_emit - C.start_win) : LZ;   // :337
        __syncthreads();
        if (use_hist) for (int b = tid; b < C.hist_nbins; b += NT) F.hist[b] = 0;     // :329
        if (tid == 0) {
            F.best = f2o(LZ); F.n_items = 0; F.n_alloc = 0; F.n_waste = 0; F.n_newact = 0;
            F.new_live = 0; F.final_key = 0ULL;                        // :316 bestFinalToken = nullToken
        }
        __syncthreads();

        // ---- phase A: doHMMInternalPropagation (:899-935) + HMMInternalPropagation (:376-484)
        float (*sh_sc)[NT] = (float (*)[NT])sh_raw;                    // old scores [state][thread]
        int nB = 0;                                                    // survivors written so far
        int ncand = 0;                                                 // live exit tokens so far
        for (int base = 0; base < n_act; base += NT) {
            const int q = base + tid;
            bool live = false, dead = false, has_exit = false;
            int slot = -1, a = -1;
            Tok ex = null_tok();
            if (q < n_act) {
                slot = act_cur[q];
                a = S.slot_arc[slot];
                const int h = S.slot_hmm[slot];
                const int n = C.hmm_n[h], tm = C.hmm_tm[h];
                const Tok *told = S.tok + ((size_t)slot * 2 + par) * MN;
                Tok *tnew = S.tok + ((size_t)slot * 2 + (par ^ 1)) * MN;
                for (int i = 0; i < n; ++i) sh_sc[i][tid] = told[i].score;
                {                                                      // :915-918 start-threshold pruning
                    const float es = sh_sc[0][tid];
                    if (es > LZ && es < startTh) sh_sc[0][tid] = LZ;
                }
                cnt[ST_INSTS]++;
                const float *trP = C.trP + (size_t)tm * MN * MN;
                const int *se = C.se32 + (size_t)tm * MN;
                Tok nw[JD_MAXN];
                nw[0] = null_tok();
                int n_live_emit = 0;
#pragma unroll
                for (int j = 1; j < JD_MAXN - 1; ++j) {
                    nw[j] = null_tok();
                    if (j < n - 1) {                                   // :387-424 emitting state j
                        const int sev = se[j];
                        const int st = sev & 0xffff, en = sev >> 16;
                        float tp = trP[st * MN + j];
                        float best = sh_sc[st][tid] + tp;
                        float btp = tp;
                        int bi = st;
                        for (int i = st + 1; i < en; ++i) {
                            tp = trP[i * MN + j];
                            const float tmp = sh_sc[i][tid] + tp;
                            if (tmp > best) { best = tmp; bi = i; btp = tp; }
                        }
                        float sc = best - normalise;                   // :408
                        if (sc > emitTh) {                             // :409
                            cnt[ST_PEMIT]++;
                            const Tok src = told[bi];
                            const float outp = llrow[C.hmm_gmm[(size_t)h * MN + j]];   // :411
                            Tok r;
                            r.score = sc + outp;
                            r.ac = (src.ac + btp) + outp;
                            r.lm = src.lm;
                            r.path = src.path;
                            nw[j] = r;
                            ++n_live_emit;
                            if (use_hist) {                            // Histogram::addScore, Histogram.cpp:64-100
                                const double ds = (double)r.score;
                                const int sci = (r.score < 0.0f) ? (int)(ds - 0.5) : (int)(ds + 0.5);
                                if (sci > C.hist_max) F.err = JD_EHIST;
                                else if (sci >= C.hist_min) atomicAdd(&F.hist[sci - C.hist_min], 1);
                            }
                            atomicMax(&F.best, f2o(r.score));          // :417-418
                        }
                    }
                }
                // exit state (:443-483) from the NEW emitting tokens
                {
                    const int sev = se[n - 1];
                    const int st = sev & 0xffff, en = sev >> 16;
                    bool first = true;
#pragma unroll
                    for (int i = 0; i < JD_MAXN - 1; ++i) {
                        if (i == st || (i > st && i < en)) {
                            const float tp = trP[i * MN + (n - 1)];
                            const float tmp = nw[i].score + tp;
                            if (first || tmp > ex.score) {
                                ex = nw[i];
                                ex.score = tmp;
                                ex.ac = nw[i].ac + tp;
                                first = false;
                            }
                        }
                    }
                    if (first || !(ex.score > LZ)) ex = null_tok();
                    has_exit = ex.score > LZ;
                }
                cnt[ST_EMIT] += n_live_emit;
                cnt[ST_END] += has_exit ? 1 : 0;
                live = n_live_emit > 0;
                dead = !live;
                if (live) {
#pragma unroll
                    for (int j = 0; j < JD_MAXN - 1; ++j)
                        if (j < n - 1) tnew[j] = nw[j];
                    tnew[n - 1] = null_tok();                          // :964 exit token leaves the instance
                }
            }
            // ballot + scan compaction of survivors / exit candidates / dead instances
            const unsigned long long bl = __ballot(live), be = __ballot(has_exit), bd = __ballot(dead);
            if (lane == 0)
                F.wsum3[wid] = (unsigned long long)__popcll(bl) | ((unsigned long long)__popcll(be) << 16) |
                               ((unsigned long long)__popcll(bd) << 32);
            __syncthreads();
            unsigned long long pre = 0, tot = 0;
            for (int w = 0; w < (NT >> 6); ++w) { const unsigned long long sv = F.wsum3[w]; if (w < wid) pre += sv; tot += sv; }
            const int pl = (int)(pre & 0xffff), pe = (int)((pre >> 16) & 0xffff), pd = (int)((pre >> 32) & 0xffff);
            if (live) act_next[nB + pl + rank_in(bl)] = slot;
            if (has_exit) {
                const int k = ncand + pe + rank_in(be);
                S.cand_tok[k] = ex;
                S.cand_arc[k] = a;
            }
            if (dead) {                                                // returnNetInst :777-797
                S.free_stk[n_free + pd + rank_in(bd)] = slot;
                S.map[a] = -1;
            }
            nB += (int)(tot & 0xffff);
            ncand += (int)((tot >> 16) & 0xffff);
            n_free += (int)((tot >> 32) & 0xffff);
            __syncthreads();
        }
        __syncthreads();
        const float bestA = o2f(F.best);
        const float endTh = (C.end_win > 0.0f) ? (bestA - C.end_win) : LZ;       // :349
        const float wordTh = (C.word_win > 0.0f) ? (bestA - C.word_win) : LZ;    // :350

        // ---- phase B round 0: doHMMExternalPropagation (:937-982) selects exit tokens
        for (int base = 0; base < ncand; base += NT) {
            const int k = base + tid;
            bool pass = false;
            Tok t = null_tok();
            int a = -1;
            if (k < ncand) {
                t = S.cand_tok[k];
                a = S.cand_arc[k];
                const int outl = C.arcs[a].out;
                pass = t.score > ((outl != 0) ? wordTh : endTh);       // :952-962
            }
            const unsigned long long bp = __ballot(pass);
            if (bp) {
                int b0 = 0;
                const int first = __ffsll((long long)bp) - 1;
                if (lane == first) b0 = atomicAdd(&F.n_items, __popcll(bp));
                b0 = __shfl(b0, first);
                if (pass) {
                    const int idx = b0 + rank_in(bp);
                    if (idx < C.cap_items) { S.item_tok[idx] = t; S.item_arc[idx] = a; cnt[ST_PEND]++; }
                    else F.err = JD_ENOMEM;
                }
            }
        }
        __syncthreads();
        const int n0 = F.n_items < C.cap_items ? F.n_items : C.cap_items;
        const int nfree0 = n_free, hw0 = hw;
        expand_and_resolve(C, S, F, sh_raw, frame, endTh, wordTh, par ^ 1, act_next, nB, n0, nfree0, hw0, cnt);

        // ---- frame epilogue (uniform)
        const int k_alloc = F.n_alloc, n_waste = F.n_waste;
        const int from_free = k_alloc < nfree0 ? k_alloc : nfree0;
        n_free = nfree0 - from_free;
        hw = hw0 + (k_alloc - from_free);
        for (int q = tid; q < n_waste; q += NT) S.free_stk[n_free + q] = S.waste[q];
        n_free += n_waste;
        n_act = nB + F.n_newact;
        best_emit = o2f(F.best);
        {
            const unsigned long long key = F.final_key;
            if (key != 0ULL) {
                const int ii = (int)(unsigned)(key & 0xffffffffULL);
                const Tok it = S.item_tok[ii];
                const float fw = C.fin_w[C.arcs[S.item_arc[ii]].to];
                best_final.score = o2f((unsigned)(key >> 32));
                best_final.ac = it.ac;
                best_final.lm = it.lm + fw;
                best_final.path = it.path;
            } else best_final = null_tok();
        }
        if (tid == 0) cnt[ST_MODELS] += nB + F.new_live;               // :981 totalActiveModels
        par ^= 1; lst ^= 1;
        __syncthreads();
    }

    // ---- write back persistent state
#pragma unroll
    for (int k = 0; k < ST_N; ++k) {
        long long v = cnt[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0 && v) atomicAdd((unsigned long long *)&sh_stats[k], (unsigned long long)v);
    }
    if (use_hist) for (int b = tid; b < C.hist_nbins; b += NT) S.hist[b] = F.hist[b];
    __syncthreads();
    if (tid == 0) {
        const bool did_init = S.needs_init != 0;
        S.par = par; S.lst = lst; S.n_act = n_act; S.hw = hw; S.n_free = n_free;
        S.n_paths = F.n_paths < C.cap_paths ? F.n_paths : C.cap_paths;
        S.frame = frame; S.best_emit = best_emit; S.best_final = best_final;
        if (F.err) S.error = F.err;
        S.needs_init = 0;
        for (int k = 0; k < ST_N; ++k) S.st[k] = (did_init ? 0 : S.st[k]) + sh_stats[k];
    }
}

// recognitionFinish (:230-309): walk the Path chain of bestFinalToken.
__global__ void jd_finish_kernel(StreamDev *streams, int s0, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    StreamDev &S = streams[s0 + s];
    const Tok best = S.best_final;
    if (!(best.score > LZ) || S.frame == 0) { S.res_n = -1; return; }
    int k = 0;
    for (int p = best.path; p >= 0; p = S.paths[p].prev) {
        if (k < S.res_cap) {
            const PathRec pr = S.paths[p];
            S.res_label[k] = pr.label; S.res_time[k] = pr.frame;
            S.res_score[k] = pr.score; S.res_ac[k] = pr.ac; S.res_lm[k] = pr.lm;
            if (k == 0) { S.res_score[0] = best.score; S.res_ac[0] = best.ac; S.res_lm[0] = best.lm; }   // :293-300
        }
        ++k;
    }
    S.res_n = k;
}

// --------------------------------------------------------------- host runtime

struct AmDevBuf {
    float *par = nullptr, *det = nullptr; int *n_mix = nullptr;
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
    return JD_OK;
}

static void free_am_gmm(AmDevBuf &b)
{
    if (b.par) (void)hipFree(b.par);
    if (b.det) (void)hipFree(b.det);
    if (b.n_mix) (void)hipFree(b.n_mix);
    b = AmDevBuf();
}

static int launch_gmm(const jd_am *a, const AmDevBuf &b, const float *d_feats, const int *d_row_src, int n_rows,
                      float *d_ll, hipStream_t st)
{
    if (n_rows <= 0) return JD_OK;
    dim3 grid((n_rows + GMM_ROWS - 1) / GMM_ROWS, (a->n_gmm + GMM_GT - 1) / GMM_GT);
    const int dp = a->D | 1;
    const size_t sm = (size_t)(GMM_ROWS * dp + GMM_ROWS * (GMM_GT + 1)) * sizeof(float);
    if (a->D == 39)
        hipLaunchKernelGGL(jd_gmm_kernel<39>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                           b.n_mix, a->n_gmm, a->max_mix, a->D, d_ll);
    else
        hipLaunchKernelGGL(jd_gmm_kernel<0>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                           b.n_mix, a->n_gmm, a->max_mix, a->D, d_ll);
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

extern "C" int jd_am_score_frames(const jd_am *a, int32_t device, const float *frames, int32_t n_frames,
                                  float *out)
{
    if (!a || !frames || !out || n_frames < 0) return jd_fail(JD_EINVAL, "jd_am_score_frames: bad argument");
    int rc = check_device(device);
    if (rc) return rc;
    if (n_frames == 0) return JD_OK;
    AmDevBuf b;
    rc = upload_am_gmm(a, b);
    if (rc) return rc;
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

struct HostResult {
    std::vector<int32_t> label, time;
    std::vector<float> score, ac, lm;
};

struct jd_dec {
    const jd_net *net = nullptr;
    const jd_am *am = nullptr;
    int device = 0, max_streams = 0, block_size = 5;
    DecConst C{};
    AmDevBuf amb;
    // device copies of static data
    int *d_row_ptr = nullptr; JdArc *d_arcs = nullptr; float *d_fin_w = nullptr;
    int *d_hmm_n = nullptr, *d_hmm_tm = nullptr, *d_hmm_gmm = nullptr, *d_se32 = nullptr;
    float *d_hmm_tee = nullptr, *d_trP = nullptr;
    // per-stream state
    StreamDev *d_streams = nullptr;
    std::vector<StreamDev> h_streams;          // host mirror (pointers + scalars)
    std::vector<void *> allocs;
    bool arenas_ready = false;
    int64_t cap_slots = 0, cap_paths = 0, cap_items = 0;
    int res_cap = 8192;
    // chunked pipeline
    int Fc = 128;
    float *d_ll[2] = {nullptr, nullptr};
    int *d_row_src = nullptr; size_t row_src_cap = 0;
    int *d_T = nullptr;
    hipStream_t s_gmm = nullptr, s_search = nullptr;
    hipEvent_t ev_gmm[2] = {nullptr, nullptr}, ev_search[2] = {nullptr, nullptr};
    // streaming API state
    std::vector<int> stream_T;                 // frames pushed so far
    std::vector<int> stream_started;
    float *d_push = nullptr; size_t push_cap = 0;
    // results
    std::vector<HostResult> results;
    jd_timing timing{};
};

template <typename T>
static int dmalloc(jd_dec *d, T **p, size_t n)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess)
        return jd_fail(JD_EHIP, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    d->allocs.push_back(q);
    *p = (T *)q;
    return JD_OK;
}

template <typename T>
static int dupload(jd_dec *d, T **p, const T *src, size_t n)
{
    int rc = dmalloc(d, p, n);
    if (rc) return rc;
    HIPCHK(hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return JD_OK;
}

extern "C" void jd_dec_destroy(jd_dec *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    for (void *p : d->allocs) (void)hipFree(p);
    free_am_gmm(d->amb);
    for (int i = 0; i < 2; ++i) {
        if (d->d_ll[i]) (void)hipFree(d->d_ll[i]);
        if (d->ev_gmm[i]) (void)hipEventDestroy(d->ev_gmm[i]);
        if (d->ev_search[i]) (void)hipEventDestroy(d->ev_search[i]);
    }
    if (d->d_row_src) (void)hipFree(d->d_row_src);
    if (d->d_push) (void)hipFree(d->d_push);
    if (d->s_gmm) (void)hipStreamDestroy(d->s_gmm);
    if (d->s_search) (void)hipStreamDestroy(d->s_search);
    delete d;
}

extern "C" int jd_dec_create(jd_dec **out, const jd_net *net, const jd_am *am, float start_beam, float main_beam,
                             float end_beam, float word_beam, int32_t max_hyps, int32_t block_size,
                             int32_t device, int32_t max_streams)
{
    if (!out || !net || !am) return jd_fail(JD_EINVAL, "jd_dec_create: null argument");
    if (block_size < 1 || block_size > 20)      // HTKFlatModels::setBlockSize, HTKFlatModels.cpp:311-312
        return jd_fail(JD_EINVAL, "HTKFlatModels::setBlockSize fnBlock should be in [1, 20]");
    if (max_streams < 1) return jd_fail(JD_EINVAL, "jd_dec_create: max_streams < 1");
    if (net->max_in > am->n_hmm)
        return jd_fail(JD_EINVAL, "network input label %d exceeds the number of HMMs %d", net->max_in, am->n_hmm);
    if (am->max_n > JD_MAXN) return jd_fail(JD_EINVAL, "HMMs with more than %d states unsupported", JD_MAXN);
    int rc = check_device(device);
    if (rc) return rc;
    jd_dec *d = new jd_dec();
    d->net = net; d->am = am; d->device = device; d->max_streams = max_streams; d->block_size = block_size;
    DecConst &C = d->C;
    C.start_win = start_beam; C.emit_win = main_beam; C.end_win = end_beam; C.word_win = word_beam;
    C.max_hyps = max_hyps;
    C.hist_min = 0; C.hist_max = 0; C.hist_nbins = 0;
    if (max_hyps > 0) {                          // WFSTDecoderLite.cpp:76-82, Histogram.cpp:29-37
        float mn = (main_beam > 0.0) ? (float)(-main_beam - 800.0) : -1000.0f;
        C.hist_min = (int)(mn - 1.0);
        C.hist_max = (int)(200.0f + 1.0);
        C.hist_nbins = C.hist_max - C.hist_min + 1;
        if (C.hist_nbins > HIST_MAX_BINS) {
            delete d;
            return jd_fail(JD_EINVAL, "mainBeam %.1f needs %d histogram bins (> %d supported)", main_beam,
                           C.hist_nbins, HIST_MAX_BINS);
        }
    }
#define TRY(x) do { rc = (x); if (rc) { jd_dec_destroy(d); return rc; } } while (0)
    TRY(dupload(d, &d->d_row_ptr, net->row_ptr.data(), net->row_ptr.size()));
    TRY(dupload(d, &d->d_arcs, net->arcs.data(), net->arcs.size()));
    TRY(dupload(d, &d->d_fin_w, net->fin_w.data(), net->fin_w.size()));
    TRY(dupload(d, &d->d_hmm_n, am->hmm_n.data(), am->hmm_n.size()));
    TRY(dupload(d, &d->d_hmm_tm, am->hmm_tm.data(), am->hmm_tm.size()));
    TRY(dupload(d, &d->d_hmm_gmm, am->hmm_gmm.data(), am->hmm_gmm.size()));
    TRY(dupload(d, &d->d_hmm_tee, am->hmm_tee.data(), am->hmm_tee.size()));
    TRY(dupload(d, &d->d_trP, am->trP.data(), am->trP.size()));
    std::vector<int> se32((size_t)am->n_tm * am->max_n);
    for (size_t i = 0; i < se32.size(); ++i)
        se32[i] = ((int)am->se[i * 2] & 0xffff) | ((int)am->se[i * 2 + 1] << 16);
    TRY(dupload(d, &d->d_se32, se32.data(), se32.size()));
    TRY(upload_am_gmm(am, d->amb));
    C.row_ptr = d->d_row_ptr; C.arcs = d->d_arcs; C.fin_w = d->d_fin_w; C.init_state = net->init;
    C.G = am->n_gmm; C.max_n = am->max_n;
    C.hmm_n = d->d_hmm_n; C.hmm_tm = d->d_hmm_tm; C.hmm_gmm = d->d_hmm_gmm; C.hmm_tee = d->d_hmm_tee;
    C.trP = d->d_trP; C.se32 = d->d_se32;
    // default arena sizes: sized for 288 GB of HBM, not for frugality
    d->cap_slots = std::min<int64_t>(net->n_arcs + 1024, 1 << 19);
    d->cap_items = 1 << 18;
    d->cap_paths = 1 << 21;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&d->s_gmm, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&d->s_search, hipStreamNonBlocking)) != hipSuccess) {
        jd_dec_destroy(d);
        return jd_fail(JD_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 2; ++i) {
        (void)hipEventCreate(&d->ev_gmm[i]);
        (void)hipEventCreate(&d->ev_search[i]);
    }
    d->stream_T.assign((size_t)max_streams, 0);
    d->stream_started.assign((size_t)max_streams, 0);
    d->results.resize((size_t)max_streams);
#undef TRY
    *out = d;
    return JD_OK;
}

extern "C" int jd_dec_set_capacity(jd_dec *d, int64_t max_slots, int64_t max_paths, int64_t max_items)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_set_capacity: null");
    if (d->arenas_ready) return jd_fail(JD_ESTATE, "jd_dec_set_capacity: arenas already allocated");
    if (max_slots > 0) d->cap_slots = max_slots;
    if (max_paths > 0) d->cap_paths = max_paths;
    if (max_items > 0) d->cap_items = max_items;
    return JD_OK;
}

static int ensure_arenas(jd_dec *d)
{
    if (d->arenas_ready) return JD_OK;
    int rc = check_device(d->device);
    if (rc) return rc;
    const int B = d->max_streams, MN = d->am->max_n;
    d->C.cap_slots = (int)d->cap_slots; d->C.cap_items = (int)d->cap_items; d->C.cap_paths = (int)d->cap_paths;
    d->h_streams.assign((size_t)B, StreamDev());
    for (int s = 0; s < B; ++s) {
        StreamDev &S = d->h_streams[(size_t)s];
        memset(&S, 0, sizeof S);
        S.needs_init = 1;
        S.best_emit = LZ;
        S.best_final.score = LZ; S.best_final.ac = LZ; S.best_final.lm = LZ; S.best_final.path = -1;
#define A(p, n) do { rc = dmalloc(d, &(p), (size_t)(n)); if (rc) return rc; } while (0)
        A(S.tok, d->cap_slots * 2 * MN);
        A(S.slot_arc, d->cap_slots); A(S.slot_hmm, d->cap_slots);
        A(S.act[0], d->cap_slots); A(S.act[1], d->cap_slots);
        A(S.free_stk, d->cap_slots); A(S.waste, d->cap_slots);
        A(S.ekey, d->cap_slots); A(S.map, d->net->n_arcs);
        A(S.item_tok, d->cap_items); A(S.item_arc, d->cap_items);
        A(S.cand_tok, d->cap_slots); A(S.cand_arc, d->cap_slots);
        A(S.paths, d->cap_paths);
        A(S.hist, HIST_MAX_BINS);
        A(S.res_label, d->res_cap); A(S.res_time, d->res_cap);
        A(S.res_score, d->res_cap); A(S.res_ac, d->res_cap); A(S.res_lm, d->res_cap);
#undef A
        S.res_cap = d->res_cap;
        HIPCHK(hipMemset(S.ekey, 0, (size_t)d->cap_slots * sizeof(unsigned long long)));
        HIPCHK(hipMemset(S.map, 0xff, (size_t)d->net->n_arcs * sizeof(int)));
        HIPCHK(hipMemset(S.hist, 0, HIST_MAX_BINS * sizeof(int)));
    }
    rc = dmalloc(d, &d->d_streams, (size_t)B);
    if (rc) return rc;
    HIPCHK(hipMemcpy(d->d_streams, d->h_streams.data(), (size_t)B * sizeof(StreamDev), hipMemcpyHostToDevice));
    rc = dmalloc(d, &d->d_T, (size_t)B);
    if (rc) return rc;
    for (int i = 0; i < 2; ++i)
        HIPCHK(hipMalloc(&d->d_ll[i], (size_t)B * d->Fc * d->am->n_gmm * sizeof(float)));
    HIPCHK(hipDeviceSynchronize());
    d->arenas_ready = true;
    return JD_OK;
}

__global__ void jd_mark_init_kernel(StreamDev *streams, int s0, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) { streams[s0 + s].needs_init = 1; streams[s0 + s].error = 0; }
}

// mark streams [s0, s0+n) for re-initialisation (IDecoder::init)
static int mark_init(jd_dec *d, int s0, int n, hipStream_t st)
{
    hipLaunchKernelGGL(jd_mark_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d->d_streams, s0, n);
    HIPCHK(hipGetLastError());
    return JD_OK;
}

static int fetch_results(jd_dec *d, int s0, int n, jd_hyp *out, int out0)
{
    std::vector<StreamDev> hs((size_t)n);
    HIPCHK(hipMemcpy(hs.data(), d->d_streams + s0, (size_t)n * sizeof(StreamDev), hipMemcpyDeviceToHost));
    int first_err = JD_OK;
    for (int i = 0; i < n; ++i) {
        const StreamDev &S = hs[(size_t)i];
        HostResult &R = d->results[(size_t)(out0 + i)];
        jd_hyp &H = out[out0 + i];
        memset(&H, 0, sizeof H);
        if (S.error && first_err == JD_OK) {
            first_err = S.error;
            if (S.error == JD_EHIST) jd_fail(JD_EHIST, "Histogram::addScore - score > maxScore (stream %d)", s0 + i);
            else jd_fail(S.error, "stream %d: device arena overflow (slots %lld / items %lld / paths %lld): "
                         "raise jd_dec_set_capacity", s0 + i, (long long)d->cap_slots, (long long)d->cap_items,
                         (long long)d->cap_paths);
        }
        if (S.error) {      // arenas may be inconsistent after an abort: wipe them for the next init
            HIPCHK(hipMemset(S.ekey, 0, (size_t)d->cap_slots * sizeof(unsigned long long)));
            HIPCHK(hipMemset(S.map, 0xff, (size_t)d->net->n_arcs * sizeof(int)));
            const int zero = 0;
            HIPCHK(hipMemcpy((char *)(d->d_streams + s0 + i) + offsetof(StreamDev, n_act), &zero, sizeof(int),
                             hipMemcpyHostToDevice));
        }
        H.stats.n_frames = S.frame;
        H.stats.tot_active_emit_hyps = S.st[ST_EMIT];
        H.stats.tot_active_end_hyps = S.st[ST_END];
        H.stats.tot_active_models = S.st[ST_MODELS];
        H.stats.tot_proc_emit_hyps = S.st[ST_PEMIT];
        H.stats.tot_proc_end_hyps = S.st[ST_PEND];
        H.stats.tot_arcs_visited = S.st[ST_ARCS];
        H.stats.tot_paths = S.st[ST_PATHS];
        H.stats.tot_insts_in = S.st[ST_INSTS];
        H.stats.ties = 0;
        int k = S.res_n;
        if (k > d->res_cap) {
            if (first_err == JD_OK) first_err = jd_fail(JD_ENOMEM, "stream %d: hypothesis has %d words (> %d)", s0 + i, k, d->res_cap);
            k = d->res_cap;
        }
        H.n = S.res_n < 0 ? -1 : k;
        const size_t kk = (size_t)std::max(k, 0);
        R.label.resize(kk); R.time.resize(kk); R.score.resize(kk); R.ac.resize(kk); R.lm.resize(kk);
        if (kk) {
            HIPCHK(hipMemcpy(R.label.data(), S.res_label, kk * sizeof(int), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(R.time.data(), S.res_time, kk * sizeof(int), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(R.score.data(), S.res_score, kk * sizeof(float), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(R.ac.data(), S.res_ac, kk * sizeof(float), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(R.lm.data(), S.res_lm, kk * sizeof(float), hipMemcpyDeviceToHost));
        }
        H.label = R.label.data(); H.time = R.time.data();
        H.score = R.score.data(); H.ac = R.ac.data(); H.lm = R.lm.data();
        if (kk) { H.tot_score = S.best_final.score; H.tot_ac = S.best_final.ac; H.tot_lm = S.best_final.lm; }
        else { H.tot_score = LZ; H.tot_ac = LZ; H.tot_lm = LZ; }      // DecHyp() defaults, DecHypHistPool.h
    }
    return first_err;
}

// Decode one wave of nb <= max_streams utterances held in device memory.
static int decode_wave(jd_dec *d, int nb, const float *d_feats, const int64_t *offs, hipStream_t user_stream)
{
    const int Fc = d->Fc, G = d->am->n_gmm;
    std::vector<int> T((size_t)nb);
    int maxT = 0;
    for (int u = 0; u < nb; ++u) {
        const int64_t t = offs[u + 1] - offs[u];
        if (t < 0 || t > 0x3fffffff) return jd_fail(JD_EINVAL, "utterance %d: bad frame count", u);
        T[(size_t)u] = (int)t;
        maxT = std::max(maxT, (int)t);
    }
    const int n_chunks = (maxT + Fc - 1) / Fc;
    // row -> source frame table for all chunks: row = (c*nb + u)*Fc + dt
    const size_t n_rows_all = (size_t)n_chunks * nb * Fc;
    if (n_rows_all > d->row_src_cap) {
        if (d->d_row_src) (void)hipFree(d->d_row_src);
        HIPCHK(hipMalloc(&d->d_row_src, std::max<size_t>(n_rows_all, 1) * sizeof(int)));
        d->row_src_cap = n_rows_all;
    }
    std::vector<int> row_src(n_rows_all);
    for (int c = 0; c < n_chunks; ++c)
        for (int u = 0; u < nb; ++u)
            for (int dt = 0; dt < Fc; ++dt) {
                const int f = c * Fc + dt;
                const int64_t src = offs[u] + f;
                if (src > 0x7fffffff) return jd_fail(JD_EINVAL, "more than 2^31 frames in one batch");
                row_src[((size_t)c * nb + u) * Fc + dt] = (f < T[(size_t)u]) ? (int)src : -1;
            }
    if (user_stream) HIPCHK(hipStreamSynchronize(user_stream));
    HIPCHK(hipMemcpyAsync(d->d_row_src, row_src.data(), n_rows_all * sizeof(int), hipMemcpyHostToDevice, d->s_gmm));
    HIPCHK(hipMemcpyAsync(d->d_T, T.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice, d->s_search));
    int rc = mark_init(d, 0, nb, d->s_search);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(d->s_gmm));
    HIPCHK(hipStreamSynchronize(d->s_search));

    std::vector<hipEvent_t> gs((size_t)n_chunks), ge((size_t)n_chunks), ss((size_t)n_chunks), se((size_t)n_chunks);
    for (int c = 0; c < n_chunks; ++c) {
        HIPCHK(hipEventCreate(&gs[(size_t)c])); HIPCHK(hipEventCreate(&ge[(size_t)c]));
        HIPCHK(hipEventCreate(&ss[(size_t)c])); HIPCHK(hipEventCreate(&se[(size_t)c]));
    }
    auto w0 = std::chrono::steady_clock::now();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c >= 2) HIPCHK(hipStreamWaitEvent(d->s_gmm, d->ev_search[buf], 0));   // ll buffer free again
        HIPCHK(hipEventRecord(gs[(size_t)c], d->s_gmm));
        rc = launch_gmm(d->am, d->amb, d_feats, d->d_row_src + (size_t)c * nb * Fc, nb * Fc, d->d_ll[buf], d->s_gmm);
        if (rc) return rc;
        HIPCHK(hipEventRecord(ge[(size_t)c], d->s_gmm));
        HIPCHK(hipEventRecord(d->ev_gmm[buf], d->s_gmm));
        HIPCHK(hipStreamWaitEvent(d->s_search, d->ev_gmm[buf], 0));
        HIPCHK(hipEventRecord(ss[(size_t)c], d->s_search));
        hipLaunchKernelGGL(jd_search_kernel, dim3(nb), dim3(NT), 0, d->s_search, d->C, d->d_streams, 0, d->d_T,
                           d->d_ll[buf], (long long)Fc * G, c * Fc, Fc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(se[(size_t)c], d->s_search));
        HIPCHK(hipEventRecord(d->ev_search[buf], d->s_search));
    }
    hipLaunchKernelGGL(jd_finish_kernel, dim3((nb + 63) / 64), dim3(64), 0, d->s_search, d->d_streams, 0, nb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->s_gmm));
    HIPCHK(hipStreamSynchronize(d->s_search));
    auto w1 = std::chrono::steady_clock::now();
    d->timing.total_ms += std::chrono::duration<double, std::milli>(w1 - w0).count();
    for (int c = 0; c < n_chunks; ++c) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, gs[(size_t)c], ge[(size_t)c]) == hipSuccess) d->timing.gmm_ms += ms;
        if (hipEventElapsedTime(&ms, ss[(size_t)c], se[(size_t)c]) == hipSuccess) d->timing.search_ms += ms;
        (void)hipEventDestroy(gs[(size_t)c]); (void)hipEventDestroy(ge[(size_t)c]);
        (void)hipEventDestroy(ss[(size_t)c]); (void)hipEventDestroy(se[(size_t)c]);
    }
    d->timing.gmm_launches += n_chunks;
    d->timing.search_launches += n_chunks;
    for (int u = 0; u < nb; ++u) d->timing.gmm_frames += T[(size_t)u];
    d->timing.gmm_states = G;
    return JD_OK;
}

extern "C" int jd_decode_batch_device(jd_dec *d, int32_t n_utts, const float *d_feats, const int64_t *offs,
                                      void *hip_stream, jd_hyp *out)
{
    if (!d || !offs || !out || n_utts < 0) return jd_fail(JD_EINVAL, "jd_decode_batch_device: bad argument");
    int rc = check_device(d->device);
    if (rc) return rc;
    rc = ensure_arenas(d);
    if (rc) return rc;
    if ((size_t)n_utts > d->results.size()) d->results.resize((size_t)n_utts);
    d->timing = jd_timing();
    int first_err = JD_OK;
    for (int u0 = 0; u0 < n_utts; u0 += d->max_streams) {
        const int nb = std::min(d->max_streams, n_utts - u0);
        rc = decode_wave(d, nb, d_feats, offs + u0, (hipStream_t)hip_stream);
        if (rc) return rc;
        rc = fetch_results(d, 0, nb, out, u0);
        if (rc && first_err == JD_OK) first_err = rc;
    }
    for (int s = 0; s < d->max_streams; ++s) { d->stream_started[(size_t)s] = 0; d->stream_T[(size_t)s] = 0; }
    return first_err;
}

extern "C" int jd_decode_batch(jd_dec *d, int32_t n_utts, const float *const *feats, const int32_t *n_frames,
                               jd_hyp *out)
{
    if (!d || !feats || !n_frames || !out || n_utts < 0) return jd_fail(JD_EINVAL, "jd_decode_batch: bad argument");
    int rc = check_device(d->device);
    if (rc) return rc;
    const int D = d->am->D;
    std::vector<int64_t> offs((size_t)n_utts + 1, 0);
    for (int u = 0; u < n_utts; ++u) {
        if (n_frames[u] < 0) return jd_fail(JD_EINVAL, "jd_decode_batch: negative frame count");
        offs[(size_t)u + 1] = offs[(size_t)u] + n_frames[u];
    }
    float *d_feats = nullptr;
    HIPCHK(hipMalloc(&d_feats, std::max<size_t>((size_t)offs[(size_t)n_utts] * D, 1) * sizeof(float)));
    for (int u = 0; u < n_utts; ++u)
        if (n_frames[u] > 0)
            HIPCHK(hipMemcpy(d_feats + (size_t)offs[(size_t)u] * D, feats[u], (size_t)n_frames[u] * D * sizeof(float),
                             hipMemcpyHostToDevice));
    rc = jd_decode_batch_device(d, n_utts, d_feats, offs.data(), nullptr, out);
    (void)hipFree(d_feats);
    return rc;
}

// ---- streaming API: IDecoder::init / processFrame / finish for one stream

extern "C" int jd_stream_init(jd_dec *d, int32_t s)
{
    if (!d || s < 0 || s >= d->max_streams) return jd_fail(JD_EINVAL, "jd_stream_init: bad stream");
    int rc = check_device(d->device);
    if (rc) return rc;
    rc = ensure_arenas(d);
    if (rc) return rc;
    rc = mark_init(d, s, 1, d->s_search);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(d->s_search));
    d->stream_T[(size_t)s] = 0;
    d->stream_started[(size_t)s] = 1;
    return JD_OK;
}

extern "C" int jd_stream_push(jd_dec *d, int32_t s, const float *frames, int32_t n_frames)
{
    if (!d || s < 0 || s >= d->max_streams || n_frames < 0 || (n_frames > 0 && !frames))
        return jd_fail(JD_EINVAL, "jd_stream_push: bad argument");
    if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_stream_push before jd_stream_init");
    int rc = check_device(d->device);
    if (rc) return rc;
    const int D = d->am->D, G = d->am->n_gmm, Fc = d->Fc;
    hipStream_t st = d->s_search;
    for (int done = 0; done < n_frames; done += Fc) {
        const int n = std::min(Fc, n_frames - done);
        if ((size_t)n * D > d->push_cap) {
            if (d->d_push) (void)hipFree(d->d_push);
            HIPCHK(hipMalloc(&d->d_push, (size_t)Fc * D * sizeof(float)));
            d->push_cap = (size_t)Fc * D;
        }
        HIPCHK(hipMemcpyAsync(d->d_push, frames + (size_t)done * D, (size_t)n * D * sizeof(float),
                              hipMemcpyHostToDevice, st));
        std::vector<int> src((size_t)Fc, -1);
        for (int i = 0; i < n; ++i) src[(size_t)i] = i;
        if ((size_t)Fc > d->row_src_cap) {
            if (d->d_row_src) (void)hipFree(d->d_row_src);
            HIPCHK(hipMalloc(&d->d_row_src, (size_t)Fc * sizeof(int)));
            d->row_src_cap = (size_t)Fc;
        }
        HIPCHK(hipMemcpyAsync(d->d_row_src, src.data(), (size_t)Fc * sizeof(int), hipMemcpyHostToDevice, st));
        const int f0 = d->stream_T[(size_t)s];
        const int Tnew = f0 + n;
        HIPCHK(hipMemcpyAsync(d->d_T + s, &Tnew, sizeof(int), hipMemcpyHostToDevice, st));
        rc = launch_gmm(d->am, d->amb, d->d_push, d->d_row_src, n, d->d_ll[0], st);
        if (rc) return rc;
        hipLaunchKernelGGL(jd_search_kernel, dim3(1), dim3(NT), 0, st, d->C, d->d_streams, s, d->d_T, d->d_ll[0],
                           (long long)Fc * G, f0, Fc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        d->stream_T[(size_t)s] = Tnew;
    }
    return JD_OK;
}

extern "C" int jd_stream_finish(jd_dec *d, int32_t s, jd_hyp *out)
{
    if (!d || s < 0 || s >= d->max_streams || !out) return jd_fail(JD_EINVAL, "jd_stream_finish: bad argument");
    if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_stream_finish before jd_stream_init");
    int rc = check_device(d->device);
    if (rc) return rc;
    if (d->stream_T[(size_t)s] == 0) {
        // init() immediately followed by finish(): run the pending init so state is defined
        const int zero = 0;
        HIPCHK(hipMemcpyAsync(d->d_T + s, &zero, sizeof(int), hipMemcpyHostToDevice, d->s_search));
        hipLaunchKernelGGL(jd_search_kernel, dim3(1), dim3(NT), 0, d->s_search, d->C, d->d_streams, s, d->d_T,
                           d->d_ll[0], 0LL, 0, d->Fc);
    }
    hipLaunchKernelGGL(jd_finish_kernel, dim3(1), dim3(64), 0, d->s_search, d->d_streams, s, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->s_search));
    // results of stream s are stored at result slot s
    std::vector<jd_hyp> tmp((size_t)d->max_streams);
    rc = fetch_results(d, s, 1, tmp.data(), s);
    *out = tmp[(size_t)s];
    return rc;
}

extern "C" int jd_dec_last_timing(const jd_dec *d, jd_timing *out)
{
    if (!d || !out) return jd_fail(JD_EINVAL, "jd_dec_last_timing: null");
    *out = d->timing;
    return JD_OK;
}
