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
#include <numeric>
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

// --------------------------------------------------------------- search kernels
//
// Lock-step frontier design.  All utterance streams of a batch advance one
// frame per "step"; each phase of WFSTDecoderLite::processFrame is ONE kernel
// whose work items are flattened over every stream (so the whole chip works on
// every phase, and kernel boundaries are the phase barriers):
//
//   k_boundary   per stream: epilogue of frame f-1 (list swap, bestFinal, stats)
//                + thresholds / histogram threshold of frame f         (:311-339)
//   k_phase_a    HMM-internal propagation over all active arc instances
//                (GS lanes per instance), ballot+scan compaction         (:376-484, :899-935)
//   k_expand<r>  frontier expansion rounds 0 and 1 over the CSR arc table,
//                64-bit atomic-max recombination per arc                 (:491-605, :937-982)
//   k_expand_tail  remaining epsilon/tee closure rounds (rare), one block per stream
//   k_resolve    winners become entry tokens; new instances attached     (:560-582, :751-774)

#define TEE_FLAG 0x40000000          // bit 30 of the device arc's in-label: the arc's HMM is a tee model
#define KT 256                       // threads per block of the small search kernels (tail)
#define KTB 256                      // threads per block of the flattened kernels (phase A, expand, resolve)
#ifndef JD_EG
#define JD_EG 16
#endif
#define EG JD_EG                     // lanes owning one frontier item (its arcs are pooled per wave)

struct DecConst {
    // network (CSR in HBM)
    const int *row_ptr; const JdArc *arcs; const float *fin_w; int init_state;
    const int *aux;     // per arc: {tmax0 bits, nStates|transMat<<8, hmm, g0, g1, g2, [g3, g4, g5], pad..}
    // models
    int G, max_n, n_tm;
    const int *hmm_n, *hmm_tm, *hmm_gmm; const float *hmm_tee; const float *hmm_tmax0;
    const float *trP; const int *se32;
    // pruning (WFSTDecoderLite ctor, WFSTDecoderLite.cpp:38-82)
    float start_win, emit_win, end_win, word_win;
    int max_hyps, hist_min, hist_max, hist_nbins;
    // arena capacities (per stream)
    int cap_slots, cap_items, cap_paths;
    int gc_threshold;   // collect Path records when more than this many are in use
    int inline_closure; // epsilon/tee closures are small (static bound): done inside k_expand_closure
    // diagnostics (jd_dec_debug_trace): per-block wall_clock64 stamps of one chosen frame
    long long *dbg; int dbg_frame;
};

enum { ST_EMIT = 0, ST_END, ST_MODELS, ST_PEMIT, ST_PEND, ST_ARCS, ST_PATHS, ST_INSTS, ST_N };

// An active arc instance (NetInst, WFSTDecoderLite.h:66-75) is ONE self-contained record,
// updated in place: header (arc, topology, tied-state ids) + its tokens.  With <= 5 HMM
// states it is exactly one 128-byte HBM line (REC_INTS = 32); up to 8 states take two lines.
//   ints [0..3]  = arc, nStates | transMat << 8, outLabel, toState
//   ints [4..7]  = g0, g1, g2, hmm          (tied-state ids of emitting states 1..3)
//   GS == 4: tokens at int offset 8;  GS == 8: ints [8..11] = g3, g4, g5, -, tokens at 12
template <int GS> struct RecLayout {
    static constexpr int REC_INTS = (GS == 4) ? 32 : 64;
    static constexpr int TOK_OFF = (GS == 4) ? 8 : 12;
};
// per-arc search state: recombination key of this frame + the instance slot (hook)
struct __align__(16) ArcState { unsigned long long key; int slot; int pad; };

// hot per-stream scalars.  Line 0 is read-mostly while the frame kernels run (written by
// k_boundary); every atomically updated counter sits on its own 128-byte line so that the
// L2 never serialises unrelated atomics (or readers of line 0) behind each other.
#define PK_SHIFT1 32
#define PK_MASK 0xffffffffULL
struct __align__(128) StreamCtl {
    // ---- line 0: persistent / per-frame constants
    int skipped_prev;   // instances whose creation was skipped last frame (still counted, see k_resolve)
    int lst;            // which active list is current
    int n_act;          // entries in the current active list
    int pad_a, pad_b;
    int frame;          // next frame to process
    int T;              // frames available
    int error, needs_init, active, started;
    float best_emit;    // bestEmitScore left by the previous frame (:321)
    float normalise, emitTh, startTh;
    int pad0[17];
    // ---- one line per atomic counter
    __align__(128) unsigned long long pkA;   // phase A: survivors (nB) | exit tokens (cnt0) << 32
    __align__(128) unsigned best;            // ordered-uint bestEmitScore of this frame
    __align__(128) int cnt1;                 // items produced by frontier round 0
    __align__(128) int cnt2;                 // items produced by frontier round 1
    __align__(128) int cnt_tail;             // items produced by the tail rounds
    __align__(128) int n_touched;
    __align__(128) int n_dirty;              // states whose closure key (skey[1]) is non-zero this frame (inline closure)
    __align__(128) int done_r;               // k_resolve blocks of this stream that have finished (fused frame boundary)
    __align__(128) int n_alloc;              // instances attached this frame (= new active entries)
    __align__(128) int n_skipped;            // hopeless instances not materialised this frame
    __align__(128) int n_paths;              // Path records in use at frame start (updated by k_boundary)
    __align__(128) int n_paths_extra;        // Path records taken by frontier rounds >= 1 this frame
    __align__(128) unsigned long long final_key;
    __align__(128) unsigned long long pkE;   // phase A: emit hyps processed | live emitting tokens << 32
    __align__(128) int fr[ST_N];             // per-frame work counters (flushed per block run)
    // ---- cold: touched by k_boundary / finish only
    __align__(128) Tok best_final;           // bestFinalToken of the last processed frame
    long long st[ST_N];
};
__device__ __forceinline__ int pk_nB(unsigned long long v) { return (int)(v & PK_MASK); }
__device__ __forceinline__ int pk_cnt0(unsigned long long v) { return (int)((v >> PK_SHIFT1) & PK_MASK); }

struct StreamDev {      // per-stream arenas (cold)
    int *rec[2];                      // the active lists ARE the instance records (RecLayout): list lst is
                                      // read by phase A, survivors + new instances are written to lst^1
    ArcState *ast;                    // per ARC: {best entry-token candidate of this frame, slot}
    unsigned long long *skey[2];      // per STATE: best frontier item arriving there (round parity)
    unsigned long long *skeyL;        // round 0 only: items whose arc carries a word label (own threshold)
    int *touched;                     // arcs whose ekey became non-zero this frame
    int *dirty;                       // inline closure: states whose skey[1] entry became non-zero this frame
    Tok *item_tok; int4 *item_info;   // frontier items: token + {arc, outLabel, toState, -}
    PathRec *paths; int *hist;
    PathRec *paths2; int *gc_idx;     // Path garbage collection: compaction target + mark / new-index array
    // result of jd_finish_kernel
    int res_n; int *res_label; int *res_time; float *res_score, *res_ac, *res_lm; int res_cap;
};

__device__ __forceinline__ Tok null_tok() { Tok t; t.score = LZ; t.ac = LZ; t.lm = LZ; t.path = -1; return t; }

// exclusive block scan (KT threads) of a per-thread count
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

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- frame boundary of one stream, executed by ONE wave: epilogue of the frame just processed
// (list swap, bestFinalToken, statistics) + start of the next (:311-339).
// Runs either as k_boundary or, fused, in the last k_resolve block of
// the stream - there the counters other workgroups have just updated (device-scope atomics) are
// read with agent-scope atomic loads; everything it writes is consumed by later kernels only.
#define WAVE_LDS_ORDER() asm volatile("" ::: "memory")
template <typename T> __device__ __forceinline__ T CL(T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void boundary_frame(const DecConst &C, StreamCtl &c, StreamDev &S, int lane, int *sh_hist)
{
    const bool use_hist = C.max_hyps > 0;
    // ---- epilogue of the frame processed in this step's predecessor kernels.  Every counter
    // lives on its own cache line: fetch them all first (independent loads in flight together).
    const int v_active = c.active, v_nalloc = CL(&c.n_alloc), v_nskip = CL(&c.n_skipped);
    const int v_skprev = c.skipped_prev, v_lst = c.lst, v_frame = c.frame, v_nact = c.n_act;
    const int v_npaths = c.n_paths + pk_cnt0(CL(&c.pkA)) + CL(&c.n_paths_extra);
    const int v_started = c.started, v_needs_init = c.needs_init, v_error = c.error, v_T = c.T;
    const unsigned long long v_pk = CL(&c.pkA), v_pe = CL(&c.pkE), v_fkey = CL(&c.final_key);
    const unsigned v_best = CL(&c.best);
    int v_fr = (lane < ST_N) ? CL(&c.fr[lane]) : 0;
    long long v_st = (lane < ST_N) ? c.st[lane] : 0;
    float best_emit = c.best_emit;
    int frame_now = v_frame;
    if (v_active == 1) {
        best_emit = o2f(v_best);
        frame_now = v_frame + 1;
        // per-frame statistics: lane k owns counter k
        if (lane == ST_MODELS) v_fr = pk_nB(v_pk) + v_nalloc + v_nskip;                     // :981
        if (lane == ST_INSTS) v_fr = v_nact + v_skprev;                                     // skipped ones die "now"
        if (lane == ST_END) v_fr = pk_cnt0(v_pk);
        if (lane == ST_PEMIT) v_fr = (int)(v_pe & 0xffffffffULL);
        if (lane == ST_EMIT) v_fr = (int)(v_pe >> 32);
        if (lane < ST_N) { c.st[lane] = v_st + v_fr; c.fr[lane] = 0; }
        if (lane == 0) {
            c.n_act = pk_nB(v_pk) + v_nalloc;
            c.best_emit = best_emit;
            if (v_fkey != 0ULL) {
                const int ii = (int)(unsigned)(v_fkey & 0xffffffffULL);
                const Tok it = S.item_tok[ii];
                const float fw = C.fin_w[S.item_info[ii].z];
                Tok bf;
                bf.score = o2f((unsigned)(v_fkey >> 32)); bf.ac = it.ac; bf.lm = it.lm + fw; bf.path = it.path;
                c.best_final = bf;
            } else c.best_final = null_tok();
            c.skipped_prev = v_nskip;
            c.lst = v_lst ^ 1;
            c.frame = frame_now;
            c.n_paths = v_npaths < C.cap_paths ? v_npaths : C.cap_paths;
        }
    }
    // ---- start of the next frame (:311-339)
    const bool go = v_started && !v_needs_init && v_error == 0 && frame_now < v_T;
    if (!go) { if (lane == 0 && v_active != 0) c.active = 0; return; }
    const float normalise = (best_emit > LZ) ? best_emit : 0.0f;                 // :321
    float emitTh = (C.emit_win > 0.0f ? -C.emit_win : LZ);                       // :331
    if (use_hist) {                                                              // Histogram::calcThresh, Histogram.cpp:134-158
        const int nb = C.hist_nbins;
        for (int b = lane; b < nb; b += 64) { sh_hist[b] = S.hist[b]; S.hist[b] = 0; }   // :329 reset
        WAVE_LDS_ORDER();                                                      // one wave: its LDS operations are ordered
        const int K = (nb + 63) >> 6;
        const int hi = nb - 1 - lane * K;
        int sum = 0;
        for (int k = 0; k < K; ++k) { int b = hi - k; if (b >= 0) sum += sh_hist[b]; }
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
                    acc += sh_hist[b];
                    res = b;
                    if (acc >= C.max_hyps) break;
                }
            }
            res = __shfl(res, L);
            th = (float)(res + C.hist_min) - 0.5f;
        }
        th -= normalise;                                                         // :325
        if (C.emit_win > 0.0f && th < -C.emit_win) th = -C.emit_win;             // :326-327
        emitTh = th;
    }
    if (lane == 0) {
        c.normalise = normalise; c.emitTh = emitTh;
        c.startTh = (C.start_win > 0.0f) ? (best_emit - C.start_win) : LZ;       // :337
        c.best = f2o(LZ); c.pkA = 0ULL; c.pkE = 0ULL;                            // :905
        c.cnt1 = 0; c.cnt2 = 0; c.cnt_tail = 0;
        c.n_alloc = 0; c.n_touched = 0; c.n_dirty = 0; c.n_skipped = 0; c.n_paths_extra = 0;
        c.final_key = 0ULL;                                                      // :316 bestFinalToken = nullToken
        c.done_r = 0;
        if (v_active != 1) c.active = 1;
    }
}

// ---- per-stream frame boundary: epilogue of the frame just processed + start of the next
// mode 0: normal step.  mode 1: recognitionStart (:139-228) part 1 (before the start-token
// expansion).  mode 2: recognitionStart part 2 (after it).
__global__ __launch_bounds__(64) void k_boundary(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0, int mode)
{
    const int s = s0 + blockIdx.x, lane = threadIdx.x;
    StreamCtl &c = ctl[s];
    StreamDev &S = streams[s];
    __shared__ int sh_hist[HIST_MAX_BINS];
    const bool use_hist = C.max_hyps > 0;

    if (mode == 1) {
        if (!c.needs_init) return;
        // drop whatever the previous utterance left behind
        const int rec_ints = (C.max_n <= 5) ? 32 : 64;
        const int *recs = S.rec[c.lst];
        for (int q = lane; q < c.n_act; q += 64) S.ast[recs[(size_t)q * rec_ints]].slot = -1;
        if (use_hist) for (int b = lane; b < C.hist_nbins; b += 64) S.hist[b] = 0;
        __syncthreads();
        if (lane == 0) {
            c.n_act = 0; c.n_paths = 0; c.frame = 0; c.error = 0;
            c.best_emit = LZ; c.normalise = 0.0f; c.emitTh = LZ; c.startTh = LZ;
            c.best = f2o(LZ); c.pkA = 1ULL << PK_SHIFT1;                         // cnt0 = 1: the start token
            c.pkE = 0ULL;
            c.cnt1 = 0; c.cnt2 = 0; c.cnt_tail = 0;
            c.n_alloc = 0; c.n_touched = 0; c.n_dirty = 0; c.final_key = 0ULL; c.n_skipped = 0; c.skipped_prev = 0;
            c.done_r = 0;
            c.n_paths_extra = 0;
            for (int k = 0; k < ST_N; ++k) { c.fr[k] = 0; c.st[k] = 0; }
            c.best_final = null_tok();
            Tok z; z.score = 0.0f; z.ac = 0.0f; z.lm = 0.0f; z.path = -1;       // :221-226
            S.item_tok[0] = z; S.item_info[0] = make_int4(-1, 0, 0, 0);
            c.active = 2;                                                        // 2 = initialising
        }
        return;
    }
    if (mode == 2) {
        if (c.active != 2) return;
        if (lane == 0) {
            c.n_act = c.n_alloc;
            c.best_emit = o2f(c.best);
            c.n_paths = c.n_paths + pk_cnt0(c.pkA) + c.n_paths_extra;
            c.n_paths_extra = 0;
            c.lst ^= 1;
            for (int k = 0; k < ST_N; ++k) { c.st[k] += c.fr[k]; c.fr[k] = 0; }
            c.st[ST_MODELS] = 0;
            c.needs_init = 0; c.active = 0;
            c.best_final = null_tok();
        }
        return;
    }

    boundary_frame(C, c, S, lane, sh_hist);
}

// ---- phase A: doHMMInternalPropagation (:899-935) + HMMInternalPropagation (:376-484)
// GS consecutive lanes own one arc instance; lane r updates emitting state r+1, lane GS-1
// builds the exit token from its neighbours' results (intra-group shuffles).  Blocks
// [sl*BPS, (sl+1)*BPS) serve stream sl; each strides over that stream's active list, so its
// work counters are flushed once per block.
template <int GS>
__global__ __launch_bounds__(KTB) void k_phase_a(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0, int BPS,
                                                const float *__restrict__ ll, long long ll_stride, int f0)
{
    __shared__ unsigned long long sh_w3[KTB / 64];
    __shared__ unsigned long long sh_pk;
    __shared__ unsigned long long sh_pe;                               // pemit | emit << 32 of this block
    __shared__ unsigned sh_bb;                                         // best emitting score of this block
    constexpr int PER = KTB / GS;                                      // instances per unit
    typedef RecLayout<GS> RL;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int MN = C.max_n;
    // static block -> stream assignment: no unit map, no search, one stream per block
    // stream-major block order: consecutive blocks (= consecutive XCDs) share a stream, so every
    // stream is spread over all 8 XCDs.  (Measured: packing a stream onto one XCD is 2.4x slower.)
    const int sl = blockIdx.x / BPS, j0 = blockIdx.x - sl * BPS;
    const int s = s0 + sl;
    StreamCtl &c = ctl[s];
    const long long t_start = wall_clock64();
    if (c.active != 1) return;
    const StreamDev &S = streams[s];
    const int units = (c.n_act + PER - 1) / PER;
    const bool dbg = C.dbg && c.frame == C.dbg_frame && tid == 0;
    if (dbg) { long long *d = C.dbg + (size_t)blockIdx.x * 4; d[0] = t_start; d[1] = wall_clock64(); d[2] = 0; d[3] = (j0 >= units) ? -1 : 0; }
    if (j0 >= units) return;
    if (tid == 0) { sh_pe = 0ULL; sh_bb = 0u; }
    __syncthreads();
    const int r = tid & (GS - 1), gb = lane & ~(GS - 1);
    const bool use_hist = C.max_hyps > 0;
    for (int u = j0; u < units; u += BPS) {
        const int n_act = c.n_act;
        const float normalise = c.normalise, emitTh = c.emitTh, startTh = c.startTh;
        const int *rec_cur = S.rec[c.lst];
        int *rec_next = S.rec[c.lst ^ 1];
        const float *llrow = ll + (size_t)sl * ll_stride + (size_t)(c.frame - f0) * C.G;
        const int q = u * PER + (tid / GS);
        const bool valid = q < n_act;
        bool emit_live = false, has_exit = false, pemit = false;
        int arc = -1, n = 0;
        Tok nw = null_tok(), ex = null_tok();
        int4 exinfo = make_int4(-1, 0, 0, 0);
        int4 h0 = make_int4(0, 0, 0, 0), h1 = h0, h2 = h0;
        const Tok *tk = nullptr;                                       // the instance's tokens in the current list
        const float *trP = C.trP;
        const int *se = C.se32;
        if (valid) {
            // the active list IS the record array: instance q of this frame sits at record q
            const int *rec = rec_cur + (size_t)q * RL::REC_INTS;
            h0 = *(const int4 *)rec; h1 = *(const int4 *)(rec + 4);
            if (GS == 8) h2 = *(const int4 *)(rec + 8);
            arc = h0.x;
            n = h0.y & 0xff;
            const int tm = h0.y >> 8;
            tk = (const Tok *)(rec + RL::TOK_OFF);
            // speculative: left-to-right HMMs read states r and r+1; issued together with the header
            const Tok spec0 = tk[r], spec1 = tk[(r + 1 < MN) ? r + 1 : r];
            trP = C.trP + (size_t)tm * MN * MN;
            se = C.se32 + (size_t)tm * MN;
            exinfo = make_int4(arc, h0.z, h0.w, 0);
            const int j = r + 1;
            if (j < n - 1) {                                           // :387-424 emitting state j
                int gmj;
                if (GS == 4) gmj = (r == 0) ? h1.x : (r == 1) ? h1.y : h1.z;
                else {
                    gmj = (r == 0) ? h1.x : (r == 1) ? h1.y : (r == 2) ? h1.z : (r == 3) ? h2.x : (r == 4) ? h2.y : h2.z;
                }
                const float outp = llrow[gmj];                         // :411
                const int sev = se[j];
                const int st = sev & 0xffff, en = sev >> 16;
                Tok src = (st == r) ? spec0 : (st == r + 1) ? spec1 : tk[st];
                if (st == 0 && src.score > LZ && src.score < startTh) src = null_tok();   // :915-918
                float btp = trP[st * MN + j];
                float best = src.score + btp;
                for (int i = st + 1; i < en; ++i) {
                    const Tok cnd = (i == r) ? spec0 : (i == r + 1) ? spec1 : tk[i];
                    const float tp = trP[i * MN + j];
                    const float tmp = cnd.score + tp;
                    if (tmp > best) { best = tmp; btp = tp; src = cnd; }
                }
                const float sc = best - normalise;                     // :408
                if (sc > emitTh) {                                     // :409
                    pemit = true;
                    nw.score = sc + outp;
                    nw.ac = (src.ac + btp) + outp;
                    nw.lm = src.lm;
                    nw.path = src.path;
                    emit_live = true;
                    if (use_hist) {                                    // Histogram::addScore, Histogram.cpp:64-100
                        const double ds = (double)nw.score;
                        const int sci = (nw.score < 0.0f) ? (int)(ds - 0.5) : (int)(ds + 0.5);
                        if (sci > C.hist_max) c.error = JD_EHIST;
                        else if (sci >= C.hist_min) atomicAdd(&S.hist[sci - C.hist_min], 1);
                    }
                }
            }
        }
        long long *dfx = C.dbg ? C.dbg + ((size_t)131072 + blockIdx.x) * 4 : nullptr;
        if (dbg && u == j0) dfx[0] = wall_clock64();                   // after loads + per-state compute
        // exit state (:443-483): lane GS-1 of the group reads the NEW tokens of its neighbours
        {
            int st = 0, en = 0;
            const bool is_exit_lane = valid && (r == GS - 1);
            if (is_exit_lane) { const int sev = se[n - 1]; st = sev & 0xffff; en = sev >> 16; }
            bool first = true;
#pragma unroll
            for (int i = 1; i < GS; ++i) {
                Tok ti;
                ti.score = __shfl(nw.score, gb + i - 1);
                ti.ac = __shfl(nw.ac, gb + i - 1);
                ti.lm = __shfl(nw.lm, gb + i - 1);
                ti.path = __shfl(nw.path, gb + i - 1);
                if (is_exit_lane && (i == st || (i > st && i < en))) {
                    const float tp = trP[i * MN + (n - 1)];
                    const float tmp = ti.score + tp;
                    if (first || tmp > ex.score) {
                        ex = ti;
                        ex.score = tmp;
                        ex.ac = ti.ac + tp;
                        first = false;
                    }
                }
            }
            if (first || !(ex.score > LZ)) ex = null_tok();
            has_exit = ex.score > LZ;
        }
        const unsigned long long bemit = __ballot(emit_live);
        const bool slot_live = ((bemit >> gb) & ((1ull << GS) - 1ull)) != 0ull;
        const bool live = valid && r == 0 && slot_live;
        const bool dead = valid && r == 0 && !slot_live;
        // block-level compaction: ONE packed returning atomic per unit
        const unsigned long long bl = __ballot(live), be = __ballot(has_exit);
        {
            unsigned mo = emit_live ? f2o(nw.score) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned y = __shfl_xor(mo, o); mo = y > mo ? y : mo; }
            const int c_pemit = __popcll(__ballot(pemit));
            if (lane == 0) {
                sh_w3[wid] = (unsigned long long)__popcll(bl) | ((unsigned long long)__popcll(be) << PK_SHIFT1);
                const unsigned long long pe = (unsigned long long)c_pemit | ((unsigned long long)__popcll(bemit) << 32);
                if (pe) atomicAdd(&sh_pe, pe);
                if (mo) atomicMax(&sh_bb, mo);
            }
        }
        if (dbg && u == j0) dfx[1] = wall_clock64();                   // before the barrier
        __syncthreads();
        unsigned long long pre = 0, tot = 0;
        for (int w = 0; w < (KTB >> 6); ++w) { const unsigned long long sv = sh_w3[w]; if (w < wid) pre += sv; tot += sv; }
        if (tid == 0) sh_pk = tot ? atomicAdd(&c.pkA, tot) : 0ULL;
        __syncthreads();
        if (dbg && u == j0) dfx[2] = wall_clock64();                   // after barrier + packed atomic
        const unsigned long long bs = sh_pk;
        // survivors are copied to compacted positions of the next list (header + new tokens, one
        // 128-byte line written by the group); the arc's hook follows the instance
        {
            int pos = live ? pk_nB(bs) + pk_nB(pre) + rank_in(bl) : -1;
            pos = __shfl(pos, gb);                                     // group leader -> whole group
            if (valid && pos >= 0) {
                if (pos >= C.cap_slots) c.error = -41;
                else {
                    int *dst = rec_next + (size_t)pos * RL::REC_INTS;
                    Tok *tn = (Tok *)(dst + RL::TOK_OFF);
                    if (r + 1 < n - 1) tn[r + 1] = nw;
                    if (r == GS - 1) {                                 // :428-436, :964 + header
                        *(int4 *)dst = h0; *(int4 *)(dst + 4) = h1;
                        if (GS == 8) *(int4 *)(dst + 8) = h2;
                        tn[0] = null_tok(); tn[n - 1] = null_tok();
                        S.ast[arc].slot = pos;
                    }
                }
            }
        }
        if (has_exit) {
            const int k = pk_cnt0(bs) + pk_cnt0(pre) + rank_in(be);
            if (k < C.cap_items) {
                S.item_tok[k] = ex; S.item_info[k] = exinfo;
                // bid for the destination state (state-level recombination, see expand_item); tokens
                // leaving word-labelled arcs face their own threshold (:952-962) -> own key class
                atomicMax((exinfo.y != 0 ? S.skeyL : S.skey[0]) + exinfo.z,
                          ((unsigned long long)f2o(ex.score) << 32) | (unsigned)k);
            } else c.error = -42;
        }
        if (dead) S.ast[arc].slot = -1;                                // returnNetInst :777-797
    }
    __syncthreads();
    if (tid == 0) {
        if (dbg) C.dbg[(size_t)blockIdx.x * 4 + 2] = wall_clock64();
        if (sh_pe) atomicAdd(&c.pkE, sh_pe);
        if (sh_bb) atomicMax(&c.best, sh_bb);                          // :417-418
        if (dbg) C.dbg[(size_t)blockIdx.x * 4 + 3] = wall_clock64();
    }
}

// ---- frontier expansion: propagateToken (WFSTDecoderLite.cpp:491-605).
// One frontier item = a token that has just traversed arc info.x (-1 = the NULL transition of
// recognitionStart).  A group of EG lanes owns one item: lane 0 records the word boundary /
// final state, all lanes walk the out-arcs of the destination state (coalesced 16-byte arc
// records).  State-level recombination: of all items that reached a state in one round only the
// best one (per threshold class) is expanded - every item would add the same arc weights, so by
// monotonicity of float addition no other item can win anything downstream.
// Block-level aggregation: Path records are allocated with one atomic per unit, first-touched
// arcs are staged in LDS and appended to the stream's list with one atomic per unit.
#define TBS_CAP 256                  // per-wave stage of first-touched arcs
#define ITS_CAP 64                   // per-wave stage of produced frontier items / inline-closure queue
#define DS_CAP 128                   // per-wave stage of dirty states (inline closure)
#define INLINE_CLOSURE_MAX (ITS_CAP / (64 / EG))   // largest closure per item the inline queue can hold
struct WaveStage { int buf[TBS_CAP]; int pfx[64 / EG + 1]; int qpos[ITS_CAP]; int dbuf[DS_CAP];
                   Tok itok[ITS_CAP]; int4 iinfo[ITS_CAP]; };
struct BlockStage { int np; int pb; WaveStage w[KTB / 64]; };
// Fill levels of the calling wave's stage.  They live in REGISTERS, computed identically by
// all 64 lanes from wave-uniform ballots: an LDS counter written by lane 0 and re-read by the
// others is a data race in the per-thread memory model (the compiler may forward a lane's own
// earlier load past another lane's store), and it measurably was one.
struct WaveFill {
    int n; int ni;                  // touched arcs / produced items staged
    int nd;                         // dirty states staged                       (inline closure)
    int qh, qt;                     // closure queue window in itok/iinfo/qpos   (inline closure)
    int cb, cl;                     // next free global item index of the wave's reserved chunk, indices left
};
// (WAVE_LDS_ORDER: LDS traffic between lanes of ONE wave needs no hardware fence - a wave's LDS
// operations execute in order - the compiler just must not move or forward accesses across it.)
// where the items produced by an expansion go: the next round's key array + the item counter
struct ItemSink { unsigned long long *sk_out; int *counter; int base; };

// The stages are wave-private (no barriers: waves never wait for each other) and are emptied
// when full and at the end of the kernel, NOT per unit: with 10^5..10^6 frontier items per
// stream-frame (wide beams) one same-address atomic per wave and unit would serialise the
// whole kernel behind the stream's two counters (~20 ns each).
__device__ __forceinline__ void stage_flush_touched(const DecConst &C, StreamCtl &c, const StreamDev &S, WaveStage &w,
                                                    WaveFill &f)
{
    const int lane = lane_id();
    const int n = f.n;
    if (n == 0) return;
    WAVE_LDS_ORDER();
    int base = 0;
    if (lane == 0) base = atomicAdd(&c.n_touched, n);
    base = __shfl(base, 0);
    for (int k = lane; k < n; k += 64) {
        if (base + k < C.cap_items) S.touched[base + k] = w.buf[k]; else c.error = -42;
    }
    WAVE_LDS_ORDER();
    f.n = 0;
}

__device__ __forceinline__ void stage_flush_items(const DecConst &C, StreamCtl &c, const StreamDev &S, WaveStage &w,
                                                  WaveFill &f, const ItemSink &sink)
{
    const int lane = lane_id();
    const int n = f.ni;
    if (n == 0) return;
    WAVE_LDS_ORDER();
    int base = 0;
    if (lane == 0) base = atomicAdd(sink.counter, n);
    base = __shfl(base, 0);
    for (int k = lane; k < n; k += 64) {
        const int pos = sink.base + base + k;
        if (pos < C.cap_items) {
            const Tok u = w.itok[k];
            const int4 ui = w.iinfo[k];
            S.item_tok[pos] = u; S.item_info[pos] = ui;
            // bid for the destination state (state-level recombination of the next round)
            atomicMax(sink.sk_out + ui.z, ((unsigned long long)f2o(u.score) << 32) | (unsigned)pos);
        } else c.error = -42;
    }
    WAVE_LDS_ORDER();
    f.ni = 0;
}

__device__ __forceinline__ void stage_touch(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &st,
                                            WaveFill &f, bool touch, int tb)
{
    const unsigned long long bt = __ballot(touch);
    if (!bt) return;
    WaveStage &w = st.w[threadIdx.x >> 6];
    const int cnt = __popcll(bt);
    if (f.n + cnt > TBS_CAP) stage_flush_touched(C, c, S, w, f);
    if (touch) w.buf[f.n + rank_in(bt)] = tb;
    f.n += cnt;
}

__device__ __forceinline__ void stage_item(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &st,
                                           WaveFill &f, const ItemSink &sink, bool mk, const Tok &u, const int4 &uinfo)
{
    const unsigned long long bm = __ballot(mk);
    if (!bm) return;
    WaveStage &w = st.w[threadIdx.x >> 6];
    const int cnt = __popcll(bm);
    if (f.ni + cnt > ITS_CAP) stage_flush_items(C, c, S, w, f, sink);
    if (mk) { const int k = f.ni + rank_in(bm); w.itok[k] = u; w.iinfo[k] = uinfo; }
    f.ni += cnt;
}

// every wave empties its own stages
__device__ __forceinline__ void stage_flush_block(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &st,
                                                  WaveFill &f, const ItemSink &sink)
{
    WaveStage &w = st.w[threadIdx.x >> 6];
    const int lane = lane_id();
    const int n = f.n, ni = f.ni;
    if ((n | ni) == 0) return;
    WAVE_LDS_ORDER();
    // both reservations in flight together: lane 0 the touched list, lane 1 the item list
    int base = 0;
    if (lane == 0 && n) base = atomicAdd(&c.n_touched, n);
    if (lane == 1 && ni) base = atomicAdd(sink.counter, ni);
    const int bt = __shfl(base, 0), bi = __shfl(base, 1);
    for (int k = lane; k < n; k += 64) {
        if (bt + k < C.cap_items) S.touched[bt + k] = w.buf[k]; else c.error = -42;
    }
    for (int k = lane; k < ni; k += 64) {
        const int pos = sink.base + bi + k;
        if (pos < C.cap_items) {
            const Tok u = w.itok[k];
            const int4 ui = w.iinfo[k];
            S.item_tok[pos] = u; S.item_info[pos] = ui;
            atomicMax(sink.sk_out + ui.z, ((unsigned long long)f2o(u.score) << 32) | (unsigned)pos);
        } else c.error = -42;
    }
    WAVE_LDS_ORDER();
    f.n = 0; f.ni = 0;
}

// ---- inline closure (DecConst::inline_closure).  When the graph's epsilon / tee closures are
// provably small (jd_dec_create bounds them), the wave that produces a closure item expands it
// itself, right after the unit that produced it, instead of handing it to another kernel:
// one launch replaces k_expand<0>, k_expand<1> and k_expand_tail.  State-level recombination
// becomes a RUNNING maximum on skey[1]: an item is expanded iff it is the best arrival at its
// state so far (checked when produced and again when taken from the queue), so the best one is
// always expanded and a state is expanded O(log arrivals) times instead of once - same results.
// Touched skey[1] entries are listed ("dirty") and zeroed by k_resolve.
__device__ __forceinline__ void stage_dirty(const DecConst &C, StreamCtl &c, const StreamDev &S, WaveStage &w,
                                            WaveFill &f, bool first, int state)
{
    const unsigned long long bf = __ballot(first);
    if (!bf) return;
    const int cnt = __popcll(bf);
    if (f.nd + cnt > DS_CAP) {
        WAVE_LDS_ORDER();
        int base = 0;
        if (lane_id() == 0) base = atomicAdd(&c.n_dirty, f.nd);
        base = __shfl(base, 0);
        for (int k = lane_id(); k < f.nd; k += 64) {
            if (base + k < C.cap_items) S.dirty[base + k] = w.dbuf[k]; else c.error = -42;
        }
        WAVE_LDS_ORDER();
        f.nd = 0;
    }
    if (first) w.dbuf[f.nd + rank_in(bf)] = state;
    f.nd += cnt;
}

__device__ __forceinline__ void closure_push(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &st,
                                             WaveFill &f, const ItemSink &sink, bool mk, const Tok &u, const int4 &uinfo)
{
    WaveStage &w = st.w[threadIdx.x >> 6];
    // cheap pre-filter: not better than the best arrival so far -> nothing downstream can win
    bool pass = false;
    unsigned so = 0;
    if (mk) {
        so = f2o(u.score);
        const unsigned long long cur = __hip_atomic_load(sink.sk_out + uinfo.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pass = so > (unsigned)(cur >> 32);
    }
    const unsigned long long bp = __ballot(pass);
    if (!bp) return;
    const int cnt = __popcll(bp);
    // global item indices (k_resolve / bestFinal find the token through them) from the wave's
    // reserved chunk: one atomic per 64 indices; unused indices are harmless holes
    if (f.cl < cnt) {
        int b = 0;
        if (lane_id() == 0) b = atomicAdd(sink.counter, 64);
        f.cb = sink.base + __shfl(b, 0); f.cl = 64;
    }
    const int pos = f.cb + rank_in(bp);
    f.cb += cnt; f.cl -= cnt;
    bool keep = false, first = false;
    if (pass) {
        if (pos < C.cap_items) {
            const unsigned long long key = ((unsigned long long)so << 32) | (unsigned)pos;
            const unsigned long long old = atomicMax(sink.sk_out + uinfo.z, key);
            keep = key > old; first = old == 0ULL;
        } else c.error = -42;
    }
    stage_dirty(C, c, S, w, f, first, uinfo.z);
    const unsigned long long bk = __ballot(keep);
    if (!bk) return;
    const int nk = __popcll(bk);
    if (f.qt + nk > ITS_CAP) { c.error = -42; return; }                // excluded by the static closure bound
    if (keep) {
        S.item_tok[pos] = u; S.item_info[pos] = uinfo;
        const int k = f.qt + rank_in(bk);
        w.itok[k] = u; w.iinfo[k] = uinfo; w.qpos[k] = pos;
    }
    f.qt += nk;
}

// arc walk of one wave.  The wave's 64/EG items pool their out-arcs: lane l takes arcs
// l, l+64, ... of the concatenated arc ranges and fetches the owning item's token from that
// item's lanes, so a state with thousands of out-arcs (trigram back-off / history states)
// occupies the whole wave instead of one EG-lane group, and items with few arcs share a pass.
// t / ii / rs / deg are uniform within an EG-lane group.
template <bool INLINE>
__device__ __forceinline__ void expand_arcs(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &stage,
                                            WaveFill &fill, const Tok &t, int ii, int rs, int deg, float endTh, float wordTh,
                                            const ItemSink &sink, int &n_arcs)
{
    constexpr int NGRP = 64 / EG;                                      // items per wave
    const int lane = lane_id();
    // exclusive prefix of the groups' degrees (group leaders carry deg, other lanes 0)
    int incl = ((lane & (EG - 1)) == 0) ? deg : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
    const int tot = __shfl(incl, 63);
    int *pfx = stage.w[threadIdx.x >> 6].pfx;                          // wave-private, NGRP + 1 entries
    WAVE_LDS_ORDER();
    if ((lane & (EG - 1)) == 0) pfx[lane / EG] = incl - deg;
    if (lane == 0) pfx[NGRP] = tot;
    WAVE_LDS_ORDER();
    for (int a0 = 0; a0 < tot; a0 += 64) {
        const int a = a0 + lane;
        int g = 0;                                                     // largest g with pfx[g] <= a
#pragma unroll
        for (int st = NGRP / 2; st > 0; st >>= 1) if (pfx[g + st] <= a) g += st;
        const int off = a - pfx[g];
        const int srcl = g * EG;
        Tok tg;
        tg.score = __shfl(t.score, srcl); tg.ac = __shfl(t.ac, srcl);
        tg.lm = __shfl(t.lm, srcl); tg.path = __shfl(t.path, srcl);
        const int iig = __shfl(ii, srcl), rsg = __shfl(rs, srcl);
        bool mk = false, touch = false;
        Tok u = null_tok();
        int4 uinfo = make_int4(-1, 0, 0, 0);
        int tb = -1;
        if (a < tot) {
            const int b = rsg + off;
            const JdArc Bk = C.arcs[b];
            ++n_arcs;
            const int inl = Bk.in & ~TEE_FLAG;
            if (inl == 0) {                                            // :533-540 epsilon input
                u = tg;
                u.score = tg.score + Bk.w;
                u.lm = tg.lm + Bk.w;
                mk = u.score > endTh;
                uinfo = make_int4(b, Bk.out, Bk.to, 0);
            } else {                                                   // :560-582 entry-token recombination
                const float ns = tg.score + Bk.w;
                const unsigned long long key = ((unsigned long long)f2o(ns) << 32) | (unsigned)iig;
                const unsigned long long old = atomicMax(&S.ast[b].key, key);
                touch = (old == 0ULL);
                tb = b;
                if (Bk.in & TEE_FLAG) {                                // :584-600 tee model
                    const float tee = C.hmm_tee[inl - 1];
                    const float ns2 = ns + tee;
                    u.score = ns2;
                    u.ac = tg.ac + tee;
                    u.lm = tg.lm + Bk.w;
                    u.path = tg.path;
                    mk = ns2 > ((Bk.out != 0) ? wordTh : endTh);
                    uinfo = make_int4(b, Bk.out, Bk.to, 0);
                }
            }
        }
        stage_touch(C, c, S, stage, fill, touch, tb);
        if (INLINE) closure_push(C, c, S, stage, fill, sink, mk, u, uinfo);
        else stage_item(C, c, S, stage, fill, sink, mk, u, uinfo);
    }
}

// One unit = KT/EG items of one stream.  All threads of the block call this.
//   sk_in_u / sk_in_l : per-state key arrays of this round (unlabelled / word-labelled class)
//   check_th          : apply the end/word threshold of doHMMExternalPropagation (:952-962) (round 0)
template <bool INLINE>
__device__ __forceinline__ void expand_unit(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &stage,
                                            WaveFill &fill, int frame, bool last_frame, bool path_direct, int path_base_extra,
                                            float endTh, float wordTh, bool check_th,
                                            bool have, int ii, unsigned long long *sk_in_u,
                                            unsigned long long *sk_in_l, const ItemSink &sink,
                                            int &n_arcs, int &n_paths_made,
                                            int &n_pend, long long *dbx = nullptr)
{
    const int lane = lane_id();
    const int er = lane & (EG - 1), eb = lane & ~(EG - 1);
    const float INF = __builtin_inff();
    Tok t = null_tok();
    int4 info = make_int4(-1, 0, 0, 0);
    int rs = 0, rs1 = 0;
    if (have) {
        info = S.item_info[ii];
        t = S.item_tok[ii];
    }
    // Path records (:497-509).  Round-0 items (exit tokens of phase A) own the record
    // n_paths + <their index>: no allocation at all (records of unlabelled / unexpanded tokens
    // stay unused; k_boundary advances n_paths by the number of exit tokens).  Items of later
    // rounds (rare: word labels on epsilon / tee arcs) reserve per block with one atomic.
    int p = -1;
    if (path_direct) {
        p = c.n_paths + ii;
    } else {
        const bool labelled = have && info.x >= 0 && info.y != 0 && er == 0;
        const unsigned long long bl = __ballot(labelled);
        int wb = 0;
        if (bl) {
            const int first = __ffsll((long long)bl) - 1;
            if (lane == first) wb = atomicAdd(&stage.np, __popcll(bl));          // LDS
            wb = __shfl(wb, first);
        }
        __syncthreads();
        if (threadIdx.x == 0) { const int np = stage.np; stage.pb = np ? atomicAdd(&c.n_paths_extra, np) : 0; }
        __syncthreads();
        p = labelled ? c.n_paths + path_base_extra + stage.pb + wb + rank_in(bl) : -1;
        p = __shfl(p, eb);
        __syncthreads();
        if (threadIdx.x == 0) stage.np = 0;
    }
    if (dbx && t.score != 12345.0f) dbx[0] = wall_clock64();          // item loaded
    if (have) {
        const int state = (info.x >= 0) ? info.z : C.init_state;
        rs = C.row_ptr[state];                                         // issued before the winner is known
        rs1 = C.row_ptr[state + 1];
        if (info.x >= 0) {
            if (check_th) {                                            // :952-962
                have = t.score > ((info.y != 0) ? wordTh : endTh);
                if (have && er == 0) ++n_pend;
            }
            unsigned long long *sk = ((info.y != 0) ? sk_in_l : sk_in_u) + info.z;
            const unsigned long long kv = __hip_atomic_load(sk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool winner = (unsigned)(kv & 0xffffffffULL) == (unsigned)ii && kv != 0ULL;
            if (winner && er == 0) __hip_atomic_store(sk, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have = have && winner;
        }
    }
    if (dbx && rs1 != -12345) dbx[1] = wall_clock64();                 // winner known, row bounds loaded
    int deg = 0;
    if (have) {
        if (info.x >= 0) {
            if (info.y != 0) {
                if (p < C.cap_paths) {
                    if (er == 0) {
                        PathRec pr;
                        pr.prev = t.path; pr.frame = frame; pr.label = info.y; pr.pad0 = 0;
                        pr.score = t.score; pr.ac = t.ac; pr.lm = t.lm; pr.pad1 = 0.0f;
                        S.paths[p] = pr;
                        S.item_tok[ii].path = p;
                        ++n_paths_made;
                    }
                    t.path = p;
                } else c.error = -43;
            }
            // :513-520 final state.  bestFinalToken is reset every frame (:316) and only read by
            // finish(), so it only has to be evaluated on the last frame that is available.
            if (er == 0 && last_frame) {
                const float fw = C.fin_w[info.z];
                if (fw < INF) {
                    const float cs = t.score + fw;
                    if (cs > LZ) atomicMax(&c.final_key, ((unsigned long long)f2o(cs) << 32) | (unsigned)ii);
                }
            }
        }
        deg = rs1 - rs;
    }
    if (dbx) dbx[2] = wall_clock64();                                  // path / final done
    expand_arcs<INLINE>(C, c, S, stage, fill, t, ii, rs, deg, endTh, wordTh, sink, n_arcs);
    if (dbx) dbx[3] = wall_clock64();                                  // arcs walked
}

// inline closure: the wave takes up to 64/EG items from its queue and expands them (which may
// queue more).  No block barriers: every wave of the block runs its own closure.
__device__ __forceinline__ void closure_step(const DecConst &C, StreamCtl &c, const StreamDev &S, BlockStage &stage,
                                             WaveFill &fill, int frame, bool last_frame, int path_base_extra,
                                             float endTh, float wordTh, const ItemSink &sink,
                                             int &n_arcs, int &n_paths_made)
{
    constexpr int NGRP = 64 / EG;
    const int lane = lane_id();
    const int er = lane & (EG - 1), eb = lane & ~(EG - 1);
    const float INF = __builtin_inff();
    WaveStage &w = stage.w[threadIdx.x >> 6];
    const int k = fill.qh + lane / EG;
    bool have = k < fill.qt;
    fill.qh = (fill.qh + NGRP < fill.qt) ? fill.qh + NGRP : fill.qt;
    Tok t = null_tok();
    int4 info = make_int4(-1, 0, 0, 0);
    int ii = 0, rs = 0, rs1 = 0;
    WAVE_LDS_ORDER();
    if (have) {
        t = w.itok[k]; info = w.iinfo[k]; ii = w.qpos[k];
        rs = C.row_ptr[info.z];
        rs1 = C.row_ptr[info.z + 1];
        const unsigned long long kv = __hip_atomic_load(sink.sk_out + info.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        have = (unsigned)(kv & 0xffffffffULL) == (unsigned)ii;         // still the best arrival at this state
    }
    // Path records of word labels on the arc just traversed (:497-509): wave-level reservation
    const bool labelled = have && info.y != 0 && er == 0;
    const unsigned long long bl = __ballot(labelled);
    int p = -1;
    if (bl) {
        const int first = __ffsll((long long)bl) - 1;
        int wb = 0;
        if (lane == first) wb = atomicAdd(&c.n_paths_extra, __popcll(bl));
        wb = __shfl(wb, first);
        p = labelled ? c.n_paths + path_base_extra + wb + rank_in(bl) : -1;
        p = __shfl(p, eb);
    }
    int deg = 0;
    if (have) {
        if (info.y != 0) {
            if (p < C.cap_paths) {
                if (er == 0) {
                    PathRec pr;
                    pr.prev = t.path; pr.frame = frame; pr.label = info.y; pr.pad0 = 0;
                    pr.score = t.score; pr.ac = t.ac; pr.lm = t.lm; pr.pad1 = 0.0f;
                    S.paths[p] = pr;
                    S.item_tok[ii].path = p;
                    ++n_paths_made;
                }
                t.path = p;
            } else c.error = -43;
        }
        if (er == 0 && last_frame) {                                   // :513-520
            const float fw = C.fin_w[info.z];
            if (fw < INF) {
                const float cs = t.score + fw;
                if (cs > LZ) atomicMax(&c.final_key, ((unsigned long long)f2o(cs) << 32) | (unsigned)ii);
            }
        }
        deg = rs1 - rs;
    }
    expand_arcs<true>(C, c, S, stage, fill, t, ii, rs, deg, endTh, wordTh, sink, n_arcs);
}

// frontier rounds, flattened over all streams.  ROUND 0 reads the live exit tokens written (and
// bid for their destination states) by phase A.  INLINE (round 0 only): every wave also runs the
// epsilon / tee closure of what it produced, so no further round is needed (k_expand_closure).
template <int ROUND, bool INLINE>
__device__ __forceinline__ void expand_body(const DecConst &C, StreamCtl *ctl, StreamDev *streams, int s0, int BPS)
{
    __shared__ BlockStage stage;
    __shared__ int sh_acc[3];                                          // ARCS, PATHS, PEND of this block
    constexpr int PER = KTB / EG;                                      // items per unit
    const int tid = threadIdx.x, lane = lane_id();
    // stream-major block order: consecutive blocks (= consecutive XCDs) share a stream, so every
    // stream is spread over all 8 XCDs.  (Measured: packing a stream onto one XCD is 2.4x slower.)
    const int sl = blockIdx.x / BPS, j0 = blockIdx.x - sl * BPS;
    const int s = s0 + sl;
    StreamCtl &c = ctl[s];
    const long long t_start = wall_clock64();
    if (c.active == 0) return;
    const StreamDev &S = streams[s];
    const int units = (((ROUND == 0) ? pk_cnt0(c.pkA) : c.cnt1) + PER - 1) / PER;
    const bool dbg = ROUND == 0 && C.dbg && c.frame == C.dbg_frame && tid == 0;
    long long *dbp = C.dbg + ((size_t)65536 + blockIdx.x) * 4;
    if (dbg) { dbp[0] = t_start; dbp[1] = wall_clock64(); dbp[2] = 0; dbp[3] = (j0 >= units) ? -1 : 0; }
    if (j0 >= units) return;
    if (tid == 0) { stage.np = 0; sh_acc[0] = sh_acc[1] = sh_acc[2] = 0; }
    __syncthreads();
    WaveFill fill = {0, 0, 0, 0, 0, 0, 0};
    // per-frame constants of the stream (nothing this kernel reads is written while it runs,
    // except by the atomics it issues itself)
    const float bestA = o2f(c.best);
    const bool init = c.active == 2;
    const float endTh = (!init && C.end_win > 0.0f) ? (bestA - C.end_win) : LZ;      // :349
    const float wordTh = (!init && C.word_win > 0.0f) ? (bestA - C.word_win) : LZ;   // :350
    const int cnt0 = pk_cnt0(c.pkA);
    const int nin = (ROUND == 0) ? cnt0 : c.cnt1;
    const int in_base = (ROUND == 0) ? 0 : cnt0;
    const int frame = c.frame;
    const bool last_frame = init || frame >= c.T - 1;
    ItemSink sink;
    sink.sk_out = S.skey[(ROUND & 1) ^ 1];
    sink.counter = (ROUND == 0) ? &c.cnt1 : &c.cnt2;
    sink.base = (ROUND == 0) ? cnt0 : cnt0 + c.cnt1;
    for (int u = j0; u < units; u += BPS) {
        const int k = u * PER + (tid / EG);
        int n_arcs = 0, n_paths_made = 0, n_pend = 0;
        expand_unit<INLINE>(C, c, S, stage, fill, frame, last_frame, ROUND == 0 && !init, cnt0, endTh, wordTh,
                            ROUND == 0 && !init, k < nin, in_base + k,
                            S.skey[ROUND & 1], (ROUND == 0) ? S.skeyL : S.skey[ROUND & 1], sink,
                            n_arcs, n_paths_made, n_pend,
                            (dbg && u == j0) ? C.dbg + ((size_t)196608 + blockIdx.x) * 4 : nullptr);
        if (INLINE) {                                                  // closure of what this wave just produced
            while (fill.qh < fill.qt)
                closure_step(C, c, S, stage, fill, frame, last_frame, cnt0, endTh, wordTh, sink, n_arcs, n_paths_made);
            fill.qh = 0; fill.qt = 0;
        }
        if (dbg) dbp[2] = wall_clock64();
        n_arcs = wave_sum(n_arcs); n_paths_made = wave_sum(n_paths_made); n_pend = wave_sum(n_pend);
        if (lane == 0) {
            if (n_arcs) atomicAdd(&sh_acc[0], n_arcs);
            if (n_paths_made) atomicAdd(&sh_acc[1], n_paths_made);
            if (n_pend) atomicAdd(&sh_acc[2], n_pend);
        }
    }
    if (INLINE) {                                                      // touched arcs + dirty states, reservations together
        WaveStage &w = stage.w[tid >> 6];
        const int n = fill.n, nd = fill.nd;
        if (n | nd) {
            WAVE_LDS_ORDER();
            int base = 0;
            if (lane == 0 && n) base = atomicAdd(&c.n_touched, n);
            if (lane == 1 && nd) base = atomicAdd(&c.n_dirty, nd);
            const int bt = __shfl(base, 0), bd = __shfl(base, 1);
            for (int k = lane; k < n; k += 64) {
                if (bt + k < C.cap_items) S.touched[bt + k] = w.buf[k]; else c.error = -42;
            }
            for (int k = lane; k < nd; k += 64) {
                if (bd + k < C.cap_items) S.dirty[bd + k] = w.dbuf[k]; else c.error = -42;
            }
        }
    } else stage_flush_block(C, c, S, stage, fill, sink);
    __syncthreads();
    if (tid == 0) {
        if (sh_acc[0]) atomicAdd(&c.fr[ST_ARCS], sh_acc[0]);
        if (sh_acc[1]) atomicAdd(&c.fr[ST_PATHS], sh_acc[1]);
        if (sh_acc[2]) atomicAdd(&c.fr[ST_PEND], sh_acc[2]);
        if (dbg) dbp[3] = wall_clock64();
    }
}

template <int ROUND>
__global__ __launch_bounds__(KTB) void k_expand(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0, int BPS)
{
    expand_body<ROUND, false>(C, ctl, streams, s0, BPS);
}

__global__ __launch_bounds__(KTB) void k_expand_closure(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0, int BPS)
{
    expand_body<0, true>(C, ctl, streams, s0, BPS);
}

// remaining closure rounds (items produced by round 1 and later): rare, one block per stream
__global__ __launch_bounds__(KT) void k_expand_tail(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0)
{
    StreamCtl &c = ctl[s0 + blockIdx.x];
    if ((c.active == 0) | (c.cnt2 == 0)) return;
    __shared__ BlockStage stage;
    const StreamDev &S = streams[s0 + blockIdx.x];
    constexpr int PER = KT / EG;
    const int tid = threadIdx.x, lane = lane_id();
    if (tid == 0) stage.np = 0;
    __syncthreads();
    WaveFill fill = {0, 0, 0, 0, 0, 0, 0};
    const float bestA = o2f(c.best);
    const bool init = c.active == 2;
    const float endTh = (!init && C.end_win > 0.0f) ? (bestA - C.end_win) : LZ;
    const float wordTh = (!init && C.word_win > 0.0f) ? (bestA - C.word_win) : LZ;
    const int base = pk_cnt0(c.pkA) + c.cnt1;
    int r0 = 0, r1 = c.cnt2;                       // window within the tail region [base + r0, base + r1)
    const int tail_base = base + c.cnt2;           // items appended here: tail_base + cnt_tail++
    int n_arcs = 0, n_paths_made = 0, n_pend = 0, parity = 0;
    while (r1 > r0) {
        for (int k0 = r0; k0 < r1; k0 += PER) {
            const int k = k0 + (tid / EG);
            ItemSink sink;
            sink.sk_out = S.skey[parity ^ 1]; sink.counter = &c.cnt_tail; sink.base = tail_base;
            expand_unit<false>(C, c, S, stage, fill, c.frame, init || c.frame >= c.T - 1, false, pk_cnt0(c.pkA), endTh, wordTh,
                        false, k < r1, base + k, S.skey[parity], S.skey[parity], sink, n_arcs, n_paths_made, n_pend);
            stage_flush_block(C, c, S, stage, fill, sink);
        }
        parity ^= 1;
        __syncthreads();
        r0 = r1;
        r1 = c.cnt2 + __hip_atomic_load(&c.cnt_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (base + r1 > C.cap_items) r1 = C.cap_items - base;
        __syncthreads();
    }
    n_arcs = wave_sum(n_arcs); n_paths_made = wave_sum(n_paths_made);
    if (lane == 0) {
        if (n_arcs) atomicAdd(&c.fr[ST_ARCS], n_arcs);
        if (n_paths_made) atomicAdd(&c.fr[ST_PATHS], n_paths_made);
    }
}

// ---- resolve: the winning candidate of every touched arc becomes the entry token of its
// instance; missing instances are attached here (attachNetInst :751-774), one lane per arc.
// fuse != 0: every block of an active stream reports when it is done and the last one runs the
// frame boundary (epilogue of this frame + start of the next) itself, so that a lock-step frame is
// k_phase_a, k_expand*, k_resolve and nothing else.
__global__ __launch_bounds__(KTB) void k_resolve(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0, int BPS, int fuse)
{
    __shared__ int sh_w[KTB / 64];
    __shared__ int sh_base;
    __shared__ unsigned sh_best;
    __shared__ int sh_skip;
    __shared__ int sh_last;
    __shared__ volatile unsigned sh_sink;
    __shared__ int sh_hist[HIST_MAX_BINS];
    const int tid = threadIdx.x, lane = lane_id();
    const int MN = C.max_n;
    const int rec_ints = (MN <= 5) ? 32 : 64, tok_off = (MN <= 5) ? 8 : 12, aux_ints = (MN <= 5) ? 8 : 12;
    // stream-major block order: consecutive blocks (= consecutive XCDs) share a stream, so every
    // stream is spread over all 8 XCDs.  (Measured: packing a stream onto one XCD is 2.4x slower.)
    const int sl = blockIdx.x / BPS, j0 = blockIdx.x - sl * BPS;
    const int s = s0 + sl;
    StreamCtl &c = ctl[s];
    if (c.active == 0) return;
    const StreamDev &S = streams[s];
    const int nt = c.n_touched < C.cap_items ? c.n_touched : C.cap_items;
    const int units = (nt + KTB - 1) / KTB;
    // blocks that take part: those with touched arcs to resolve, at least block 0.  Only they may
    // read the stream's counters: a surplus block can start after the stream's last participating
    // block has already run the fused boundary, which resets them.
    const int nrep = units < 1 ? 1 : (units < BPS ? units : BPS);
    if (j0 >= nrep) return;
    // inline closure: zero the per-state closure keys that were used this frame
    const int nd = C.inline_closure ? (c.n_dirty < C.cap_items ? c.n_dirty : C.cap_items) : 0;
    for (int i = j0 * KTB + tid; i < nd; i += nrep * KTB) S.skey[1][S.dirty[i]] = 0ULL;
    if (tid == 0) { sh_best = 0u; sh_skip = 0; sh_last = 0; }
    __syncthreads();
    for (int u = j0; u < units; u += BPS) {
        const int q = u * KTB + tid;
        int *rec_next = S.rec[c.lst ^ 1];
        const int nB = pk_nB(c.pkA);
        // An instance whose entry token provably fails next frame's emit threshold is not
        // materialised: next frame normalises by bestEmitScore >= bestA (the phase-A best, final
        // now) and emitTh >= -mainBeam, and float add/sub are monotone, so
        //     (entry + max_j trP[0][j]) - bestA <= -mainBeam   ==>   pruned at :409 next frame.
        // The reference would attach it, count it and let it die; we only count it.
        const float bestA = o2f(c.best);
        const bool can_skip = C.emit_win > 0.0f && bestA > LZ && c.active == 1;
        bool need = false, win = false, skip = false;
        int b = -1, slot = -1, ii = 0;
        float sc = LZ;
        JdArc Bk{0, 0.0f, 0, 0};
        int4 ax0 = make_int4(0, 0, 0, 0), ax1 = make_int4(0, 0, 0, 0), ax2 = make_int4(0, 0, 0, 0);
        if (q < nt) {
            b = S.touched[q];
            ArcState *as = S.ast + b;
            // one hop from the arc id: key (exchanged), slot, arc record and instance template together
            const unsigned long long key = atomicExch(&as->key, 0ULL);
            slot = as->slot;
            Bk = C.arcs[b];
            const int4 *ap = (const int4 *)(C.aux + (size_t)b * aux_ints);
            ax0 = ap[0]; ax1 = ap[1];
            if (aux_ints == 12) ax2 = ap[2];
            sc = o2f((unsigned)(key >> 32));
            if (sc > LZ) {
                win = true;
                ii = (int)(unsigned)(key & 0xffffffffULL);
                need = slot < 0;
                if (need && can_skip) {
                    const float tmax = __int_as_float(ax0.x);
                    if ((sc + tmax) - bestA <= -C.emit_win) { skip = true; need = false; }
                }
            } else slot = -1;
        }
        const int nskip = __popcll(__ballot(skip));
        if (lane == 0 && nskip) atomicAdd(&sh_skip, nskip);
        // block-aggregated allocation: one returning atomic per unit
        int tot_need;
        const int myk = block_excl_scan(need ? 1 : 0, sh_w, tot_need);
        if (tid == 0) sh_base = tot_need ? atomicAdd(&c.n_alloc, tot_need) : 0;
        __syncthreads();
        unsigned mo = win ? f2o(sc) : 0u;                              // :572-573 (skipped ones can never raise it)
        if (need) {
            const int ns_ = nB + sh_base + myk;                        // appended to the next list
            if (ns_ >= C.cap_slots) { c.error = -41; slot = -1; }
            else {                                                     // attachNetInst :751-774
                slot = ns_;
                const int n = ax0.y & 0xff;
                int *rec = rec_next + (size_t)slot * rec_ints;
                *(int4 *)rec = make_int4(b, ax0.y, Bk.out, Bk.to);
                if (rec_ints == 32) *(int4 *)(rec + 4) = make_int4(ax0.w, ax1.x, ax1.y, ax0.z);
                else {
                    *(int4 *)(rec + 4) = make_int4(ax0.w, ax1.x, ax1.y, ax0.z);
                    *(int4 *)(rec + 8) = make_int4(ax1.z, ax1.w, ax2.x, 0);
                }
                Tok *tp = (Tok *)(rec + tok_off);
                for (int qq = 1; qq < n; ++qq) tp[qq] = null_tok();
                S.ast[b].slot = slot;
            }
        }
        if (win && slot >= 0) {
            const Tok it = S.item_tok[ii];
            Tok e;
            e.score = sc; e.ac = it.ac; e.lm = it.lm + Bk.w; e.path = it.path;
            *(Tok *)(rec_next + (size_t)slot * rec_ints + tok_off) = e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned y = __shfl_xor(mo, o); mo = y > mo ? y : mo; }
        if (lane == 0 && mo) atomicMax(&sh_best, mo);
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        unsigned r = 0;
        if (sh_best) r += atomicMax(&c.best, sh_best);
        if (sh_skip) r += (unsigned)atomicAdd(&c.n_skipped, sh_skip);
        if (fuse) {
            // the atomics above and this block's n_alloc reservations have RETURNED, i.e. were
            // performed, before the block reports itself done (the volatile store consumes r)
            sh_sink = r;
            (void)sh_sink;
            sh_last = atomicAdd(&c.done_r, 1) == nrep - 1;
        }
    }
    if (!fuse) return;
    __syncthreads();
    if (sh_last && tid < 64) boundary_frame(C, c, streams[s], tid, sh_hist);
}

// Path garbage collection = collectPaths (WFSTDecoderLite.cpp:699-747) as a mark-compact: records
// reachable from a live token (or bestFinalToken) are kept, everything else - including the
// indices reserved for exit tokens that were never expanded - is dropped.  No effect on results.
// One 1024-thread block per stream, run between frames; a no-op below the threshold.
__global__ __launch_bounds__(1024) void k_gc(DecConst C, StreamCtl *ctl, StreamDev *streams, int s0)
{
    StreamCtl &c = ctl[s0 + blockIdx.x];
    StreamDev &S = streams[s0 + blockIdx.x];
    const int np = c.n_paths;
    if (!c.started || c.needs_init || np <= C.gc_threshold) return;
    __shared__ int sh_w[16];
    __shared__ int sh_carry;
    const int tid = threadIdx.x, NTH = blockDim.x;
    const int MN = C.max_n, rec_ints = (MN <= 5) ? 32 : 64, tok_off = (MN <= 5) ? 8 : 12;
    int *idx = S.gc_idx;
    for (int p = tid; p < np; p += NTH) idx[p] = 0;
    __syncthreads();
    // mark: walk the chain of every stored token until an already marked record is met
    const int *recs = S.rec[c.lst];
    const int n_tok = c.n_act * MN;
    for (int k = tid; k < n_tok + 1; k += NTH) {
        int p;
        if (k == n_tok) p = c.best_final.path;
        else {
            const int q = k / MN, i = k - q * MN;
            const int n = recs[(size_t)q * rec_ints + 1] & 0xff;
            p = (i < n) ? ((const Tok *)(recs + (size_t)q * rec_ints + tok_off))[i].path : -1;
            if (i < n && !(((const Tok *)(recs + (size_t)q * rec_ints + tok_off))[i].score > LZ)) p = -1;
        }
        while (p >= 0 && atomicExch(&idx[p], 1) == 0) p = S.paths[p].prev;
    }
    __syncthreads();
    // exclusive scan of the marks -> new indices (idx[p] = new index, -1 if dropped)
    if (tid == 0) sh_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < np; b0 += NTH) {
        const int p = b0 + tid;
        const int m = (p < np) ? idx[p] : 0;
        int tot;
        const int ex = block_excl_scan(m, sh_w, tot);
        const int base = sh_carry;
        if (p < np) idx[p] = m ? base + ex : -1;
        __syncthreads();
        if (tid == 0) sh_carry = base + tot;
        __syncthreads();
    }
    const int kept = sh_carry;
    // compact into the second arena, remapping prev (prev < p, so its new index is final)
    for (int p = tid; p < np; p += NTH) {
        const int ni = idx[p];
        if (ni >= 0) {
            PathRec pr = S.paths[p];
            pr.prev = (pr.prev >= 0) ? idx[pr.prev] : -1;
            S.paths2[ni] = pr;
        }
    }
    // remap the tokens
    int *recw = S.rec[c.lst];
    for (int k = tid; k < n_tok; k += NTH) {
        const int q = k / MN, i = k - q * MN;
        const int n = recw[(size_t)q * rec_ints + 1] & 0xff;
        if (i < n) {
            Tok *t = (Tok *)(recw + (size_t)q * rec_ints + tok_off) + i;
            const int p = t->path;
            if (p >= 0) t->path = (t->score > LZ) ? idx[p] : -1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (c.best_final.path >= 0) c.best_final.path = idx[c.best_final.path];
        PathRec *tmp = S.paths; S.paths = S.paths2; S.paths2 = tmp;
        c.n_paths = kept;
    }
}

// recognitionFinish (:230-309): walk the Path chain of bestFinalToken.
__global__ void jd_finish_kernel(StreamCtl *ctl, StreamDev *streams, int s0, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    StreamDev &S = streams[s0 + s];
    const StreamCtl &c = ctl[s0 + s];
    const Tok best = c.best_final;
    if (!(best.score > LZ) || c.frame == 0) { S.res_n = -1; return; }
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

__global__ void jd_mark_init_kernel(StreamCtl *ctl, int s0, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) { ctl[s0 + s].needs_init = 1; ctl[s0 + s].started = 1; ctl[s0 + s].error = 0; ctl[s0 + s].T = 0; }
}

__global__ void jd_set_T_kernel(StreamCtl *ctl, int s0, int n, const int *T)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) ctl[s0 + s].T = T[s];
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

// max_blocks > 0 bounds the grid (the kernel strides over the tiles): next to the search, a
// chip-filling scoring launch holds every wave slot for milliseconds and the latency-bound search
// kernels, which need slots for microseconds at a time, all but stop (measured: 25 ms of
// scoring cost the search 20 ms).  A bounded grid scores in the background instead.
// skip_unused: row_src marks unused rows with -1 in whole-tile runs (decode_wave's stream slots).
static int launch_gmm(const jd_am *a, const AmDevBuf &b, const float *d_feats, const int *d_row_src, int n_rows,
                      float *d_ll, hipStream_t st, int max_blocks = 0, int skip_unused = 0)
{
    if (n_rows <= 0) return JD_OK;
    const long long tiles = (long long)((n_rows + GMM_ROWS - 1) / GMM_ROWS) * ((a->n_gmm + GMM_GT - 1) / GMM_GT);
    dim3 grid((unsigned)((max_blocks > 0 && tiles > max_blocks) ? max_blocks : tiles));
    const int dp = a->D | 1;
    const size_t sm = (size_t)(GMM_ROWS * dp + GMM_ROWS * (GMM_GT + 1)) * sizeof(float);
    if (a->D == 39)
        hipLaunchKernelGGL(jd_gmm_kernel<39>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                           b.n_mix, a->n_gmm, a->max_mix, a->D, d_ll, skip_unused);
    else
        hipLaunchKernelGGL(jd_gmm_kernel<0>, grid, dim3(256), sm, st, d_feats, d_row_src, n_rows, b.par, b.det,
                           b.n_mix, a->n_gmm, a->max_mix, a->D, d_ll, skip_unused);
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
    int *d_row_ptr = nullptr; JdArc *d_arcs = nullptr; float *d_fin_w = nullptr; int *d_aux = nullptr;
    int *d_hmm_n = nullptr, *d_hmm_tm = nullptr, *d_hmm_gmm = nullptr, *d_se32 = nullptr;
    float *d_hmm_tee = nullptr, *d_trP = nullptr, *d_hmm_tmax0 = nullptr;
    // per-stream state
    StreamDev *d_streams = nullptr;
    StreamCtl *d_ctl = nullptr;
    std::vector<StreamDev> h_streams;          // host mirror (arena pointers)
    std::vector<void *> allocs;
    bool arenas_ready = false;
    int64_t cap_slots = 0, cap_paths = 0, cap_items = 0;
    int res_cap = 8192;
    int *d_res = nullptr;                 // result arena, see ensure_arenas
    long long max_closure = 0;            // static closure bound of the network (see jd_dec_create)
    int gmm_bg_blocks = 384;              // grid bound of a scoring launch that overlaps the search (see launch_gmm);
                                          // 1.5 per CU, doubled whenever the scoring turns out to be the bottleneck
    int n_cus = 256;
    // chunked pipeline
    int Fc = 128;
    float *d_ll[2] = {nullptr, nullptr};
    int *d_row_src = nullptr; size_t row_src_cap = 0;
    int *d_T = nullptr;
    hipStream_t s_gmm = nullptr, s_search = nullptr;
    hipEvent_t ev_gmm[2] = {nullptr, nullptr}, ev_search[2] = {nullptr, nullptr};
    // sampled per-kernel timing: every KSAMPLE_EVERY-th step records events around each launch
    std::vector<hipEvent_t> kev;               // KSAMPLE_MAX x 8 events
    int kev_used = 0;
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
    for (auto &e : d->kev) (void)hipEventDestroy(e);
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
    if (const char *e = getenv("JD_FC")) { const int v = atoi(e); if (v >= 16 && v <= 4096) d->Fc = v; }   // development: frames per scoring chunk
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) d->n_cus = prop.multiProcessorCount;
        d->gmm_bg_blocks = d->n_cus + d->n_cus / 2;
    }
    if (const char *e = getenv("JD_GMM_BLOCKS")) d->gmm_bg_blocks = atoi(e);                                  // development: 0 = unbounded
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
    {   // device arc table: bit 30 of the in-label marks arcs whose HMM is a tee model
        std::vector<JdArc> darcs(net->arcs);
        for (JdArc &a : darcs)
            if (a.in > 0 && am->hmm_tee[(size_t)a.in - 1] > LZ) a.in |= TEE_FLAG;
        TRY(dupload(d, &d->d_arcs, darcs.data(), darcs.size()));
    }
    TRY(dupload(d, &d->d_fin_w, net->fin_w.data(), net->fin_w.size()));
    TRY(dupload(d, &d->d_hmm_n, am->hmm_n.data(), am->hmm_n.size()));
    TRY(dupload(d, &d->d_hmm_tm, am->hmm_tm.data(), am->hmm_tm.size()));
    TRY(dupload(d, &d->d_hmm_gmm, am->hmm_gmm.data(), am->hmm_gmm.size()));
    TRY(dupload(d, &d->d_hmm_tee, am->hmm_tee.data(), am->hmm_tee.size()));
    {   // largest log transition probability out of the entry state of every HMM (see k_resolve)
        std::vector<float> tmax((size_t)am->n_hmm, LZ);
        for (int h = 0; h < am->n_hmm; ++h) {
            const float *t0 = am->trP.data() + (size_t)am->hmm_tm[(size_t)h] * am->max_n * am->max_n;
            for (int j = 0; j < am->hmm_n[(size_t)h]; ++j) tmax[(size_t)h] = std::max(tmax[(size_t)h], t0[j]);
        }
        TRY(dupload(d, &d->d_hmm_tmax0, tmax.data(), tmax.size()));
    }
    TRY(dupload(d, &d->d_trP, am->trP.data(), am->trP.size()));
    std::vector<int> se32((size_t)am->n_tm * am->max_n);
    for (size_t i = 0; i < se32.size(); ++i)
        se32[i] = ((int)am->se[i * 2] & 0xffff) | ((int)am->se[i * 2 + 1] << 16);
    TRY(dupload(d, &d->d_se32, se32.data(), se32.size()));
    TRY(upload_am_gmm(am, d->amb));
    C.row_ptr = d->d_row_ptr; C.arcs = d->d_arcs; C.fin_w = d->d_fin_w; C.init_state = net->init;
    C.G = am->n_gmm; C.max_n = am->max_n; C.n_tm = am->n_tm;
    C.hmm_n = d->d_hmm_n; C.hmm_tm = d->d_hmm_tm; C.hmm_gmm = d->d_hmm_gmm; C.hmm_tee = d->d_hmm_tee;
    C.hmm_tmax0 = d->d_hmm_tmax0;
    {   // per-arc instance template: everything k_resolve needs to attach an instance, one hop from the arc id
        const int AI = (am->max_n <= 5) ? 8 : 12;
        std::vector<int> aux((size_t)net->n_arcs * AI, 0);
        for (int64_t b = 0; b < net->n_arcs; ++b) {
            const int in = net->arcs[(size_t)b].in;
            if (in <= 0) continue;
            const int hm = in - 1, n = am->hmm_n[(size_t)hm];
            const float *t0 = am->trP.data() + (size_t)am->hmm_tm[(size_t)hm] * am->max_n * am->max_n;
            float tmax = LZ;
            for (int j = 0; j < n; ++j) tmax = std::max(tmax, t0[j]);
            int *a = aux.data() + (size_t)b * AI;
            memcpy(&a[0], &tmax, sizeof(float));
            a[1] = n | (am->hmm_tm[(size_t)hm] << 8);
            a[2] = hm;
            for (int j = 1; j < n - 1 && j <= (AI == 8 ? 3 : 6); ++j) a[2 + j] = am->hmm_gmm[(size_t)hm * am->max_n + j];
        }
        TRY(dupload(d, &d->d_aux, aux.data(), aux.size()));
        C.aux = d->d_aux;
    }
    C.trP = d->d_trP; C.se32 = d->d_se32;
    {   // Static bound on the epsilon / tee closure one frontier item can cause: the number of
        // epsilon-input or tee-model arcs on all paths of such arcs below a state (a tee arc
        // forwards the token to its end state in the same frame, WFSTDecoderLite.cpp:584-600).
        // Small and acyclic -> the closure runs inline in k_expand_closure (4 launches per
        // frame); otherwise the rounds are separate kernels (6 launches per frame).
        const int S_ = net->n_states;
        const long long LIMIT = INLINE_CLOSURE_MAX;
        std::vector<long long> sz((size_t)S_, -1);                     // -1 unknown, -2 on the DFS stack
        long long worst = 0;
        auto passes = [&](const JdArc &a) { return a.in == 0 || (a.in - 1 < am->n_hmm && am->hmm_tee[(size_t)a.in - 1] > LZ); };
        std::vector<std::pair<int, int>> stk;                          // (state, next arc)
        for (int s0_ = 0; s0_ < S_ && worst <= LIMIT; ++s0_) {
            if (sz[(size_t)s0_] >= 0) continue;
            stk.push_back({s0_, net->row_ptr[(size_t)s0_]});
            sz[(size_t)s0_] = -2;
            std::vector<long long> acc(1, 0);
            while (!stk.empty() && worst <= LIMIT) {
                auto &top = stk.back();
                const int st_ = top.first;
                if (top.second == net->row_ptr[(size_t)st_ + 1]) {
                    sz[(size_t)st_] = acc.back();
                    worst = std::max(worst, acc.back());
                    const long long done = acc.back();
                    acc.pop_back(); stk.pop_back();
                    if (!acc.empty()) acc.back() += 1 + done;           // the arc that led here + its closure
                    continue;
                }
                const JdArc &a = net->arcs[(size_t)top.second++];
                if (!passes(a)) continue;
                if (sz[(size_t)a.to] == -2) { worst = LIMIT + 1; break; }   // epsilon cycle
                if (sz[(size_t)a.to] >= 0) { acc.back() += 1 + sz[(size_t)a.to]; continue; }
                sz[(size_t)a.to] = -2;
                stk.push_back({a.to, net->row_ptr[(size_t)a.to]});
                acc.push_back(0);
            }
            stk.clear();
        }
        d->max_closure = worst;
        C.inline_closure = worst <= LIMIT ? 1 : 0;
        if (const char *e = getenv("JD_INLINE_CLOSURE")) {              // development: 0 forces the staged kernels
            if (atoi(e) == 0) C.inline_closure = 0;
        }
    }
    // arena capacities: 0 = sized from the free HBM when the arenas are allocated (ensure_arenas)
    d->cap_slots = d->cap_items = d->cap_paths = 0;
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

// reset a stream's per-arc search state: no candidate, no instance
__global__ void jd_reset_ast_kernel(ArcState *ast, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ast[i] = ArcState{0ULL, -1, 0};
}
static void reset_ast(ArcState *ast, int64_t n_arcs)
{
    hipLaunchKernelGGL(jd_reset_ast_kernel, dim3((unsigned)((n_arcs + 255) / 256)), dim3(256), 0, 0, ast, (long long)n_arcs);
}

static int ensure_arenas(jd_dec *d)
{
    if (d->arenas_ready) return JD_OK;
    int rc = check_device(d->device);
    if (rc) return rc;
    const int B = d->max_streams, MN = d->am->max_n;
    {   // Capacities the caller did not set: sized for 288 GB of HBM, not for frugality.  Half of the
        // free memory is split over the streams; of a stream's share (after its per-arc / per-state
        // tables) 50% goes to instance records, 20% to frontier items, 30% to Path records - each
        // between a floor that suits narrow beams and the most the graph can ever need.
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const double n_arcs = (double)d->net->n_arcs, n_states = (double)d->net->n_states;
        const double fixed = n_arcs * sizeof(ArcState) + n_states * 24.0 + 2.0 * d->Fc * d->am->n_gmm * sizeof(float);
        const double budget = std::max(0.0, 0.5 * (double)free_b / B - fixed);
        const double rec_b = 2.0 * ((MN <= 5) ? 128 : 256), item_b = 4.0 + sizeof(Tok) + sizeof(int4);
        const double path_b = 2.0 * sizeof(PathRec) + 4.0;
        auto pick = [](double share, int64_t lo, int64_t hi) {
            return std::max<int64_t>(std::min<int64_t>(hi, (int64_t)share), std::min(lo, hi));
        };
        if (d->cap_slots <= 0) d->cap_slots = pick(0.5 * budget / rec_b, 1 << 19, d->net->n_arcs + 1024);
        if (d->cap_items <= 0) d->cap_items = pick(0.2 * budget / item_b, 1 << 21, std::max<int64_t>(2 * d->net->n_arcs + 1024, 1 << 16));
        if (d->cap_paths <= 0) d->cap_paths = pick(0.3 * budget / path_b, 1 << 21, 1 << 26);
        const int64_t lim = 0x7fffff00;
        if (d->cap_slots > lim || d->cap_items > lim || d->cap_paths > lim)
            return jd_fail(JD_EINVAL, "arena capacity above 2^31 records");
    }
    d->C.cap_slots = (int)d->cap_slots; d->C.cap_items = (int)d->cap_items; d->C.cap_paths = (int)d->cap_paths;
    d->C.gc_threshold = (int)(d->cap_paths / 2);
    d->h_streams.assign((size_t)B, StreamDev());
    rc = dmalloc(d, &d->d_res, (size_t)B * 5 * d->res_cap);
    if (rc) return rc;
    for (int s = 0; s < B; ++s) {
        StreamDev &S = d->h_streams[(size_t)s];
        memset(&S, 0, sizeof S);
#define A(p, n) do { rc = dmalloc(d, &(p), (size_t)(n)); if (rc) return rc; } while (0)
        A(S.rec[0], d->cap_slots * ((MN <= 5) ? 32 : 64));
        A(S.rec[1], d->cap_slots * ((MN <= 5) ? 32 : 64));
        A(S.ast, d->net->n_arcs);
        A(S.skey[0], d->net->n_states); A(S.skey[1], d->net->n_states); A(S.skeyL, d->net->n_states);
        A(S.touched, d->cap_items);
        A(S.dirty, d->cap_items);
        A(S.item_tok, d->cap_items); A(S.item_info, d->cap_items);
        A(S.paths, d->cap_paths); A(S.paths2, d->cap_paths); A(S.gc_idx, d->cap_paths);
        A(S.hist, HIST_MAX_BINS);
        {   // the five result arrays of all streams live in one arena [stream][array][res_cap]:
            // fetch_results brings a whole wave back with a single strided copy
            int *base = d->d_res + (size_t)s * 5 * d->res_cap;
            S.res_label = base; S.res_time = base + d->res_cap;
            S.res_score = (float *)(base + 2 * (size_t)d->res_cap); S.res_ac = (float *)(base + 3 * (size_t)d->res_cap);
            S.res_lm = (float *)(base + 4 * (size_t)d->res_cap);
        }
#undef A
        S.res_cap = d->res_cap;
        reset_ast(S.ast, d->net->n_arcs);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemset(S.skey[0], 0, (size_t)d->net->n_states * sizeof(unsigned long long)));
        HIPCHK(hipMemset(S.skey[1], 0, (size_t)d->net->n_states * sizeof(unsigned long long)));
        HIPCHK(hipMemset(S.skeyL, 0, (size_t)d->net->n_states * sizeof(unsigned long long)));
        HIPCHK(hipMemset(S.hist, 0, HIST_MAX_BINS * sizeof(int)));
    }
    rc = dmalloc(d, &d->d_streams, (size_t)B);
    if (rc) return rc;
    HIPCHK(hipMemcpy(d->d_streams, d->h_streams.data(), (size_t)B * sizeof(StreamDev), hipMemcpyHostToDevice));
    rc = dmalloc(d, &d->d_T, (size_t)B);
    if (rc) return rc;
    rc = dmalloc(d, &d->d_ctl, (size_t)B);
    if (rc) return rc;
    {
        std::vector<StreamCtl> hc((size_t)B);
        memset(hc.data(), 0, hc.size() * sizeof(StreamCtl));
        for (auto &c : hc) {
            c.needs_init = 1; c.best_emit = LZ;
            c.best_final.score = LZ; c.best_final.ac = LZ; c.best_final.lm = LZ; c.best_final.path = -1;
        }
        HIPCHK(hipMemcpy(d->d_ctl, hc.data(), hc.size() * sizeof(StreamCtl), hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 2; ++i)
        HIPCHK(hipMalloc(&d->d_ll[i], (size_t)B * d->Fc * d->am->n_gmm * sizeof(float)));
    HIPCHK(hipDeviceSynchronize());
    d->arenas_ready = true;
    return JD_OK;
}

// mark streams [s0, s0+n) for re-initialisation (IDecoder::init)
static int mark_init(jd_dec *d, int s0, int n, hipStream_t st)
{
    hipLaunchKernelGGL(jd_mark_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d->d_ctl, s0, n);
    HIPCHK(hipGetLastError());
    return JD_OK;
}

// results of streams [s0, s0+n) -> out[out_idx[i]] (out_idx == nullptr: out[out0 + i])
static int fetch_results(jd_dec *d, int s0, int n, jd_hyp *out, int out0, const int *out_idx = nullptr)
{
    std::vector<StreamDev> hs((size_t)n);
    std::vector<StreamCtl> hc((size_t)n);
    HIPCHK(hipMemcpy(hs.data(), d->d_streams + s0, (size_t)n * sizeof(StreamDev), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hc.data(), d->d_ctl + s0, (size_t)n * sizeof(StreamCtl), hipMemcpyDeviceToHost));
    int first_err = JD_OK;
    int kmax = 0;
    for (int i = 0; i < n; ++i) kmax = std::max(kmax, std::min(hs[(size_t)i].res_n, d->res_cap));
    std::vector<int> hres((size_t)n * 5 * std::max(kmax, 0));
    if (kmax > 0)                                                      // rows = (stream, array), first kmax words of each
        HIPCHK(hipMemcpy2D(hres.data(), (size_t)kmax * 4, d->d_res + (size_t)s0 * 5 * d->res_cap, (size_t)d->res_cap * 4,
                           (size_t)kmax * 4, (size_t)n * 5, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        const StreamDev &S = hs[(size_t)i];
        const StreamCtl &K = hc[(size_t)i];
        const int oi = out_idx ? out_idx[i] : out0 + i;
        HostResult &R = d->results[(size_t)oi];
        jd_hyp &H = out[oi];
        memset(&H, 0, sizeof H);
        if (K.error && first_err == JD_OK) {
            if (K.error == JD_EHIST)
                first_err = jd_fail(JD_EHIST, "Histogram::addScore - score > maxScore (stream %d)", s0 + i);
            else {
                const char *what = K.error == -41 ? "instance slots" : K.error == -42 ? "frontier items"
                                 : K.error == -43 ? "Path records" : "arena";
                const long long cap = K.error == -41 ? d->cap_slots : K.error == -42 ? d->cap_items : d->cap_paths;
                first_err = jd_fail(JD_ENOMEM, "stream %d: device arena overflow at frame %d: %s (capacity %lld); "
                                    "raise it with jd_dec_set_capacity", s0 + i, K.frame, what, cap);
            }
        }
        if (K.error) {      // arenas may be inconsistent after an abort: wipe them for the next init
            reset_ast(S.ast, d->net->n_arcs);
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemset(S.skey[0], 0, (size_t)d->net->n_states * sizeof(unsigned long long)));
            HIPCHK(hipMemset(S.skey[1], 0, (size_t)d->net->n_states * sizeof(unsigned long long)));
            HIPCHK(hipMemset(S.skeyL, 0, (size_t)d->net->n_states * sizeof(unsigned long long)));
            const int zero = 0;
            HIPCHK(hipMemcpy((char *)(d->d_ctl + s0 + i) + offsetof(StreamCtl, n_act), &zero, sizeof(int),
                             hipMemcpyHostToDevice));
        }
        H.stats.n_frames = K.frame;
        H.stats.tot_active_emit_hyps = K.st[ST_EMIT];
        H.stats.tot_active_end_hyps = K.st[ST_END];
        H.stats.tot_active_models = K.st[ST_MODELS];
        H.stats.tot_proc_emit_hyps = K.st[ST_PEMIT];
        H.stats.tot_proc_end_hyps = K.st[ST_PEND];
        H.stats.tot_arcs_visited = K.st[ST_ARCS];
        H.stats.tot_paths = K.st[ST_PATHS];
        H.stats.tot_insts_in = K.st[ST_INSTS];
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
            const int *row = hres.data() + (size_t)i * 5 * kmax;
            memcpy(R.label.data(), row, kk * 4); memcpy(R.time.data(), row + kmax, kk * 4);
            memcpy(R.score.data(), row + 2 * (size_t)kmax, kk * 4); memcpy(R.ac.data(), row + 3 * (size_t)kmax, kk * 4);
            memcpy(R.lm.data(), row + 4 * (size_t)kmax, kk * 4);
        }
        H.label = R.label.data(); H.time = R.time.data();
        H.score = R.score.data(); H.ac = R.ac.data(); H.lm = R.lm.data();
        if (kk) { H.tot_score = K.best_final.score; H.tot_ac = K.best_final.ac; H.tot_lm = K.best_final.lm; }
        else { H.tot_score = LZ; H.tot_ac = LZ; H.tot_lm = LZ; }      // DecHyp() defaults, DecHypHistPool.h
    }
    return first_err;
}

#define KSAMPLE_EVERY 32
#define KSAMPLE_MAX 96
// blocks per stream of the flattened kernels.  The base values suit a full batch (64 streams);
// with few streams each one gets more blocks so that a launch still covers the chip
// (256 CUs x 8 resident blocks).  Blocks beyond a stream's work return at once.
#define BPS_A 224      // phase A (64 instances per unit; blocks loop beyond 14k instances)
#define BPS_X 128      // frontier rounds (16 items per unit; measured: 64 -> 128 = +5% frames/s on configs[1])
#define BPS_R 64       // resolve (256 touched arcs per unit)
static inline int bps_for(int base, int nb)
{
    static const int ov_a = getenv("JD_BPS_A") ? atoi(getenv("JD_BPS_A")) : 0;      // tuning knobs (development)
    static const int ov_x = getenv("JD_BPS_X") ? atoi(getenv("JD_BPS_X")) : 0;
    static const int ov_r = getenv("JD_BPS_R") ? atoi(getenv("JD_BPS_R")) : 0;
    const int ov = base == BPS_A ? ov_a : base == BPS_X ? ov_x : ov_r;
    if (ov > 0) base = ov;
    return std::max(base, (2048 + nb - 1) / nb);
}

// recognitionStart for every stream of [s0, s0+nb) that is flagged needs_init
static void launch_init(jd_dec *d, int nb, int s0, hipStream_t st)
{
    const int bx = bps_for(BPS_X, nb), br = bps_for(BPS_R, nb);
    hipLaunchKernelGGL(k_boundary, dim3(nb), dim3(64), 0, st, d->C, d->d_ctl, d->d_streams, s0, 1);
    if (d->C.inline_closure)
        hipLaunchKernelGGL(k_expand_closure, dim3(nb * bx), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, bx);
    else {
        hipLaunchKernelGGL(k_expand<0>, dim3(nb * bx), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, bx);
        hipLaunchKernelGGL(k_expand<1>, dim3(nb * bx), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, bx);
        hipLaunchKernelGGL(k_expand_tail, dim3(nb), dim3(KT), 0, st, d->C, d->d_ctl, d->d_streams, s0);
    }
    hipLaunchKernelGGL(k_resolve, dim3(nb * br), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, br, 0);
    hipLaunchKernelGGL(k_boundary, dim3(nb), dim3(64), 0, st, d->C, d->d_ctl, d->d_streams, s0, 2);
}

// one lock-step frame for streams [s0, s0+nb); ev != nullptr: 7 events bracketing the launches.
// lead: the frame boundary has not been run yet (first frame of a run of steps) -> k_boundary
// first; otherwise the previous step's k_resolve has already done it (fused, see k_resolve).
static void launch_step(jd_dec *d, int nb, int s0, const float *ll, long long ll_stride, int f0, hipStream_t st,
                        bool lead, hipEvent_t *ev = nullptr)
{
#define EV(i) do { if (ev) (void)hipEventRecord(ev[i], st); } while (0)
    const int ba = bps_for(BPS_A, nb), bx = bps_for(BPS_X, nb), br = bps_for(BPS_R, nb);
    EV(0);
    if (lead) hipLaunchKernelGGL(k_boundary, dim3(nb), dim3(64), 0, st, d->C, d->d_ctl, d->d_streams, s0, 0);
    EV(1);
    if (d->am->max_n <= 5)
        hipLaunchKernelGGL(k_phase_a<4>, dim3(nb * ba), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, ba, ll,
                           ll_stride, f0);
    else
        hipLaunchKernelGGL(k_phase_a<8>, dim3(nb * ba), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, ba, ll,
                           ll_stride, f0);
    EV(2);
    if (d->C.inline_closure) {
        hipLaunchKernelGGL(k_expand_closure, dim3(nb * bx), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, bx);
    } else {                                                           // (events 3 and 4 stay unused inline)
        hipLaunchKernelGGL(k_expand<0>, dim3(nb * bx), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, bx);
        EV(3);
        hipLaunchKernelGGL(k_expand<1>, dim3(nb * bx), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, bx);
        EV(4);
        hipLaunchKernelGGL(k_expand_tail, dim3(nb), dim3(KT), 0, st, d->C, d->d_ctl, d->d_streams, s0);
    }
    EV(5);
    hipLaunchKernelGGL(k_resolve, dim3(nb * br), dim3(KTB), 0, st, d->C, d->d_ctl, d->d_streams, s0, br, 1);
    EV(6);
#undef EV
}

// Path garbage collection between two frames (the last frame has been closed by the fused
// boundary of its k_resolve, or by launch_close)
static void launch_gc(jd_dec *d, int nb, int s0, hipStream_t st)
{
    hipLaunchKernelGGL(k_gc, dim3(nb), dim3(1024), 0, st, d->C, d->d_ctl, d->d_streams, s0);
}

// closes the last frame of a run of steps (epilogue only: no stream has frames left)
static void launch_close(jd_dec *d, int nb, int s0, hipStream_t st)
{
    hipLaunchKernelGGL(k_boundary, dim3(nb), dim3(64), 0, st, d->C, d->d_ctl, d->d_streams, s0, 0);
}

// Decode one wave of nb <= max_streams utterances held in device memory.
// Stream u decodes frames [ustart[u], ustart[u] + ulen[u]) of d_feats.
static int decode_wave(jd_dec *d, int nb, const float *d_feats, const int64_t *ustart, const int64_t *ulen,
                       hipStream_t user_stream)
{
    const int Fc = d->Fc, G = d->am->n_gmm;
    std::vector<int> T((size_t)nb);
    int maxT = 0;
    for (int u = 0; u < nb; ++u) {
        const int64_t t = ulen[u];
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
                const int64_t src = ustart[u] + f;
                if (src > 0x7fffffff) return jd_fail(JD_EINVAL, "more than 2^31 frames in one batch");
                row_src[((size_t)c * nb + u) * Fc + dt] = (f < T[(size_t)u]) ? (int)src : -1;
            }
    if (user_stream) HIPCHK(hipStreamSynchronize(user_stream));
    HIPCHK(hipMemcpyAsync(d->d_row_src, row_src.data(), n_rows_all * sizeof(int), hipMemcpyHostToDevice, d->s_gmm));
    HIPCHK(hipMemcpyAsync(d->d_T, T.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice, d->s_search));
    int rc = mark_init(d, 0, nb, d->s_search);
    if (rc) return rc;
    hipLaunchKernelGGL(jd_set_T_kernel, dim3((nb + 63) / 64), dim3(64), 0, d->s_search, d->d_ctl, 0, nb, d->d_T);
    launch_init(d, nb, 0, d->s_search);
    HIPCHK(hipGetLastError());
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
        // chunk 0 is on the critical path (whole chip); later chunks score in the background
        rc = launch_gmm(d->am, d->amb, d_feats, d->d_row_src + (size_t)c * nb * Fc, nb * Fc, d->d_ll[buf], d->s_gmm,
                        c == 0 ? 0 : d->gmm_bg_blocks, (Fc % GMM_ROWS) == 0);
        if (rc) return rc;
        HIPCHK(hipEventRecord(ge[(size_t)c], d->s_gmm));
        HIPCHK(hipEventRecord(d->ev_gmm[buf], d->s_gmm));
        HIPCHK(hipStreamWaitEvent(d->s_search, d->ev_gmm[buf], 0));
        HIPCHK(hipEventRecord(ss[(size_t)c], d->s_search));
        {
            const int nsteps = std::min(Fc, maxT - c * Fc);
            for (int k = 0; k < nsteps; ++k) {
                if ((c > 0 || k > 0) && (k % 8) == 0) launch_gc(d, nb, 0, d->s_search);    // no-op below the threshold
                hipEvent_t *ev = nullptr;
                if (((c * Fc + k) % KSAMPLE_EVERY) == KSAMPLE_EVERY / 2 && d->kev_used < KSAMPLE_MAX) {
                    if (d->kev.empty()) {
                        d->kev.resize((size_t)KSAMPLE_MAX * 8);
                        for (auto &e : d->kev) (void)hipEventCreate(&e);
                    }
                    ev = d->kev.data() + (size_t)d->kev_used++ * 8;
                }
                launch_step(d, nb, 0, d->d_ll[buf], (long long)Fc * G, c * Fc, d->s_search, c == 0 && k == 0, ev);
            }
            if (c == n_chunks - 1) launch_close(d, nb, 0, d->s_search);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(se[(size_t)c], d->s_search));
        HIPCHK(hipEventRecord(d->ev_search[buf], d->s_search));
    }
    hipLaunchKernelGGL(jd_finish_kernel, dim3((nb + 63) / 64), dim3(64), 0, d->s_search, d->d_ctl, d->d_streams, 0, nb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->s_gmm));
    HIPCHK(hipStreamSynchronize(d->s_search));
    auto w1 = std::chrono::steady_clock::now();
    d->timing.total_ms += std::chrono::duration<double, std::milli>(w1 - w0).count();
    double waited_ms = 0.0, searched_ms = 0.0;
    for (int c = 0; c < n_chunks; ++c) {
        float gms = 0.0f, sms = 0.0f, gap = 0.0f;
        if (hipEventElapsedTime(&gms, gs[(size_t)c], ge[(size_t)c]) == hipSuccess) d->timing.gmm_ms += gms;
        if (hipEventElapsedTime(&sms, ss[(size_t)c], se[(size_t)c]) == hipSuccess) { d->timing.search_ms += sms; searched_ms += sms; }
        // chunk c >= 1 is scored while chunk c-1 is searched; the search of chunk c starts when both
        // are done: time between the end of search c-1 and the start of search c = waiting for scores
        if (c >= 1 && hipEventElapsedTime(&gap, se[(size_t)c - 1], ss[(size_t)c]) == hipSuccess) waited_ms += gap;
    }
    for (int c = 0; c < n_chunks; ++c) {
        (void)hipEventDestroy(gs[(size_t)c]); (void)hipEventDestroy(ge[(size_t)c]);
        (void)hipEventDestroy(ss[(size_t)c]); (void)hipEventDestroy(se[(size_t)c]);
    }
    if (waited_ms > 0.05 * searched_ms && d->gmm_bg_blocks > 0) {       // give the scoring more of the chip next time
        d->gmm_bg_blocks *= 2;
        if (d->gmm_bg_blocks > 4 * d->n_cus) d->gmm_bg_blocks = 0;     // unbounded
    }
    for (int i = 0; i < d->kev_used; ++i)
        for (int k = 0; k < 6; ++k) {
            float ms = 0.0f;
            int k1 = k + 1;
            if (k == 0) continue;                                      // sampled steps never lead: no k_boundary launch in slot 0
            if (d->C.inline_closure) {                                 // slot 2 spans events 2 -> 5, slots 3 and 4 are empty
                if (k == 3 || k == 4) continue;
                if (k == 2) k1 = 5;
            }
            if (hipEventElapsedTime(&ms, d->kev[(size_t)i * 8 + k], d->kev[(size_t)i * 8 + k1]) == hipSuccess)
                d->timing.kernel_us[k] += 1e3 * ms;
        }
    d->timing.kernel_samples += d->kev_used;
    d->kev_used = 0;
    d->timing.search_steps += maxT;
    d->timing.gmm_launches += n_chunks;
    d->timing.search_launches += n_chunks;
    for (int u = 0; u < nb; ++u) d->timing.gmm_frames += T[(size_t)u];
    d->timing.gmm_states = G;
    d->timing.closure_inline = d->C.inline_closure;
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
    // More utterances than streams: successive lock-step waves.  A wave lasts as long as its
    // longest utterance, so the waves are formed from the utterances sorted by length (results do
    // not depend on which utterances share a wave).
    std::vector<int> order((size_t)n_utts);
    std::iota(order.begin(), order.end(), 0);
    if (n_utts > d->max_streams)
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return offs[a + 1] - offs[a] > offs[b + 1] - offs[b]; });
    std::vector<int64_t> ustart((size_t)d->max_streams), ulen((size_t)d->max_streams);
    for (int u0 = 0; u0 < n_utts; u0 += d->max_streams) {
        const int nb = std::min(d->max_streams, n_utts - u0);
        for (int i = 0; i < nb; ++i) {
            const int u = order[(size_t)(u0 + i)];
            ustart[(size_t)i] = offs[u]; ulen[(size_t)i] = offs[u + 1] - offs[u];
        }
        rc = decode_wave(d, nb, d_feats, ustart.data(), ulen.data(), (hipStream_t)hip_stream);
        if (rc) return rc;
        rc = fetch_results(d, 0, nb, out, 0, order.data() + u0);
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
        hipLaunchKernelGGL(jd_set_T_kernel, dim3(1), dim3(64), 0, st, d->d_ctl, s, 1, d->d_T + s);
        rc = launch_gmm(d->am, d->amb, d->d_push, d->d_row_src, n, d->d_ll[0], st);
        if (rc) return rc;
        launch_init(d, 1, s, st);                      // no-op unless the stream is flagged needs_init
        for (int k = 0; k < n; ++k) {
            if ((k % 8) == 0) launch_gc(d, 1, s, st);
            launch_step(d, 1, s, d->d_ll[0], (long long)Fc * G, f0, st, k == 0);
        }
        launch_close(d, 1, s, st);
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
    if (d->stream_T[(size_t)s] == 0) launch_init(d, 1, s, d->s_search);   // init() directly followed by finish()
    hipLaunchKernelGGL(jd_finish_kernel, dim3(1), dim3(64), 0, d->s_search, d->d_ctl, d->d_streams, s, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->s_search));
    // results of stream s are stored at result slot s
    std::vector<jd_hyp> tmp((size_t)d->max_streams);
    rc = fetch_results(d, s, 1, tmp.data(), s);
    *out = tmp[(size_t)s];
    return rc;
}

// Diagnostics: record per-block wall-clock stamps (100 MHz) of k_phase_a and k_expand<0> for
// lock-step frame `frame` of subsequent decodes; fetch copies 2 x 65536 x 4 stamps.
extern "C" int jd_dec_debug_trace(jd_dec *d, int32_t frame, int64_t *fetch)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_debug_trace: null");
    const size_t n = (size_t)4 * 65536 * 4;
    if (!d->C.dbg) {
        long long *p = nullptr;
        HIPCHK(hipMalloc(&p, n * sizeof(long long)));
        HIPCHK(hipMemset(p, 0, n * sizeof(long long)));
        d->allocs.push_back(p);
        d->C.dbg = p;
    }
    d->C.dbg_frame = frame;
    if (fetch) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(fetch, d->C.dbg, n * sizeof(long long), hipMemcpyDeviceToHost));
    }
    return JD_OK;
}

// Test hook for the bit-exactness claim of jd_expf: evaluates it for x[0..n) on HIP device
// `device`, or - device == -1 - its host twin compiled from the same source.
__global__ void jd_debug_expf_kernel(const float *x, long long n, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = jd_expf(x[i]);
}
extern "C" int jd_debug_expf(int32_t device, const float *x, int64_t n, float *out)
{
    if (!x || !out || n < 0) return jd_fail(JD_EINVAL, "jd_debug_expf: bad argument");
    if (device == -1) {
        for (int64_t i = 0; i < n; ++i) out[i] = jd_expf_impl(x[i], jd_exp2f_tab_host);
        return JD_OK;
    }
    int rc = check_device(device);
    if (rc) return rc;
    float *dx = nullptr, *dy = nullptr;
    HIPCHK(hipMalloc(&dx, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    HIPCHK(hipMalloc(&dy, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    HIPCHK(hipMemcpy(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(jd_debug_expf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, (long long)n, dy);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dy, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return JD_OK;
}

extern "C" int jd_dec_last_timing(const jd_dec *d, jd_timing *out)
{
    if (!d || !out) return jd_fail(JD_EINVAL, "jd_dec_last_timing: null");
    *out = d->timing;
    return JD_OK;
}
