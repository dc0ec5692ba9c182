// jd_search.h - the token-passing search of juicer_amd as ONE persistent kernel per chunk of
// frames (included by jd_device.hip; gfx950 only).
//
// Reference: WFSTDecoderLite::processFrame (src/WFSTDecoderLite.cpp:311-372) =
// doHMMInternalPropagation (:899-935, :376-484) + doHMMExternalPropagation (:937-982) +
// propagateToken (:491-605), recognitionStart (:139-228), Histogram (src/Histogram.cpp).
//
// Design.  Utterance streams are independent, so nothing forces them to advance in lock-step:
// every stream is served by a CLUSTER of Cw workgroups (1024 threads each) that runs all frames
// of a chunk without returning to the host.  A frame is two phases separated by cluster barriers
// (a monotonic counter per stream, ~1-2 us), not kernel boundaries (~35 us with their launch
// chains):
//
//   phase A  HMM-internal propagation of every active arc instance AND of every arc that was
//            first entered in the previous frame.  An instance pulls its entry token itself: the
//            previous frame's expansion left the best candidate of each arc in a 64-bit key
//            (ordered score << 32 | frontier item) - so there is no separate "resolve" pass, and a
//            new instance is only ever written if something in it survives its first frame.
//   phase X  frontier expansion (propagateToken): exit tokens, then epsilon / tee closure rounds
//            (one barrier per non-empty round), atomic-max recombination into the per-arc keys.
//
// No list is shared for appending: every wave owns a segment of each output list (instance
// records, frontier items, newly entered arcs) and publishes its fill count at the end of the
// phase; readers turn the counts into a prefix over fixed-size chunks and take chunks
// round-robin, so reading is balanced whatever the writers did.  The only returning atomics left
// are the per-arc recombination (first-touch detection) and the Path-record reservation.
//
// Memory model (MI355X_MICROARCH.md, inter-workgroup visibility): every mutable per-stream word
// is written with agent-scope (sc1, write-through) stores or atomics and read with sc1 loads, which
// bypass the non-coherent per-CU L1; each storing wave drains (s_waitcnt vmcnt(0)) before it
// arrives at a barrier.  Static data (graph, models, likelihoods) uses plain cached loads.
#pragma once

#define SW 16                        // waves per search workgroup
#define SNT (SW * 64)                // threads per search workgroup
#define MAXW 1024                    // most waves one cluster may have (64 workgroups per stream)
#define MAXCW (MAXW / SW)
#define TEE_FLAG 0x40000000          // bit 30 of the device arc's in-label: the arc's HMM is a tee model
#define EG 16                        // lanes owning one frontier item (its arcs are pooled per wave)
#define NGRP (64 / EG)               // frontier items per wave pass
#define TRP_LDS_MAX 4096             // floats of transition tables cached in LDS (else read from HBM)

enum { ST_EMIT = 0, ST_END, ST_MODELS, ST_PEMIT, ST_PEND, ST_ARCS, ST_PATHS, ST_INSTS, ST_N };
enum { JDE_SLOTS = -41, JDE_ITEMS = -42, JDE_PATHS = -43, JDE_NEW = -44, JDE_BARRIER = -50 };

struct DecConst {
    // network (CSR in HBM)
    const int *row_ptr; const JdArc *arcs; const float *fin_w; int init_state;
    const int *aux;     // per arc: {nStates | transMat << 8, g0, g1, g2} (+ {g3, g4, g5, 0} for > 5 states)
    // models
    int G, max_n, n_tm;
    const float *hmm_tee;
    const float *trP; const int *se32;
    // pruning (WFSTDecoderLite ctor, WFSTDecoderLite.cpp:38-82)
    float start_win, emit_win, end_win, word_win;
    int max_hyps, hist_min, hist_max, hist_nbins;
    // arena capacities (per stream, in records)
    unsigned cap_slots, cap_items, cap_new; int cap_paths;
    int gc_threshold;   // a launch stops early (for k_gc) when more Path records than this are in use
};

// An active arc instance (NetInst, WFSTDecoderLite.h:66-75) is ONE self-contained record: header
// (arc, topology, tied-state ids, arc weight) + the tokens of its emitting states.  With <= 5 HMM
// states it is one 128-byte line, up to 8 states take two.  Entry and exit tokens are never
// stored: the entry token is pulled from the arc's key in phase A, the exit token is consumed by
// phase X of the same frame (:964).
//   ints [0..3] = arc, nStates | transMat << 8, outLabel, toState
//   ints [4..7] = g0, g1, g2, arc weight     (tied-state ids of emitting states 1..3)
//   GS == 8: ints [8..11] = g3, g4, g5, -;   token of state i at int TOK_OFF + 4 i
template <int GS> struct RecLayout {
    static constexpr int REC_INTS = (GS == 4) ? 32 : 64;
    static constexpr int REC_BYTES = REC_INTS * 4;
    static constexpr int TOK_OFF = (GS == 4) ? 8 : 12;
    static constexpr int K = 64 / GS;                  // instances per wave pass (= chunk of a record / new-arc list)
};

// per-arc search state: recombination key of this frame + "an instance of this arc is in the list"
struct __align__(16) ArcState { unsigned long long key; int live; int pad; };

// per-stream scalars.  Line 0 is written by the host-side helper kernels and by workgroup 0 of the
// stream's cluster at the END of a launch (nobody reads it while a launch runs, except at its
// start); every word that is updated atomically while a launch runs has its own 128-byte line.
struct __align__(128) StreamCtl {
    int frame;          // next frame to process
    int T;              // frames available
    int error, needs_init, started;
    int lst_nw;         // number of wave segments the current lists were written with
    int n_rec_hint;     // instances in the current list (statistics / capacity planning only)
    float best_emit;    // bestEmitScore left by the last processed frame (:321)
    int pad0[24];
    __align__(128) unsigned bar;             // cluster barrier (zeroed by the host before every launch)
    __align__(128) unsigned bestA[2];        // ordered-uint best emitting score of phase A, by frame parity
    __align__(128) unsigned bestX[2];        // ... best entry-token candidate of phase X
    __align__(128) int n_paths;              // Path records in use
    __align__(128) unsigned long long final_key;
    __align__(128) int err[2];               // first error raised during a frame of that parity
    __align__(128) long long st[ST_N];       // statistics (WFSTDecoderLite.cpp:231-241 + build counters)
    __align__(128) Tok best_final;           // bestFinalToken of the last processed frame
};

struct StreamDev {      // per-stream arenas
    int *rec[2];                      // instance records, by frame parity: list f&1 is read by frame f
    ArcState *ast;                    // per ARC
    unsigned long long *skey[2];      // per STATE: best frontier item arriving there (round parity)
    unsigned long long *skeyL;        // round 0 only: items whose arc carries a word label (own threshold)
    Tok *item_tok[2]; int4 *item_info[2];   // frontier items of a frame, by frame parity: token + {arc, out, to, -}
    int *newl;                        // arcs entered for the first time this frame (no live instance)
    int *tot;                         // published per-wave fill counts: [rec0 | rec1 | new | exit | closure0 | closure1][MAXW]
    int *item_end;                    // per wave: items written in the last processed frame (k_gc)
    PathRec *paths; int *hist;        // hist: [2][HIST_MAX_BINS] by frame parity
    PathRec *paths2; int *gc_idx;     // Path garbage collection: compaction target + mark / new-index array
    // result of jd_finish_kernel
    int res_n; int *res_label; int *res_time; float *res_score, *res_ac, *res_lm; int res_cap;
};
enum { TOT_REC0 = 0, TOT_REC1 = 1, TOT_NEW = 2, TOT_EXIT = 3, TOT_CL0 = 4, TOT_CL1 = 5, TOT_N = 6 };

struct SearchArgs {
    DecConst C;
    StreamCtl *ctl; StreamDev *streams;
    const int2 *work;        // {stream, likelihood slot} of every stream this launch advances
    int n_work;
    int Cw;                  // workgroups per cluster
    int n_slots;             // clusters in the grid (slot q serves work items q, q + n_slots, ...)
    int pack;                // 1: a cluster's workgroups are n_slots blocks apart (same XCD when n_slots % 8 == 0)
    const float *ll; long long ll_stride; int f0;   // likelihoods: ll[slot * ll_stride + (f - f0) * G + g]
    int f_end;               // process frames < min(T, f_end)
    int *status;             // += 1 for every stream that stopped early (Path garbage collection needed)
    long long *dbg;          // optional: per-workgroup cycle accounting [phase A, barrier, phase X, barrier, frames]
};

// ------------------------------------------------------------------ device helpers

typedef int v4i __attribute__((ext_vector_type(4)));
#define AUX_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *p, unsigned long long bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(unsigned)(bytes > 0xffffffffULL ? 0xffffffffULL : bytes), 0x00020000);
}
// 16-byte agent-scope (sc1) accesses through a wave-uniform buffer descriptor (out-of-range
// offsets read 0 / are dropped by the hardware bounds check)
__device__ __forceinline__ v4i ld16(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX_SC1); }
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned off, v4i v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, AUX_SC1); }
template <typename T> __device__ __forceinline__ T CL(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void CS(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Tok as_tok(v4i v) { Tok t; t.score = __int_as_float(v.x); t.ac = __int_as_float(v.y); t.lm = __int_as_float(v.z); t.path = v.w; return t; }
__device__ __forceinline__ v4i as_v4(const Tok &t) { v4i v; v.x = __float_as_int(t.score); v.y = __float_as_int(t.ac); v.z = __float_as_int(t.lm); v.w = t.path; return v; }
__device__ __forceinline__ Tok null_tok() { Tok t; t.score = LZ; t.ac = LZ; t.lm = LZ; t.path = -1; return t; }
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned y = __shfl_xor(v, o); v = y > v ? y : v; }
    return v;
}

struct SearchShared {
    int pfx_a[MAXW + 1]; int cnt_a[MAXW];      // chunk prefix / fill counts of list a (records; items)
    int pfx_b[MAXW + 1]; int cnt_b[MAXW];      // ... of list b (newly entered arcs)
    int start[MAXW];                           // per writer wave: where its items of the current round begin
    float trP[TRP_LDS_MAX]; int se[TRP_LDS_MAX / 4];
    int hist[HIST_MAX_BINS];                   // this workgroup's share of the frame's histogram
    int hprev[HIST_MAX_BINS];                  // the stream's bins of the previous frame
    int wsum[SW];
    unsigned best;
    int abort;
    int stat[ST_N];                            // this workgroup's counters of the current frame
    long long clk[4];
};

// exclusive prefix over the per-wave chunk counts of a published list.  All SNT threads call it.
// K = records per chunk; segcap = capacity of one wave segment (counts are clamped to it).
__device__ __forceinline__ int build_prefix(SearchShared &sh, int *pfx, int *cnt, const int *tot, int nw, int K, unsigned segcap)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int c = 0;
    if (tid < nw) { c = CL(tot + tid); if (c < 0) c = 0; if ((unsigned)c > segcap) c = (int)segcap; }
    const int n = (c + K - 1) / K;
    int x = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) sh.wsum[wid] = x;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SW; ++w) { const int s = sh.wsum[w]; if (w < wid) base += s; total += s; }
    if (tid < nw) { pfx[tid] = base + x - n; cnt[tid] = c; }
    if (tid == 0) pfx[nw] = total;
    __syncthreads();
    return total;
}

// largest w in [0, nw) with pfx[w] <= r (r < pfx[nw], wave-uniform): two ballot steps.  Empty
// segments (equal prefix values) are skipped because the LAST of equal entries is returned.
__device__ __forceinline__ int find_seg(const int *pfx, int nw, int r)
{
    const int lane = threadIdx.x & 63;
    const int stride = (nw + 63) >> 6;
    int i1 = lane * stride; if (i1 > nw) i1 = nw;
    const unsigned long long m1 = __ballot(pfx[i1] <= r);
    const int base = (__popcll(m1) - 1) * stride;
    int i2 = base + lane; if (i2 > nw) i2 = nw;
    const unsigned long long m2 = __ballot(lane < stride && pfx[i2] <= r);
    return base + __popcll(m2) - 1;
}

// ---- cluster barrier: all Cw workgroups of one stream.  target = Cw * (number of this barrier).
__device__ __forceinline__ void cluster_barrier(SearchShared &sh, StreamCtl &c, int Cw, unsigned &nbar, long long t_limit)
{
    // every wave: its write-through stores and atomics have been performed before anybody is told
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ++nbar;
    if (Cw > 1 && threadIdx.x == 0) {
        __hip_atomic_fetch_add(&c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = nbar * (unsigned)Cw;
        unsigned spins = 0;
        while (CL(&c.bar) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && wall_clock64() > t_limit) { sh.abort = 1; break; }
        }
    }
    __syncthreads();
}

// ---- Histogram::calcThresh (Histogram.cpp:134-158) over the bins of the previous frame, by one wave
__device__ __forceinline__ float hist_threshold(const DecConst &C, const int *sh_hist, int lane)
{
    const int nb = C.hist_nbins;
    const int K = (nb + 63) >> 6;
    const int hi = nb - 1 - lane * K;
    int sum = 0;
    for (int k = 0; k < K; ++k) { const int b = hi - k; if (b >= 0) sum += sh_hist[b]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    const int total = __shfl(inc, 63);
    if (total <= C.max_hyps) return (float)C.hist_min - 0.5f;
    const unsigned long long m = __ballot(inc >= C.max_hyps);
    const int L = __ffsll((long long)m) - 1;
    int res = 0;
    if (lane == L) {
        int acc = inc - sum;
        for (int k = 0; k < K; ++k) {
            const int b = hi - k;
            if (b < 0) break;
            acc += sh_hist[b];
            res = b;
            if (acc >= C.max_hyps) break;
        }
    }
    res = __shfl(res, L);
    return (float)(res + C.hist_min) - 0.5f;
}

struct Geo {            // geometry of the wave-segmented lists
    int nw;             // writer waves (= Cw * SW of the launch that wrote the list)
    unsigned seg_rec, seg_item, seg_new;    // records per wave segment
};
__device__ __forceinline__ Geo make_geo(const DecConst &C, int nw)
{
    Geo g; g.nw = nw;
    g.seg_rec = C.cap_slots / (unsigned)nw; g.seg_item = C.cap_items / (unsigned)nw; g.seg_new = C.cap_new / (unsigned)nw;
    return g;
}

struct StreamView {     // wave-uniform descriptors of one stream's arenas
    __amdgpu_buffer_rsrc_t rec[2], itok[2], iinfo[2];
    ArcState *ast; unsigned long long *skey[2], *skeyL; int *newl; int *tot; PathRec *paths; int *hist;
};

// ------------------------------------------------------------------ phase A
//
// doHMMInternalPropagation (:899-935) + HMMInternalPropagation (:376-484).  GS consecutive lanes
// own one instance: lane r updates emitting state r+1, lane GS-1 builds the exit token from its
// neighbours' results.  A wave takes chunks (K instances of ONE writer segment) round-robin;
// survivors and exit tokens go to the wave's own output segments - no atomics, no barriers.
template <int GS, bool TRPL>
__device__ __forceinline__ void phase_a(const DecConst &C, SearchShared &sh, StreamCtl &c, const StreamView &V,
                                        const Geo &gin, const Geo &gout, int Qr, int Qn, int gw, int NW, int f,
                                        float normalise, float emitTh, float startTh, const float *llrow,
                                        int &out_cnt, int &exit_cnt)
{
    const float *trP_all = TRPL ? sh.trP : C.trP;                      // transition tables: LDS copy when they fit
    const int *se_all = TRPL ? sh.se : C.se32;
    typedef RecLayout<GS> RL;
    constexpr int K = RL::K;
    const int lane = threadIdx.x & 63;
    const int r = lane & (GS - 1), gb = lane & ~(GS - 1), grp = lane / GS;
    const int MN = C.max_n;
    const int p = f & 1;
    const bool use_hist = C.max_hyps > 0;
    const __amdgpu_buffer_rsrc_t rec_cur = V.rec[p], rec_next = V.rec[p ^ 1];
    const __amdgpu_buffer_rsrc_t itok_prev = V.itok[p ^ 1];
    const __amdgpu_buffer_rsrc_t itok_cur = V.itok[p], iinfo_cur = V.iinfo[p];
    const unsigned rec_base = (unsigned)gw * gout.seg_rec, item_base = (unsigned)gw * gout.seg_item;
    int c_insts = 0, c_pemit = 0, c_emit = 0, c_end = 0, c_surv = 0;
    unsigned mo = 0u;
    for (int u = gw; u < Qr + Qn; u += NW) {
        const bool is_new = u >= Qr;
        const int ru = is_new ? u - Qr : u;
        const int *pfx = is_new ? sh.pfx_b : sh.pfx_a;
        const int *cnt = is_new ? sh.cnt_b : sh.cnt_a;
        const int w = find_seg(pfx, gin.nw, ru);
        const int ci = ru - pfx[w];
        int fill = cnt[w] - ci * K; if (fill > K) fill = K;
        const bool valid = grp < fill;
        v4i h0 = {0, 0, 0, 0}, h1 = h0, h2 = h0;
        Tok spec0 = null_tok(), spec1 = null_tok();
        unsigned roff = 0;
        if (valid) {
            if (!is_new) {
                roff = ((unsigned)w * gin.seg_rec + (unsigned)(ci * K + grp)) * RL::REC_BYTES;
                h0 = ld16(rec_cur, roff); h1 = ld16(rec_cur, roff + 16);
                if (GS == 8) h2 = ld16(rec_cur, roff + 32);
                // left-to-right HMMs read states r and r+1: issued together with the header
                if (r >= 1) spec0 = as_tok(ld16(rec_cur, roff + (RL::TOK_OFF + 4 * r) * 4));
                if (r + 1 < MN) spec1 = as_tok(ld16(rec_cur, roff + (RL::TOK_OFF + 4 * (r + 1)) * 4));
            } else {                                                   // attachNetInst :751-774, from the arc's template
                const int b = CL(V.newl + (size_t)w * gin.seg_new + (unsigned)(ci * K + grp));
                const JdArc Bk = C.arcs[b];
                const int4 a0 = ((const int4 *)C.aux)[(GS == 4) ? b : 2 * b];
                h0 = (v4i){b, a0.x, Bk.out, Bk.to};
                h1 = (v4i){a0.y, a0.z, a0.w, __float_as_int(Bk.w)};
                if (GS == 8) { const int4 a1 = ((const int4 *)C.aux)[2 * b + 1]; h2 = (v4i){a1.x, a1.y, a1.z, 0}; }
            }
        }
        const int arc = h0.x;
        const int n = h0.y & 0xff;
        const int tm = h0.y >> 8;
        // entry token = the best candidate phase X of the previous frame left in the arc's key (:560-582)
        Tok entry = null_tok();
        if (valid && r == 0) {
            ArcState *as = V.ast + arc;
            const unsigned long long kv = CL(&as->key);
            if (kv != 0ULL) {
                CS(&as->key, 0ULL);
                const Tok it = as_tok(ld16(itok_prev, (unsigned)(kv & 0xffffffffULL) * 16u));
                entry.score = o2f((unsigned)(kv >> 32));
                entry.ac = it.ac; entry.lm = it.lm + __int_as_float(h1.w); entry.path = it.path;
            }
        }
        entry.score = __shfl(entry.score, gb); entry.ac = __shfl(entry.ac, gb);
        entry.lm = __shfl(entry.lm, gb); entry.path = __shfl(entry.path, gb);
        if (entry.score > LZ && entry.score < startTh) entry = null_tok();                  // :915-918
        const float *trP = trP_all + (size_t)tm * MN * MN;
        const int *se = se_all + (size_t)tm * MN;
        auto tok_at = [&](int i) -> Tok {
            if (i == 0) return entry;
            if (is_new) return null_tok();
            if (i == r) return spec0;
            if (i == r + 1) return spec1;
            return as_tok(ld16(rec_cur, roff + (RL::TOK_OFF + 4 * i) * 4));
        };
        bool emit_live = false, pemit = false;
        Tok nw = null_tok();
        const int j = r + 1;
        if (valid && j < n - 1) {                                      // :387-424 emitting state j
            int gmj;
            if (GS == 4) gmj = (r == 0) ? h1.x : (r == 1) ? h1.y : h1.z;
            else gmj = (r == 0) ? h1.x : (r == 1) ? h1.y : (r == 2) ? h1.z : (r == 3) ? h2.x : (r == 4) ? h2.y : h2.z;
            const float outp = llrow[gmj];                             // :411
            const int sev = se[j];
            const int st = sev & 0xffff, en = sev >> 16;
            Tok src = tok_at(st);
            float btp = trP[st * MN + j];
            float best = src.score + btp;
            for (int i = st + 1; i < en; ++i) {
                const Tok cnd = tok_at(i);
                const float tp = trP[i * MN + j];
                const float tmp = cnd.score + tp;
                if (tmp > best) { best = tmp; btp = tp; src = cnd; }
            }
            const float sc = best - normalise;                         // :408
            if (sc > emitTh) {                                         // :409
                pemit = true;
                nw.score = sc + outp;
                nw.ac = (src.ac + btp) + outp;
                nw.lm = src.lm;
                nw.path = src.path;
                emit_live = true;
                if (use_hist) {                                        // Histogram::addScore, Histogram.cpp:64-100
                    const double ds = (double)nw.score;
                    const int sci = (nw.score < 0.0f) ? (int)(ds - 0.5) : (int)(ds + 0.5);
                    if (sci > C.hist_max) CS(&c.err[p], (int)JD_EHIST);
                    else if (sci >= C.hist_min) atomicAdd(&sh.hist[sci - C.hist_min], 1);
                }
                const unsigned so = f2o(nw.score);
                mo = so > mo ? so : mo;
            }
        }
        // exit state (:443-483): lane GS-1 of the group reads the NEW tokens of its neighbours
        Tok ex = null_tok();
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
        }
        const bool has_exit = ex.score > LZ;
        const unsigned long long bemit = __ballot(emit_live);
        const bool slot_live = ((bemit >> gb) & ((1ull << GS) - 1ull)) != 0ull;
        const bool lead = valid && r == 0;
        const unsigned long long bl = __ballot(lead && slot_live), be = __ballot(has_exit);
        c_insts += __popcll(__ballot(lead));
        c_pemit += __popcll(__ballot(pemit));
        c_emit += __popcll(bemit);
        // survivors: header + new tokens to this wave's segment of the next list
        {
            int pos = (lead && slot_live) ? out_cnt + rank_in(bl) : -1;
            pos = __shfl(pos, gb);
            const int nsurv = __popcll(bl);
            if (out_cnt + nsurv > (int)gout.seg_rec) { if (lane == 0) CS(&c.err[p], (int)JDE_SLOTS); }
            else {
                if (valid && pos >= 0) {
                    const unsigned doff = (rec_base + (unsigned)pos) * RL::REC_BYTES;
                    if (j < n - 1) st16(rec_next, doff + (RL::TOK_OFF + 4 * j) * 4, as_v4(nw));
                    if (r == GS - 1) {
                        st16(rec_next, doff, h0); st16(rec_next, doff + 16, h1);
                        if (GS == 8) st16(rec_next, doff + 32, h2);
                    }
                }
                out_cnt += nsurv;
                c_surv += nsurv;
            }
            // the arc's "has an instance" flag changes at birth and death only (returnNetInst :777-797)
            if (lead && slot_live && is_new) CS(&V.ast[arc].live, 1);
            if (lead && !slot_live && !is_new) CS(&V.ast[arc].live, 0);
        }
        // exit tokens: frontier items of round 0 in this wave's item segment, bidding for their
        // destination state (state-level recombination, see phase X); tokens leaving word-labelled
        // arcs face their own threshold (:952-962) -> own key class
        {
            const int nex = __popcll(be);
            if (exit_cnt + nex > (int)gout.seg_item) { if (lane == 0) CS(&c.err[p], (int)JDE_ITEMS); }
            else {
                if (has_exit) {
                    const unsigned k = item_base + (unsigned)(exit_cnt + rank_in(be));
                    st16(itok_cur, k * 16u, as_v4(ex));
                    st16(iinfo_cur, k * 16u, (v4i){arc, h0.z, h0.w, 0});
                    atomicMax((h0.z != 0 ? V.skeyL : V.skey[0]) + h0.w, ((unsigned long long)f2o(ex.score) << 32) | k);
                }
                exit_cnt += nex;
                c_end += nex;
            }
        }
    }
    // per-wave totals -> workgroup counters (LDS)
    mo = wave_umax(mo);
    if (lane == 0) {
        if (mo) atomicMax(&sh.best, mo);
        if (c_insts) atomicAdd(&sh.stat[ST_INSTS], c_insts);
        if (c_pemit) atomicAdd(&sh.stat[ST_PEMIT], c_pemit);
        if (c_emit) atomicAdd(&sh.stat[ST_EMIT], c_emit);
        if (c_end) atomicAdd(&sh.stat[ST_END], c_end);
        if (c_surv) atomicAdd(&sh.stat[ST_MODELS], c_surv);
    }
}

// ------------------------------------------------------------------ phase X
//
// propagateToken (:491-605) for one round of frontier items.  A group of EG lanes owns one item; the
// NGRP items of a wave pool their out-arcs (lane l takes arcs l, l+64, ... of the concatenated
// ranges), so a history state with thousands of out-arcs occupies the whole wave.  State-level
// recombination: of all items that reached a state in one round only the best (per threshold
// class) is expanded - every item would add the same arc weights, and float addition is monotone,
// so no other item can win anything downstream.
struct XOut { int item_cnt; int new_cnt; };
__device__ __forceinline__ void phase_x(const DecConst &C, SearchShared &sh, StreamCtl &c, const StreamView &V,
                                        const Geo &gin, const Geo &gout, int Q, int round, int gw, int NW,
                                        int p, int pframe, bool init, bool last_frame, float endTh, float wordTh,
                                        XOut &out, int &round_items)
{
    const int lane = threadIdx.x & 63;
    const int er = lane & (EG - 1), eb = lane & ~(EG - 1), grp = lane / EG;
    const float INF = __builtin_inff();
    const __amdgpu_buffer_rsrc_t itok = V.itok[p], iinfo = V.iinfo[p];
    unsigned long long *sk_in_u = V.skey[round & 1];
    unsigned long long *sk_in_l = (round == 0) ? V.skeyL : V.skey[round & 1];
    unsigned long long *sk_out = V.skey[(round & 1) ^ 1];
    const bool check_th = round == 0 && !init;                         // :952-962
    const unsigned item_base = (unsigned)gw * gout.seg_item, new_base = (unsigned)gw * gout.seg_new;
    int c_arcs = 0, c_paths = 0, c_pend = 0, c_new = 0;
    unsigned mo = 0u;
    for (int u = gw; u < Q; u += NW) {
        const int w = find_seg(sh.pfx_a, gin.nw, u);
        const int ci = u - sh.pfx_a[w];
        int fill = sh.cnt_a[w] - ci * NGRP; if (fill > NGRP) fill = NGRP;
        const bool valid = grp < fill;
        bool have = valid;
        const unsigned ii = (unsigned)w * gin.seg_item + (unsigned)(sh.start[w] + ci * NGRP + grp);
        Tok t = null_tok();
        v4i info = {-1, 0, 0, 0};
        if (have) { info = ld16(iinfo, ii * 16u); t = as_tok(ld16(itok, ii * 16u)); }
        int rs = 0, rs1 = 0;
        bool labelled = false;
        if (have) {
            const int state = (info.x >= 0) ? info.z : C.init_state;
            rs = C.row_ptr[state];                                     // issued before the winner is known
            rs1 = C.row_ptr[state + 1];
            if (info.x >= 0) {
                if (check_th) {
                    have = t.score > ((info.y != 0) ? wordTh : endTh);
                    if (have && er == 0) ++c_pend;
                }
                labelled = have && info.y != 0 && er == 0;
            }
        }
        // Path records (:497-509) are reserved for every labelled item that passed its threshold,
        // winner or not, so that the reservation is in flight together with the key load
        const unsigned long long blab = __ballot(labelled);
        int pbase = 0;
        if (blab) {
            const int first = __ffsll((long long)blab) - 1;
            if (lane == first) pbase = atomicAdd(&c.n_paths, __popcll(blab));
            pbase = __shfl(pbase, first);
        }
        if (valid && info.x >= 0) {
            // every state that received a bid is cleaned up by its winner, expanded or not (an item
            // below its threshold still holds the key of its state if it was the best one there)
            unsigned long long *sk = ((info.y != 0) ? sk_in_l : sk_in_u) + info.z;
            const unsigned long long kv = CL(sk);
            const bool winner = (unsigned)(kv & 0xffffffffULL) == ii && kv != 0ULL;
            if (winner && er == 0) CS(sk, 0ULL);
            have = have && winner;
        }
        int deg = 0;
        if (have) {
            if (info.x >= 0) {
                if (info.y != 0) {
                    int pp = labelled ? pbase + rank_in(blab) : -1;
                    pp = __shfl(pp, eb);
                    if (pp < C.cap_paths) {
                        if (er == 0) {
                            PathRec pr;
                            pr.prev = t.path; pr.frame = pframe; pr.label = info.y; pr.pad0 = 0;
                            pr.score = t.score; pr.ac = t.ac; pr.lm = t.lm; pr.pad1 = 0.0f;
                            V.paths[pp] = pr;                          // read by later launches only
                            Tok tn = t; tn.path = pp;
                            st16(itok, ii * 16u, as_v4(tn));           // the candidates of this item carry the new history
                            ++c_paths;
                        }
                        t.path = pp;
                    } else if (er == 0) CS(&c.err[p], (int)JDE_PATHS);
                }
                // :513-520 final state.  bestFinalToken is reset every frame (:316) and only read by
                // finish(), so it only has to be evaluated on the last frame that is available.
                if (er == 0 && last_frame && !init) {
                    const float fw = C.fin_w[info.z];
                    if (fw < INF) {
                        const float cs = t.score + fw;
                        if (cs > LZ) atomicMax(&c.final_key, ((unsigned long long)f2o(cs) << 32) | ii);
                    }
                }
            }
            deg = rs1 - rs;
        }
        // ---- pooled arc walk
        const int d0 = __shfl(deg, 0), d1 = __shfl(deg, EG), d2 = __shfl(deg, 2 * EG), d3 = __shfl(deg, 3 * EG);
        const int p1 = d0, p2 = d0 + d1, p3 = p2 + d2, tot = p3 + d3;
        for (int a0 = 0; a0 < tot; a0 += 64) {
            const int a = a0 + lane;
            const int g = (a >= p1) + (a >= p2) + (a >= p3);
            const int off = a - ((g == 0) ? 0 : (g == 1) ? p1 : (g == 2) ? p2 : p3);
            const int srcl = g * EG;
            Tok tg;
            tg.score = __shfl(t.score, srcl); tg.ac = __shfl(t.ac, srcl);
            tg.lm = __shfl(t.lm, srcl); tg.path = __shfl(t.path, srcl);
            const unsigned iig = (unsigned)__shfl((int)ii, srcl);
            const int rsg = __shfl(rs, srcl);
            bool mk = false, touch = false;
            Tok un = null_tok();
            v4i uinfo = {-1, 0, 0, 0};
            int tb = -1;
            if (a < tot) {
                const int b = rsg + off;
                const JdArc Bk = C.arcs[b];
                ++c_arcs;
                const int inl = Bk.in & ~TEE_FLAG;
                if (inl == 0) {                                        // :533-540 epsilon input
                    un = tg;
                    un.score = tg.score + Bk.w;
                    un.lm = tg.lm + Bk.w;
                    mk = un.score > endTh;
                    uinfo = (v4i){b, Bk.out, Bk.to, 0};
                } else {                                               // :560-582 entry-token recombination
                    const float ns = tg.score + Bk.w;
                    const unsigned so = f2o(ns);
                    ArcState *as = V.ast + b;
                    const int lv = CL(&as->live);                      // in flight together with the atomic
                    const unsigned long long old = atomicMax(&as->key, ((unsigned long long)so << 32) | iig);
                    touch = (old == 0ULL) && (lv == 0);                // first candidate of an arc without an instance
                    tb = b;
                    mo = so > mo ? so : mo;                            // :572-573
                    if (Bk.in & TEE_FLAG) {                            // :584-600 tee model
                        const float tee = C.hmm_tee[inl - 1];
                        const float ns2 = ns + tee;
                        un.score = ns2;
                        un.ac = tg.ac + tee;
                        un.lm = tg.lm + Bk.w;
                        un.path = tg.path;
                        mk = ns2 > ((Bk.out != 0) ? wordTh : endTh);
                        uinfo = (v4i){b, Bk.out, Bk.to, 0};
                    }
                }
            }
            // newly entered arcs -> this wave's segment of the new list
            const unsigned long long bt = __ballot(touch);
            if (bt) {
                const int nt = __popcll(bt);
                if (out.new_cnt + nt > (int)gout.seg_new) { if (lane == 0) CS(&c.err[p], (int)JDE_NEW); }
                else {
                    if (touch) CS(V.newl + (size_t)new_base + (unsigned)(out.new_cnt + rank_in(bt)), tb);
                    out.new_cnt += nt; c_new += nt;
                }
            }
            // closure items -> this wave's item segment (next round), bidding for their state
            const unsigned long long bm = __ballot(mk);
            if (bm) {
                const int nm = __popcll(bm);
                if (out.item_cnt + nm > (int)gout.seg_item) { if (lane == 0) CS(&c.err[p], (int)JDE_ITEMS); }
                else {
                    if (mk) {
                        const unsigned k = item_base + (unsigned)(out.item_cnt + rank_in(bm));
                        st16(itok, k * 16u, as_v4(un));
                        st16(iinfo, k * 16u, uinfo);
                        atomicMax(sk_out + uinfo.z, ((unsigned long long)f2o(un.score) << 32) | k);
                    }
                    out.item_cnt += nm; round_items += nm;
                }
            }
        }
    }
    mo = wave_umax(mo);
    c_arcs = wave_sum(c_arcs); c_paths = wave_sum(c_paths); c_pend = wave_sum(c_pend);
    if (lane == 0) {
        if (mo) atomicMax(&sh.best, mo);
        if (c_arcs) atomicAdd(&sh.stat[ST_ARCS], c_arcs);
        if (c_paths) atomicAdd(&sh.stat[ST_PATHS], c_paths);
        if (c_pend) atomicAdd(&sh.stat[ST_PEND], c_pend);
        if (c_new) atomicAdd(&sh.stat[ST_MODELS], c_new);              // attached instances count as active models (:981)
    }
}

// ------------------------------------------------------------------ one stream, one launch

template <int GS>
__device__ void run_stream(const SearchArgs &A, SearchShared &sh, int s, int ll_slot, int jw)
{
    typedef RecLayout<GS> RL;
    const DecConst &C = A.C;
    StreamCtl &c = A.ctl[s];
    const StreamDev &S = A.streams[s];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Cw = A.Cw, NW = Cw * SW, gw = jw * SW + wid;
    const int MN = C.max_n;
    // ---- launch-constant state (line 0 of the control block is not written while the launch runs)
    int f = c.frame;
    const int T = c.T;
    const bool needs_init = c.needs_init != 0;
    if (!c.started || c.error != 0) return;
    const int f_stop = T < A.f_end ? T : A.f_end;
    if (!needs_init && f >= f_stop) return;
    float best_emit = c.best_emit;
    Geo gin = make_geo(C, c.lst_nw > 0 ? c.lst_nw : NW);
    const Geo gout = make_geo(C, NW);
    StreamView V;
    V.rec[0] = mk_rsrc(S.rec[0], (unsigned long long)C.cap_slots * RL::REC_BYTES);
    V.rec[1] = mk_rsrc(S.rec[1], (unsigned long long)C.cap_slots * RL::REC_BYTES);
    V.itok[0] = mk_rsrc(S.item_tok[0], (unsigned long long)C.cap_items * 16u);
    V.itok[1] = mk_rsrc(S.item_tok[1], (unsigned long long)C.cap_items * 16u);
    V.iinfo[0] = mk_rsrc(S.item_info[0], (unsigned long long)C.cap_items * 16u);
    V.iinfo[1] = mk_rsrc(S.item_info[1], (unsigned long long)C.cap_items * 16u);
    V.ast = S.ast; V.skey[0] = S.skey[0]; V.skey[1] = S.skey[1]; V.skeyL = S.skeyL; V.newl = S.newl; V.tot = S.tot;
    V.paths = S.paths; V.hist = S.hist;
    const bool use_hist = C.max_hyps > 0;
    const bool trp_lds = (size_t)C.n_tm * MN * MN <= TRP_LDS_MAX && (size_t)C.n_tm * MN <= TRP_LDS_MAX / 4;
    __syncthreads();                                                   // the previous stream of this slot is done with LDS
    if (trp_lds) {
        for (int i = tid; i < C.n_tm * MN * MN; i += SNT) sh.trP[i] = C.trP[i];
        for (int i = tid; i < C.n_tm * MN; i += SNT) sh.se[i] = C.se32[i];
    }
    if (tid == 0) { sh.abort = 0; sh.best = 0u; for (int k = 0; k < ST_N; ++k) sh.stat[k] = 0; for (int k = 0; k < 4; ++k) sh.clk[k] = 0; }
    if (use_hist) for (int b = tid; b < C.hist_nbins; b += SNT) sh.hist[b] = 0;
    __syncthreads();
    unsigned nbar = 0;
    const long long t_limit = wall_clock64() + 3000000000LL;           // 30 s at 100 MHz: a lost workgroup, not a slow one
    int frames_done = 0;
    int my_item_end = 0;                                               // items this wave wrote in the last processed frame
    long long st_acc[ST_N];
#pragma unroll
    for (int k = 0; k < ST_N; ++k) st_acc[k] = 0;
    bool aborted = false, failed = false;

    // =============================================================== recognitionStart (:139-228)
    if (needs_init) {
        // drop whatever the previous utterance left behind: instance flags and pending candidates
        {
            const int p0 = f & 1;
            const int Qr = build_prefix(sh, sh.pfx_a, sh.cnt_a, V.tot + (size_t)(TOT_REC0 + p0) * MAXW, gin.nw, RL::K, gin.seg_rec);
            const int Qn = build_prefix(sh, sh.pfx_b, sh.cnt_b, V.tot + (size_t)TOT_NEW * MAXW, gin.nw, RL::K, gin.seg_new);
            const int grp = lane / GS, r = lane & (GS - 1);
            for (int u = gw; u < Qr + Qn; u += NW) {
                const bool is_new = u >= Qr;
                const int ru = is_new ? u - Qr : u;
                const int *pfx = is_new ? sh.pfx_b : sh.pfx_a;
                const int *cnt = is_new ? sh.cnt_b : sh.cnt_a;
                const int w = find_seg(pfx, gin.nw, ru);
                const int ci = ru - pfx[w];
                int fill = cnt[w] - ci * RL::K; if (fill > RL::K) fill = RL::K;
                if (grp < fill && r == 0) {
                    int b;
                    if (!is_new) b = ld16(V.rec[p0], ((unsigned)w * gin.seg_rec + (unsigned)(ci * RL::K + grp)) * RL::REC_BYTES).x;
                    else b = CL(V.newl + (size_t)w * gin.seg_new + (unsigned)(ci * RL::K + grp));
                    CS(&V.ast[b].key, 0ULL); CS(&V.ast[b].live, 0);
                }
            }
            if (use_hist) for (int b = jw * SNT + tid; b < 2 * HIST_MAX_BINS; b += Cw * SNT) CS(V.hist + b, 0);
            if (jw == 0 && tid == 0) {
                CS(&c.bestA[0], 0u); CS(&c.bestA[1], 0u); CS(&c.bestX[0], 0u); CS(&c.bestX[1], 0u);
                CS(&c.n_paths, 0); CS(&c.final_key, 0ULL); CS(&c.err[0], 0); CS(&c.err[1], 0);
                for (int k = 0; k < ST_N; ++k) CS(&c.st[k], 0LL);
                c.best_final = null_tok();
                // the start token (:221-226) is the only item of round 0, in wave 0's segment (parity 1)
                Tok z; z.score = 0.0f; z.ac = 0.0f; z.lm = 0.0f; z.path = -1;
                st16(V.itok[1], 0u, as_v4(z)); st16(V.iinfo[1], 0u, (v4i){-1, 0, 0, 0});
            }
            cluster_barrier(sh, c, Cw, nbar, t_limit);
            aborted = sh.abort != 0;
            // every workgroup has read the old lists (possibly written with another geometry): they are
            // gone, this launch's geometry applies and every segment starts empty.  (Visible to the
            // others after the barrier that ends round 0 of the start expansion; the closure and
            // new-arc counts are published by every wave in every round anyway.)
            gin = gout;
            if (lane == 0) {
                CS(V.tot + (size_t)TOT_REC0 * MAXW + gw, 0); CS(V.tot + (size_t)TOT_REC1 * MAXW + gw, 0);
                CS(V.tot + (size_t)TOT_EXIT * MAXW + gw, 0);
            }
        }
        f = 0;
        // propagate the start token from the initial state: item / key parity 1 (the frame "before" 0)
        if (!aborted) {
            XOut xo = {0, 0};
            if (gw == 0) xo.item_cnt = 1;                              // the start item occupies wave 0's first slot
            for (int round = 0; !aborted; ++round) {
                int Q;
                if (round == 0) {
                    if (tid < gin.nw) { sh.pfx_a[tid] = tid == 0 ? 0 : 1; sh.cnt_a[tid] = tid == 0 ? 1 : 0; sh.start[tid] = 0; }
                    if (tid == 0) sh.pfx_a[gin.nw] = 1;
                    __syncthreads();
                    Q = 1;
                } else {
                    if (tid < gin.nw) sh.start[tid] += sh.cnt_a[tid];
                    __syncthreads();
                    Q = build_prefix(sh, sh.pfx_a, sh.cnt_a, V.tot + (size_t)(TOT_CL0 + (round & 1)) * MAXW, gin.nw, NGRP, gin.seg_item);
                    if (Q == 0) break;
                }
                int round_items = 0;
                phase_x(C, sh, c, V, gin, gout, Q, round, gw, NW, 1, 0, true, false, LZ, LZ, xo, round_items);
                if (lane == 0) {
                    CS(V.tot + (size_t)(TOT_CL0 + ((round + 1) & 1)) * MAXW + gw, round_items);
                    CS(V.tot + (size_t)TOT_NEW * MAXW + gw, xo.new_cnt);
                }
                __syncthreads();
                if (tid == 0 && sh.best) { atomicMax(&c.bestX[1], sh.best); sh.best = 0u; }
                cluster_barrier(sh, c, Cw, nbar, t_limit);
                aborted = sh.abort != 0;
            }
            my_item_end = xo.item_cnt;
            if (tid == 0)                                              // totalActiveModels starts with frame 0 (:981)
                for (int k = 0; k < ST_N; ++k) { if (k != ST_MODELS) st_acc[k] += sh.stat[k]; sh.stat[k] = 0; }
            if (!aborted) {
                const unsigned bx = CL(&c.bestX[1]);
                best_emit = bx ? o2f(bx) : LZ;
                failed = CL(&c.err[1]) != 0;
            }
        }
    }

    // =============================================================== frames
    while (!aborted && !failed && f < f_stop) {
        const int p = f & 1;
        // stop early when the Path arena needs collecting (k_gc runs between launches); n_paths only
        // changes in phase X, so every workgroup of the cluster reads the same value here
        if (frames_done > 0 && CL(&c.n_paths) > C.gc_threshold) break;
        long long t0 = 0, t1 = 0, t2 = 0;
        if (A.dbg && tid == 0) t0 = wall_clock64();
        // ---- frame start (:311-339): thresholds + the work lists of phase A
        const float normalise = (best_emit > LZ) ? best_emit : 0.0f;                 // :321
        float emitTh = (C.emit_win > 0.0f ? -C.emit_win : LZ);                       // :331
        if (use_hist) {
            // bins of the previous frame (parity p^1); every workgroup evaluates the same threshold
            for (int b = tid; b < C.hist_nbins; b += SNT) sh.hprev[b] = CL(V.hist + (size_t)(p ^ 1) * HIST_MAX_BINS + b);
            __syncthreads();
            float th = hist_threshold(C, sh.hprev, lane);
            th -= normalise;                                                         // :325
            if (C.emit_win > 0.0f && th < -C.emit_win) th = -C.emit_win;             // :326-327
            emitTh = th;
        }
        const float startTh = (C.start_win > 0.0f) ? (best_emit - C.start_win) : LZ; // :337
        const int Qr = build_prefix(sh, sh.pfx_a, sh.cnt_a, V.tot + (size_t)(TOT_REC0 + p) * MAXW, gin.nw, RL::K, gin.seg_rec);
        const int Qn = build_prefix(sh, sh.pfx_b, sh.cnt_b, V.tot + (size_t)TOT_NEW * MAXW, gin.nw, RL::K, gin.seg_new);
        const float *llrow = A.ll + (size_t)ll_slot * A.ll_stride + (size_t)(f - A.f0) * C.G;
        int out_cnt = 0, exit_cnt = 0;
        if (trp_lds) phase_a<GS, true>(C, sh, c, V, gin, gout, Qr, Qn, gw, NW, f, normalise, emitTh, startTh, llrow, out_cnt, exit_cnt);
        else phase_a<GS, false>(C, sh, c, V, gin, gout, Qr, Qn, gw, NW, f, normalise, emitTh, startTh, llrow, out_cnt, exit_cnt);
        if (lane == 0) {
            CS(V.tot + (size_t)(TOT_REC0 + (p ^ 1)) * MAXW + gw, out_cnt);
            CS(V.tot + (size_t)TOT_EXIT * MAXW + gw, exit_cnt);
        }
        __syncthreads();
        if (use_hist)                                                  // Histogram of this frame: workgroup bins -> stream bins
            for (int b = tid; b < C.hist_nbins; b += SNT) {
                const int v = sh.hist[b];
                if (v) { atomicAdd(V.hist + (size_t)p * HIST_MAX_BINS + b, v); sh.hist[b] = 0; }
            }
        if (tid == 0 && sh.best) { atomicMax(&c.bestA[p], sh.best); sh.best = 0u; }
        if (A.dbg && tid == 0) { t1 = wall_clock64(); sh.clk[0] += t1 - t0; }
        cluster_barrier(sh, c, Cw, nbar, t_limit);
        if (sh.abort) { aborted = true; break; }
        gin = gout;                                                    // every list read from here on was written by this launch
        if (A.dbg && tid == 0) { t2 = wall_clock64(); sh.clk[1] += t2 - t1; }
        // ---- phase X
        const unsigned ba = CL(&c.bestA[p]);
        const float bestA = ba ? o2f(ba) : LZ;
        const float endTh = (C.end_win > 0.0f) ? (bestA - C.end_win) : LZ;           // :349
        const float wordTh = (C.word_win > 0.0f) ? (bestA - C.word_win) : LZ;        // :350
        const bool last_frame = f >= T - 1;
        if (jw == 0) {                                                 // housekeeping for the frame after this one
            if (tid == 0) { CS(&c.bestA[p ^ 1], 0u); CS(&c.bestX[p ^ 1], 0u); }
            if (use_hist) for (int b = tid; b < C.hist_nbins; b += SNT) CS(V.hist + (size_t)(p ^ 1) * HIST_MAX_BINS + b, 0);
        }
        XOut xo = {exit_cnt, 0};
        for (int round = 0;; ++round) {
            int Q;
            if (round == 0) {
                if (tid < gin.nw) sh.start[tid] = 0;
                Q = build_prefix(sh, sh.pfx_a, sh.cnt_a, V.tot + (size_t)TOT_EXIT * MAXW, gin.nw, NGRP, gin.seg_item);
            } else {
                if (tid < gin.nw) sh.start[tid] += sh.cnt_a[tid];      // items of round r begin where those of round r-1 ended
                __syncthreads();
                Q = build_prefix(sh, sh.pfx_a, sh.cnt_a, V.tot + (size_t)(TOT_CL0 + (round & 1)) * MAXW, gin.nw, NGRP, gin.seg_item);
                if (Q == 0) break;
            }
            int round_items = 0;
            phase_x(C, sh, c, V, gin, gout, Q, round, gw, NW, p, f, false, last_frame, endTh, wordTh, xo, round_items);
            if (lane == 0) {
                CS(V.tot + (size_t)(TOT_CL0 + ((round + 1) & 1)) * MAXW + gw, round_items);
                CS(V.tot + (size_t)TOT_NEW * MAXW + gw, xo.new_cnt);
            }
            __syncthreads();
            if (tid == 0 && sh.best) { atomicMax(&c.bestX[p], sh.best); sh.best = 0u; }
            long long t3 = 0;
            if (A.dbg && tid == 0) { t3 = wall_clock64(); sh.clk[2] += t3 - t2; }
            cluster_barrier(sh, c, Cw, nbar, t_limit);
            if (sh.abort) { aborted = true; break; }
            if (A.dbg && tid == 0) { t2 = wall_clock64(); sh.clk[3] += t2 - t3; }
        }
        if (aborted) break;
        my_item_end = xo.item_cnt;
        // ---- frame end
        {
            const unsigned bx = CL(&c.bestX[p]);
            const unsigned bb = ba > bx ? ba : bx;
            best_emit = bb ? o2f(bb) : LZ;                             // :417-418, :572-573
        }
        if (tid == 0)
            for (int k = 0; k < ST_N; ++k) { st_acc[k] += sh.stat[k]; sh.stat[k] = 0; }
        if (last_frame && jw == 0 && tid == 0) {                       // bestFinalToken of this frame (:513-520)
            const unsigned long long fk = CL(&c.final_key);
            Tok bf = null_tok();
            if (fk != 0ULL) {
                const unsigned fi = (unsigned)(fk & 0xffffffffULL);
                const Tok it = as_tok(ld16(V.itok[p], fi * 16u));
                const float fw = C.fin_w[ld16(V.iinfo[p], fi * 16u).z];
                bf.score = o2f((unsigned)(fk >> 32)); bf.ac = it.ac; bf.lm = it.lm + fw; bf.path = it.path;
                CS(&c.final_key, 0ULL);
            }
            c.best_final = bf;
        }
        ++f; ++frames_done;
        if (CL(&c.err[p]) != 0) failed = true;                         // raised before this frame's last barrier: seen by all
    }

    if (aborted) {
        if (tid == 0 && jw == 0) { c.error = JDE_BARRIER; c.needs_init = 0; }
        return;
    }
    // ---- end of the launch: persist the stream state (read by the next launch / the host kernels)
    if (lane == 0) CS(S.item_end + gw, my_item_end);
    if (tid == 0) {
        for (int k = 0; k < ST_N; ++k) if (st_acc[k]) atomicAdd((unsigned long long *)&c.st[k], (unsigned long long)st_acc[k]);
        if (A.dbg) {
            long long *d = A.dbg + (size_t)blockIdx.x * 8;
            d[0] += sh.clk[0]; d[1] += sh.clk[1]; d[2] += sh.clk[2]; d[3] += sh.clk[3]; d[4] += frames_done;
        }
        if (jw == 0) {
            const int e0 = CL(&c.err[0]), e1 = CL(&c.err[1]);
            c.frame = f; c.best_emit = best_emit; c.lst_nw = NW; c.needs_init = 0;
            if (e0 | e1) c.error = e0 ? e0 : e1;
            else if (f < f_stop) atomicAdd(A.status, 1);               // stopped early: collect Path records, then go on
        }
    }
}

// Workgroup b belongs to cluster slot q = b / Cw (spread: consecutive blocks, i.e. consecutive
// XCDs, serve one stream) or q = b % n_slots (pack: a stream's workgroups are n_slots apart and
// share an XCD when n_slots is a multiple of 8).  All workgroups of the grid must be resident at
// once - the host sizes the grid to the device (one 1024-thread workgroup per CU).
template <int GS>
__global__ __launch_bounds__(SNT) void k_search(SearchArgs A)
{
    __shared__ SearchShared sh;
    const int q = A.pack ? (int)(blockIdx.x % (unsigned)A.n_slots) : (int)(blockIdx.x / (unsigned)A.Cw);
    const int jw = A.pack ? (int)(blockIdx.x / (unsigned)A.n_slots) : (int)(blockIdx.x % (unsigned)A.Cw);
    for (int k = q; k < A.n_work; k += A.n_slots) run_stream<GS>(A, sh, A.work[k].x, A.work[k].y, jw);
}
