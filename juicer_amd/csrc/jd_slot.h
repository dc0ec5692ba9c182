// jd_slot.h - the search of ONE utterance stream by ONE workgroup that takes half a CU: the slot kernel
// (included by jd_device.hip behind jd_search.h and jd_resident.h; gfx950 only).
//
// Reference: the same as jd_search.h - WFSTDecoderLite::processFrame (src/WFSTDecoderLite.cpp:311-372) =
// doHMMInternalPropagation (:899-935, :376-484) + doHMMExternalPropagation (:937-982) + propagateToken (:491-605),
// recognitionStart (:139-228), Histogram (src/Histogram.cpp).  The ALGORITHM, the arithmetic and every data format in
// HBM (instance records, frontier items, StateRec, wave-segmented lists, StreamCtl) are those of jd_search.h: a stream
// may be served by either kernel, the collection / finish / init kernels do not know the difference.
//
// Why a second kernel.  run_stream (jd_search.h) serves a stream with a CLUSTER of workgroups: its lists are sized for
// 512 writer waves, its counters live in HBM behind agent-scope atomics, a frame is a chain of cluster barriers and
// work-list round trips, and inlined it needs 256 VGPRs and 86 KB of LDS - one workgroup per CU, two waves per SIMD,
// SQ_WAIT_ANY two thirds of the wave cycles (round 4's verdict).  The batch pipeline (jd_device.hip, JD_FLOW_RESIDENT)
// runs every stream on ONE workgroup, which talks to nobody:
//   * every per-frame word (best scores, error, Path reservation cursor, arcs entered, histogram, the fill counts of the
//     lists) lives in LDS; HBM sees them once per command (128 frames);
//   * a frame is workgroup barriers only: one at its start, one behind phase A, one behind every round of the
//     expansion - the readers of a list build its prefix themselves from eight counts in LDS;
//   * phase A is NOT software-pipelined: the pipeline of jd_search.h (next record + next keys in flight behind the
//     current chunk) is what two waves per SIMD need and what costs 60 VGPRs; here the kernel is compiled for FOUR waves
//     per SIMD (<= 128 VGPRs, <= 80 KB LDS: half a CU).  What runs on the other half: a second slot (k_slot_batch: batches of
//     more streams than CUs) - or the scoring kernel's waves, arithmetic beside a search that waits for memory, which is how
//     the pipeline's slots are dealt: one per CU (jd_host_resident.h: jd_res_start).
//   * phase X walks only the arcs of a state that the arriving token can enter: the decoder keeps a state's model arcs in
//     descending order of the bound the pruning test runs on (XState, jd_search.h).
#pragma once

#define XLW 4                        // phase X: 8-byte words of a row's instance flags requested at once (XState)
#ifndef SLOT_WPE
#define SLOT_WPE 4                   // waves per SIMD the kernel is compiled for (2 workgroups of SW = 8 waves per CU)
#endif
#define SLOT_WG_PER_CU (SLOT_WPE * 4 / SW)
#define JDE_GEOM -47                 // a stream in the middle of an utterance whose lists were written with another geometry
#define SLOT_TRP_MAX 2048            // floats of transition tables cached in LDS (else read from HBM)
#ifndef SLOT_REC_AHEAD
#define SLOT_REC_AHEAD 0             // development: the next chunk's record requested one pass ahead - measured: phase A 52.6 -> 62.6 us per frame
#endif
#define SLOT_LL_MAX 3072             // tied states whose likelihoods of the frame are staged in LDS (else gathered from HBM)

struct SlotShared {
    // fill counts of the stream's wave-segmented lists (what StreamDev::tot holds between commands)
    int c_rec[2][SW], c_new[SW], c_dirty[2][SW], c_exit[SW], c_cl[2][SW], c_cls[2][SW];
    int pfx[3][SW + 1];                        // phase A: entries before every segment of its three lists (records, new arcs, keys to zero)
    int hist[2][HIST_MAX_BINS];                // Histogram bins by frame parity (this frame's, the previous frame's)
    float trP[SLOT_TRP_MAX]; int se[SLOT_TRP_MAX / 4];
    float tee[TEE_LDS_MAX];                    // per HMM: tee transition log-probability ...
    float tmax[TEE_LDS_MAX];                   // ... and the largest log transition probability out of the entry state (phase X's filter)
    // The frame's likelihood row.  Phase A gathers three values per instance from it, each a wave instruction that touches up to 64
    // cache lines of a 12 KB row - and the vector L1 looks up ONE line per cycle: under the counters (profiles/r05_slot_pmc) the L1 of
    // a CU was busy 88 % of the launch and the gathers a third of phase A's lines.  So the row is brought into LDS once per frame -
    // by LDS-DMA (global_load_lds: no registers), issued behind phase A for the NEXT frame, landing while the expansion runs.
    float ll[SLOT_LL_MAX];
    int wpfx[SW][64];                          // phase X: per wave, prefix of the out-degrees of its 64 items
    v4i qtok[SW][QCAP], qinfo[SW][QCAP];       // phase X: per wave, closure items it will expand itself
    int2 qrow[SW][QCAP];
    int nextA, nextX[2];                       // chunk hand-out (phase A; the rounds of phase X alternately)
    unsigned bestA[2], bestX[2];               // ordered-uint best scores of the frame, by frame parity
    float emitTh;                              // this frame's emitting threshold (Histogram::calcThresh by one wave)
    int err;                                   // first error raised (sticky)
    int new_all;                               // arcs entered without an instance in the frame under way
    int n_paths, n_paths_ref;                  // Path records in use (the reservation cursor) / the reference's count of its objects
    unsigned long long final_key;
    int stat[ST_N]; long long acc[ST_N];
    unsigned long long cntA, cntX;             // per frame: new-arc entries taken up | entry items << 32; arcs loaded | closure items << 32 (ST_NEWL ..)
    long long clk[8];
};

__device__ __forceinline__ void slot_err(SlotShared &sh, int code) { atomicCAS(&sh.err, 0, code); }
// The thread's number, opaque to the optimiser.  Under the mailbox loop of k_slot everything a command computes once from the kernel's
// arguments and the thread's number - loop bounds, list offsets, table addresses - is invariant ACROSS commands, and the compiler hoists
// it out of that loop and keeps it in registers over everything (53 VGPRs spilled where the plain launch, the same code without a loop
// around it, spills 7).  What depends on this value is computed where it is used.
#ifndef SLOT_PHASE_TID
#define SLOT_PHASE_TID ((int)threadIdx.x)
#endif
__device__ __forceinline__ int slot_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ int slot_grab(int *next)
{
    int k = 0;
    if ((threadIdx.x & 63) == 0) k = atomicAdd(next, 1);
    return RFL(k);
}
// lanes 0 .. SW-1 of the calling wave hold a value each: exclusive prefix over them (lane SW: the total)
__device__ __forceinline__ int slot_prefix8(int v, int &total)
{
    const int lane = threadIdx.x & 63;
    int x = lane < SW ? v : 0;
#pragma unroll
    for (int o = 1; o < SW; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    total = __shfl(x, SW - 1);
    const int ex = __shfl_up(x, 1);
    return lane == 0 ? 0 : ex;                                         // (lane SW holds the total)
}

// ------------------------------------------------------------------ phase A (see jd_search.h: phase_a)
template <int NE, bool TRPL, bool LR, bool LLL>
__device__ __forceinline__ void slot_phase_a(const DecConst &C, SlotShared &sh, const StreamView &V, const Geo &g, int Q0, int Q1, int Q2,
                                             int n0, int n1, int n2, int p, float normalise, float emitTh, float startTh,
                                             const float *llrow, int &out_cnt, int &exit_cnt)
{
    constexpr bool XL_ = true;
    typedef RecLayout<NE> RL;
    constexpr int HF = RL::HF;
    const int tid_ = SLOT_PHASE_TID;
    const int lane = tid_ & 63, wid = RFL(tid_ >> 6);
    const int MN = C.max_n;
    const bool use_hist = C.max_hyps > 0;
    const float *trP_all = TRPL ? sh.trP : C.trP;
    const int *se_all = TRPL ? sh.se : C.se32;
    const unsigned rcur = p ? V.rec_par : 0u, rnext = p ? 0u : V.rec_par;
    const unsigned iprev = p ? 0u : V.item_par, icur = p ? V.item_par : 0u;
    const unsigned item_base = (unsigned)wid * g.seg_item;
    const int Q01 = Q0 + Q1, Qall = Q01 + Q2;
    int *const hist = sh.hist[p];
    int c_insts = 0, c_pemit = 0, c_emit = 0, c_end = 0, c_surv = 0;
    unsigned mo = 0u;
    // entry `lane` of chunk ru of packed list k: its writer segment and its index there (eight segments: seven compares)
    auto locate = [&](int k, int ru, int total, bool &valid, int &w, int &idx) __attribute__((always_inline)) {
        const int gi = (ru << 6) + lane;
        valid = gi < total;
        w = 0;
#pragma unroll
        for (int j = 1; j < SW; ++j) w += (gi >= sh.pfx[k][j]) ? 1 : 0;
        idx = gi - sh.pfx[k][w];
    };
    // the record of a chunk of list 0.  (SLOT_REC_AHEAD: requested ONE PASS AHEAD, so that the first of a pass's dependent round trips
    // runs beside the pass before it - what jd_search.h's phase A does at two waves per SIMD; at four it LOSES, 52.6 -> 62.6 us per
    // frame: the pass is bound by the L1's line rate, not by the round trip, and twenty more live registers cost more than they hide)
    auto load_rec = [&](int uu, bool &valid, v4i &h0, v4i &h1, v4i &h2, Tok (&tk)[NE + 1]) __attribute__((always_inline)) {
        int w, idx;
        locate(0, uu, n0, valid, w, idx);
        const unsigned off = valid ? rcur + rec_chunk_off<NE>(g.seg_rec, w, idx >> 6) + (unsigned)(idx & 63) * 16u : OOB_OFF;
        h0 = ld16(V.rec, off); h1 = ld16(V.rec, off + 1024u);
        if (NE == 6) h2 = ld16(V.rec, off + 2048u);
#pragma unroll
        for (int j = 1; j <= NE; ++j) tk[j] = as_tok(ld16(V.rec, off + (unsigned)(HF + j - 1) * 1024u));
    };
    int u = slot_grab(&sh.nextA);
    // (80-byte records only: the nine fields of the 144-byte layout ahead cost 36 registers and 300 spilled ones)
    constexpr bool AHEAD = SLOT_REC_AHEAD != 0 && NE == 3;
    bool pvalid = false;
    v4i ph0 = {0, 0, 0, 0}, ph1 = {0, 0, 0, 0}, ph2 = {0, 0, 0, 0};
    Tok ptk[NE + 1];
    if (AHEAD && u < Q0) load_rec(u, pvalid, ph0, ph1, ph2, ptk);
#pragma nounroll
    while (u < Q01) {
        const int un = slot_grab(&sh.nextA);                           // (its LDS round trip runs beside this pass)
        const bool is_new = u >= Q0;                                   // wave-uniform
        bool valid;
        int w, idx;
        v4i h0, h1, h2 = {0, 0, 0, 0};
        Tok tk[NE + 1];
        if (!is_new) {
            if constexpr (AHEAD) {
                valid = pvalid; h0 = ph0; h1 = ph1; h2 = ph2;
#pragma unroll
                for (int j = 1; j <= NE; ++j) tk[j] = ptk[j];
                if (un < Q0) load_rec(un, pvalid, ph0, ph1, ph2, ptk);
            } else load_rec(u, valid, h0, h1, h2, tk);
        } else {                                                       // attachNetInst :751-774, from the arc's template
            locate(1, u - Q0, n1, valid, w, idx);
            const unsigned long long e = CL(V.newl + (valid ? (size_t)w * g.seg_new + (unsigned)idx : (size_t)0));
            const int2 nb = valid ? make_int2((int)(unsigned)e, (int)(unsigned)(e >> 32)) : make_int2(0, 0);   // {arc, source state}
            const JdArc Bk = C.arcs[nb.x];
            const int hm = max((Bk.in & ~ARC_FLAGS) - 1, 0);            // (arcs on the new list carry a model; idle lanes read arc 0)
            const int4 a0 = ((const int4 *)C.aux_h)[(NE == 3) ? hm : 2 * hm];
            h0 = (v4i){nb.x, valid ? (a0.x | (Bk.out != 0 ? REC_LABELLED : 0) | ((Bk.in & SOLE_FLAG) ? REC_SOLE : 0)) : 0, nb.y, Bk.to};
            h1 = (v4i){a0.y, a0.z, a0.w, __float_as_int(Bk.w)};
            if (NE == 6) { const int4 a1 = ((const int4 *)C.aux_h)[2 * hm + 1]; h2 = (v4i){a1.x, a1.y, a1.z, 0}; }
#pragma unroll
            for (int j = 1; j <= NE; ++j) tk[j] = null_tok();
        }
        const int arc = h0.x;
        const int n = h0.y & 0xff;                                     // 0 for lanes without an instance
        const int tm = (h0.y >> 8) & 0x1fffff;
        // the best arrival at the source state (StateRec::e[p ^ 1]) and the likelihoods: in flight together
        unsigned long long kv;
        float outp[NE];
        {
            const unsigned long long ev = ld8(V.srec_r, valid ? SREC_E_OFF(C, h0.z, p ^ 1) : OOB_OFF);
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int gj = (j == 0) ? h1.x : (j == 1) ? h1.y : (j == 2) ? h1.z : (j == 3) ? h2.x : (j == 4) ? h2.y : h2.z;
                outp[j] = LLL ? sh.ll[(j + 1 < n - 1) ? gj : 0] : llrow[(j + 1 < n - 1) ? gj : 0];   // :411
            }
            kv = ev;
        }
        // entry token = the best token that arrived at the arc's source state in the previous frame, over the arc (:560-582)
        const v4i itv = ld16(V.items, kv != 0ULL ? iprev + (unsigned)(kv & 0xffffffffULL) * 32u : OOB_OFF);
        // The item is a THIRD round trip behind the record and the key, and all that it brings is the entry token's history (ac, lm,
        // path): its score is in the key.  Everything that decides - maxima, thresholds, the histogram - runs on scores; with plain
        // left-to-right models only state 1 can take the entry token, so the item is taken up BEHIND the arithmetic (below), its
        // round trip running beside it.  (General topologies: any state may take it - they wait for it here.)
        tk[0] = null_tok();
        if (kv != 0ULL) {
            tk[0].score = o2f((unsigned)(kv >> 32)) + __int_as_float(h1.w);   // :562 newScore = tok.score + weight
            if (!LR) {
                const Tok it = as_tok(itv);
                tk[0].ac = it.ac; tk[0].lm = it.lm + __int_as_float(h1.w); tk[0].path = it.path;
            }
            if (tk[0].score < startTh) tk[0] = null_tok();            // :915-918 (a candidate is never LOG_ZERO)
        }
        Tok nw[NE + 1];
        int live_mask = 0;
        Tok ex = null_tok();
        auto emit = [&](int j, float best, float btp, const Tok &src) __attribute__((always_inline)) {   // :408-424
            const float sc = best - normalise;                         // :408
            if (sc > emitTh) {                                         // :409
                ++c_pemit;
                nw[j].score = sc + outp[j - 1];
                nw[j].ac = (src.ac + btp) + outp[j - 1];
                nw[j].lm = src.lm;
                nw[j].path = src.path;
                live_mask |= 1 << j;
                if (use_hist) {                                        // Histogram::addScore, Histogram.cpp:64-100
                    const double ds = (double)nw[j].score;
                    const int sci = (nw[j].score < 0.0f) ? (int)(ds - 0.5) : (int)(ds + 0.5);
                    if (sci > C.hist_max) slot_err(sh, (int)JD_EHIST);
                    else if (sci >= C.hist_min) atomicAdd(&hist[sci - C.hist_min], 1);
                }
                const unsigned so = f2o(nw[j].score);
                mo = so > mo ? so : mo;
            }
        };
        if (LR) {
            constexpr int LRW = (NE == 3) ? 8 : 16;                    // a_1 .. a_{NE+1}, s_1 .. s_NE
            const float4 *lt = (const float4 *)(sh.trP + tm * LRW);
            float tw[LRW];
#pragma unroll
            for (int q = 0; q < LRW / 4; ++q) {
                const float4 v = lt[q];
                tw[4 * q] = v.x; tw[4 * q + 1] = v.y; tw[4 * q + 2] = v.z; tw[4 * q + 3] = v.w;
            }
            bool entry_won = false;
#pragma unroll
            for (int j = 1; j <= NE; ++j) {                            // :387-424 emitting state j: predecessors j-1 and j
                nw[j] = null_tok();
                const float a = tw[j - 1], sf = tw[NE + j];
                const float c0 = tk[j - 1].score + a, c1 = tk[j].score + sf;
                const bool self = c1 > c0;                             // the lower predecessor wins ties (:401)
                Tok src;
                src.score = 0.0f; src.ac = self ? tk[j].ac : tk[j - 1].ac; src.lm = self ? tk[j].lm : tk[j - 1].lm;
                src.path = self ? tk[j].path : tk[j - 1].path;
                if (j < n - 1) emit(j, self ? c1 : c0, self ? sf : a, src);
                if (j == 1) entry_won = !self;
            }
            if ((live_mask & 2) && entry_won) {                        // state 1 took the entry token: its history, from the item (:562-566)
                const Tok it = as_tok(itv);
                nw[1].ac = (it.ac + tw[0]) + outp[0];
                nw[1].lm = it.lm + __int_as_float(h1.w);
                nw[1].path = it.path;
            }
            // exit state (:443-483): entered from the last emitting state only
            Tok le = null_tok();
            float ax = 0.0f;
#pragma unroll
            for (int i = 1; i <= NE; ++i) if (i == n - 2) { le = nw[i]; ax = tw[i]; }
            if (le.score > LZ) { ex = le; ex.score = le.score + ax; ex.ac = le.ac + ax; if (!(ex.score > LZ)) ex = null_tok(); }
        } else {
            // general topologies, branch-free: every (predecessor, state) pair is evaluated and selected
            const float *trP = trP_all + (size_t)tm * MN * MN;
            const int *se = se_all + (size_t)tm * MN;
#pragma unroll
            for (int j = 1; j <= NE; ++j) {                            // :387-424 emitting state j
                nw[j] = null_tok();
                const int sev = se[j < MN ? j : 0];
                const int st = sev & 0xffff, en = sev >> 16;
                float best = 0.0f, btp = 0.0f;
                Tok src = null_tok();
                bool have = false;
#pragma unroll
                for (int i = 0; i <= NE; ++i) {                        // predecessors in ascending order, the first wins ties
                    const bool v = (i == st) | ((i > st) & (i < en));
                    const float tp = trP[(i < MN ? i : 0) * MN + (j < MN ? j : 0)];
                    const float tmp = tk[i].score + tp;
                    const bool take = v & (!have | (tmp > best));
                    best = take ? tmp : best; btp = take ? tp : btp;
                    src.ac = take ? tk[i].ac : src.ac; src.lm = take ? tk[i].lm : src.lm; src.path = take ? tk[i].path : src.path;
                    have |= v;
                }
                if (have & (j < n - 1)) emit(j, best, btp, src);
            }
            // exit state (:443-483) from the NEW tokens
            {
                const int sev = se[n >= 2 ? n - 1 : 0];
                const int st = sev & 0xffff, en = sev >> 16;
                bool have = false;
#pragma unroll
                for (int i = 1; i <= NE; ++i) {
                    const bool v = (i == st) | ((i > st) & (i < en));
                    const float tp = trP[(i < MN ? i : 0) * MN + (n >= 2 ? n - 1 : 0)];
                    const float tmp = nw[i].score + tp;
                    const bool take = v & (!have | (tmp > ex.score));
                    ex.score = take ? tmp : ex.score; ex.ac = take ? nw[i].ac + tp : ex.ac;
                    ex.lm = take ? nw[i].lm : ex.lm; ex.path = take ? nw[i].path : ex.path;
                    have |= v;
                }
                if (!(have & (n >= 2)) || !(ex.score > LZ)) ex = null_tok();
            }
        }
        c_emit += __popc(live_mask);
        const bool has_exit = ex.score > LZ;
        const bool slot_live = live_mask != 0;
        const unsigned long long bl = __ballot(slot_live), be = __ballot(has_exit);
        if (!is_new) c_insts += __popcll(__ballot(valid));             // (new arcs are counted when they are entered)
        JD_COUNT(const unsigned long long cnt_a_ = (unsigned long long)(is_new ? __popcll(__ballot(valid)) : 0) | ((unsigned long long)__popcll(__ballot(valid && kv != 0ULL)) << 32);
                 if (lane == 0) atomicAdd(&sh.cntA, cnt_a_));         // (the ballots are the whole wave's: taken outside the one-lane branch)
        // survivors: header + new tokens to this wave's segment of the next list
        {
            const int nsurv = __popcll(bl);
            if (out_cnt + nsurv > (int)g.seg_rec) { if (lane == 0) slot_err(sh, (int)JDE_SLOTS); }
            else {
                const int pos = out_cnt + rank_in(bl);
                const unsigned doff = slot_live ? rnext + rec_chunk_off<NE>(g.seg_rec, wid, pos >> 6) + (unsigned)(pos & 63) * 16u : OOB_OFF;
                st16(V.rec, doff, h0); st16(V.rec, doff + 1024u, h1);
                if (NE == 6) st16(V.rec, doff + 2048u, h2);
#pragma unroll
                for (int j = 1; j <= NE; ++j) st16(V.rec, doff + (unsigned)(HF + j - 1) * 1024u, as_v4(nw[j]));
                out_cnt += nsurv;
                c_surv += nsurv;
            }
            // the arc's "has an instance" flag changes at birth and death only (returnNetInst :777-797)
            if (valid && is_new && slot_live) CS(&V.live[arc], (unsigned char)1);
            if (valid && !slot_live && !is_new) CS(&V.live[arc], (unsigned char)0);
        }
        // exit tokens: frontier items of round 0 in this wave's item segment, bidding for their destination state
        {
            const int nex = __popcll(be);
            if (exit_cnt + nex > (int)g.seg_item) { if (lane == 0) slot_err(sh, (int)JDE_ITEMS); }
            else {
                const unsigned k = item_base + (unsigned)(exit_cnt + rank_in(be));
                const unsigned ioff = has_exit ? icur + k * 32u : OOB_OFF;
                st16(V.items, ioff, as_v4(ex));
                const int lab = (h0.y & REC_LABELLED) ? 1 : 0;
                const int sole = (h0.y & REC_SOLE) ? ITEM_SOLE : 0;      // (jd_search.h: REC_SOLE - nobody to recombine with, no bid)
                st16(V.items, ioff + 16u, (v4i){arc, lab, h0.w, sole});
                if (has_exit && !sole) GMAX((lab ? &SREC_BID(V.srec, C, h0.w).keyL : &SREC_BID(V.srec, C, h0.w).key0), ((unsigned long long)f2o(ex.score) << 32) | k);
                JD_COUNT(const int nbid_ = __popcll(__ballot(has_exit && !sole)); if (lane == 0 && nbid_) atomicAdd(&sh.stat[ST_BIDS], nbid_));
                exit_cnt += nex;
                c_end += nex;
            }
        }
        u = un;
    }
    // key clean-up: the arrival keys e[p] of the frame before the previous one (see jd_search.h)
#pragma nounroll
    for (; u < Qall; u = slot_grab(&sh.nextA)) {
        bool on;
        int w, idx;
        locate(2, u - Q01, n2, on, w, idx);
        if (on) {
            const int b = CL(V.dirtyl + (p ? V.dirty_par : 0u) + (size_t)w * g.seg_new + (unsigned)idx);
            CS(&SREC_E(V.srec, C, b, p), 0ULL);
        }
    }
    mo = wave_umax(mo);
    c_pemit = wave_sum(c_pemit); c_emit = wave_sum(c_emit);
    if (lane == 0) {
        if (mo) atomicMax(&sh.bestA[p], mo);
        if (c_insts) { atomicAdd(&sh.stat[ST_INSTS], c_insts); atomicAdd(&sh.stat[ST_RECS], c_insts); }
        if (c_pemit) atomicAdd(&sh.stat[ST_PEMIT], c_pemit);
        if (c_emit) atomicAdd(&sh.stat[ST_EMIT], c_emit);
        if (c_end) atomicAdd(&sh.stat[ST_END], c_end);
        if (c_surv) { atomicAdd(&sh.stat[ST_MODELS], c_surv); atomicAdd(&sh.stat[ST_SURV], c_surv); }
    }
}

// ------------------------------------------------------------------ phase X (see jd_search.h: phase_x)
// xp / xc / xs: lanes 0 .. SW-1 of the calling wave hold, per writer segment, the chunks before it, its items and where
// they start (built by the caller from the counts in LDS: every wave has its own copy - no barrier, no shared table)
__device__ __forceinline__ void slot_phase_x(const DecConst &C, SlotShared &sh, const StreamView &V, const Geo &g, int Q, int KX, int round,
                                             int xp, int xc, int xs, int *next, int p, int pframe, bool init, bool last_frame,
                                             float endTh, float wordTh, float bestA, XOut &out, int &deferred)
{
    constexpr bool XL_ = true;
    const int tid_ = SLOT_PHASE_TID;
    const int lane = tid_ & 63;
    const int wid = RFL(tid_ >> 6);
    const float INF = __builtin_inff();
    const unsigned icur = p ? V.item_par : 0u;
    const bool can_filter = !init && C.emit_win > 0.0f && bestA > LZ;
    const bool tee_lds = C.n_hmm <= TEE_LDS_MAX;
    const unsigned item_base = (unsigned)wid * g.seg_item, new_base = (unsigned)wid * g.seg_new;
    GAS int *const dirty_seg = V.dirtyl + (p ? V.dirty_par : 0u) + (size_t)new_base;
    int *wpfx = sh.wpfx[wid];
    v4i *qtok = sh.qtok[wid], *qinfo = sh.qinfo[wid];
    int2 *qrow = sh.qrow[wid];
    int q_n = 0;
    int c_arcs = 0, c_paths = 0, c_pend = 0, c_new = 0, c_ref = 0;
    unsigned mo = 0u;
    auto list_dirty = [&](bool first, int state) __attribute__((always_inline)) {
        const unsigned long long bf = __ballot(first);
        if (bf) {
            const int nf = __popcll(bf);
            if (out.dirty_cnt + nf > (int)g.seg_new) { if (lane == 0) slot_err(sh, (int)JDE_NEW); }
            else {
                if (first) CS(dirty_seg + (unsigned)(out.dirty_cnt + rank_in(bf)), state);
                out.dirty_cnt += nf;
            }
        }
    };
#pragma nounroll
    for (;;) {
        bool valid, exit_kind;
        unsigned ii;
        Tok t;
        v4i info;
        int slice_no = 0;
        const bool from_q = q_n > 0;                                   // (wave-uniform)
        int2 row_q = make_int2(0, 0);
        if (from_q) {
            valid = lane < q_n;
            exit_kind = false;
            t = as_tok(qtok[lane & (QCAP - 1)]);
            info = qinfo[lane & (QCAP - 1)];
            row_q = qrow[lane & (QCAP - 1)];
            ii = (unsigned)info.w;
            info.w = 0;
            q_n = 0;
        } else {
            const int u = slot_grab(next);
            if (u >= Q) break;
            // the chunk's writer segment: the last one whose chunk prefix is <= u (empty segments share their successor's prefix)
            const unsigned long long mw = __ballot(lane < SW && xp <= u);
            const int w = RFL(63 - __clzll((long long)mw));
            const int ci = u - __builtin_amdgcn_readlane(xp, w);
            valid = lane < KX && ci * KX + lane < __builtin_amdgcn_readlane(xc, w);
            ii = (unsigned)w * g.seg_item + (unsigned)(__builtin_amdgcn_readlane(xs, w) + ci * KX + lane);
            const unsigned ioff = valid ? icur + ii * 32u : OOB_OFF;
            t = as_tok(ld16(V.items, ioff));
            info = ld16(V.items, ioff + 16u);
            exit_kind = round == 0;
            if (!exit_kind && (info.w & 3) == 1) valid = false;        // expanded by its producer / superseded
            if (!exit_kind && (info.w & 3) == 2) slice_no = info.w >> 2;
        }
        const unsigned ioff = valid ? icur + ii * 32u : OOB_OFF;
        const bool start_tok = valid && exit_kind && info.x < 0;       // recognitionStart's token: it has traversed no arc
        JD_COUNT(const int cnt_i_ = __popcll(__ballot(valid)); if (lane == 0) atomicAdd(&sh.stat[ST_XITEMS], cnt_i_));
        const bool real = valid && !start_tok && slice_no == 0;
        const int state = !valid ? 0 : start_tok ? C.init_state : info.z;
        // the state's static record (XState): requested here, used when the item is known to go on
        const int4 *xq = (const int4 *)(C.xst + state);
        const int4 x0 = xq[0], x1 = xq[1], x2 = xq[2], x3 = xq[3];
        bool have = valid;
        if (real && exit_kind && !init) {                              // :952-962
            have = t.score > ((info.y != 0) ? wordTh : endTh);
            if (have) ++c_pend;
        }
        if (C.pcount != nullptr && ((real && exit_kind && have) || start_tok))
            c_ref += (start_tok ? 0 : (info.y != 0 ? 1 : 0)) + C.pcount[state];
        // Path records (:497-509) are reserved for every labelled item that passed its threshold, winner or not: the cursor is in LDS
        const bool labelled = real && have && info.y != 0;
        const unsigned long long blab = __ballot(labelled);
        int pbase = 0;
        if (blab) {
            const int first = __ffsll((long long)blab) - 1;
            if (lane == first) pbase = atomicAdd(&sh.n_paths, __popcll(blab));
            pbase = __shfl(pbase, first);
        }
        int rs, rs1;
        unsigned long long kv = 0ULL;
        int label = exit_kind ? 0 : info.y;
        const bool sole = exit_kind && (info.w & ITEM_SOLE) != 0;
        if (from_q) { rs = row_q.x; rs1 = row_q.x + row_q.y; }
        else {
            const unsigned soff = valid ? SREC_BID_OFF(C, state) : OOB_OFF;
            const v4i sk = ld16(V.srec_r, (real && exit_kind && !sole) ? soff : OOB_OFF);   // {key0, keyL}
            const int sti = valid ? state : 0;
            const int2 srow = make_int2(C.row_ptr[sti], C.row_ptr[sti + 1]);
            const bool lab_on = exit_kind && real && info.y != 0;
            const int lb = C.arcs[lab_on ? info.x : 0].out;
            label = lab_on ? lb : label;
            rs = srow.x; rs1 = srow.y;
            kv = ((unsigned long long)(unsigned)(info.y != 0 ? sk.w : sk.y) << 32) | (unsigned)(info.y != 0 ? sk.z : sk.x);
        }
        if (real) {
            const bool winner = !exit_kind || sole || ((unsigned)(kv & 0xffffffffULL) == ii && kv != 0ULL);
            if (winner && exit_kind && !sole) CS(info.y != 0 ? &SREC_BID(V.srec, C, state).keyL : &SREC_BID(V.srec, C, state).key0, 0ULL);
            have = have && winner;
        }
        if (have && real) {
            if (info.y != 0) {
                const int pp = pbase + rank_in(blab);
                if (pp < C.cap_paths) {
                    V.paths[2 * (size_t)pp] = (v4i){t.path, pframe, label, 0};
                    V.paths[2 * (size_t)pp + 1] = (v4i){__float_as_int(t.score), __float_as_int(t.ac), __float_as_int(t.lm), 0};
                    t.path = pp;
                    st16(V.items, ioff, as_v4(t));                     // the tokens pulled from this item carry the new history
                    ++c_paths;
                } else slot_err(sh, (int)JDE_PATHS);
            }
            if (last_frame) {                                          // :513-520 final state
                const float fw = C.fin_w[info.z];
                if (fw < INF) {
                    const float cs = t.score + fw;
                    if (cs > LZ) atomicMax(&sh.final_key, ((unsigned long long)f2o(cs) << 32) | ii);
                }
            }
        }
        unsigned eo = exit_kind ? 0u : (unsigned)info.x;               // ordered score of the best arrival before this one (0: none)
        unsigned long long eold = 0ULL;
        const bool arrive = have && exit_kind;
        // The arcs of the state that enter a model stand in descending order of w + tmax behind the ones every arrival walks (XState):
        // this item can only enter a PREFIX of them - the rest fails the "hopeless candidate" test below whatever its flag says - and
        // the prefix's upper bound comes from the samples in the state's record.  (Conservative by a margin far above the rounding of
        // the sums: the test itself still decides inside the prefix.)  What the walk did for the arcs left out: they count as visited,
        // the best entry-token candidate of the WHOLE row is score + wmax (float addition is monotone), and the arcs entered without
        // an instance are the row's model arcs less the instance flags set in it - one byte per arc, the row's side by side -
        // counted behind the arrival below.
        int x_new = 0;
        // (rows of up to 8 * 2 * XLW - 7 arcs: their flags are one or two batches of loads; a longer row is walked whole and counted arc by arc)
        const bool xitem = have && slice_no == 0 && rs1 - (rs & ~7) <= 16 * XLW;
        if (xitem) {
            const int n_entry = x0.y, n_model = x0.w;
            if (n_model > 0) {
                const unsigned sw = f2o(t.score + __int_as_float(x0.z));
                mo = sw > mo ? sw : mo;
            }
            const int a8 = rs & ~7;
            const GAS unsigned long long *lw = (const GAS unsigned long long *)(V.live + a8);
            auto in_row = [&](int base) __attribute__((always_inline)) {   // the bytes of the word at `base` that belong to the row
                const int lo = max(rs - base, 0), hi = min(rs1 - base, 8);
                const unsigned long long mh = hi >= 8 ? ~0ULL : ((1ULL << (8 * max(hi, 0))) - 1ULL);
                const unsigned long long ml = (1ULL << (8 * lo)) - 1ULL;
                return 0x0101010101010101ULL & mh & ~ml;
            };
            int lv_row = 0;
            {
                unsigned long long w8[XLW];
#pragma unroll
                for (int i = 0; i < XLW; ++i) w8[i] = (n_model > 0 && a8 + 8 * i < rs1) ? CL(lw + i) : 0ULL;
#pragma unroll
                for (int i = 0; i < XLW; ++i) lv_row += __popcll(w8[i] & in_row(a8 + 8 * i));
            }
            if (__ballot(n_model > 0 && a8 + 8 * XLW < rs1)) {          // (some lane's row goes on: the second batch)
                unsigned long long w8[XLW];
#pragma unroll
                for (int i = 0; i < XLW; ++i) w8[i] = (n_model > 0 && a8 + 8 * (XLW + i) < rs1) ? CL(lw + XLW + i) : 0ULL;
#pragma unroll
                for (int i = 0; i < XLW; ++i) lv_row += __popcll(w8[i] & in_row(a8 + 8 * (XLW + i)));
            }
            x_new = n_model - lv_row;
            if (can_filter && n_entry > 0) {
                const float lim = (bestA - C.emit_win) - (1.0f + 1e-5f * (fabsf(bestA) + fabsf(t.score)));
                const int kx[XNCAND] = {x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
                int P = n_entry;
#pragma unroll
                for (int i = XNCAND - 1; i >= 0; --i)
                    if (xcand(i) < n_entry && t.score + __int_as_float(kx[i]) <= lim) P = xcand(i);
                c_arcs += n_entry - P;
                rs1 -= n_entry - P;
            }
        }
        int alo = rs, ahi = rs1;
        if (slice_no > 0) { alo = rs + slice_no * X_SLICE; ahi = min(rs1, alo + X_SLICE); }
        int n_slices = 0;
        if (have && slice_no == 0 && rs1 - rs > X_SLICE) { n_slices = (rs1 - rs - 1) / X_SLICE; ahi = rs + X_SLICE; }
        const int deg = have ? ahi - alo : 0;
        int incl = deg;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const int tot = __shfl(incl, 63);
        wpfx[lane] = incl - deg;
        auto owner_of = [&](int a) __attribute__((always_inline)) { int gg = 0;
#pragma unroll
            for (int stp = 32; stp > 0; stp >>= 1) if (wpfx[gg + stp] <= a) gg += stp;
            return gg; };
        int g_nx = owner_of(lane);
        int b_nx = __shfl(alo, g_nx) + (lane - wpfx[g_nx]);
        JdArc Bk_nx = {0, 0.0f, 0, 0};
        int lv_nx = 0;
        { const int bq = lane < tot ? b_nx : 0; Bk_nx = C.arcs[bq]; lv_nx = CL(V.live + bq); }
if (arrive) { eold = GMAX(&SREC_E(V.srec, C, state, p), ((unsigned long long)f2o(t.score) << 32) | ii); eo = (unsigned)(eold >> 32); }
        list_dirty(arrive && eold == 0ULL, state);
        if (eo == 0u) c_new += x_new;                                  // (the first arrival at the state in this frame: :899-935 tries them all)
        if (__ballot(n_slices > 0)) {
            for (unsigned long long bs = __ballot(n_slices > 0); bs; bs &= bs - 1) {
                const int src = __ffsll((long long)bs) - 1;
                const int ns = __shfl(n_slices, src);
                const v4i tv = {__shfl(__float_as_int(t.score), src), __shfl(__float_as_int(t.ac), src),
                                __shfl(__float_as_int(t.lm), src), __shfl(t.path, src)};
                const int sx = __shfl((int)eo, src), sy = __shfl(label, src), sz = __shfl(state, src);
                for (int j0 = 0; j0 < ns; j0 += 64) {
                    const int nj = min(64, ns - j0);
                    if (out.item_cnt + nj > (int)g.seg_item) { if (lane == 0) slot_err(sh, (int)JDE_ITEMS); break; }
                    if (lane < nj) {
                        const unsigned k = item_base + (unsigned)(out.item_cnt + lane);
                        st16(V.items, icur + k * 32u, tv);
                        st16(V.items, icur + k * 32u + 16u, (v4i){sx, sy, sz, 2 | ((j0 + lane + 1) << 2)});
                    }
                    out.item_cnt += nj; deferred += nj;
                }
            }
        }
#pragma nounroll
        for (int a0 = 0; a0 < tot; a0 += 64) {
            const int a = a0 + lane;
            const int gg = g_nx, b = b_nx;
            const JdArc Bk = Bk_nx;
            const int lv = lv_nx;
            g_nx = owner_of(a + 64);
            const int alo_nx = __shfl(alo, g_nx);
            b_nx = (a + 64 < tot) ? alo_nx + (a + 64 - wpfx[g_nx]) : 0;
            Tok tg;
            tg.score = __shfl(t.score, gg); tg.ac = __shfl(t.ac, gg);
            tg.lm = __shfl(t.lm, gg); tg.path = __shfl(t.path, gg);
            const unsigned eog = (unsigned)__shfl((int)eo, gg);
            const int sgx = __shfl(state | (xitem ? (int)0x80000000 : 0), gg);   // (+ the owner's "counted per state" flag)
            const int sg = sgx & 0x7fffffff;
            bool mk = false, touch = false;
            Tok un = null_tok();
            const bool on = a < tot;
            const int inl = Bk.in & ~ARC_FLAGS;
            const bool entry = on && inl != 0;
            const bool is_tee = entry && (Bk.in & TEE_FLAG) != 0;
            const float ns = tg.score + Bk.w;                          // (:535 / :562: the same sum either way)
            const unsigned so = f2o(ns);
            unsigned long long skc = 0ULL;
            const float tmax = tee_lds ? sh.tmax[entry ? inl - 1 : 0] : C.hmm_tmax0[entry ? inl - 1 : 0];
            int2 nrow = make_int2(0, 0);
            {
                const unsigned doff = ((on && inl == 0) || is_tee) ? SREC_E_OFF(C, Bk.to, p) : OOB_OFF;
                const unsigned long long se = ld8(V.srec_r, doff);
                { const int ti = doff != OOB_OFF ? Bk.to : 0; const int r0 = C.row_ptr[ti]; nrow = make_int2(r0, C.row_ptr[ti + 1] - r0); }
                Bk_nx = C.arcs[b_nx]; lv_nx = CL(V.live + b_nx);
                skc = se;
            }
            if (on) ++c_arcs;
            JD_COUNT(const unsigned long long cnt_w_ = (unsigned long long)__popcll(__ballot(on)); if (lane == 0) atomicAdd(&sh.cntX, cnt_w_));
            if (on && inl == 0) {                                      // :533-540 epsilon input
                un = tg;
                un.score = ns;
                un.lm = tg.lm + Bk.w;
                mk = un.score > endTh;
            } else if (is_tee) {                                       // :584-600 tee model
                const float tee = tee_lds ? sh.tee[inl - 1] : CL(C.hmm_tee + (inl - 1));
                const float ns2 = ns + tee;
                un.score = ns2;
                un.ac = tg.ac + tee;
                un.lm = tg.lm + Bk.w;
                un.path = tg.path;
                mk = ns2 > ((Bk.out != 0) ? wordTh : endTh);
            }
            if (entry) {                                               // :560-582 entry-token recombination: pulled by the next phase A
                mo = so > mo ? so : mo;                                // :572-573
                if (lv == 0) {                                         // no instance: attachNetInst :751-774
                    if (sgx >= 0 && eog == 0u) ++c_new;                // (long rows: counted arc by arc; else per state, above)
                    if (can_filter) {
                        const bool mine = (ns + tmax) - bestA > -C.emit_win;
                        const bool before = eog != 0u && ((o2f(eog) + Bk.w) + tmax) - bestA > -C.emit_win;
                        touch = mine && !before;
                    } else touch = eog == 0u;
                }
            }
            const unsigned long long bt = __ballot(touch);
            if (bt) {
                const int nt = __popcll(bt);
                if (out.new_cnt + nt > (int)g.seg_new) { if (lane == 0) slot_err(sh, (int)JDE_NEW); }
                else {
                    if (touch) CS(V.newl + (size_t)new_base + (unsigned)(out.new_cnt + rank_in(bt)),
                                  ((unsigned long long)(unsigned)sg << 32) | (unsigned)b);
                    out.new_cnt += nt;
                }
            }
            if (__ballot(mk)) {
                const unsigned sou = f2o(un.score);
                const bool pass = mk && sou > (unsigned)(skc >> 32);
                const unsigned long long bp = __ballot(pass);
                const int np = __popcll(bp);
                JD_COUNT(if (lane == 0 && np) atomicAdd(&sh.cntX, (unsigned long long)np << 32));
                if (out.item_cnt + np > (int)g.seg_item) { if (lane == 0) slot_err(sh, (int)JDE_ITEMS); }
                else if (np) {
                    const unsigned k = item_base + (unsigned)(out.item_cnt + rank_in(bp));
                    bool keep = false, first = false;
                    unsigned ceo = 0u;
                    if (pass) {
                        const unsigned long long key = ((unsigned long long)sou << 32) | k;
                        const unsigned long long cold = GMAX(&SREC_E(V.srec, C, Bk.to, p), key);
                        keep = key > cold; first = cold == 0ULL; ceo = (unsigned)(cold >> 32);
                    }
                    const unsigned long long bk = __ballot(keep);
                    const int room = QCAP - q_n;
                    const bool inq = keep && rank_in(bk) < room;
                    if (pass) {
                        st16(V.items, icur + k * 32u, as_v4(un));
                        st16(V.items, icur + k * 32u + 16u, (v4i){(int)ceo, Bk.out, Bk.to, (keep && !inq) ? 0 : 1});
                    }
                    if (inq) {
                        const int qi = q_n + rank_in(bk);
                        qtok[qi] = as_v4(un); qinfo[qi] = (v4i){(int)ceo, Bk.out, Bk.to, (int)k}; qrow[qi] = nrow;
                    }
                    const int nk = __popcll(bk);
                    const int n_inq = nk < room ? nk : room;
                    q_n += n_inq; deferred += nk - n_inq;
                    out.item_cnt += np;
                    list_dirty(first, Bk.to);
                }
            }
        }
    }
    mo = wave_umax(mo);
    c_arcs = wave_sum(c_arcs); c_paths = wave_sum(c_paths); c_pend = wave_sum(c_pend); c_new = wave_sum(c_new);
    if (C.pcount != nullptr) {
        c_ref = wave_sum(c_ref);
        if (lane == 0 && c_ref) atomicAdd(&sh.n_paths_ref, c_ref);
    }
    if (lane == 0) {
        if (mo) atomicMax(&sh.bestX[p], mo);
        if (c_arcs) atomicAdd(&sh.stat[ST_ARCS], c_arcs);
        if (c_paths) atomicAdd(&sh.stat[ST_PATHS], c_paths);
        if (c_pend) atomicAdd(&sh.stat[ST_PEND], c_pend);
        if (c_new) { atomicAdd(&sh.stat[ST_MODELS], c_new); atomicAdd(&sh.new_all, c_new); }
    }
}

// ------------------------------------------------------------------ one stream, one command (see jd_search.h: run_stream)
template <int NE>
__device__ __forceinline__ void slot_run(const SearchArgs &A, SlotShared &sh, int s, int ll_slot)
{
    constexpr bool XL_ = true;
    typedef RecLayout<NE> RL;
    const DecConst &C = A.C;
    StreamCtl &c = A.ctl[s];
    const StreamDev &S = A.streams[s];
    const int tid = slot_tid(), lane = tid & 63, wid = RFL(tid >> 6);
    const int MN = C.max_n;
    int f = RFL(c.frame);
    const int T = RFL(c.T);
    const bool needs_init = RFL(c.needs_init) != 0;
    if (!c.started || c.error != 0) return;
    const int f_stop = T < A.f_end ? T : A.f_end;
    if (!needs_init && f >= f_stop) return;
    float best_emit = __int_as_float(RFL(__float_as_int(c.best_emit)));
    const int old_nw = RFL(c.lst_nw);
    const int dn0 = RFL(c.dirty_nw[0]), dn1 = RFL(c.dirty_nw[1]);
    const bool ref_rule = C.pcount != nullptr;
    const int path_new = needs_init ? 0 : RFL(ref_rule ? c.path_new_ref : c.path_new);
    const Geo g = make_geo(C, SW);
    StreamView V;
    V.rec = mk_rsrc(S.rec, 2ULL * C.cap_slots * RL::REC_BYTES);
    V.items = mk_rsrc(S.items, 2ULL * C.cap_items * 32u);
    V.rec_par = C.cap_slots * (unsigned)RL::REC_BYTES; V.item_par = C.cap_items * 32u;
    V.live = (GAS unsigned char *)S.live; V.srec = (GAS StateRec *)S.srec;
    V.srec_r = mk_rsrc(S.srec, (unsigned long long)C.n_states * sizeof(StateRec));
    V.newl = (GAS unsigned long long *)S.newl; V.dirtyl = (GAS int *)S.dirtyl; V.dirty_par = C.cap_new;
    V.tot = (GAS int *)S.tot; V.paths = (GAS v4i *)S.paths; V.hist = (GAS int *)S.hist;
    const bool use_hist = C.max_hyps > 0;
    const bool lr = C.lrt != nullptr && C.n_tm * ((NE == 3) ? 8 : 16) <= SLOT_TRP_MAX;
    const bool trp_lds = !lr && (size_t)C.n_tm * MN * MN <= SLOT_TRP_MAX && (size_t)C.n_tm * MN <= SLOT_TRP_MAX / 4;
    const bool ll_lds = C.G <= SLOT_LL_MAX;
    // frame fr's likelihood row -> LDS, by this wave's share of LDS-DMA loads (64 floats each; nothing is waited for here)
    auto row_of = [&](int fr) __attribute__((always_inline)) { return A.ll + ((long long)ll_slot * A.ll_stride + (long long)(fr - A.f0) * (long long)C.G); };
    auto load_row = [&](int fr) __attribute__((always_inline)) {
        const float *row = row_of(fr);
        for (int c0 = wid * 64; c0 < C.G; c0 += SW * 64) {
            const int i = min(c0 + lane, C.G - 1);
            __builtin_amdgcn_global_load_lds((const GAS float *)(row + i), (__attribute__((address_space(3))) float *)(sh.ll + c0), 4, 0, 0);
        }
    };
    int row_in_lds = -1;                                               // the frame whose row is in sh.ll (or on its way)
    auto tot_of = [&](int k) __attribute__((always_inline)) { return V.tot + (size_t)k * MAXW; };
    __syncthreads();                                                   // the previous command of this slot is done with LDS
#define SLOT_LOOP _Pragma("clang loop unroll(disable) vectorize(disable)")
    if (lr) SLOT_LOOP for (int i = tid; i < C.n_tm * ((NE == 3) ? 8 : 16); i += SNT) sh.trP[i] = C.lrt[i];
    if (C.n_hmm <= TEE_LDS_MAX) SLOT_LOOP for (int i = tid; i < C.n_hmm; i += SNT) { sh.tee[i] = C.hmm_tee[i]; sh.tmax[i] = C.hmm_tmax0[i]; }
    if (trp_lds) {
        SLOT_LOOP for (int i = tid; i < C.n_tm * MN * MN; i += SNT) sh.trP[i] = C.trP[i];
        SLOT_LOOP for (int i = tid; i < C.n_tm * MN; i += SNT) sh.se[i] = C.se32[i];
    }
    if (tid == 0) {
        sh.err = 0; sh.bestA[0] = sh.bestA[1] = 0u; sh.bestX[0] = sh.bestX[1] = 0u; sh.nextA = 0; sh.nextX[0] = sh.nextX[1] = 0;
        sh.final_key = 0ULL;
        sh.n_paths = needs_init ? 0 : CL(&c.n_paths); sh.n_paths_ref = needs_init ? 0 : CL(&c.n_paths_ref);
        sh.new_all = needs_init ? 0 : CL(&c.new_all[(f & 1) ^ 1]);      // arcs entered in the last frame of the command before
        for (int k = 0; k < ST_N; ++k) { sh.stat[k] = 0; sh.acc[k] = 0; }
        sh.cntA = 0ULL; sh.cntX = 0ULL;
        for (int k = 0; k < 8; ++k) sh.clk[k] = 0;
        // a stream in the middle of an utterance: its lists are this kernel's (eight wave segments)
        if (!needs_init && (old_nw != SW || dn0 != SW || dn1 != SW)) sh.err = (int)JDE_GEOM;
    }
    if (tid < SW) {                                                    // the fill counts of the lists the command before left
        const unsigned cr = g.seg_rec, cn = g.seg_new;
        auto clampc = [](int v, unsigned cap) { return v < 0 ? 0 : ((unsigned)v > cap ? (int)cap : v); };
        sh.c_rec[0][tid] = needs_init ? 0 : clampc(CL(tot_of(TOT_REC0) + tid), cr);
        sh.c_rec[1][tid] = needs_init ? 0 : clampc(CL(tot_of(TOT_REC1) + tid), cr);
        sh.c_new[tid] = needs_init ? 0 : clampc(CL(tot_of(TOT_NEW) + tid), cn);
        sh.c_dirty[0][tid] = needs_init ? 0 : clampc(CL(tot_of(TOT_DIRTY0) + tid), cn);
        sh.c_dirty[1][tid] = needs_init ? 0 : clampc(CL(tot_of(TOT_DIRTY1) + tid), cn);
        sh.c_exit[tid] = 0; sh.c_cl[0][tid] = sh.c_cl[1][tid] = 0; sh.c_cls[0][tid] = sh.c_cls[1][tid] = 0;
    }
    if (use_hist) {                                                    // the bins of the previous frame (the other parity's are this frame's: empty)
        const int pp = (f & 1) ^ 1;
        SLOT_LOOP for (int b = tid; b < C.hist_nbins; b += SNT) {
            sh.hist[pp][b] = needs_init ? 0 : CL(V.hist + (size_t)pp * HIST_MAX_BINS + b);
            sh.hist[pp ^ 1][b] = 0;
        }
    }
    __syncthreads();
    if (sh.err != 0) {                                                 // (JDE_GEOM: the host mixed the kernels up under a running utterance)
        if (tid == 0) { c.error = sh.err; c.needs_init = 0; }
        return;
    }
    int frames_done = 0;
    int my_item_end = 0;
    bool failed = false;
    bool init_pending = needs_init;

    // =============================================================== recognitionStart (:139-228), part 1
    if (needs_init) {
        // drop whatever the previous utterance left behind: instance flags, arrival keys (both parities) - lists of ANY geometry
        const int p0 = f & 1;
        const Geo go = make_geo(C, old_nw > 0 ? old_nw : SW), gd0 = make_geo(C, dn0 > 0 ? dn0 : SW), gd1 = make_geo(C, dn1 > 0 ? dn1 : SW);
        for (int kind = 0; kind < 3; ++kind) {
            const Geo &gk = kind == 0 ? go : kind == 1 ? gd0 : gd1;
            const GAS int *tk = tot_of(kind == 0 ? TOT_REC0 + p0 : kind == 1 ? TOT_DIRTY0 : TOT_DIRTY1);
            const unsigned cap = kind == 0 ? gk.seg_rec : gk.seg_new;
            for (int w = wid; w < gk.nw; w += SW) {
                int cnt = RFL(CL(tk + w));
                cnt = cnt < 0 ? 0 : ((unsigned)cnt > cap ? (int)cap : cnt);
                for (int i0 = 0; i0 < cnt; i0 += 64) {
                    if (i0 + lane < cnt) {
                        if (kind == 0) {
                            const unsigned roff = (p0 ? V.rec_par : 0u) + rec_chunk_off<NE>(gk.seg_rec, w, i0 >> 6) + (unsigned)lane * 16u;
                            CS(&V.live[ld16(V.rec, roff).x], (unsigned char)0);
                        } else {
                            const int b = CL(V.dirtyl + (kind == 2 ? V.dirty_par : 0u) + (size_t)w * gk.seg_new + (unsigned)(i0 + lane));
                            CS(&SREC_E(V.srec, C, b, 0), 0ULL); CS(&SREC_E(V.srec, C, b, 1), 0ULL);
                        }
                    }
                }
            }
        }
        if (tid == 0) {
            c.path_new = 0; c.path_new_ref = 0; c.n_collect = 0;
            for (int k = 0; k < ST_N; ++k) CS(&c.st[k], 0LL);
            c.best_final = null_tok();
            // the start token (:221-226) is the only item of round 0, in wave 0's segment (parity 1)
            Tok z; z.score = 0.0f; z.ac = 0.0f; z.lm = 0.0f; z.path = -1;
            st16(V.items, V.item_par, as_v4(z)); st16(V.items, V.item_par + 16u, (v4i){-1, 0, 0, 0});
            sh.c_exit[0] = 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        f = 0;
        best_emit = LZ;
    }

    // =============================================================== frames (the first pass may be recognitionStart part 2:
    // the expansion of the start token, a frame without phase A that uses item / key parity 1 like a frame "-1")
    int np_seen = needs_init ? 0 : sh.n_paths, npr_seen = needs_init ? 0 : sh.n_paths_ref;
    while (!failed) {
        const bool init = init_pending;
        if (!init && f >= f_stop) break;
        const int p = init ? 1 : (f & 1);
        if (!init && frames_done > 0 &&
            (np_seen > C.gc_threshold || (C.path_rule && path_rule_fires(ref_rule ? npr_seen : np_seen, path_new)))) break;
        long long t0 = 0;
        const bool clk_on = A.dbg != nullptr && tid == 0;
#define SCLK(slot) do { if (clk_on) { const long long tn_ = wall_clock64(); sh.clk[slot] += tn_ - t0; t0 = tn_; } } while (0)
        if (clk_on) t0 = wall_clock64();
        int exit_cnt = (init && wid == 0) ? 1 : 0;
        if (!init) {
            // ---- frame start (:311-339): thresholds + the work lists of phase A
            const float normalise = (best_emit > LZ) ? best_emit : 0.0f;             // :321
            if (ll_lds && row_in_lds != f) {                           // (a command's first frame: nobody has asked for its row yet)
                load_row(f);
                row_in_lds = f;
            }
            if (wid == 0) {                                            // entries before every segment of the three lists
                int t3;
                const int e0 = slot_prefix8(lane < SW ? sh.c_rec[p][lane] : 0, t3);
                if (lane <= SW) sh.pfx[0][lane] = lane == SW ? t3 : e0;
                const int e1 = slot_prefix8(lane < SW ? sh.c_new[lane] : 0, t3);
                if (lane <= SW) sh.pfx[1][lane] = lane == SW ? t3 : e1;
                const int e2 = slot_prefix8(lane < SW ? sh.c_dirty[p][lane] : 0, t3);
                if (lane <= SW) sh.pfx[2][lane] = lane == SW ? t3 : e2;
                if (lane == 0) {
                    atomicAdd(&sh.stat[ST_INSTS], sh.new_all);         // arcs entered in the previous frame are instances of this one (:899-935)
                    sh.new_all = 0; sh.nextA = 0; sh.nextX[0] = 0;
                }
            } else if (wid == 1) {
                float emitTh = (C.emit_win > 0.0f ? -C.emit_win : LZ);                   // :331
                if (use_hist) {                                        // bins of the previous frame (parity p ^ 1)
                    float th = hist_threshold(C, sh.hist[p ^ 1], lane);
                    th -= normalise;                                                     // :325
                    if (C.emit_win > 0.0f && th < -C.emit_win) th = -C.emit_win;         // :326-327
                    emitTh = th;
                }
                if (lane == 0) sh.emitTh = emitTh;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (this wave's share of the likelihood row has landed)
            __syncthreads();
            const int n0 = RFL(sh.pfx[0][SW]), n1 = RFL(sh.pfx[1][SW]), n2 = RFL(sh.pfx[2][SW]);
            const int Q0 = (n0 + 63) >> 6, Q1 = (n1 + 63) >> 6, Q2 = (n2 + 63) >> 6;
            const float emitTh = __int_as_float(RFL(__float_as_int(sh.emitTh)));
            const float startTh = (C.start_win > 0.0f) ? (best_emit - C.start_win) : LZ;   // :337
            const float *llrow = row_of(f);
            int out_cnt = 0;
            SCLK(0);
#define SLOT_A_ARGS C, sh, V, g, Q0, Q1, Q2, n0, n1, n2, p, normalise, emitTh, startTh, llrow, out_cnt, exit_cnt
#if defined(SLOT_EXP_NO_A)
#elif defined(SLOT_EXP_ONLY_LR)
            slot_phase_a<NE, true, true, true>(SLOT_A_ARGS);
#else
            if (ll_lds) {
                if (lr) slot_phase_a<NE, true, true, true>(SLOT_A_ARGS);
                else if (trp_lds) slot_phase_a<NE, true, false, true>(SLOT_A_ARGS);
                else slot_phase_a<NE, false, false, true>(SLOT_A_ARGS);
            } else {
                if (lr) slot_phase_a<NE, true, true, false>(SLOT_A_ARGS);
                else if (trp_lds) slot_phase_a<NE, true, false, false>(SLOT_A_ARGS);
                else slot_phase_a<NE, false, false, false>(SLOT_A_ARGS);
            }
#endif
#undef SLOT_A_ARGS
            SCLK(1);
            if (lane == 0) { sh.c_rec[p ^ 1][wid] = out_cnt; sh.c_exit[wid] = exit_cnt; }
            if (use_hist) SLOT_LOOP for (int b = tid; b < C.hist_nbins; b += SNT) sh.hist[p ^ 1][b] = 0;   // (the next frame's bins: read above, by wave 1, before the barrier)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // records, exit items and bids are out before anybody reads them
            __syncthreads();
            if (ll_lds && f + 1 < f_stop) {                            // the NEXT frame's row: nobody reads sh.ll before that frame's first barrier
                load_row(f + 1);
                row_in_lds = f + 1;
            }
            SCLK(2);
        }
        // ---- phase X
        const unsigned ba = (unsigned)RFL((int)(init ? 0u : sh.bestA[p]));
        const float bestA = ba ? o2f(ba) : LZ;
        const float endTh = (!init && C.end_win > 0.0f) ? (bestA - C.end_win) : LZ;     // :349
        const float wordTh = (!init && C.word_win > 0.0f) ? (bestA - C.word_win) : LZ;  // :350
        const bool last_frame = !init && f >= T - 1;
        XOut xo = {exit_cnt, 0, 0};
        for (int round = 0;; ++round) {
            // this round's items, per writer segment: exit tokens (round 0), else what the round before left for it
            const int rb = round & 1;
            const int cnt = lane < SW ? (round == 0 ? sh.c_exit[lane] : sh.c_cl[rb][lane]) : 0;
            const int start = lane < SW ? (round == 0 ? 0 : sh.c_cls[rb][lane]) : 0;
            int n_items;
            (void)slot_prefix8(cnt, n_items);
            if (round > 0 && n_items == 0) break;
            if (tid == 0) {                                            // housekeeping for what comes next (nobody reads these now)
                sh.nextX[rb ^ 1] = 0;
                if (round == 0) { sh.bestA[p ^ 1] = 0u; sh.bestX[p ^ 1] = 0u; }
            }
            int KX = 64;
            if (((n_items + 63) >> 6) < SW * C.x_chunks)
                while (KX > 4 && n_items < KX * SW * C.x_chunks) KX >>= 1;                 // a few chunks per wave
            int Q;
            const int xp = slot_prefix8((cnt + KX - 1) / KX, Q);
            SCLK(4);
            const int round_start = xo.item_cnt;
            int deferred = 0;
#ifndef SLOT_EXP_NO_X
            slot_phase_x(C, sh, V, g, Q, KX, round, xp, cnt, start, &sh.nextX[rb], p, init ? 0 : f, init, last_frame, endTh, wordTh, bestA, xo, deferred);
#endif
            SCLK(5);
            if (lane == 0) {
                sh.c_cl[rb ^ 1][wid] = deferred > 0 ? xo.item_cnt - round_start : 0;
                sh.c_cls[rb ^ 1][wid] = round_start;
                sh.c_new[wid] = xo.new_cnt;
                sh.c_dirty[p][wid] = xo.dirty_cnt;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            SCLK(6);
        }
        my_item_end = xo.item_cnt;
        // ---- frame end (every wave reads the same words behind the last barrier; they are reset two barriers later at the earliest)
        {
            const unsigned bx = (unsigned)RFL((int)sh.bestX[p]);
            const unsigned bb = ba > bx ? ba : bx;
            best_emit = bb ? o2f(bb) : LZ;                             // :417-418, :572-573
            np_seen = RFL(sh.n_paths); npr_seen = RFL(sh.n_paths_ref);
            if (RFL(sh.err) != 0) failed = true;
        }
        if (tid == 0) {                                                // totalActiveModels starts with frame 0 (:981)
            const unsigned long long ca = sh.cntA, cx = sh.cntX;
            sh.cntA = 0ULL; sh.cntX = 0ULL;
            sh.stat[ST_NEWL] = (int)(unsigned)ca; sh.stat[ST_KEYS] = (int)(unsigned)(ca >> 32);
            sh.stat[ST_WALK] = (int)(unsigned)cx; sh.stat[ST_CLOS] = (int)(unsigned)(cx >> 32);
            for (int k = 0; k < ST_N; ++k) { if (!init || k != ST_MODELS) sh.acc[k] += sh.stat[k]; sh.stat[k] = 0; }
        }
        if (last_frame && tid == 0) {                                  // bestFinalToken of this frame (:513-520)
            const unsigned long long fk = sh.final_key;
            Tok bf = null_tok();
            if (fk != 0ULL) {
                const unsigned fi = (unsigned)(fk & 0xffffffffULL);
                const unsigned ic = p ? V.item_par : 0u;
                const Tok it = as_tok(ld16(V.items, ic + fi * 32u));
                const int fst = ld16(V.items, ic + fi * 32u + 16u).z;
                const float fw = C.fin_w[fst];
                bf.score = o2f((unsigned)(fk >> 32)); bf.ac = it.ac; bf.lm = it.lm + fw; bf.path = it.path;
                sh.final_key = 0ULL;
            }
            c.best_final = bf;
        }
        if (init) init_pending = false;
        else { ++f; ++frames_done; }
    }
#undef SCLK
    // ---- end of the command: persist the stream state (read by the next command / the host kernels)
    __syncthreads();                                                   // (thread 0's last sums are in)
    const int plast = (f - 1) & 1;                                     // parity of the last frame processed (recognitionStart's pass: 1)
    if (lane == 0) {
        CS((GAS int *)S.item_end + wid, my_item_end);
        CS(tot_of(TOT_REC0) + wid, sh.c_rec[0][wid]); CS(tot_of(TOT_REC1) + wid, sh.c_rec[1][wid]);
        CS(tot_of(TOT_NEW) + wid, sh.c_new[wid]);
        CS(tot_of(TOT_DIRTY0) + wid, sh.c_dirty[0][wid]); CS(tot_of(TOT_DIRTY1) + wid, sh.c_dirty[1][wid]);
        CS(tot_of(TOT_EXIT) + wid, sh.c_exit[wid]);
        CS(tot_of(TOT_CL0) + wid, 0); CS(tot_of(TOT_CL1) + wid, 0);
    }
    if (use_hist)                                                      // Histogram: the last frame's bins where its parity has them, the other parity clear
        SLOT_LOOP for (int b = tid; b < C.hist_nbins; b += SNT) {
            CS(V.hist + (size_t)plast * HIST_MAX_BINS + b, sh.hist[plast][b]);
            CS(V.hist + (size_t)(plast ^ 1) * HIST_MAX_BINS + b, 0);
        }
    if (tid == 0) {
        for (int k = 0; k < ST_N; ++k) if (sh.acc[k]) atomicAdd((unsigned long long *)&c.st[k], (unsigned long long)sh.acc[k]);
        if (A.dbg) {
            long long *d = A.dbg + (size_t)blockIdx.x * 16;
            for (int k = 0; k < 8; ++k) d[k] += sh.clk[k];
            d[8] += frames_done;
        }
        CS(&c.n_paths, sh.n_paths); CS(&c.n_paths_ref, sh.n_paths_ref);
        CS(&c.new_all[plast], sh.new_all); CS(&c.new_all[plast ^ 1], 0);
        CS(&c.bestA[0], 0u); CS(&c.bestA[1], 0u); CS(&c.bestX[0], 0u); CS(&c.bestX[1], 0u);
        CS(&c.final_key, 0ULL); CS(&c.err[0], 0); CS(&c.err[1], 0);
        c.frame = f; c.best_emit = best_emit; c.lst_nw = SW; c.needs_init = 0;
        c.dirty_nw[0] = SW; c.dirty_nw[1] = SW;
        if (sh.err) c.error = sh.err;
        else if (f < f_stop && A.status) atomicAdd(A.status, 1);       // stopped early (a Path collection is due): the host collects and goes on
    }
}

// The slot kernel as an ordinary launch: one workgroup per stream of the work list, each runs its stream as far as the launch
// goes (frames < min(T, f_end), or up to a Path collection) and leaves.  Nothing has to be resident at once - the workgroups talk
// to nobody - so a batch with more utterances than the chip has room for needs no slots dealt by the host: the dispatcher puts the
// next workgroup on a CU the moment one leaves, two per CU.  (launch_search: batches of more streams than CUs; and what the PMC
// passes of profiles/ count - under the counters kernels run one after the other, and k_slot waits for the kernels beside it.)
template <int NE>
__global__ __launch_bounds__(SNT, SLOT_WPE) void k_slot_batch(SearchArgs A)
{
    __shared__ SlotShared sh;
    const int k = (int)blockIdx.x;
    if (k >= A.n_work) return;
    slot_run<NE>(A, sh, RFL(A.work[k].x), RFL(A.work[k].y));
}

// The slot kernel under the mailbox of jd_resident.h (the same commands, reports, ready numbers and host heartbeat as
// k_resident; grid = streams, one workgroup each, SLOT_WG_PER_CU of them per CU - all resident at once).
// started (host-mapped, or null): counted up by every workgroup when it is on its CU - the host releases the parked CUs then (jd_park_kernel)
template <int NE>
__global__ __launch_bounds__(SNT, SLOT_WPE) void k_slot(SearchArgs A, const ResPost *post, const unsigned *ready, ResDone *done, const unsigned *beat,
                                                        unsigned *started)
{
    __shared__ SlotShared sh;
    __shared__ unsigned long long sh_word;
    __shared__ int sh_exit;
    const int s = (int)blockIdx.x;
    const int tid = threadIdx.x;
    StreamCtl &c = A.ctl[s];
    unsigned seen = 0u;
    if (started && tid == 0) (void)__hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
        if (tid == 0) {
            long long t_idle = wall_clock64() + RES_IDLE_TICKS;
            unsigned last_beat = __hip_atomic_load(beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            auto host_gone = [&]() __attribute__((always_inline)) {
                if (wall_clock64() <= t_idle) return false;
                const unsigned bt = __hip_atomic_load(beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (bt == last_beat) return true;
                last_beat = bt; t_idle = wall_clock64() + RES_IDLE_TICKS;
                return false;
            };
            unsigned long long w = 0ULL;
            int ex = 0;
            unsigned spins = 0;
            for (;;) {
                w = __hip_atomic_load(&post[s].word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(w >> 32) != seen) break;
                ex = __hip_atomic_load(&post[s].exit_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ex) break;
                __builtin_amdgcn_s_sleep(64);
                if ((++spins & 255u) == 0 && host_gone()) { ex = 1; break; }
            }
            if (!ex) {
                const int T = __hip_atomic_load(&post[s].T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned rid = __hip_atomic_load(&post[s].ready_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                while ((int)(__hip_atomic_load(&ready[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - rid) < 0) {
                    __builtin_amdgcn_s_sleep(16);
                    if ((++spins & 255u) == 0 && host_gone()) { ex = 1; break; }
                }
                if (!ex) {
                    if (__hip_atomic_load(&post[s].init, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                        __hip_atomic_store(&c.needs_init, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&c.started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&c.error, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __hip_atomic_store(&c.T, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            sh_word = w; sh_exit = ex;
        }
        __syncthreads();
        const unsigned long long w = sh_word;
        const int ex = sh_exit;
        __syncthreads();
        if (ex) {
            if (tid == 0) __hip_atomic_store(&done[s].left, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const unsigned seq = (unsigned)(w >> 32);
        const int ll_slot = (int)(unsigned)(w & 0xffffffffULL);
        // what other kernels wrote since the last command - the likelihood rows, arenas swapped by a collection,
        // recognitionStart's mark - is read from memory, not from what this CU or this XCD's L2 still holds
        const long long t_cmd = wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_dcache_inv();
        slot_run<NE>(A, sh, s, ll_slot);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                               // every wave's end-of-command words are written before the host hears of it
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int fr = __hip_atomic_load(&c.frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int er = __hip_atomic_load(&c.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            done[s].frame = fr; done[s].error = er; done[s].run_ticks = wall_clock64() - t_cmd;
            __hip_atomic_store(&done[s].seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        seen = seq;
    }
}

// Where the slots go - a development tool now (JD_SLOT_KEEP_SE): the pipeline deals its slots ONE per CU and the scoring kernel's
// workgroups take the other half of the same CUs, which is the dispatcher's own arrangement for a grid of at most one workgroup
// per CU and measured 18 % faster than what this kernel is for (jd_host_resident.h: jd_res_start): TWO slots per CU on some
// CUs, the other CUs left whole to the scoring.  The dispatcher deals a grid smaller than twice the chip one workgroup per CU
// first; a CU-masked stream would change that, but hipExtStreamCreateWithCUMask only makes BLOCKING streams, and the legacy default
// stream then waits for a kernel that stays.  So the CUs the scoring is to keep are PARKED while the slots are dealt: this kernel asks
// for a CU's whole LDS - one workgroup per CU, nothing fits beside it - and of the workgroups that reach an XCD the first
// quota[xcd] stay until the host releases them; the others leave at once.  The slot kernel, launched then, finds room on the
// CUs that are left only - two workgroups each - and stays there for its life; the parked CUs are released when every slot
// has reported that it is on its CU (k_slot: started).  state (host-mapped): [0] parked, [1] left, [2] release.
#define PARK_LDS_BYTES 163840
// quota: workgroups that stay per SHADER ENGINE (the dispatcher deals a grid round robin over the XCDs and their shader engines and
// waits for room in the engine whose turn it is: an engine with fewer free CUs than the others holds the whole grid up - measured
// with quotas per XCD only: half of the slots were still waiting for a CU when the others had theirs)
__global__ __launch_bounds__(64) void jd_park_kernel(unsigned *cnt, int quota, unsigned *state)
{
    __shared__ unsigned char whole_cu[PARK_LDS_BYTES];
    whole_cu[threadIdx.x * 997 % PARK_LDS_BYTES] = (unsigned char)threadIdx.x;   // (the allocation is what matters)
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0x7u;       // HW_REG_XCC_ID[2:0]
    const unsigned se = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 13) & 0x7u; // HW_REG_HW_ID: SE_ID
    const unsigned r = __hip_atomic_fetch_add(&cnt[xcc * 8u + se], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)r >= quota) {
        (void)__hip_atomic_fetch_add(&state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    (void)__hip_atomic_fetch_add(&state[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t_lim = wall_clock64() + 100000000LL;                        // (a second: the host releases within milliseconds)
    while (__hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u && wall_clock64() < t_lim) __builtin_amdgcn_s_sleep(127);
    if (whole_cu[1] == 255 && whole_cu[2] == 254) state[3] = 1u;                 // (never: keeps the array)
}
