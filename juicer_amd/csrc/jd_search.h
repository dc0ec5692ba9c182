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
//            first entered in the previous frame.  An instance PULLS its entry token itself: the
//            previous frame's expansion left the best token that arrived at each STATE in a 64-bit
//            key (ordered score << 32 | frontier item), and all tokens at a state add the same arc
//            weight - so there is no separate "resolve" pass, no per-arc word is ever written, and
//            a new instance is only ever written if something in it survives its first frame.
//   phase X  frontier expansion (propagateToken): exit tokens and the epsilon / tee closure of what
//            a wave produces (expanded by that wave right away: a per-wave queue in LDS; what does
//            not fit goes to a further round behind a barrier) ARRIVE at their states - one atomic
//            max per arrival on the state's key is the whole Viterbi recombination (:560-582).
//
// No list is shared for appending: every wave owns a segment of each output list (instance
// records, frontier items, newly entered arcs) and publishes its fill count at the end of the
// phase; readers turn the counts into a prefix over fixed-size chunks and take chunks
// round-robin, so reading is balanced whatever the writers did.  The only returning atomics left
// are the arrivals (one per token and state, not per arc) and the Path-record reservation.
//
// Memory model (MI355X_MICROARCH.md, inter-workgroup visibility): every mutable per-stream word
// is written with agent-scope (sc1, write-through) stores or atomics and read with sc1 loads, which
// bypass the non-coherent per-CU L1; each storing wave drains (s_waitcnt vmcnt(0)) before it
// arrives at a barrier.  Static data (graph, models, likelihoods) uses plain cached loads.
#pragma once

#ifndef SW
#define SW 8                         // waves per search workgroup
#endif
#ifndef JD_KBOUNDS
#define JD_KBOUNDS __launch_bounds__(SNT, WG_PER_CU)
#endif
#ifndef WG_PER_CU
#define WG_PER_CU 1                  // search workgroups resident per CU
#endif
#define SNT (SW * 64)                // threads per search workgroup
#define MAXW SNT                      // most waves one cluster may have (one thread per wave when the lists are set up)
#define MAXCW (MAXW / SW)
#define TEE_FLAG 0x40000000          // bit 30 of the device arc's in-label: the arc's HMM is a tee model
#define SOLE_FLAG 0x20000000         // bit 29: the arc is the ONLY arc that leads to its destination state (jd_dec_create) - see REC_SOLE
#define ARC_FLAGS (TEE_FLAG | SOLE_FLAG)
#define TRP_LDS_MAX 4096             // floats of transition tables cached in LDS (else read from HBM)
#ifndef TEE_LDS_MAX
#define TEE_LDS_MAX 2048             // HMMs whose tee log-probability is cached in LDS
#endif

// the counters of what the kernels really touched (ST_RECS ..) can be compiled out (development: -DJD_COUNTERS=0, to measure what they cost)
#ifndef JD_COUNTERS
#define JD_COUNTERS 1
#endif
#if JD_COUNTERS
#define JD_COUNT(...) __VA_ARGS__
#else
#define JD_COUNT(...)
#endif
enum { ST_EMIT = 0, ST_END, ST_MODELS, ST_PEMIT, ST_PEND, ST_ARCS, ST_PATHS, ST_INSTS,
       // what the kernels really touched (jd_stats: tot_recs_read ..): the reference's figures above price the REFERENCE's work -
       // hopeless candidates that never become records, arcs a prefix walk accounts for without reading them
       ST_RECS,     // phase A: instance records read
       ST_NEWL,     // phase A: entries of the new-arc list taken up (attachNetInst from the arc's template)
       ST_SURV,     // phase A: records written to the next list
       ST_KEYS,     // phase A: entry tokens pulled (an arrival key that was set: the winning item is gathered)
       ST_XITEMS,   // phase X: frontier items taken up (exit tokens, closure items, slices)
       ST_WALK,     // phase X: arc records loaded by the walks
       ST_CLOS,     // phase X: closure items written
       ST_BIDS,     // phase A: bids placed for a destination state (exit tokens that are not alone into it, REC_SOLE)
       ST_N };
enum { JDE_SLOTS = -41, JDE_ITEMS = -42, JDE_PATHS = -43, JDE_NEW = -44, JDE_LAZY = -45, JDE_LAZY_INV = -46, JDE_BARRIER = -50 };

struct DecConst {
    // network (CSR in HBM)
    const int *row_ptr; const JdArc *arcs; const float *fin_w; int init_state; int n_states;
    const struct XState *xst;     // per state: the decoder's arc order and what phase X needs to cut a walk short (null: lazily composed networks)
    int xcut;                     // k_search cuts walks short too: most of the graph's model arcs sit in rows the cut applies to (jd_dec_create)
    // models
    int G, max_n, n_tm;
    const float *hmm_tee; int n_hmm;
    const struct LazyDev *lazy;   // search-driven composition (jd_lazy.h): the graph grows while the search runs; else null
    const int *aux_h;             // the instance template of an arc, by HMM: {nStates | transMat << 8, g0, g1, g2} (+ {g3, g4, g5, 0} for > 5 states)
    const float *hmm_tmax0;   // per HMM: largest log transition probability out of the entry state
    const float *trP; const int *se32;
    const float *lrt;   // left-to-right topologies only (else null): per transMat a_1.., s_1.. (see phase A)
    // pruning (WFSTDecoderLite ctor, WFSTDecoderLite.cpp:38-82)
    float start_win, emit_win, end_win, word_win;
    int max_hyps, hist_min, hist_max, hist_nbins;
    // arena capacities (per stream, in records)
    unsigned cap_slots, cap_items, cap_new; int cap_paths;
    int gc_threshold;   // a launch stops early (for the collection, k_gc_*) when more Path records than this are in use
    int x_chunks;       // phase X: chunks per wave the item lists are cut into (dynamic hand-out balances the arc walks)
    int exp;            // development experiments (JD_EXP)
    unsigned srec_stride, srec_arr, srec_estride, srec_par;   // the layout of a stream's per-state words (see StateRec): bytes between two states' bids; where e[0] of state 0
                        // sits; bytes between two states' arrival keys; bytes from a state's e[0] to its e[1] - joint 32 / 16 / 32 / 8, split 16 / 16 n / 16 / 8, split8 16 / 16 n / 8 / 8 n
    int path_rule;      // PARTIAL_DECODING is on: a stream also stops for a collection by the reference's count rule (path_rule_fires)
    const int *pcount;  // ... on the REFERENCE's Path counts: per state, the Path objects the reference creates for one token that
                        // arrives there (the labelled epsilon / tee arcs of its closure, with multiplicity; jd_dec_set_partial_interval).
                        // null: the rule runs on this build's own records (end / word beams on, lazily composed network)
};

// An active arc instance (NetInst, WFSTDecoderLite.h:66-75) is a record of 16-byte fields: header
// (arc, topology, tied-state ids, arc weight) + the tokens of its emitting states.  Entry and exit
// tokens are never stored: the entry token is pulled from the source state's arrival key in phase A, the exit token
// is consumed by phase X of the same frame (:964).  ONE LANE owns one instance, so records are
// stored in chunks of 64 as structure-of-arrays: field f of record l of a chunk sits at byte
// f * 1024 + l * 16 of the chunk - every load / store of a wave covers 1 KiB of consecutive bytes.
//   field 0 = arc, nStates | transMat << 8 | REC_LABELLED, source state, destination state
//   field 1 = g0, g1, g2, arc weight          (tied-state ids of emitting states 1..3)
//   NE == 6: field 2 = g3, g4, g5, -
//   then one field per emitting state: its token
// NE = 3 serves HMMs of up to 5 states (80-byte records), NE = 6 up to 8 states (144 bytes).
template <int NE> struct RecLayout {
    static constexpr int HF = (NE == 3) ? 2 : 3;       // header fields
    static constexpr int FIELDS = HF + NE;
    static constexpr int REC_BYTES = FIELDS * 16;
    static constexpr int CHUNK_BYTES = FIELDS * 1024;  // 64 records
};
#define OOB_OFF 0xf0000000u          // byte offset beyond every arena: buffer loads return 0, stores are dropped
#define REC_LABELLED 0x40000000      // bit 30 of a record's second header word: the instance's arc carries a word label
#define REC_SOLE 0x20000000          // bit 29: no other arc leads to the arc's destination state.  Its exit tokens have nobody to recombine
                                     // with (:560-582 compares the tokens that ARRIVE at a state): they place no bid in phase A and read,
                                     // win and reset none in phase X - an atomic, a load and a store less per exit token, each a 64-byte
                                     // sector of a state's words for 8 useful bytes.  Most states of a C.L.G are such states (the inside
                                     // of a word's chain, the nodes of a lexicon tree).  ITEM_SOLE: the same bit on the exit item.
#define ITEM_SOLE 4

// per-state search state, ONE 32-byte record (half a memory sector): the recombination keys of a state.
//   key0  best exit token arriving at the state this frame (bid in phase A, reset by its winner in phase X)
//   keyL  ... among the tokens whose arc carries a word label (own threshold, :952-962)
//   e[p]  best ARRIVAL at the state in a frame of parity p: (ordered score << 32) | frontier item.  Every token that
//         reaches the state in phase X - exit tokens past their threshold, epsilon / tee closure items - raises it
//         with an atomic max; it is what the arcs leaving the state PULL their entry tokens from in the next frame's
//         phase A (all arrivals at a state add the same arc weight and float addition is monotone, so the best arrival
//         is the best candidate of every out-arc: propagateToken's per-arc comparison :560-582, done once per state).
//         Frame f writes e[f & 1], frame f + 1 reads it, frame f + 2 zeroes it through the dirty list of that parity.
// The state's CSR row comes from row_ptr (shared by the streams, cache-resident; a per-stream copy inside this
// record was measured: no faster, and 64-byte records cost configs[3] 5 %).
// Round 2 kept one 16-byte {key, flag} record per ARC instead: an atomic, a poll and a reset per visited arc and frame,
// each a 64-byte sector for 8 useful bytes - the memory side carried out 13-15 G atomics / s on the heavy workloads,
// about what it can do (tools/traffic_probe: 16-17 G / s).
struct __align__(32) StateRec { unsigned long long key0, keyL, e[2]; };
// Where a state's 32 bytes sit in the stream's StateRec allocation is the DECODER's choice (DecConst::srec_stride / srec_arr, round 6):
//   joint  one 32-byte record per state (stride 32, arrival keys at +16): an exit token's bid and its arrival touch ONE line;
//   split  two arrays - the bids {key0, keyL} of all states, then the arrival keys e[2] of all states (stride 16, arrival keys behind
//          n_states bids): the arrival keys are what phase A gathers for every instance - the scattered HBM misses of the heavy
//          graphs - and 16-byte records put four states' keys in a 64-byte line where the joint record puts two.
// Which one pays is a property of the graph's numbering (jd_dec_create: the share of arcs that lead to the NEXT state number - the
// chains of a lexicon laid out state after state): measured on one box, split against joint: the 14 M-arc trigram-shaped graph +5.5 %,
// configs[3] +2.7 %, configs[1] with two batches in flight +2 %, the slot kernel's legs +-0, the device-composed configs[4] graph -3.4 %
// (canonical numbering: neighbours in number are no neighbours in time, and every exit token pays a second line).  The other direction
// was measured too: 64-byte records, -5 % (docs/state_pay_experiment.patch).
struct __align__(16) SBid { unsigned long long key0, keyL; };
struct __align__(16) SArr { unsigned long long e[2]; };
//   split8 (the default where split pays) goes one step further: the arrival keys of each frame PARITY in an array of their own - 8 bytes per
//          state, eight states to a line; a frame pulls from one parity and writes the other, never both of a state.
#define SREC_BID_OFF(C, st) ((unsigned)(st) * (C).srec_stride)
#define SREC_E_OFF(C, st, q) ((C).srec_arr + (unsigned)(q) * (C).srec_par + (unsigned)(st) * (C).srec_estride)
#define SREC_BID(base, C, st) (*(GAS SBid *)((GAS char *)(base) + SREC_BID_OFF(C, st)))
#define SREC_E(base, C, st, q) (*(GAS unsigned long long *)((GAS char *)(base) + SREC_E_OFF(C, st, q)))
// per-state STATIC record of the decoder's own copy of the graph (shared by the streams; jd_dec_create).  The decoder keeps the
// arcs of a state in an order of its own: first the arcs every arrival has to walk (epsilon inputs, tee models: n_always of
// them), then the arcs that enter a model, by DESCENDING w + tmax (arc weight + the model's largest entry transition) - the
// quantity phase X's "hopeless candidate" test runs on.  An arrival of score s can only enter the arcs of a PREFIX of that order;
// k[] samples the order at the positions xcand() so that an item finds an upper bound of its prefix from this one record instead
// of looking at every arc: the slot kernel (jd_slot.h: phase X) does not walk the arcs behind it at all.  What the walk did for
// them besides is accounted from here: n_model (arcs that carry a model, tee models included) less the instance flags set in the
// state's row (StreamDev::live: one byte per arc, a row's flags side by side) gives the arcs entered without an instance, wmax the
// best entry-token candidate.  (k_search walks every arc, in this order, and does not read this record.)
#define XNCAND 12
struct __align__(64) XState { int n_always, n_entry; float wmax; int n_model; float k[XNCAND]; };
__host__ __device__ __forceinline__ constexpr int xcand(int i)
{
    return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : i == 3 ? 3 : i == 4 ? 4 : i == 5 ? 6 : i == 6 ? 8 : i == 7 ? 12 : i == 8 ? 16 : i == 9 ? 24 : i == 10 ? 32 : 64;
}

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
    int dirty_nw[2];    // number of wave segments the dirty list of each frame parity was written with (it lives two frames)
    int path_new;       // Path records the last collection kept (nPathNew, WFSTDecoderLite.cpp:745; 0: none yet in this utterance)
    int n_collect;      // collections in this utterance (DecConst::pcount: those the reference runs too - its two triggers)
    int path_new_ref;   // DecConst::pcount: Path objects the reference's last collection kept (nPathNew, :745)
    int pad0[19];
    __align__(128) int new_all[2];           // arcs entered without an instance in a frame of that parity (listed or not)
    __align__(128) unsigned bar;             // cluster barrier (zeroed by the host before every launch)
    __align__(128) unsigned xbar, xmask;     // placement handshake of an XCD-local launch (agent scope; zeroed with bar)
    __align__(128) unsigned bestA[2];        // ordered-uint best emitting score of phase A, by frame parity
    __align__(128) unsigned bestX[2];        // ... best entry-token candidate of phase X
    __align__(128) int n_paths;              // Path records in use
    int n_paths_ref;                         // (same line) DecConst::pcount: Path objects the reference holds (nPath, :613 / :626)
    __align__(128) unsigned long long final_key;
    __align__(128) int err[2];               // first error raised during a frame of that parity
    int stop_req;                            // (same line) the cluster is to stop after this frame: the launch is being re-planned
    __align__(128) long long st[ST_N];       // statistics (WFSTDecoderLite.cpp:231-241 + build counters)
    __align__(128) Tok best_final;           // bestFinalToken of the last processed frame
};

struct StreamDev {      // per-stream arenas
    int *rec;                         // instance records, [2][cap_slots] by frame parity: list f&1 is read by frame f
    unsigned char *live;              // per ARC: 1 = an instance of this arc is in the list (set at its birth, cleared at its death)
    StateRec *srec;                   // per STATE: recombination keys + the CSR row (see StateRec)
    int4 *items;                      // frontier items of a frame, [2][cap_items] by frame parity: 32 bytes each,
                                      // token + {arc, out, to, flag}; flag 1 = a closure item that needs no
                                      // expansion in a later round (done by its producer, or superseded)
    int2 *newl;                       // {arc, source state}: arcs without an instance whose source state received a token this
                                      // frame that may survive the next one (they are tried in the next phase A)
    int *dirtyl;                      // [2][cap_new] by frame parity: states whose e[parity] became non-zero in that frame
    int *tot;                         // published per-wave fill counts, TOT_N arrays of MAXW
    int *item_end;                    // per wave: items written in the last processed frame (k_gc_*)
    PathRec *paths; int *hist;        // hist: [2][HIST_MAX_BINS] by frame parity
    PathRec *paths2; int *gc_idx;     // Path garbage collection: compaction target + mark / new-index array
    int *gc_state;                    // ... and its per-stream bookkeeping (GcState, jd_device.hip)
    // result of jd_finish_kernel
    int res_n; int *res_label; int *res_time; float *res_score, *res_ac, *res_lm; int res_cap;
};
enum { TOT_REC0 = 0, TOT_REC1 = 1, TOT_NEW = 2, TOT_DIRTY0 = 3, TOT_DIRTY1 = 4, TOT_EXIT = 5,
       TOT_CL0 = 6, TOT_CL1 = 7,        // closure items left for the next round: size of the range that holds them (0: none)
       TOT_CLS0 = 8, TOT_CLS1 = 9,      // ... and where that range starts in the wave's item segment
       TOT_N = 10 };

struct SearchArgs {
    DecConst C;
    StreamCtl *ctl; StreamDev *streams;
    const int4 *work;        // {stream, likelihood slot, first workgroup, workgroups} of every stream this launch advances
    int n_work;
    int Cw;                  // uniform mode: workgroups per cluster
    int n_slots;             // uniform mode: clusters in the grid (slot q serves work items q, q + n_slots, ...);
                             // 0 = weighted mode: work item k owns workgroups [work[k].z, work[k].z + work[k].w)
    const float *ll; long long ll_stride; int f0;   // likelihoods: ll[slot * ll_stride + (f - f0) * G + g]
    int f_end;               // process frames < min(T, f_end)
    int xl_selftest;         // test knob: an XCD-local launch numbers its workgroups in dispatch order, which puts every
                             // cluster on several XCDs - the placement check has to catch it
    int rebalance_at;        // weighted mode: when this many workgroups of the grid have nothing left to do (their stream is
                             // through, or stopped), every cluster stops after its frame and the host plans the rest anew (0: never)
    unsigned *cells;         // diagnostics (or null): one bit per cell of the likelihood slab, set when phase A adds the cell's value
                             // to a token (jd_dec_debug_cells: what part of a table the search reads - SURVEY.md 8d's Ug)
    int n_prio;              // weighted mode: n_prio work items (bit 30 of their workgroup count) are the batch the caller is waiting for; the others
                             // belong to the batch BEHIND it (searched ahead on workgroups the plan leaves, jd_device.hip
                             // "two batches in flight") and stop after their frame once all of these are through (0: none)
    int *status;             // [0] += 1 for every stream that stopped early (Path garbage collection, or a re-plan);
                             // [1] += 1 for every cluster of an XCD-local launch that found itself on several XCDs;
                             // [2] += the workgroups of every cluster that has left its stream; [3] += 1 per stream stopped for a re-plan;
                             // [4] != 0: the next batch's table is being scored on the CUs finished clusters left - no re-plan meanwhile
                             // [5] += 1 for every stream of the first n_prio that is through (or failed);
                             // [6] += 1 for every other stream that stopped because they all were
    long long *dbg;          // optional: per-workgroup cycle accounting (jd_dec_debug_trace)
    int *resident; int launch_seq;   // host-mapped word: the LAST workgroup of the grid writes launch_seq into it when it starts
                             // (workgroups are dispatched in order: the launch is then resident; jd_dec_prefetch_scores)
};

// ------------------------------------------------------------------ device helpers

typedef int v4i __attribute__((ext_vector_type(4)));
#define AUX_SC1 16
#define X_SLICE 256                  // phase X: arcs of one state a wave walks itself; the rest becomes slices for the next round
#define QCAP 64                      // closure items a wave keeps for itself (inline closure queue)
// The descriptor inputs go through readfirstlane so that the compiler can PROVE them wave-uniform;
// otherwise it wraps every buffer access in a "waterfall" loop (cdna_hip_programming.md, T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *p, unsigned long long bytes)
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bytes > 0xffffffffULL ? 0xffffffffULL : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)n, 0x00020000);
}
// 16-byte agent-scope (sc1) accesses through a wave-uniform buffer descriptor (out-of-range
// offsets read 0 / are dropped by the hardware bounds check)
#include "jd_lazy.h"

// Two ways of making a stream's mutable words visible to the other workgroups of its cluster.
//   XL = false  agent scope: `sc1` (write-through) stores, agent-scope atomics; right wherever the
//               workgroups run.  An `sc1` store (and an agent-scope atomic) drops its line from the XCD's
//               L2, so even a reader on the same XCD goes to memory for it.
//   XL = true   the cluster sits on ONE XCD (checked at run time, see run_stream): plain stores keep their
//               lines in that XCD's L2 and workgroup-scope atomics execute there - every hop between the
//               cluster's workgroups is an L2 hit.  Loads are `sc1` (L1 bypass, L2-served) either way: a
//               CU's vector L1 is never refreshed by another CU's stores.
// The functions below are used through the macros st16 / CS / GMAX / GADD, which pick the flavour of
// the enclosing function's `XL_`.
__device__ __forceinline__ v4i ld16(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX_SC1); }
typedef int v2i_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long ld8(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const v2i_ v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, AUX_SC1);
    return ((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x;
}
template <typename T> __device__ __forceinline__ T CL(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A pointer the kernel LOADS from memory (the per-stream arena pointers of StreamDev) is a generic pointer to the
// compiler, and every access through it a FLAT instruction: one that may touch LDS, so it counts on the LDS counter
// as well - and since the LDS counter cannot tell a flat load's return from a ds_read's, every wait for an LDS result
// (list look-ups, the queue, shuffles through ds_bpermute) while a flat load is in flight becomes a wait for that
// load: the software pipelines of both phases stalled at their first LDS operation (s_waitcnt vmcnt(0) lgkmcnt(0)
// right behind the arc walk's "one pass ahead" loads).  Hence: the arena pointers are address_space(1) (GLOBAL)
// pointers in the kernel (StreamView), and these are the accessors for them.
#define GAS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T CL(const GAS T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool XL> __device__ __forceinline__ void st16x(__amdgpu_buffer_rsrc_t r, unsigned off, v4i v)
{
    if (XL) __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, AUX_SC1);
}
template <bool XL, typename T> __device__ __forceinline__ void CSx(T *p, T v)
{
    if (XL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL, typename T> __device__ __forceinline__ void CSx(GAS T *p, T v)
{
    if (XL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL, typename T, typename U> __device__ __forceinline__ T GMAXx(GAS T *p, U v)
{
    return XL ? __hip_atomic_fetch_max(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
              : __hip_atomic_fetch_max(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL, typename T, typename U> __device__ __forceinline__ T GADDx(GAS T *p, U v)
{
    return XL ? __hip_atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
              : __hip_atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL, typename T, typename U> __device__ __forceinline__ T GMAXx(T *p, U v)
{
    return XL ? __hip_atomic_fetch_max(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
              : __hip_atomic_fetch_max(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL, typename T, typename U> __device__ __forceinline__ T GADDx(T *p, U v)
{
    return XL ? __hip_atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
              : __hip_atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define st16(r, off, ...) st16x<XL_>(r, off, __VA_ARGS__)
#define CS(p, ...) CSx<XL_>(p, __VA_ARGS__)
#define GMAX(p, ...) GMAXx<XL_>(p, __VA_ARGS__)
#define GADD(p, ...) GADDx<XL_>(p, __VA_ARGS__)
__device__ __forceinline__ Tok as_tok(v4i v)
{
    Tok t;
    t.score = __int_as_float(v.x); t.ac = __int_as_float(v.y); t.lm = __int_as_float(v.z); t.path = v.w;
    return t;
}
__device__ __forceinline__ v4i as_v4(const Tok &t)
{
    v4i v;
    v.x = __float_as_int(t.score); v.y = __float_as_int(t.ac); v.z = __float_as_int(t.lm); v.w = t.path;
    return v;
}
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
#define RFL(x) __builtin_amdgcn_readfirstlane(x)

#define NLISTS 4
#ifndef JD_PRECOUNT
#define JD_PRECOUNT 1               // development: 0 = the next frame's list counts are read at its start
#endif
struct SearchShared {
    // slot 0: the item list of phase X (chunk prefix / fill counts per writer segment, build_lists); slots 1..3: the
    // lists of phase A, PACKED (build_packed: entry prefix / writer segment of every non-empty segment)
    int pfx[NLISTS][MAXW + 1]; int cnt[NLISTS][MAXW];
    int start[MAXW];                           // per writer wave: where its items of the current round begin
    int hist[HIST_MAX_BINS];                   // this workgroup's share of the frame's histogram
    int hprev[HIST_MAX_BINS];                  // the stream's bins of the previous frame
    float trP[TRP_LDS_MAX]; int se[TRP_LDS_MAX / 4];   // transition tables (when they fit)
    float tee[TEE_LDS_MAX];                    // tee transition log-probability per HMM (when they fit)
    int wpfx[SW][64];                          // phase X: per wave, prefix of the out-degrees of its 64 items
    v4i qtok[SW][QCAP], qinfo[SW][QCAP];       // phase X: per wave, closure items it will expand itself
    int2 qrow[SW][QCAP];                       // ... and the CSR rows of their states (they came with the closure key)
    int wsum[NLISTS][SW], wsum2[NLISTS][SW];
    int next;                                  // next chunk (of this workgroup's share) to hand to a wave
    unsigned best;
    int abort;
    int new_all;                               // arcs entered without an instance (this workgroup, this frame)
    int stat[ST_N];                            // this workgroup's counters of the current frame
    long long acc[ST_N];                       // ... summed over the frames of the launch
    long long clk[8];
#ifdef JD_FINE
    long long fclk[7];                         // development build: hop timing inside the phases (FINE)
#endif
};

// Turn the published per-wave fill counts of up to NLISTS lists into chunk prefixes (LDS).  The
// loads of all lists are in flight together and the scans share two barriers.  All SNT threads
// call it.  K = records per chunk; segcap = capacity of a wave segment (counts are clamped to it); nw = the wave
// segments the list was written with (lists of different ages may come from launches of different geometries).
struct ListSrc { const GAS int *tot; int K; unsigned segcap; int nw; };
template <int N>
__device__ __forceinline__ void build_lists(SearchShared &sh, const ListSrc (&src)[N], int (&Q)[N], int (&items)[N])
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int c[N], n[N], x[N], y[N];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = (tid < src[k].nw) ? CL(src[k].tot + tid) : 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (c[k] < 0) c[k] = 0;
        if ((unsigned)c[k] > src[k].segcap) c[k] = (int)src[k].segcap;
        n[k] = (c[k] + src[k].K - 1) / src[k].K;
        x[k] = n[k]; y[k] = c[k];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int u = __shfl_up(x[k], o), v = __shfl_up(y[k], o);
            if (lane >= o) { x[k] += u; y[k] += v; }
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < N; ++k) { sh.wsum[k][wid] = x[k]; sh.wsum2[k][wid] = y[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int base = 0, total = 0, it = 0;
#pragma unroll
        for (int w = 0; w < SW; ++w) { const int s = sh.wsum[k][w]; if (w < wid) base += s; total += s; it += sh.wsum2[k][w]; }
        if (tid < src[k].nw) { sh.pfx[k][tid] = base + x[k] - n[k]; sh.cnt[k][tid] = c[k]; }
        if (tid == 0) sh.pfx[k][src[k].nw] = total;
        Q[k] = RFL(total); items[k] = RFL(it);
    }
    __syncthreads();
}
// the same list again with another chunk size (counts are already in LDS)
__device__ __forceinline__ int rebuild_list(SearchShared &sh, int k, int nw, int K)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = (tid < nw) ? sh.cnt[k][tid] : 0;
    const int n = (c + K - 1) / K;
    int x = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(x, o); if (lane >= o) x += u; }
    if (lane == 63) sh.wsum[k][wid] = x;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SW; ++w) { const int s = sh.wsum[k][w]; if (w < wid) base += s; total += s; }
    if (tid < nw) sh.pfx[k][tid] = base + x - n;
    if (tid == 0) sh.pfx[k][nw] = total;
    __syncthreads();
    return RFL(total);
}

// Phase A reads its lists PACKED: a chunk is 64 consecutive entries of the concatenation of the writers' segments,
// whatever segments they sit in.  (Chunks of ONE segment, as phase X takes them, leave every writer wave's last
// chunk partly empty: with 32-64 writer waves per stream and a few hundred new arcs per frame most chunks of the
// new list held a handful of entries, and a fifth of the record chunks were tails.)  Per list: the non-empty
// segments in order - sh.cnt[slot][i] = the i-th one's writer wave, sh.pfx[slot][i] = entries before it,
// sh.pfx[slot][ns] = all entries.  Loads of all lists in flight together, two barriers; all SNT threads call it.
// (the counts come from packed_counts: requested by the caller, one frame ahead when it can - see run_stream)
template <int N>
__device__ __forceinline__ void packed_counts(const ListSrc (&src)[N], int (&c)[N])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = (tid < src[k].nw) ? CL(src[k].tot + tid) : 0;
}
template <int N>
__device__ __forceinline__ void build_packed(SearchShared &sh, const ListSrc (&src)[N], int (&c)[N], int slot0, int (&Q)[N],
                                             int (&items)[N], int (&ns)[N])
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int x[N], y[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (c[k] < 0) c[k] = 0;
        if ((unsigned)c[k] > src[k].segcap) c[k] = (int)src[k].segcap;
        x[k] = c[k] > 0 ? 1 : 0; y[k] = c[k];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int u = __shfl_up(x[k], o), v = __shfl_up(y[k], o);
            if (lane >= o) { x[k] += u; y[k] += v; }
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < N; ++k) { sh.wsum[k][wid] = x[k]; sh.wsum2[k][wid] = y[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int bx = 0, tx = 0, by = 0, ty = 0;
#pragma unroll
        for (int w = 0; w < SW; ++w) {
            const int s = sh.wsum[k][w], t = sh.wsum2[k][w];
            if (w < wid) { bx += s; by += t; }
            tx += s; ty += t;
        }
        if (c[k] > 0) { const int i = bx + x[k] - 1; sh.cnt[slot0 + k][i] = tid; sh.pfx[slot0 + k][i] = by + y[k] - c[k]; }
        if (tid == 0) sh.pfx[slot0 + k][tx] = ty;
        Q[k] = RFL((ty + 63) >> 6); items[k] = RFL(ty); ns[k] = RFL(tx);
    }
    __syncthreads();
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
    return RFL(base + __popcll(m2) - 1);
}

// Entry `lane` of chunk ru of a packed list (build_packed): its writer segment and its index there.  The segments
// in the table are non-empty, so the 64 boundaries behind the chunk's first segment cover the chunk: the lanes mark
// the boundaries that fall into it in a wave-private scratch row, and a lane's segment is the first one plus the
// boundaries at or before it.
__device__ __forceinline__ void packed_lane(const int *cp, const int *sg, int ns, int total, int ru, int *scratch,
                                            bool &valid, int &w, int &idx)
{
    const int lane = threadIdx.x & 63;
    const int base = ru << 6;
    const int w0 = max(find_seg(cp, ns, base), 0);                     // (an empty list: nothing is valid, the look-ups stay in range)
    const int j = w0 + 1 + lane;
    const int r = (j < ns ? cp[j] : 0x3fffffff) - base;               // (> 0: segments are non-empty)
    // (lanes talk to each other through the row: wavefront-scope atomics, or the compiler - which sees one thread -
    // concludes that a lane that marks nothing reads back its own zero and skips the load)
    __hip_atomic_store(&scratch[lane], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (r < 64) __hip_atomic_store(&scratch[r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned long long B = __ballot(__hip_atomic_load(&scratch[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) != 0);
    const int g = base + lane;
    valid = g < total;
    const int wi = valid ? w0 + __popcll(B & (~0ULL >> (63 - lane))) : w0;
    w = sg[wi]; idx = g - cp[wi];
}

// Chunks of a phase are dealt to workgroups round-robin (chunk u belongs to workgroup u % Cw) and,
// inside a workgroup, handed to whichever wave is free next - the waves of a workgroup wait for each
// other at the end of the phase anyway.  sh.next is reset (and a barrier passed) before the phase.
__device__ __forceinline__ int grab_chunk(SearchShared &sh, int jw, int Cw)
{
    int k = 0;
    if ((threadIdx.x & 63) == 0) k = atomicAdd(&sh.next, 1);
    return jw + RFL(k) * Cw;
}

// collectPaths' count trigger (WFSTDecoderLite.cpp:360-362): nPath / nPathNew > 12 and nPath > 10000, with the IEEE float
// division of the reference (nPathNew = 0 before the first collection: the ratio is +inf).  The counts are this
// reference's where DecConst::pcount is there (n_paths_ref / path_new_ref: a Path per labelled propagateToken call, winner
// or not, and what collectPaths keeps), else this build's own records - at most the reference's, which also creates
// records for tokens that lose their state's recombination.
__device__ __host__ __forceinline__ bool path_rule_fires(int n_path, int n_path_new)
{
    return n_path > 10000 && (float)n_path / (float)n_path_new > 12.0f;
}

// ---- cluster barrier: all Cw workgroups of one stream.  target = Cw * (number of this barrier).
template <bool XL>
__device__ __forceinline__ void cluster_barrier(SearchShared &sh, StreamCtl &c, int Cw, unsigned &nbar, long long t_limit)
{
    constexpr bool XL_ = XL;
    // every wave: its write-through stores and atomics have been performed before anybody is told
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ++nbar;
    if (threadIdx.x == 0) {
        sh.next = 0;
        if (Cw > 1) {
            GADD(&c.bar, 1u);
            const unsigned target = nbar * (unsigned)Cw;
            unsigned spins = 0;
            while (CL(&c.bar) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && wall_clock64() > t_limit) { sh.abort = 1; break; }
            }
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
    unsigned seg_rec, seg_item, seg_new;    // records per wave segment (seg_rec: a multiple of 64)
};
__device__ __host__ __forceinline__ Geo make_geo(const DecConst &C, int nw)
{
    Geo g; g.nw = nw;
    g.seg_rec = (C.cap_slots / (unsigned)nw) & ~63u; g.seg_item = C.cap_items / (unsigned)nw; g.seg_new = C.cap_new / (unsigned)nw;
    return g;
}

struct StreamView {     // wave-uniform descriptors of one stream's arenas
    __amdgpu_buffer_rsrc_t rec, items;          // both frame parities in one descriptor each
    unsigned rec_par, item_par;                 // byte offset of parity 1 in them
    __amdgpu_buffer_rsrc_t lrows, larcs;        // lazy graphs: the rows and the arc arena (read with `sc1` loads)
    __amdgpu_buffer_rsrc_t srec_r;              // the per-state records, for 16-byte loads
    GAS unsigned char *live; GAS StateRec *srec; GAS unsigned long long *newl; GAS int *dirtyl; unsigned dirty_par; GAS int *tot;
    GAS v4i *paths; GAS int *hist;              // (global pointers, not generic ones: see GAS)
};

// byte offset of chunk ci of wave segment w in a record list (parity offset added by the caller)
template <int NE>
__device__ __forceinline__ unsigned rec_chunk_off(unsigned seg_rec, int w, int ci)
{
    return ((unsigned)w * (seg_rec >> 6) + (unsigned)ci) * (unsigned)RecLayout<NE>::CHUNK_BYTES;
}

// ------------------------------------------------------------------ phase A
//
// doHMMInternalPropagation (:899-935) + HMMInternalPropagation (:376-484).  One lane owns one
// instance and updates its emitting states in turn, so a wave has 64 instances in flight and all
// loads of a pass are issued back to back (no divergent load branches: lanes without work read
// out of range and get zeros).  Waves take chunks of 64 (of ONE writer segment); survivors and exit
// tokens go to the wave's own output segments - no atomics, no barriers.  Work items, in this
// order: chunks of instance records (list 0), of newly entered arcs (1), of states whose arrival keys
// of the frame before the previous one need zeroing (2).  An instance PULLS its entry token: the best
// arrival at its arc's source state in the previous frame (StateRec::e) plus the arc's weight.
//
// LR: every transition matrix of the model set is plain left-to-right (state j is entered from j-1 and
// itself, the exit state from the last emitting state; no skips): the predecessor loops become one
// comparison per state, on a compact table a_k = log P(k-1 -> k), s_k = log P(k -> k) in LDS.
#ifdef JD_FINE
// development build (-DJD_FINE=1: phase A, =2: phase X): drains the memory counters and charges the time
// since the last mark to slot k; thread 0 of every workgroup only.  Serialises the hops it measures -
// not for benchmarks.
#define FINE_START_() long long ft_ = 0; do { if (threadIdx.x == 0) ft_ = wall_clock64(); } while (0)
#define FINE_(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                      if (threadIdx.x == 0) { const long long tn_ = wall_clock64(); sh.fclk[k] += tn_ - ft_; ft_ = tn_; } } while (0)
#define FINE_COUNT_(k) do { if (threadIdx.x == 0) sh.fclk[k] += 1; } while (0)
#endif
#if defined(JD_FINE) && JD_FINE == 1
#define FINE_START() FINE_START_()
#define FINE(k) FINE_(k)
#else
#define FINE_START() do { } while (0)
#define FINE(k) do { } while (0)
#endif
#if defined(JD_FINE) && JD_FINE == 2
#define XFINE_START() FINE_START_()
#define XFINE(k) FINE_(k)
#define XFINE_COUNT(k) FINE_COUNT_(k)
#else
#define XFINE_START() do { } while (0)
#define XFINE(k) do { } while (0)
#define XFINE_COUNT(k) do { } while (0)
#endif

template <int NE, bool TRPL, bool LR, bool XL, bool LZY>
__device__ __forceinline__ void phase_a(const DecConst &C, SearchShared &sh, StreamCtl &c, const StreamView &V,
                                        const Geo &gin, const Geo &gd, const Geo &gout, const int (&Q)[3], const int (&LN)[3], const int (&NS)[3],
                                        int jw, int Cw, int gw, int p,
                                        float normalise, float emitTh, float startTh, const float *llrow, unsigned *cells, long long cell0,
                                        int &out_cnt, int &exit_cnt)
{
    constexpr bool XL_ = XL;
    typedef RecLayout<NE> RL;
    constexpr int HF = RL::HF;
    const int lane = threadIdx.x & 63;
    int *const scratch = sh.wpfx[RFL(threadIdx.x >> 6)];               // (phase X's row of this wave: free during phase A)
    const int MN = C.max_n;
    const bool use_hist = C.max_hyps > 0;
    const float *trP_all = TRPL ? sh.trP : C.trP;
    const int *se_all = TRPL ? sh.se : C.se32;
    const unsigned rcur = p ? V.rec_par : 0u, rnext = p ? 0u : V.rec_par;           // byte offsets of the two lists
    const unsigned iprev = p ? 0u : V.item_par, icur = p ? V.item_par : 0u;
    const unsigned item_base = (unsigned)gw * gout.seg_item;
    const int Q01 = Q[0] + Q[1], Qall = Q01 + Q[2];
    int c_insts = 0, c_pemit = 0, c_emit = 0, c_end = 0, c_surv = 0, c_recs = 0, c_keys = 0;
    unsigned mo = 0u;
    // The pass loop is software-pipelined two deep.  A pass is a chain of dependent memory round trips
    // (record -> source state's arrival key + likelihoods -> winning item) followed by arithmetic and stores, and with
    // two waves per SIMD nothing else hides them; the memory counter is in-order, so a wait for a
    // load also waits for every store issued before it.  Hence: the NEXT chunk's record is requested
    // while this chunk's item is in flight (stage R), and its key + likelihoods right BEFORE this
    // chunk's stores (stage K) - by the time they are needed they are there, and no wait has a store
    // in front of it.  Chunks come in increasing order per workgroup (records, then new arcs, then
    // the clean-up list), so the clean-up chunks form a plain loop of their own at the end.
    auto stage_r = [&](int u, bool &is_new, bool &valid, int2 &nb, v4i &h0, v4i &h1, v4i &h2, Tok (&tk)[NE + 1])
        __attribute__((always_inline)) {
        // Every load of this stage is issued whatever kind of chunk this is, and even when the lists are through (u >= Q01):
        // lanes without a record read out of range (zeros, no memory access), lanes without a new arc read entry 0.  A
        // load inside a branch - even a wave-uniform one - reaches the registers that carry it to the next pass through
        // a copy at the join, and the compiler waits for the load right there: the pipeline below would be no pipeline.
        const bool any = u < Q01;
        is_new = u >= Q[0];
        const int ql = is_new ? Q[1] : Q[0];
        int ru = is_new ? u - Q[0] : u;
        ru = ru < ql ? ru : ql - 1;
        ru = ru > 0 ? ru : 0;
        int w, idx;                                                    // (per lane: the writer segment and the entry's index in it)
        packed_lane(sh.pfx[is_new ? 2 : 1], sh.cnt[is_new ? 2 : 1], is_new ? NS[1] : NS[0], is_new ? LN[1] : LN[0], ru, scratch, valid, w, idx);
        valid = valid && any;
        const bool rec = valid && !is_new, fresh = valid && is_new;
        const unsigned off = rec ? rcur + rec_chunk_off<NE>(gin.seg_rec, w, idx >> 6) + (unsigned)(idx & 63) * 16u : OOB_OFF;
        h0 = ld16(V.rec, off); h1 = ld16(V.rec, off + 1024u);
        if (NE == 6) h2 = ld16(V.rec, off + 2048u);
#pragma unroll
        for (int j = 1; j <= NE; ++j) tk[j] = as_tok(ld16(V.rec, off + (unsigned)(HF + j - 1) * 1024u));
        const unsigned long long e = CL(V.newl + (fresh ? (size_t)w * gin.seg_new + (unsigned)idx : (size_t)0));
        nb = fresh ? make_int2((int)(unsigned)e, (int)(unsigned)(e >> 32)) : make_int2(0, 0);   // {arc, source state}
    };
    auto stage_k = [&](bool is_new, bool valid, int2 nb, v4i &h0, v4i &h1, v4i &h2, Tok (&tk)[NE + 1],
                       unsigned long long &kv, float (&outp)[NE]) __attribute__((always_inline)) {
        if (is_new) {                                                  // attachNetInst :751-774, from the arc's template
            JdArc Bk;
            int4 a0, a1 = make_int4(0, 0, 0, 0);
            if (LZY) {                                                 // the arena, and the template by HMM
                const v4i r = ld16(V.larcs, (unsigned)nb.x * 16u);
                Bk = JdArc{r.x, __int_as_float(r.y), r.z, r.w};
            } else Bk = C.arcs[nb.x];
            {   // the template by HMM (a table of a few tens of KB: L2 hits; a per-arc copy would be a second random sector)
                const int hm = max((Bk.in & ~ARC_FLAGS) - 1, 0);        // (arcs on the new list carry a model; idle lanes read arc 0)
                a0 = ((const int4 *)C.aux_h)[(NE == 3) ? hm : 2 * hm];
                if (NE == 6) a1 = ((const int4 *)C.aux_h)[2 * hm + 1];
            }
            // header: arc, nStates | transMat << 8 | (the arc carries a word label) << 30, source state, destination state
            h0 = (v4i){nb.x, valid ? (a0.x | (Bk.out != 0 ? REC_LABELLED : 0) | ((Bk.in & SOLE_FLAG) ? REC_SOLE : 0)) : 0, nb.y, Bk.to};
            h1 = (v4i){a0.y, a0.z, a0.w, __float_as_int(Bk.w)};
            if (NE == 6) h2 = (v4i){a1.x, a1.y, a1.z, 0};
#pragma unroll
            for (int j = 1; j <= NE; ++j) tk[j] = null_tok();
        }
        const int n = h0.y & 0xff;
        {   // the best arrival at the source state (StateRec::e[p ^ 1]) and the likelihoods: in flight together (no branch:
            // lanes without an instance read out of range)
            kv = ld8(V.srec_r, valid ? SREC_E_OFF(C, h0.z, p ^ 1) : OOB_OFF);
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int gj = (j == 0) ? h1.x : (j == 1) ? h1.y : (j == 2) ? h1.z : (j == 3) ? h2.x : (j == 4) ? h2.y : h2.z;
            outp[j] = llrow[(j + 1 < n - 1) ? gj : 0];                 // :411
        }
    };
    int u = grab_chunk(sh, jw, Cw);
    bool is_new = false, valid = false;
    int2 nb = make_int2(0, 0);
    v4i h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0}, h2 = {0, 0, 0, 0};
    Tok tk[NE + 1];
    unsigned long long kv = 0ULL;
    float outp[NE];
    stage_r(u, is_new, valid, nb, h0, h1, h2, tk);
    stage_k(is_new, valid, nb, h0, h1, h2, tk, kv, outp);
#pragma nounroll
    while (u < Q01) {
        FINE_START();
        const int un = grab_chunk(sh, jw, Cw);
        FINE(0);                                                       // (development build) the wait for stage K
        const int arc = h0.x;
        const int n = h0.y & 0xff;                                     // 0 for lanes without an instance
        const int tm = (h0.y >> 8) & 0x1fffff;
        // entry token = the best token that arrived at the arc's source state in the previous frame, over the arc (:560-582)
        const v4i itv = ld16(V.items, kv != 0ULL ? iprev + (unsigned)(kv & 0xffffffffULL) * 32u : OOB_OFF);   // (no branch, see stage_r)
        // stage R of the next chunk (issued after the item load: the wait for the item leaves it in flight)
        bool n_is_new = false, n_valid = false;
        int2 n_nb = make_int2(0, 0);
        v4i nh0 = {0, 0, 0, 0}, nh1 = {0, 0, 0, 0}, nh2 = {0, 0, 0, 0};
        Tok ntk[NE + 1];
        stage_r(un, n_is_new, n_valid, n_nb, nh0, nh1, nh2, ntk);
        FINE(1);                                                       // the winning item (+ the next record)
        tk[0] = null_tok();
        if (kv != 0ULL) {
            const Tok it = as_tok(itv);
            tk[0].score = o2f((unsigned)(kv >> 32)) + __int_as_float(h1.w);   // :562 newScore = tok.score + weight
            tk[0].ac = it.ac; tk[0].lm = it.lm + __int_as_float(h1.w); tk[0].path = it.path;
            if (tk[0].score < startTh) tk[0] = null_tok();            // :915-918 (a candidate is never LOG_ZERO)
        }
        Tok nw[NE + 1];
        int live_mask = 0;
        Tok ex = null_tok();
        auto emit = [&](int j, float best, float btp, const Tok &src) __attribute__((always_inline)) {   // :408-424
            const float sc = best - normalise;                         // :408
            if (sc > emitTh) {                                         // :409
                ++c_pemit;
                if (cells) {                                           // (diagnostics: this cell of the table is read, :411)
                    const int gj = (j == 1) ? h1.x : (j == 2) ? h1.y : (j == 3) ? h1.z : (j == 4) ? h2.x : (j == 5) ? h2.y : h2.z;
                    const long long cell = cell0 + gj;
                    atomicOr(cells + (cell >> 5), 1u << (cell & 31));
                }
                nw[j].score = sc + outp[j - 1];
                nw[j].ac = (src.ac + btp) + outp[j - 1];
                nw[j].lm = src.lm;
                nw[j].path = src.path;
                live_mask |= 1 << j;
                if (use_hist) {                                        // Histogram::addScore, Histogram.cpp:64-100
                    const double ds = (double)nw[j].score;
                    const int sci = (nw[j].score < 0.0f) ? (int)(ds - 0.5) : (int)(ds + 0.5);
                    if (sci > C.hist_max) CS(&c.err[p], (int)JD_EHIST);
                    else if (sci >= C.hist_min) atomicAdd(&sh.hist[sci - C.hist_min], 1);
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
        FINE(2);                                                       // arithmetic
        // stage K of the next chunk: its record has arrived during the arithmetic
        unsigned long long nkv = 0ULL;
        float noutp[NE];
        stage_k(n_is_new, n_valid, n_nb, nh0, nh1, nh2, ntk, nkv, noutp);
        c_emit += __popc(live_mask);
        const bool has_exit = ex.score > LZ;
        const bool slot_live = live_mask != 0;
        const unsigned long long bl = __ballot(slot_live), be = __ballot(has_exit);
        if (!is_new) c_insts += __popcll(__ballot(valid));             // (new arcs are counted when they are entered)
        JD_COUNT(if (is_new) c_recs += __popcll(__ballot(valid)); c_keys += __popcll(__ballot(valid && kv != 0ULL)));
        // survivors: header + new tokens to this wave's segment of the next list
        {
            const int nsurv = __popcll(bl);
            if (out_cnt + nsurv > (int)gout.seg_rec) { if (lane == 0) CS(&c.err[p], (int)JDE_SLOTS); }
            else {
                const int pos = out_cnt + rank_in(bl);
                const unsigned doff = slot_live ? rnext + rec_chunk_off<NE>(gout.seg_rec, gw, pos >> 6) + (unsigned)(pos & 63) * 16u : OOB_OFF;
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
        // exit tokens: frontier items of round 0 in this wave's item segment, bidding for their
        // destination state (state-level recombination, see phase X); tokens leaving word-labelled
        // arcs face their own threshold (:952-962) -> own key class
        {
            const int nex = __popcll(be);
            if (exit_cnt + nex > (int)gout.seg_item) { if (lane == 0) CS(&c.err[p], (int)JDE_ITEMS); }
            else {
                const unsigned k = item_base + (unsigned)(exit_cnt + rank_in(be));
                const unsigned ioff = has_exit ? icur + k * 32u : OOB_OFF;
                st16(V.items, ioff, as_v4(ex));
                const int lab = (h0.y & REC_LABELLED) ? 1 : 0;         // (the label itself is read from the arc when a Path record is written)
                const int sole = (h0.y & REC_SOLE) ? ITEM_SOLE : 0;
                st16(V.items, ioff + 16u, (v4i){arc, lab, h0.w, sole});
                if (has_exit && !sole) GMAX((lab ? &SREC_BID(V.srec, C, h0.w).keyL : &SREC_BID(V.srec, C, h0.w).key0), ((unsigned long long)f2o(ex.score) << 32) | k);
                JD_COUNT(const int nbid_ = __popcll(__ballot(has_exit && !sole)); if (lane == 0 && nbid_) atomicAdd(&sh.stat[ST_BIDS], nbid_));
                exit_cnt += nex;
                c_end += nex;
            }
        }
#if defined(JD_FINE) && JD_FINE == 1
        // issue of stage K + stores (not drained)
        if (threadIdx.x == 0) { const long long tn_ = wall_clock64(); sh.fclk[3] += tn_ - ft_; sh.fclk[4] += 1; }
#endif
        // the next chunk becomes the current one
        u = un; is_new = n_is_new; valid = n_valid; nb = n_nb; h0 = nh0; h1 = nh1; h2 = nh2; kv = nkv;
#pragma unroll
        for (int j = 1; j <= NE; ++j) tk[j] = ntk[j];
#pragma unroll
        for (int j = 0; j < NE; ++j) outp[j] = noutp[j];
    }
    // key clean-up: the arrival keys e[p] of the frame before the previous one have been pulled from (by the
    // previous frame's phase A) and are written again by this frame's phase X, behind the barrier
#pragma nounroll
    for (; u < Qall; u = grab_chunk(sh, jw, Cw)) {
        bool on;
        int w, idx;
        packed_lane(sh.pfx[3], sh.cnt[3], NS[2], LN[2], u - Q01, scratch, on, w, idx);
        if (on) {                                                      // (gd: the geometry this list was written with, two frames ago)
            const int b = CL(V.dirtyl + (p ? V.dirty_par : 0u) + (size_t)w * gd.seg_new + (unsigned)idx);
            CS(&SREC_E(V.srec, C, b, p), 0ULL);
        }
    }
    // per-wave totals -> workgroup counters (LDS)
    mo = wave_umax(mo);
    c_pemit = wave_sum(c_pemit); c_emit = wave_sum(c_emit);
    if (lane == 0) {
        if (mo) atomicMax(&sh.best, mo);
        if (c_insts) { atomicAdd(&sh.stat[ST_INSTS], c_insts); atomicAdd(&sh.stat[ST_RECS], c_insts); }
        if (c_pemit) atomicAdd(&sh.stat[ST_PEMIT], c_pemit);
        if (c_emit) atomicAdd(&sh.stat[ST_EMIT], c_emit);
        if (c_end) atomicAdd(&sh.stat[ST_END], c_end);
        if (c_surv) { atomicAdd(&sh.stat[ST_MODELS], c_surv); atomicAdd(&sh.stat[ST_SURV], c_surv); }
        if (c_recs) atomicAdd(&sh.stat[ST_NEWL], c_recs);
        if (c_keys) atomicAdd(&sh.stat[ST_KEYS], c_keys);
    }
}

// ------------------------------------------------------------------ phase X
//
// propagateToken (:491-605).  One lane owns one frontier item (threshold, winner check, Path
// record, final state); the out-arcs of the wave's items are then pooled: lane l takes arcs
// l, l+64, ... of the concatenated ranges, so a history state with thousands of out-arcs occupies
// the whole wave and items with few arcs share a pass.
//
// State-level recombination.  All tokens at one state add the same arc weights and float addition
// is monotone, so only the best token at a state can win anything downstream.  Exit tokens (round
// 0) bid for their state in phase A; after the barrier exactly the best one (per threshold class)
// finds its own index in the state's key, and if it passes its threshold it ARRIVES: an atomic max
// on the state's arrival key e[p] (StateRec).  Closure items (a token that has just traversed an
// epsilon arc or a tee model, :533-540 / :584-600) arrive the same way when they are produced, and are
// expanded iff they are the best arrival at their state so far (a running maximum).
// The arrival key is all that the next frame needs of this expansion: phase A of the next frame
// pulls the entry token of every arc leaving the state from it.  What an arrival still does here:
//   * it walks the state's arcs to send its token on through epsilon arcs and tee models (closure
//     items; a wave expands the ones it produces itself right away, QCAP of them wait in LDS, the
//     rest is left for a further round) - only the best arrival SO FAR does (running maximum: the
//     best one always does, a state is walked O(log arrivals) times, results are the same);
//   * it lists the arcs of the state that have no instance yet for the next phase A (new list), so
//     that they are tried - attachNetInst :751-774 happens there, and only if the instance survives
//     its first frame.  No per-arc word is written: which arrival lists an arc is decided by the
//     chain of maxima the atomic on e[p] orders the arrivals in (below).
//
// Hopeless candidates.  An entry token with (score + max_j trP[0][j]) - bestA <= -mainBeam
// (bestA = this frame's best emitting score) fails :409 next frame whatever happens: that frame
// normalises by bestEmitScore >= bestA, its emit threshold is >= -mainBeam, float ops are
// monotone.  The reference attaches an instance for it, counts it and lets it die.  Here the arc
// is counted (new_all, by the FIRST arrival at its state) but only listed by the first arrival
// whose token is not hopeless for it: an arrival knows the best score before it (the atomic's old
// value) and lists exactly the arcs that are hopeful for its own score and were not for that one -
// every arc once, whatever order the arrivals are expanded in.
//
// Items (32 bytes): token + {x, label, state, flags}.  Exit tokens (round 0): x = their arc (the start
// token: -1), label = 1 if the arc carries a word label (the label is read from the arc when the Path
// record is written).  Closure items and slices: x = ordered score of the best arrival BEFORE this one
// (0: none - this one is the first), label = the arc's word label.  flags & 3: 0 = to be expanded in
// a later round, 1 = done (expanded by its producer, or superseded), 2 = a slice (>> 2: its number).
struct XOut { int item_cnt; int new_cnt; int dirty_cnt; };
template <bool XL, bool LZY>
__device__ __forceinline__ void phase_x(const DecConst &C, SearchShared &sh, StreamCtl &c, const StreamView &V,
                                        const Geo &gin, const Geo &gout, int Q, int KX, int round, int jw, int Cw, int gw,
                                        int p, int pframe, bool init, bool last_frame, float endTh, float wordTh,
                                        float bestA, XOut &out, int &deferred)
{
    constexpr bool XL_ = XL;
    const int lane = threadIdx.x & 63;
    const int wid = RFL(threadIdx.x >> 6);
    const float INF = __builtin_inff();
    const unsigned icur = p ? V.item_par : 0u;
    const bool can_filter = !init && C.emit_win > 0.0f && bestA > LZ;
    const bool tee_lds = C.n_hmm <= TEE_LDS_MAX;
    const unsigned item_base = (unsigned)gw * gout.seg_item, new_base = (unsigned)gw * gout.seg_new;
    GAS int *const dirty_seg = V.dirtyl + (p ? V.dirty_par : 0u) + (size_t)new_base;
    int *wpfx = sh.wpfx[wid];
    v4i *qtok = sh.qtok[wid], *qinfo = sh.qinfo[wid];
    int2 *qrow = sh.qrow[wid];
    int q_n = 0;                                                       // closure items waiting in this wave's queue
    int c_arcs = 0, c_paths = 0, c_pend = 0, c_new = 0, c_xitems = 0, c_walk = 0, c_clos = 0;
    int c_ref = 0;                                                     // Path objects the reference creates for this wave's exit tokens
    unsigned mo = 0u;
    // states whose arrival key became non-zero: zeroed by the phase A of the frame after the next one
    auto list_dirty = [&](bool first, int state) __attribute__((always_inline)) {
        const unsigned long long bf = __ballot(first);
        if (bf) {
            const int nf = __popcll(bf);
            if (out.dirty_cnt + nf > (int)gout.seg_new) { if (lane == 0) CS(&c.err[p], (int)JDE_NEW); }
            else {
                if (first) CS(dirty_seg + (unsigned)(out.dirty_cnt + rank_in(bf)), state);
                out.dirty_cnt += nf;
            }
        }
    };
#pragma nounroll
    for (;;) {
        // ---- a batch of up to 64 items: the wave's own closure queue first, else the next chunk
        XFINE_START();
        XFINE_COUNT(5);
        bool valid, exit_kind;
        unsigned ii;
        Tok t;
        v4i info;
        int slice_no = 0;                                              // > 0: this item is a slice of a state with many arcs
        const bool from_q = q_n > 0;                                   // (wave-uniform)
        int2 row_q = make_int2(0, 0);
        if (from_q) {
            valid = lane < q_n;
            exit_kind = false;
            t = as_tok(qtok[lane & (QCAP - 1)]);
            info = qinfo[lane & (QCAP - 1)];
            row_q = qrow[lane & (QCAP - 1)];
            ii = (unsigned)info.w;                                     // (the queue keeps the item's index here)
            info.w = 0;
            q_n = 0;
        } else {
            const int u = grab_chunk(sh, jw, Cw);
            if (u >= Q) break;
            const int w = find_seg(sh.pfx[0], gin.nw, u);
            const int ci = u - RFL(sh.pfx[0][w]);
            valid = lane < KX && ci * KX + lane < RFL(sh.cnt[0][w]);
            ii = (unsigned)w * gin.seg_item + (unsigned)(RFL(sh.start[w]) + ci * KX + lane);
            const unsigned ioff = valid ? icur + ii * 32u : OOB_OFF;
            t = as_tok(ld16(V.items, ioff));
            info = ld16(V.items, ioff + 16u);
            exit_kind = round == 0;
            if (!exit_kind && (info.w & 3) == 1) valid = false;        // expanded by its producer / superseded
            if (!exit_kind && (info.w & 3) == 2) slice_no = info.w >> 2;
        }
        XFINE(0);                                                      // hop 1: the items
        const unsigned ioff = valid ? icur + ii * 32u : OOB_OFF;
        const bool start_tok = valid && exit_kind && info.x < 0;       // recognitionStart's token: it has traversed no arc
        JD_COUNT(c_xitems += __popcll(__ballot(valid)));
        const bool real = valid && !start_tok && slice_no == 0;        // an item that traversed an arc (a slice has been through all this)
        const int state = !valid ? 0 : start_tok ? C.init_state : info.z;
        // (the state's static record, XState: requested here, used when the item is known to go on; graphs of long rows - C.xcut
        // off - ask for state 0's every time: no branch around the loads, one cached line)
        const bool xcut = !LZY && C.xcut != 0;
        int4 x0 = make_int4(0, 0, 0, 0), x1 = x0, x2 = x0, x3 = x0;
        if (!LZY) { const int4 *xq = (const int4 *)(C.xst + (xcut ? state : 0)); x0 = xq[0]; x1 = xq[1]; x2 = xq[2]; x3 = xq[3]; }
        // second level, in flight together: the state's keys, its CSR row, the word label
        // of an exit token's arc, the Path reservation.  A closure item this wave queued for itself brought its row
        // along - the record was read when it arrived - and has just been found the best arrival at its state: it
        // needs no load at all.
        bool have = valid;
        if (real && exit_kind && !init) {                              // :952-962
            have = t.score > ((info.y != 0) ? wordTh : endTh);
            if (have) ++c_pend;
        }
        // the reference's own Path count (collectPaths' trigger, :360-362): propagateToken makes one for the arc's label
        // and one for every labelled epsilon / tee arc of the closure behind it, for EVERY token it is called with -
        // recombination happens at the entry states only (:560) - where this build expands a state's best arrival alone
        if (C.pcount != nullptr && ((real && exit_kind && have) || start_tok))
            c_ref += (start_tok ? 0 : (info.y != 0 ? 1 : 0)) + C.pcount[state];
        // Path records (:497-509) are reserved for every labelled item that passed its threshold, winner or not, so that
        // the reservation is in flight together with the loads below: it is issued BEHIND them (the compiler waits
        // for a returning atomic where it stands, and that wait then is the wait for the loads as well)
        const bool labelled = real && have && info.y != 0;
        const unsigned long long blab = __ballot(labelled);
        int pbase = 0;
        auto reserve = [&]() __attribute__((always_inline)) {
            if (blab) {
                const int first = __ffsll((long long)blab) - 1;
                if (lane == first) pbase = GADD(&c.n_paths, __popcll(blab));
                pbase = __shfl(pbase, first);
            }
        };
        int rs, rs1;
        float fin_lazy = 0.0f;
        unsigned long long kv = 0ULL;
        int label = exit_kind ? 0 : info.y;
        const bool carried = !LZY && from_q;                           // (wave-uniform)
        const bool sole = exit_kind && (info.w & ITEM_SOLE) != 0;      // (REC_SOLE: the only arc into its state - it placed no bid)
        if (carried) { rs = row_q.x; rs1 = row_q.x + row_q.y; reserve(); }
        else {
            // (no load sits in a branch of its own: a load inside a branch is waited for at the branch's end, and these
            // would be three round trips one after the other instead of one)
            const unsigned soff = (real || (valid && !LZY)) ? SREC_BID_OFF(C, state) : OOB_OFF;
            const v4i sk = ld16(V.srec_r, (real && exit_kind && !sole) ? soff : OOB_OFF);   // {key0, keyL}
            int2 srow = make_int2(0, 0);
            // (static, shared by the streams: cached loads)
            if (!LZY) { const int sti = valid ? state : 0; srow = make_int2(C.row_ptr[sti], C.row_ptr[sti + 1]); }
            const bool lab_on = exit_kind && real && info.y != 0;      // (labelled exit tokens: a few per cent of the items)
            int lb;
            if (LZY) lb = ld16(V.larcs, lab_on ? (unsigned)info.x * 16u : OOB_OFF).w;
            else lb = C.arcs[lab_on ? info.x : 0].out;
            v4i lr = {0, 0, 0, 0};
            if (LZY) lr = ld16(V.lrows, (unsigned)state * 16u);        // {first arc, arcs, status, final weight}: ready by the invariant
            reserve();
            label = lab_on ? lb : label;
            if (LZY) {
                rs = lr.x; rs1 = lr.x + lr.y; fin_lazy = __int_as_float(lr.w);
                if (valid && lr.z < LZ_EXPANDED) CS(&c.err[p], (int)JDE_LAZY_INV);   // (cannot happen: the invariant of jd_lazy.h)
            } else { rs = srow.x; rs1 = srow.y; }
            kv = ((unsigned long long)(unsigned)(info.y != 0 ? sk.w : sk.y) << 32) | (unsigned)(info.y != 0 ? sk.z : sk.x);
        }
        XFINE(1);                                                      // hop 2: state record, label, Path reservation
        if (real) {
            // (a closure item was the best arrival at its state when it was produced - else it was never listed for a
            // later round - and that makes it responsible for the arcs its score was the first to make hopeful, see
            // above: it is expanded even if a better arrival has come since)
            const bool winner = !exit_kind || sole || ((unsigned)(kv & 0xffffffffULL) == ii && kv != 0ULL);
            // every state that received exit-token bids is cleaned up by its winner, expanded or not (an
            // item below its threshold still holds the key of its state if it was the best one there)
            if (winner && exit_kind && !sole) CS(info.y != 0 ? &SREC_BID(V.srec, C, state).keyL : &SREC_BID(V.srec, C, state).key0, 0ULL);
            have = have && winner;
        }
        if (have && real) {
            if (info.y != 0) {
                const int pp = pbase + rank_in(blab);
                if (pp < C.cap_paths) {
                    // PathRec {prev, frame, label, -; score, ac, lm, -}: two plain 16-byte stores (read by later launches only)
                    V.paths[2 * (size_t)pp] = (v4i){t.path, pframe, label, 0};
                    V.paths[2 * (size_t)pp + 1] = (v4i){__float_as_int(t.score), __float_as_int(t.ac), __float_as_int(t.lm), 0};
                    t.path = pp;
                    st16(V.items, ioff, as_v4(t));                     // the tokens pulled from this item carry the new history
                    ++c_paths;
                } else CS(&c.err[p], (int)JDE_PATHS);
            }
            // :513-520 final state.  bestFinalToken is reset every frame (:316) and only read by
            // finish(), so it only has to be evaluated on the last frame that is available.
            if (last_frame) {
                const float fw = LZY ? fin_lazy : C.fin_w[info.z];
                if (fw < INF) {
                    const float cs = t.score + fw;
                    if (cs > LZ) GMAX(&c.final_key, ((unsigned long long)f2o(cs) << 32) | ii);
                }
            }
        }
        // ---- arrival of an exit token (and of the start token) at its state: the atomic's old value is the best
        // arrival before it.  (Closure items arrived when they were produced and carry that value; a slice carries
        // its item's.)  The answer is first needed by the arc passes: its round trip runs beside the first arcs'.
        unsigned eo = exit_kind ? 0u : (unsigned)info.x;               // ordered score of the best arrival before this one (0: none)
        unsigned long long eold = 0ULL;
        const bool arrive = have && exit_kind;                         // (the atomic itself: behind the first arcs' loads, below)
        XFINE(2);                                                      // winners: key reset, Path record, final state
        // ---- The prefix walk of jd_slot.h's phase X (see there): a row of up to 57 arcs has its model arcs in descending order of
        // w + tmax behind the arcs every arrival walks; the item walks the prefix it can enter, by the samples of the state's XState,
        // and accounts for the rest from that record and the row's instance flags.
        int x_new = 0;
        const bool xitem = xcut && have && slice_no == 0 && rs1 - (rs & ~7) <= 64;
        if (xitem) {
            const int n_entry = x0.y, n_model = x0.w;
            if (n_model > 0) { const unsigned sw = f2o(t.score + __int_as_float(x0.z)); mo = sw > mo ? sw : mo; }
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
                unsigned long long w8[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w8[i] = (n_model > 0 && a8 + 8 * i < rs1) ? CL(lw + i) : 0ULL;
#pragma unroll
                for (int i = 0; i < 4; ++i) lv_row += __popcll(w8[i] & in_row(a8 + 8 * i));
            }
            if (__ballot(n_model > 0 && a8 + 32 < rs1)) {               // (some lane's row goes on: the second batch)
                unsigned long long w8[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w8[i] = (n_model > 0 && a8 + 8 * (4 + i) < rs1) ? CL(lw + 4 + i) : 0ULL;
#pragma unroll
                for (int i = 0; i < 4; ++i) lv_row += __popcll(w8[i] & in_row(a8 + 8 * (4 + i)));
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
        // ---- A state with thousands of out-arcs (a history with 10^4 successors) would keep this wave busy
        // for hundreds of passes while the cluster waits at the barrier: the wave walks the first X_SLICE
        // arcs itself and hands the rest on as SLICES - items of the next round (flag 2 + slice number)
        // that carry the token as it stands now and skip everything above; the cluster shares them out.
        int alo = rs, ahi = rs1;
        if (slice_no > 0) { alo = rs + slice_no * X_SLICE; ahi = min(rs1, alo + X_SLICE); }
        int n_slices = 0;
        if (have && slice_no == 0 && rs1 - rs > X_SLICE) { n_slices = (rs1 - rs - 1) / X_SLICE; ahi = rs + X_SLICE; }
        // ---- pooled arc walk: exclusive prefix of the items' out-degrees
        const int deg = have ? ahi - alo : 0;
        int incl = deg;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const int tot = __shfl(incl, 63);
        wpfx[lane] = incl - deg;                                       // wave-private: a wave's LDS operations are ordered
        // owner of pooled arc a = largest g with wpfx[g] <= a; its arc record is fetched one pass ahead
        auto owner_of = [&](int a) __attribute__((always_inline)) { int g = 0;
#pragma unroll
            for (int stp = 32; stp > 0; stp >>= 1) if (wpfx[g + stp] <= a) g += stp;
            return g; };
        int g_nx = owner_of(lane);
        int b_nx = __shfl(alo, g_nx) + (lane - wpfx[g_nx]);
        JdArc Bk_nx = {0, 0.0f, 0, 0};
        auto arc_at = [&](int b) __attribute__((always_inline)) -> JdArc {
            if (LZY) { const v4i r = ld16(V.larcs, (unsigned)b * 16u); return JdArc{r.x, __int_as_float(r.y), r.z, r.w}; }
            return C.arcs[b];
        };
        // ... and so is its "has an instance" flag (a byte per arc: the arcs of a state share a sector).  These loads are
        // UNCONDITIONAL (lanes without an arc read arc 0): a load inside a branch reaches the loop-carried registers
        // through a copy at the join, and the compiler waits for it right there - the "pass ahead" was a pass behind.
        int lv_nx = 0;
        { const int bq = lane < tot ? b_nx : 0; Bk_nx = arc_at(bq); lv_nx = CL(V.live + bq); }
        // the arrival (see above), issued behind the first arcs' loads: the compiler waits for a returning atomic where it
        // stands, so this way the two round trips are one
        if (arrive) {
            const unsigned long long akey = ((unsigned long long)f2o(t.score) << 32) | ii;
            if (sole) CS(&SREC_E(V.srec, C, state, p), akey);          // (REC_SOLE: the frame's only arrival at the state - a store, nothing to read back: +1-2 % on
                                                                       // the heavy graphs; the slot kernel keeps the atomic - the branch cost its headline 1.5 %)
            else { eold = GMAX(&SREC_E(V.srec, C, state, p), akey); eo = (unsigned)(eold >> 32); }
        }
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
                    if (out.item_cnt + nj > (int)gout.seg_item) { if (lane == 0) CS(&c.err[p], (int)JDE_ITEMS); break; }
                    if (lane < nj) {
                        const unsigned k = item_base + (unsigned)(out.item_cnt + lane);
                        st16(V.items, icur + k * 32u, tv);
                        st16(V.items, icur + k * 32u + 16u, (v4i){sx, sy, sz, 2 | ((j0 + lane + 1) << 2)});
                    }
                    out.item_cnt += nj; deferred += nj;
                }
            }
        }

        XFINE(3);                                                      // prefix + hop 3: the first 64 arcs (+ the arrival's answer)
#pragma nounroll
        for (int a0 = 0; a0 < tot; a0 += 64) {
            XFINE_COUNT(6);
            const int a = a0 + lane;
            const int g = g_nx, b = b_nx;
            const JdArc Bk = Bk_nx;
            const int lv = lv_nx;
            // the next pass's arcs (LDS look-ups only; the loads are issued BEHIND this pass's own, see below)
            g_nx = owner_of(a + 64);
            const int alo_nx = __shfl(alo, g_nx);                      // (every lane takes part: its owner may be a lane that has no next arc itself)
            b_nx = (a + 64 < tot) ? alo_nx + (a + 64 - wpfx[g_nx]) : 0;
            Tok tg;
            tg.score = __shfl(t.score, g); tg.ac = __shfl(t.ac, g);
            tg.lm = __shfl(t.lm, g); tg.path = __shfl(t.path, g);
            const unsigned eog = (unsigned)__shfl((int)eo, g);         // best arrival at the owner's state before it (ordered; 0: none)
            const int sgx = __shfl(state | (xitem ? (int)0x80000000 : 0), g);   // (+ the owner's "counted per state" flag)
            const int sg = sgx & 0x7fffffff;
            bool mk = false, touch = false;
            Tok un = null_tok();
            // Everything a pass READS is requested before anything is waited for - one memory round trip: the
            // model's constant of an entry arc, and the state record of the destination of every arc that can
            // produce a closure item (its arrival key as a pre-filter: hot history states receive many arrivals,
            // and an atomic on a contended key costs far more than this load; its row for the item to carry).
            const bool on = a < tot;
            const int inl = Bk.in & ~ARC_FLAGS;
            const bool entry = on && inl != 0;
            const bool is_tee = entry && (Bk.in & TEE_FLAG) != 0;
            const float ns = tg.score + Bk.w;                          // (:535 / :562: the same sum either way)
            const unsigned so = f2o(ns);
            unsigned long long skc = 0ULL;
            const float tmax = C.hmm_tmax0[entry ? inl - 1 : 0];       // (used for entry arcs without an instance; unconditional, see above)
            int2 nrow = make_int2(0, 0);
            {
                const unsigned doff = ((on && inl == 0) || is_tee) ? SREC_E_OFF(C, Bk.to, p) : OOB_OFF;
                const unsigned long long se = ld8(V.srec_r, doff);
                if (!LZY) { const int ti = doff != OOB_OFF ? Bk.to : 0; const int r0 = C.row_ptr[ti]; nrow = make_int2(r0, C.row_ptr[ti + 1] - r0); }
                // the next pass's arc records and flags: in flight during this pass, and - issued behind the loads this
                // pass waits for (loads return in order) - not waited for before the next one
                Bk_nx = arc_at(b_nx); lv_nx = CL(V.live + b_nx);
                skc = se;
            }
            if (on) ++c_arcs;
            JD_COUNT(c_walk += __popcll(__ballot(on)));
            if (on && inl == 0) {                                      // :533-540 epsilon input
                un = tg;
                un.score = ns;
                un.lm = tg.lm + Bk.w;
                mk = un.score > endTh;
            } else if (is_tee) {                                       // :584-600 tee model
                // (an atomic load: never merged with the LDS one into a flat load)
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
                    if (sgx >= 0 && eog == 0u) ++c_new;                // (counted once, by the first arrival at the state; prefix walks: per state, above)
                    if (can_filter) {
                        const bool mine = (ns + tmax) - bestA > -C.emit_win;
                        const bool before = eog != 0u && ((o2f(eog) + Bk.w) + tmax) - bestA > -C.emit_win;
                        touch = mine && !before;                       // the first arrival whose candidate may survive
                    } else touch = eog == 0u;
                }
            }
            // arcs to be tried in the next phase A -> this wave's segment of the new list
            const unsigned long long bt = __ballot(touch);
            if (bt) {
                const int nt = __popcll(bt);
                if (out.new_cnt + nt > (int)gout.seg_new) { if (lane == 0) CS(&c.err[p], (int)JDE_NEW); }
                else {
                    if (touch) CS(V.newl + (size_t)new_base + (unsigned)(out.new_cnt + rank_in(bt)),
                                  ((unsigned long long)(unsigned)sg << 32) | (unsigned)b);
                    out.new_cnt += nt;
                }
            }
            // closure items: the best arrival at its state so far is kept (running maximum), written to
            // this wave's item segment and - if the wave's queue has room - expanded by this wave itself
            if (__ballot(mk)) {
                const unsigned sou = f2o(un.score);
                const bool pass = mk && sou > (unsigned)(skc >> 32);   // cheap pre-filter before an index is spent
                const unsigned long long bp = __ballot(pass);
                const int np = __popcll(bp);
                JD_COUNT(c_clos += np);
                if (out.item_cnt + np > (int)gout.seg_item) { if (lane == 0) CS(&c.err[p], (int)JDE_ITEMS); }
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
                    const bool inq = keep && rank_in(bk) < room;       // expanded by this wave, right after this batch
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
        XFINE(4);                                                      // the arc passes of this batch
    }
    mo = wave_umax(mo);
    c_arcs = wave_sum(c_arcs); c_paths = wave_sum(c_paths); c_pend = wave_sum(c_pend); c_new = wave_sum(c_new);
    if (C.pcount != nullptr) {
        c_ref = wave_sum(c_ref);
        if (lane == 0 && c_ref) (void)GADD(&c.n_paths_ref, c_ref);
    }
    if (lane == 0) {
        if (mo) atomicMax(&sh.best, mo);
        if (c_arcs) atomicAdd(&sh.stat[ST_ARCS], c_arcs);
        if (c_paths) atomicAdd(&sh.stat[ST_PATHS], c_paths);
        if (c_pend) atomicAdd(&sh.stat[ST_PEND], c_pend);
        if (c_xitems) atomicAdd(&sh.stat[ST_XITEMS], c_xitems);
        if (c_walk) atomicAdd(&sh.stat[ST_WALK], c_walk);
        if (c_clos) atomicAdd(&sh.stat[ST_CLOS], c_clos);
        if (c_new) { atomicAdd(&sh.stat[ST_MODELS], c_new); atomicAdd(&sh.new_all, c_new); }   // attached instances are active models (:981)
    }
}

// ------------------------------------------------------------------ one stream, one launch

// nbar_io (or null): the cluster's barrier count so far, when the stream's barrier word is NOT zeroed between calls (the
// resident kernel, jd_resident.h); updated on every normal return.
template <int NE, bool XL, bool LZY>
__device__ __forceinline__ void run_stream(const SearchArgs &A, SearchShared &sh, int s, int ll_slot, int jw, int Cw, bool prio,
                                           unsigned *nbar_io = nullptr)
{
    constexpr bool XL_ = XL;
    typedef RecLayout<NE> RL;
    const DecConst &C = A.C;
    StreamCtl &c = A.ctl[s];
    const StreamDev &S = A.streams[s];
    const int tid = threadIdx.x, lane = tid & 63, wid = RFL(tid >> 6);     // wave-uniform values live in SGPRs
    const int NW = Cw * SW, gw = jw * SW + wid;
    const int MN = C.max_n;
    // ---- launch-constant state (line 0 of the control block is not written while the launch runs)
    int f = RFL(c.frame);
    const int T = RFL(c.T);
    const bool needs_init = RFL(c.needs_init) != 0;
    if (!c.started || c.error != 0) return;
    const int f_stop = T < A.f_end ? T : A.f_end;
    if (!needs_init && f >= f_stop) return;
    float best_emit = __int_as_float(RFL(__float_as_int(c.best_emit)));
    const int old_nw = RFL(c.lst_nw);
    const bool ref_rule = C.pcount != nullptr;                          // collectPaths' count rule on the reference's counts
    const int path_new = needs_init ? 0 : RFL(ref_rule ? c.path_new_ref : c.path_new);
    Geo gin = make_geo(C, old_nw > 0 ? old_nw : NW);
    const Geo gout = make_geo(C, NW);
    // geometry of the two dirty lists (each lives two frames, so it may come from the launch before the previous one)
    const int dn0 = RFL(c.dirty_nw[0]), dn1 = RFL(c.dirty_nw[1]);
    Geo gd0 = make_geo(C, dn0 > 0 ? dn0 : NW), gd1 = make_geo(C, dn1 > 0 ? dn1 : NW);
    StreamView V;
    V.rec = mk_rsrc(S.rec, 2ULL * C.cap_slots * RL::REC_BYTES);
    V.items = mk_rsrc(S.items, 2ULL * C.cap_items * 32u);
    V.rec_par = C.cap_slots * (unsigned)RL::REC_BYTES; V.item_par = C.cap_items * 32u;
    V.live = (GAS unsigned char *)S.live; V.srec = (GAS StateRec *)S.srec;
    V.srec_r = mk_rsrc(S.srec, (unsigned long long)C.n_states * sizeof(StateRec));
    V.newl = (GAS unsigned long long *)S.newl; V.dirtyl = (GAS int *)S.dirtyl; V.dirty_par = C.cap_new;
    V.tot = (GAS int *)S.tot; V.paths = (GAS v4i *)S.paths; V.hist = (GAS int *)S.hist;
    if (LZY) {
        V.lrows = mk_rsrc(C.lazy->rows, (unsigned long long)C.lazy->max_states * 16ULL);
        V.larcs = mk_rsrc(C.lazy->arcs, (unsigned long long)C.lazy->max_arcs * 16ULL);
    }
    const bool use_hist = C.max_hyps > 0;
    const bool lr = C.lrt != nullptr;                                  // plain left-to-right topologies (compact table in LDS)
    const bool trp_lds = !lr && (size_t)C.n_tm * MN * MN <= TRP_LDS_MAX && (size_t)C.n_tm * MN <= TRP_LDS_MAX / 4;
    __syncthreads();                                                   // the previous stream of this slot is done with LDS
    if (lr) for (int i = tid; i < C.n_tm * ((NE == 3) ? 8 : 16); i += SNT) sh.trP[i] = C.lrt[i];
    if (C.n_hmm <= TEE_LDS_MAX) for (int i = tid; i < C.n_hmm; i += SNT) sh.tee[i] = C.hmm_tee[i];
    if (trp_lds) {
        for (int i = tid; i < C.n_tm * MN * MN; i += SNT) sh.trP[i] = C.trP[i];
        for (int i = tid; i < C.n_tm * MN; i += SNT) sh.se[i] = C.se32[i];
    }
    if (tid == 0) {
        sh.abort = 0; sh.best = 0u; sh.new_all = 0; sh.next = 0;
        for (int k = 0; k < ST_N; ++k) { sh.stat[k] = 0; sh.acc[k] = 0; }
        for (int k = 0; k < 8; ++k) sh.clk[k] = 0;
#ifdef JD_FINE
        for (int k = 0; k < 7; ++k) sh.fclk[k] = 0;
#endif
    }
    if (use_hist) for (int b = tid; b < C.hist_nbins; b += SNT) sh.hist[b] = 0;
    __syncthreads();
    unsigned nbar = nbar_io ? *nbar_io : 0u;
    const long long t_limit = wall_clock64() + 3000000000LL;           // 30 s at 100 MHz: a lost workgroup, not a slow one
    if (XL && Cw > 1) {
        // XCD-local launch: nothing promises where workgroups run, so the cluster first makes sure it does
        // sit on one XCD - every workgroup ORs its XCC id into a mask and arrives at a counter (agent-scope
        // read-modify-writes: performed at the memory side, whatever the placement), then reads the mask
        // the same way.  A cluster that is spread out leaves the stream untouched; the host repeats the
        // launch with the agent-scope kernel.
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;   // HW_REG_XCC_ID[3:0]
            (void)__hip_atomic_fetch_or(&c.xmask, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            (void)__hip_atomic_fetch_add(&c.xbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_fetch_add(&c.xbar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)Cw) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0 && wall_clock64() > t_limit) { sh.abort = 1; break; }
            }
            const unsigned mask = __hip_atomic_fetch_or(&c.xmask, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__popc(mask) != 1) sh.abort = 2;
        }
        __syncthreads();
        if (sh.abort == 2) {                                           // every workgroup of the cluster sees the same mask
            if (jw == 0 && tid == 0) atomicAdd(A.status + 1, 1);
            return;
        }
        if (sh.abort) { if (tid == 0 && jw == 0) { c.error = JDE_BARRIER; c.needs_init = 0; } return; }
    }
    int frames_done = 0;
    int my_item_end = 0;                                               // items this wave wrote in the last processed frame
    bool aborted = false, failed = false;
    bool init_pending = needs_init;
    auto tot_of = [&](int k) __attribute__((always_inline)) { return V.tot + (size_t)k * MAXW; };

    // =============================================================== recognitionStart (:139-228), part 1
    if (needs_init) {
        // drop whatever the previous utterance left behind: instance flags, arrival keys (both parities)
        const int p0 = f & 1;
        const ListSrc src[3] = {{tot_of(TOT_REC0 + p0), 64, gin.seg_rec, gin.nw}, {tot_of(TOT_DIRTY0), 64, gd0.seg_new, gd0.nw},
                                {tot_of(TOT_DIRTY1), 64, gd1.seg_new, gd1.nw}};
        int Q[3], items[3];
        build_lists<3>(sh, src, Q, items);
        const int Qall = Q[0] + Q[1] + Q[2];
        for (int u = gw; u < Qall; u += NW) {
            const int kind = (u >= Q[0] + Q[1]) ? 2 : (u >= Q[0]) ? 1 : 0;
            const int ru = u - (kind == 2 ? Q[0] + Q[1] : kind == 1 ? Q[0] : 0);
            const Geo &gk = kind == 0 ? gin : kind == 1 ? gd0 : gd1;
            const int w = find_seg(sh.pfx[kind], gk.nw, ru);
            const int ci = ru - RFL(sh.pfx[kind][w]);
            if (ci * 64 + lane < RFL(sh.cnt[kind][w])) {
                if (kind == 0) {
                    const unsigned roff = (p0 ? V.rec_par : 0u) + rec_chunk_off<NE>(gin.seg_rec, w, ci) + (unsigned)lane * 16u;
                    CS(&V.live[ld16(V.rec, roff).x], (unsigned char)0);
                }
                else {
                    const int b = CL(V.dirtyl + (kind == 2 ? V.dirty_par : 0u) + (size_t)w * gk.seg_new + (unsigned)(ci * 64 + lane));
                    CS(&SREC_E(V.srec, C, b, 0), 0ULL); CS(&SREC_E(V.srec, C, b, 1), 0ULL);
                }
            }
        }
        if (use_hist) for (int b = jw * SNT + tid; b < 2 * HIST_MAX_BINS; b += Cw * SNT) CS(V.hist + b, 0);
        if (jw == 0 && tid == 0) {
            CS(&c.bestA[0], 0u); CS(&c.bestA[1], 0u); CS(&c.bestX[0], 0u); CS(&c.bestX[1], 0u);
            CS(&c.new_all[0], 0); CS(&c.new_all[1], 0);
            CS(&c.n_paths, 0); CS(&c.n_paths_ref, 0); CS(&c.final_key, 0ULL); CS(&c.err[0], 0); CS(&c.err[1], 0);
            c.path_new = 0; c.path_new_ref = 0; c.n_collect = 0;
            for (int k = 0; k < ST_N; ++k) CS(&c.st[k], 0LL);
            c.best_final = null_tok();
            // the start token (:221-226) is the only item of round 0, in wave 0's segment (parity 1)
            Tok z; z.score = 0.0f; z.ac = 0.0f; z.lm = 0.0f; z.path = -1;
            st16(V.items, V.item_par, as_v4(z)); st16(V.items, V.item_par + 16u, (v4i){-1, 0, 0, 0});
        }
        cluster_barrier<XL>(sh, c, Cw, nbar, t_limit);
        aborted = sh.abort != 0;
        // every workgroup has read the old lists (possibly written with another geometry): they are
        // gone, this launch's geometry applies and every segment starts empty - except wave 0's
        // "exit" segment, which holds the start token.  (Visible to the others after the next barrier.)
        gin = gout; gd0 = gout; gd1 = gout;
        if (lane == 0) {
            CS(tot_of(TOT_REC0) + gw, 0); CS(tot_of(TOT_REC1) + gw, 0);
            CS(tot_of(TOT_DIRTY0) + gw, 0); CS(tot_of(TOT_DIRTY1) + gw, 0);
            CS(tot_of(TOT_EXIT) + gw, gw == 0 ? 1 : 0);
        }
        cluster_barrier<XL>(sh, c, Cw, nbar, t_limit);
        aborted = aborted || sh.abort != 0;
        f = 0;
    }

    // =============================================================== frames (the first pass may be
    // recognitionStart part 2: the expansion of the start token, a frame without phase A that
    // uses item / key parity 1 like a frame "-1")
    int pre_cnt[3] = {0, 0, 0}, pre_new = 0;                           // next frame's list counts, requested one frame ahead (below)
    bool pre_ok = false;
    int np_seen = 0;                                                   // Path records in use, as of the last frame end
    int npr_seen = 0;                                                  // ... and the reference's count of them (ref_rule)
    int stop_seen = 0;                                                 // the host wants to re-plan the launch (SearchArgs::rebalance_at): 1;
                                                                       // the batch this stream runs ahead of is through: 2
    while (!aborted && !failed) {
        const bool init = init_pending;
        if (!init && f >= f_stop) break;
        const int p = init ? 1 : (f & 1);
        // stop early when the Path arena needs collecting (k_gc_* run between launches); n_paths only
        // changes in phase X, so every workgroup of the cluster reads the same value here
        if (!init && frames_done > 0 &&
            (np_seen > C.gc_threshold || stop_seen || (C.path_rule && path_rule_fires(ref_rule ? npr_seen : np_seen, path_new)))) break;
        long long t0 = 0;
        const bool clk_on = A.dbg != nullptr && tid == 0;
#define CLK(slot) do { if (clk_on) { const long long tn_ = wall_clock64(); sh.clk[slot] += tn_ - t0; t0 = tn_; } } while (0)
        if (clk_on) t0 = wall_clock64();
        int exit_cnt = (init && gw == 0) ? 1 : 0;
        unsigned ba = 0u;
        if (!init) {
            // ---- frame start (:311-339): thresholds + the work lists of phase A
            const float normalise = (best_emit > LZ) ? best_emit : 0.0f;             // :321
            float emitTh = (C.emit_win > 0.0f ? -C.emit_win : LZ);                   // :331
            if (use_hist)                                              // bins of the previous frame (parity p^1)
                for (int b = tid; b < C.hist_nbins; b += SNT) sh.hprev[b] = CL(V.hist + (size_t)(p ^ 1) * HIST_MAX_BINS + b);
            const ListSrc src[3] = {{tot_of(TOT_REC0 + p), 64, gin.seg_rec, gin.nw}, {tot_of(TOT_NEW), 64, gin.seg_new, gin.nw},
                                    {tot_of(TOT_DIRTY0 + p), 64, (p ? gd1 : gd0).seg_new, (p ? gd1 : gd0).nw}};
            int Q[3], items[3], nseg[3];
            // the lists' fill counts and the arcs entered in the previous frame: requested behind the previous frame's last
            // barrier, together with what that frame read there anyway (pre_ok) - else now
            if (!pre_ok) {
                if (jw == 0 && tid == 0) pre_new = CL(&c.new_all[p ^ 1]);
                packed_counts<3>(src, pre_cnt);
            }
            const int new_prev = pre_new;
            pre_ok = false;
            build_packed<3>(sh, src, pre_cnt, 1, Q, items, nseg);
            if (use_hist) {                                            // every workgroup evaluates the same threshold
                float th = hist_threshold(C, sh.hprev, lane);
                th -= normalise;                                                     // :325
                if (C.emit_win > 0.0f && th < -C.emit_win) th = -C.emit_win;         // :326-327
                emitTh = th;
            }
            const float startTh = (C.start_win > 0.0f) ? (best_emit - C.start_win) : LZ;   // :337
            // arcs entered in the previous frame are instances of this one, tried or not (:899-935)
            if (jw == 0 && tid == 0) atomicAdd(&sh.stat[ST_INSTS], new_prev);
            // (signed: streams that sit at different frames share one table - jd_streams_push - and a stream's "slot"
            // is then its first row minus its first frame)
            const float *llrow = A.ll + ((long long)ll_slot * A.ll_stride + (long long)(f - A.f0) * (long long)C.G);
            int out_cnt = 0;
            CLK(0);                                                    // thresholds + work lists
            const long long cell0 = (long long)(llrow - A.ll);       // (diagnostics: the row's first cell in the slab)
#define PHASE_A_ARGS C, sh, c, V, gin, (p ? gd1 : gd0), gout, Q, items, nseg, jw, Cw, gw, p, normalise, emitTh, startTh, llrow, A.cells, cell0, \
                     out_cnt, exit_cnt
            if (lr) phase_a<NE, true, true, XL, LZY>(PHASE_A_ARGS);
            else if (trp_lds) phase_a<NE, true, false, XL, LZY>(PHASE_A_ARGS);
            else phase_a<NE, false, false, XL, LZY>(PHASE_A_ARGS);
#undef PHASE_A_ARGS
            CLK(1);                                                    // phase A (wave 0's share)
            if (lane == 0) {
                CS(tot_of(TOT_REC0 + (p ^ 1)) + gw, out_cnt);
                CS(tot_of(TOT_EXIT) + gw, exit_cnt);
            }
            __syncthreads();
            if (use_hist)                                              // Histogram of this frame: workgroup bins -> stream bins
                for (int b = tid; b < C.hist_nbins; b += SNT) {
                    const int v = sh.hist[b];
                    if (v) { GADD(V.hist + (size_t)p * HIST_MAX_BINS + b, v); sh.hist[b] = 0; }
                }
            if (tid == 0 && sh.best) { GMAX(&c.bestA[p], sh.best); sh.best = 0u; }
            CLK(2);                                                    // waiting for the workgroup's other waves + publishing
            cluster_barrier<XL>(sh, c, Cw, nbar, t_limit);
            if (sh.abort) { aborted = true; break; }
            gin = gout;                                                // every list read from here on was written by this launch
            CLK(3);                                                    // cluster barrier 1
        }
        // ---- phase X.  (Words that every workgroup reads after a barrier - the frame's best scores, the
        // error flag, the Path count - are requested together with the next work list's counts: one
        // round trip, not one each.)
        float bestA = LZ, endTh = LZ, wordTh = LZ;
        unsigned bx_raw = 0u;
        int err_raw = 0, np_raw = 0, npr_raw = 0, stop_raw = 0;
        const bool last_frame = !init && f >= T - 1;
        XOut xo = {exit_cnt, 0, 0};
        for (int round = 0;; ++round) {
            // Items are taken in chunks of KX per wave pass.  The arcs of a chunk are walked by ONE wave,
            // so when there are fewer than 64 items per wave the chunks shrink to spread the arc walk
            // over the cluster (the item stage just leaves lanes idle).
            int KX = 64;
            int Q1[1], n1[1];
            if (round == 0) {
                unsigned ba_raw = 0u;
                if (!init) ba_raw = CL(&c.bestA[p]);
                if (tid < gin.nw) sh.start[tid] = 0;
                const ListSrc src[1] = {{tot_of(TOT_EXIT), KX, gin.seg_item, gin.nw}};
                build_lists<1>(sh, src, Q1, n1);
                if (jw == 0 && !init) {                                // housekeeping for the frame after this one (stores:
                    // after the list's loads, so that nothing waits for their acknowledgement)
                    if (tid == 0) {
                        CS(&c.bestA[p ^ 1], 0u); CS(&c.bestX[p ^ 1], 0u); CS(&c.new_all[p ^ 1], 0);
                        // enough of the grid idles: this cluster stops after this frame (every workgroup of it reads the
                        // request behind this round's barrier, i.e. in the same frame)
                        if (A.rebalance_at > 0 && CL(A.status + 2) >= A.rebalance_at && CL(A.status + 4) == 0) CS(&c.stop_req, 1);
                        // searched ahead of its batch's turn: the batch the caller waits for is through - so is this launch
                        if (!prio && A.n_prio > 0 && CL(A.status + 5) >= A.n_prio) CS(&c.stop_req, 2);
                    }
                    if (use_hist) for (int b = tid; b < C.hist_nbins; b += SNT) CS(V.hist + (size_t)(p ^ 1) * HIST_MAX_BINS + b, 0);
                }
                ba = (unsigned)RFL((int)ba_raw);
                bestA = ba ? o2f(ba) : LZ;
                endTh = (!init && C.end_win > 0.0f) ? (bestA - C.end_win) : LZ;     // :349
                wordTh = (!init && C.word_win > 0.0f) ? (bestA - C.word_win) : LZ;  // :350
            } else {
                bx_raw = CL(&c.bestX[p]); err_raw = CL(&c.err[p]); np_raw = CL(&c.n_paths);   // final if this round has nothing to do
                if (ref_rule) npr_raw = CL(&c.n_paths_ref);                                   // (exit tokens are round 0's)
                stop_raw = CL(&c.stop_req);
                {   // ... and so are the lists the NEXT frame's phase A reads (parity p ^ 1: the frame after this one, or
                    // frame 0 after recognitionStart's pass): their counts ride on this round trip instead of one of their own
                    const int pn = p ^ 1;
                    const ListSrc nsrc[3] = {{tot_of(TOT_REC0 + pn), 64, gin.seg_rec, gin.nw}, {tot_of(TOT_NEW), 64, gin.seg_new, gin.nw},
                                             {tot_of(TOT_DIRTY0 + pn), 64, (pn ? gd1 : gd0).seg_new, (pn ? gd1 : gd0).nw}};
                    pre_new = 0;
                    if (jw == 0 && tid == 0) pre_new = CL(&c.new_all[p]);
                    packed_counts<3>(nsrc, pre_cnt);
                }
                if (tid < gin.nw) sh.start[tid] = CL(tot_of(TOT_CLS0 + (round & 1)) + tid);
                const ListSrc src[1] = {{tot_of(TOT_CL0 + (round & 1)), KX, gin.seg_item, gin.nw}};
                build_lists<1>(sh, src, Q1, n1);
                if (Q1[0] == 0) { pre_ok = JD_PRECOUNT != 0; break; }
            }
            int Q = Q1[0];
            if (Q < NW * C.x_chunks) {
                while (KX > 4 && n1[0] < KX * NW * C.x_chunks) KX >>= 1;   // a few chunks per wave
                if (KX < 64) Q = rebuild_list(sh, 0, gin.nw, KX);
            }
            CLK(4);                                                    // phase X work lists
            const int round_start = xo.item_cnt;
            int deferred = 0;
            phase_x<XL, LZY>(C, sh, c, V, gin, gout, Q, KX, round, jw, Cw, gw, p, init ? 0 : f, init, last_frame, endTh, wordTh, bestA, xo, deferred);
            CLK(5);                                                    // phase X (wave 0's share)
            if (lane == 0) {
                CS(tot_of(TOT_CL0 + ((round + 1) & 1)) + gw, deferred > 0 ? xo.item_cnt - round_start : 0);
                CS(tot_of(TOT_CLS0 + ((round + 1) & 1)) + gw, round_start);
                CS(tot_of(TOT_NEW) + gw, xo.new_cnt);
                CS(tot_of(TOT_DIRTY0 + p) + gw, xo.dirty_cnt);
            }
            __syncthreads();
            if (tid == 0) {
                if (sh.best) { GMAX(&c.bestX[p], sh.best); sh.best = 0u; }
                if (sh.new_all) { GADD(&c.new_all[p], sh.new_all); sh.new_all = 0; }
            }
            CLK(6);                                                    // waiting for the workgroup's other waves + publishing
            cluster_barrier<XL>(sh, c, Cw, nbar, t_limit);
            if (sh.abort) { aborted = true; break; }
            CLK(7);                                                    // cluster barriers of phase X
        }
        if (aborted) break;
        if constexpr (LZY) {
            __shared__ int2 lzq[SW][LZQ];                              // per wave: the stack of lz_close (only in the lazy kernels' LDS)
            // search-driven composition: the arcs this wave entered in this frame lead to states that phase X
            // of a later frame will expand - close them now, epsilon / tee closure included (jd_lazy.h)
            const LazyDev &L = *C.lazy;
            bool ok = true;
            const size_t nbase = (size_t)gw * gout.seg_new;
            for (int i0 = 0; i0 < xo.new_cnt && ok; i0 += 64) {
                int dest = -1;
                if (i0 + lane < xo.new_cnt) {
                    const int b = (int)(unsigned)CL(V.newl + nbase + (unsigned)(i0 + lane));
                    dest = ld16(V.larcs, (unsigned)b * 16u).x;
                    if (lz_status(L, dest) == LZ_CLOSED) dest = -1;
                }
                for (unsigned long long bm = __ballot(dest >= 0); bm && ok; bm &= bm - 1)
                    ok = lz_close(L, C.hmm_tee, __shfl(dest, __ffsll((long long)bm) - 1), lzq[wid], t_limit);
            }
            // (raised behind the frame's last barrier, so into BOTH parities: the next frame's error check reads the other one)
            if ((!ok || lz_failed(L)) && lane == 0) { CS(&c.err[p], (int)JDE_LAZY); CS(&c.err[p ^ 1], (int)JDE_LAZY); }
        }
        my_item_end = xo.item_cnt;
        if (p) gd1 = gout; else gd0 = gout;                                              // (this frame's dirty list: written by this launch)
        // ---- frame end
        {
            const unsigned bx = (unsigned)RFL((int)bx_raw);
            const unsigned bb = ba > bx ? ba : bx;
            best_emit = bb ? o2f(bb) : LZ;                             // :417-418, :572-573
            np_seen = RFL(np_raw);                                     // (n_paths only changes in phase X)
            npr_seen = RFL(npr_raw);
            stop_seen = RFL(stop_raw);
        }
        if (tid == 0)                                                  // totalActiveModels starts with frame 0 (:981)
            for (int k = 0; k < ST_N; ++k) { if (!init || k != ST_MODELS) sh.acc[k] += sh.stat[k]; sh.stat[k] = 0; }
        if (last_frame && jw == 0 && tid == 0) {                       // bestFinalToken of this frame (:513-520)
            const unsigned long long fk = CL(&c.final_key);
            Tok bf = null_tok();
            if (fk != 0ULL) {
                const unsigned fi = (unsigned)(fk & 0xffffffffULL);
                const unsigned ic = p ? V.item_par : 0u;
                const Tok it = as_tok(ld16(V.items, ic + fi * 32u));
                const int fst = ld16(V.items, ic + fi * 32u + 16u).z;
                const float fw = LZY ? __int_as_float(ld16(V.lrows, (unsigned)fst * 16u).w) : C.fin_w[fst];
                bf.score = o2f((unsigned)(fk >> 32)); bf.ac = it.ac; bf.lm = it.lm + fw; bf.path = it.path;
                CS(&c.final_key, 0ULL);
            }
            c.best_final = bf;
        }
        if (RFL(err_raw) != 0) failed = true;                          // raised before this frame's last barrier: seen by all
        if (init) init_pending = false;
        else { ++f; ++frames_done; }
    }

    if (aborted) {
        if (tid == 0 && jw == 0) { c.error = JDE_BARRIER; c.needs_init = 0; }
        return;
    }
    // ---- end of the launch: persist the stream state (read by the next launch / the host kernels)
    if (nbar_io) *nbar_io = nbar;
    if (lane == 0) CS((GAS int *)S.item_end + gw, my_item_end);
    if (tid == 0) {
        for (int k = 0; k < ST_N; ++k) if (sh.acc[k]) atomicAdd((unsigned long long *)&c.st[k], (unsigned long long)sh.acc[k]);
        if (A.dbg) {
            long long *d = A.dbg + (size_t)blockIdx.x * 16;
            for (int k = 0; k < 8; ++k) d[k] += sh.clk[k];
            d[8] += frames_done;
#ifdef JD_FINE
            for (int k = 0; k < 7; ++k) d[9 + k] += sh.fclk[k];
#endif
        }
        if (jw == 0) {
            const int e0 = CL(&c.err[0]), e1 = CL(&c.err[1]);
            c.frame = f; c.best_emit = best_emit; c.lst_nw = NW; c.needs_init = 0;
            c.dirty_nw[0] = gd0.nw; c.dirty_nw[1] = gd1.nw;
            // (a network out of room is the cause; "a token on an unexpanded state" what follows from it)
            if (e0 | e1) c.error = (e0 == (int)JDE_LAZY || e1 == (int)JDE_LAZY) ? (int)JDE_LAZY : (e0 ? e0 : e1);
            else if (f < f_stop) {                                     // stopped early: collect Path records / re-plan, then go on
                atomicAdd(A.status, 1);
                if (stop_seen == 1) atomicAdd(A.status + 3, 1);
                if (stop_seen == 2) atomicAdd(A.status + 6, 1);
            }
            if (prio && ((e0 | e1) || f >= f_stop)) atomicAdd(A.status + 5, 1);   // (one of the streams the launch is there for is through)
            if (A.n_slots == 0) atomicAdd(A.status + 2, Cw);           // this cluster's workgroups have nothing left to do in this launch
        }
    }
}

// Uniform mode: workgroup b belongs to cluster slot q = b / Cw (consecutive blocks, i.e. consecutive
// XCDs, serve one stream); a slot serves its streams one after the other.  Weighted mode (at most
// one stream per workgroup): stream k owns the workgroups [first_k, first_k + n_k) - the host sizes
// the clusters by the streams' recent load.  All workgroups of the grid must be resident at once:
// the host sizes the grid to the device (one 512-thread workgroup per CU).
template <int NE, bool XL, bool LZY>
__global__ JD_KBOUNDS void k_search(SearchArgs A)
{
    __shared__ SearchShared sh;
    int k, kstep, jw, Cw;
    if (A.resident && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(A.resident, A.launch_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // XCD-local launch: workgroup b is expected on XCD b % 8 (observed dispatch order - run_stream checks),
    // so the workgroups are numbered XCD by XCD and the host keeps every cluster inside one eighth of the grid
    const unsigned wg = (XL && !A.xl_selftest) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (A.n_slots > 0) {
        k = (int)(wg / (unsigned)A.Cw); jw = (int)(wg % (unsigned)A.Cw); Cw = A.Cw; kstep = A.n_slots;
    } else {
        int lo = 0, hi = A.n_work - 1;                                 // last k with first_k <= wg
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (RFL(A.work[mid].z) <= (int)wg) lo = mid; else hi = mid - 1; }
        const int first = RFL(A.work[lo].z);
        Cw = RFL(A.work[lo].w) & 0xffff; jw = (int)wg - first;             // (bit 30: one of the streams the launch is there for, n_prio)
        k = (jw < Cw) ? lo : A.n_work; kstep = A.n_work;
    }
    for (; k < A.n_work; k += kstep)
        run_stream<NE, XL, LZY>(A, sh, RFL(A.work[k].x), RFL(A.work[k].y), jw, Cw, A.n_slots == 0 && (RFL(A.work[k].w) & 0x40000000) != 0);
}
