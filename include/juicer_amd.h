/*
 * juicer_amd.h - C ABI of the MI355X-native WFST Viterbi decoder.
 *
 * This is the drop-in boundary for ONE path of idiap/juicer: the
 * WFSTDecoderLite token-passing search + HTKFlatModels diagonal-GMM scoring,
 * behind Juicer's IDecoder / DecoderBatchTest seam.  Every entry point cites
 * the reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every function that can fail returns 0 on success or a negative JD_E*
 *     code; jd_last_error() returns a human readable message for the calling
 *     thread's most recent failure.  (The reference calls Torch3 error() =
 *     print + exit(); a library must not exit, so the same conditions are
 *     reported as codes instead.)
 *   - All "host" pointers are only read during the call; nothing is retained.
 *   - There is NO CPU fallback: if no gfx950 device / HIP runtime is usable
 *     jd_dec_create() fails with JD_ENODEV.
 */
#ifndef JUICER_AMD_H
#define JUICER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JD_OK        0
#define JD_EINVAL   -1   /* bad argument                                        */
#define JD_ENODEV   -2   /* no usable HIP device                                */
#define JD_EHIP     -3   /* HIP runtime call failed                             */
#define JD_ENOMEM   -4   /* a device arena (slots/paths/frontier) overflowed     */
#define JD_EHIST    -5   /* Histogram::addScore - score > maxScore
                            (Histogram.cpp:78-79 is fatal in the reference)     */
#define JD_ESTATE   -6   /* call order violated (e.g. push before init)          */
#define JD_EFORMAT  -7   /* text loader: malformed FSM / MMF input               */

/* LOG_ZERO of the reference: Torch3 defines it as -FLT_MAX (not -inf). */
#define JD_LOG_ZERO (-3.402823466e+38f)

typedef struct jd_net jd_net;   /* replaces Juicer::WFSTNetwork  (WFSTNetwork.h:110-245)   */
typedef struct jd_am  jd_am;    /* replaces Juicer::HTKFlatModels (HTKFlatModels.h)         */
typedef struct jd_dec jd_dec;   /* replaces Juicer::WFSTDecoderLite (WFSTDecoderLite.h:77) */

/* ------------------------------------------------------------------ network */

/*
 * Build a network from arcs in FSM *file order* with the raw file weights,
 * applying the load-time arithmetic of WFSTNetwork::WFSTNetwork(text)
 * (WFSTNetwork.cpp:371-616):
 *     weight = (float)(-w_file * lm_scale); if (out > 0) weight += ins_penalty
 *     final weight = (float)(-w_file * lm_scale)
 *     initial state = source state of the first arc
 * Arcs of one source state must be contiguous (getTransitions(),
 * WFSTNetwork.cpp:709-721, returns "first arc + count"); JD_EFORMAT otherwise.
 * in/out labels are integers; 0 is epsilon; in-label i>0 selects HMM i-1
 * (WFSTDecoderLite.cpp:754).  Auxiliary symbols must already be removed.
 */
int jd_net_create_arcs(jd_net **out, int64_t n_arcs,
                       const int32_t *from, const int32_t *to,
                       const int32_t *in, const int32_t *outl,
                       const float *w_file,
                       int32_t n_final, const int32_t *fstate, const float *fweight_file,
                       float lm_scale, float ins_penalty);

/*
 * Build a network from an already prepared CSR (weights are used as given,
 * i.e. they are the reference's in-memory transitions[].weight values).
 * row_ptr has n_states+1 entries.
 */
int jd_net_create_csr(jd_net **out, int32_t n_states, int32_t init_state,
                      const int32_t *row_ptr, const int32_t *to, const float *w,
                      const int32_t *in, const int32_t *outl,
                      int32_t n_final, const int32_t *fstate, const float *fweight);

/* AT&T text FSM file (WFSTNetwork.cpp:403-560); symbol files are optional and
 * only used for the range checks of WFSTNetwork.cpp:566-584. */
int jd_net_load_fsm(jd_net **out, const char *fsm_path, const char *insyms_path,
                    const char *outsyms_path, float lm_scale, float ins_penalty);

/* Juicer's binary network cache "<fsm>.bin" (JWNT), which juicer.cpp:854-866 prefers over the
 * text FSM when it exists: WFSTNetwork(lmScale, insPenalty) + readBinary (WFSTNetwork.cpp:333-359,
 * 1228-1365).  Arc weights are stored without scale and penalty and get `w *= lm_scale`, then
 * `+= ins_penalty` on arcs with an output label; final-state weights are used as stored.
 * Embedded alphabets are skipped (auxiliary '#' symbols are an error, as for the text loader). */
int jd_net_load_jwnt(jd_net **out, const char *path, float lm_scale, float ins_penalty);
/* WFSTNetwork::writeBinary (WFSTNetwork.cpp:1106-1225, juicer.cpp:879-882 -writeBinaryFiles):
 * takes the penalty and the scale the network was built with off the arc weights again;
 * no alphabets are written. */
int jd_net_save_jwnt(const jd_net *n, const char *path);

/* Read back the prepared CSR (host copies; any pointer may be NULL) - used by parity tests.
 * fin_w has n_states entries, +inf for non-final states. */
int jd_net_get_csr(const jd_net *n, int32_t *row_ptr, int32_t *to, float *w, int32_t *in, int32_t *outl,
                   float *fin_w);

int64_t jd_net_num_arcs(const jd_net *n);     /* WFSTNetwork::getNumTransitions */
int32_t jd_net_num_states(const jd_net *n);   /* WFSTNetwork::getNumStates      */
int32_t jd_net_init_state(const jd_net *n);   /* WFSTNetwork::getInitState      */
void    jd_net_destroy(jd_net *n);

/*
 * Dynamic composition, first step (WFSTOnTheFlyDecoder, juicer.cpp:594-598): C.L and G
 * are handed over apart - cl loaded like the reference's clNetwork (scale 1.0, juicer.cpp:933-940), g like
 * its gNetwork (lmScaleFactor; arcs of a state sorted by input label, one per label, the back-off
 * epsilon first: WFSTSortedInLabelNetwork, WFSTNetwork.cpp:2693-2710) - and composed ON THE DEVICE into
 * an ordinary network for jd_dec_create: reachable (C.L state, G state) pairs only, G advanced by a
 * binary search for the C.L arc's output label (WFSTOnTheFlyDecoder.cpp:3106-3159), back-off epsilons
 * taken right after a word only (WFSTOnTheFlyDecoder.cpp:1590-1622), lexicon branches that hold no word
 * the G state has an arc for left out (label look-ahead, WFSTNetwork.cpp:1505-2590, on label intervals:
 * tight when words are numbered in the lexicon tree's depth-first order).  Exact definition and the
 * state / arc numbering: csrc/jd_compose.hip.  max_states / max_arcs bound the result (0: a default
 * derived from the inputs); JD_ENOMEM names the one that was too small.  There is no CPU path.
 * pushing: the reference's -pushing (doLabelAndWeightPushing, juicer.cpp:240, 931-935) is JD_PUSH_WEIGHTS |
 * JD_PUSH_LABELS.  JD_PUSH_WEIGHTS: the best grammar weight reachable below a lexicon-tree node is paid on the way
 * into it, so the beam prunes on it; path totals are the same up to float association.  JD_PUSH_LABELS: C.L is
 * composed with its output labels pushed towards the initial state (jd_net_push_labels).
 */
#define JD_PUSH_WEIGHTS 1
#define JD_PUSH_LABELS 2
int jd_net_compose(jd_net **out, const jd_net *cl, const jd_net *g, int32_t device,
                   int64_t max_states, int64_t max_arcs, int32_t pushing);

/*
 * Label pushing on a C.L transducer (the reference: WFSTLabelPushingNetwork gives every transition the set of output
 * labels that can follow it, WFSTNetwork.cpp:1643-1764, and the on-the-fly decoder takes the G transition as soon as
 * the set is one label): every output label moves towards the initial state, up to the first arc behind which it is
 * the only label that can follow.  The result has the same states, arcs and weights and accepts the same label
 * sequences at the same costs; a hypothesis' word-end times become the frames in which its words were identified.
 * Host code (no device needed); exact rule: csrc/jd_compose.hip, cl_push_labels.  n_moved (may be NULL): labels moved.
 */
int jd_net_push_labels(jd_net **out, const jd_net *cl, int64_t *n_moved);

/*
 * Dynamic composition proper (WFSTOnTheFlyDecoder: C.L o G expanded where the search goes, never as a
 * whole): the network this returns holds only its start state; decoders created on it expand a composed
 * state - same expansion step as jd_net_compose, one state at a time - when a token first enters an arc
 * that leads to it (csrc/jd_lazy.h).  What has been expanded stays, shared by every decoder and stream on
 * the network, so later utterances find most of what they need.  `am` says which input labels are tee
 * models; max_states / max_arcs are the room the network may grow into (0: 2^22 states, 2^24 arcs;
 * see jd_net_lazy_set_high_water for what happens when it fills up).  The decoder must be created on the same device; the
 * network has no arc table to read back (jd_net_get_csr / jd_net_save_jwnt refuse it).  Results are those
 * of decoding on jd_net_compose's network (pushing: as there, applied arc by arc as states are expanded).
 * There is no CPU path.
 */
int jd_net_create_lazy(jd_net **out, const jd_net *cl, const jd_net *g, const jd_am *am, int32_t device,
                       int64_t max_states, int64_t max_arcs, int32_t pushing);
/* Bounded look-ahead memory (the reference: an LRU cache of composed transitions, WFSTOnTheFlyDecoder.h:210-371).
 * Eviction here is by arena GENERATION: when an utterance (or batch) begins while no stream of any decoder on the
 * network is inside one, and the arena is past its high-water mark - `fraction` of either capacity, default 0.9 -
 * or has run out of room, everything expanded so far is dropped and the arena starts again from the start state.
 * A batch that runs out of room under way is decoded again on a fresh generation, so JD_ENOMEM means ONE batch
 * needs more than the capacities; it is never a sticky state of the network.  jd_net_lazy_generation counts the
 * generations begun so far (0: nothing has been dropped yet). */
int jd_net_lazy_set_high_water(jd_net *n, double fraction);
int jd_net_lazy_generation(const jd_net *n, int64_t *generation);
/* The same by hand: forget everything expanded so far.  JD_ESTATE while a stream of a decoder on the network is
 * inside an utterance. */
int jd_net_lazy_reset(jd_net *n);
/* composed states and arcs materialised so far */
int jd_net_lazy_size(const jd_net *n, int64_t *states, int64_t *arcs);

/* ---------------------------------------------------------- acoustic models */

/*
 * Build models from HTK-level parameters, applying the load-time arithmetic of
 * HTKModels::addVarVec (HTKModels.cpp:835-870), addGMM (:600-676),
 * addTransMatrix (:873-974), addHMM tee detection (:581-593),
 * createTrPandSEIndex (:2330-2390) and HTKFlatModels::init
 * (HTKFlatModels.cpp:94-177):
 *     det  = -0.5*(D*LOG_2_PI + sum_k logf(var_k))  [float accumulation] + logf(weight)
 *     ivar = (float)(1.0/var)
 *     trP[i][j] = logf(a_ij) if a_ij > 0 else LOG_ZERO; SEIndex; teeWeight
 * Layout: weight[n_gmm*max_mix], mean/var[n_gmm*max_mix*D] (component major),
 * n_mix[n_gmm]; hmm_nstates[n_hmm] (incl. entry+exit), hmm_gmm[n_hmm*max_n]
 * (gmm id per state, -1 for entry/exit), hmm_tm[n_hmm]; tm_nstates[n_tm],
 * transp[n_tm*max_n*max_n] row-major a_ij.
 */
int jd_am_create_htk(jd_am **out, int32_t D, int32_t n_gmm, int32_t max_mix,
                     const int32_t *n_mix, const float *weight,
                     const float *mean, const float *var,
                     int32_t n_hmm, int32_t max_n, const int32_t *hmm_nstates,
                     const int32_t *hmm_gmm, const int32_t *hmm_tm,
                     int32_t n_tm, const int32_t *tm_nstates, const float *transp);

/*
 * Build models from the PREPARED arrays of a loaded HTKFlatModels / HTKModels object - what
 * `WFSTDecoderLite(network, models, ...)` (WFSTDecoderLite.h:81-89) is handed by juicer.cpp:577-586:
 * det = fDets, mean = fMeans, ivar = fVars (inverse variances) of HTKFlatModels.h:48-52 re-laid as
 * [n_gmm][max_mix] / [n_gmm][max_mix][D]; hmm_tee[h] = getTeeLogProb(h) (Models.h:61); trP[t] =
 * getTransMat(h) rows (Models.h:63, [n_tm][max_n][max_n], LOG_ZERO where there is no transition),
 * se[t][j] = getSEIndex(h)[j] (Models.h:64, [n_tm][max_n][2]).  Nothing is recomputed.
 * include/juicer_amd_decoder.hpp's exact-signature constructor fills these from an IModels*.
 */
int jd_am_create_flat(jd_am **out, int32_t D, int32_t n_gmm, int32_t max_mix, const int32_t *n_mix,
                      const float *det, const float *mean, const float *ivar,
                      int32_t n_hmm, int32_t max_n, const int32_t *hmm_nstates,
                      const int32_t *hmm_gmm, const int32_t *hmm_tm, const float *hmm_tee,
                      int32_t n_tm, const int32_t *tm_nstates, const float *trP, const int16_t *se);

/* HTK MMF text (HTKModels::Load, HTKModels.cpp:221-282): the subset the reference's
 * flex/bison front-end accepts (htkparse.l.lpp:21-268, htkparse.y.ypp:113-147,414-685):
 * ~o, ~v (ignored), ~s, ~t, ~h with shared or inline states / <TRANSP>, <NUMMIXES>/<MIXTURE>
 * or the implicit single mixture, optional <GCONST>.  HMM index = order of the ~h macros. */
int jd_am_load_mmf(jd_am **out, const char *mmf_path);
/* HTKModels::Load(phonesListFName, priorsFName, statesPerModel) (HTKModels.cpp:74-218): hybrid ANN / HMM
 * models - one HMM per phone, states_per_model states each (entry and exit included), one shared
 * transition matrix (1.0 into the first emitting state, 0.5 / 0.5 self loop / forward); the feature vector
 * holds one log posterior per phone and an emitting state scores x[phone] - log(prior[phone])
 * (HTKModels.cpp:481-512, HTKFlatModels.cpp:190-222).  priors: n_phones values as read from the priors
 * file (the phone list only gives names). */
int jd_am_create_hybrid(jd_am **out, int32_t n_phones, const float *priors, int32_t states_per_model);

/* Juicer's binary model cache "<mmf>.bin" (JMBI), preferred by juicer.cpp:778-784:
 * HTKModels::readBinary (HTKModels.cpp:1110-1233) + HTKFlatModels::init.  The derived values in
 * the file (sumLogVarPlusNObsLog2Pi, logCompWeights, transition logProbs) are used as stored. */
int jd_am_load_jmbi(jd_am **out, const char *path);
/* HTKModels::output(fName, true) (HTKModels.cpp:1044-1105, juicer.cpp:791-795). */
int jd_am_save_jmbi(const jd_am *a, const char *path);

int32_t jd_am_num_hmms(const jd_am *a);       /* IModels::getNumHMMs       (Models.h:57) */
int32_t jd_am_num_gmms(const jd_am *a);
int32_t jd_am_vec_size(const jd_am *a);       /* IModels::getInputVecSize  (Models.h:60) */
int32_t jd_am_max_states(const jd_am *a);
int32_t jd_am_max_mix(const jd_am *a);
int32_t jd_am_num_transmats(const jd_am *a);
/* topology read-back: hmm_nstates[n_hmm], hmm_gmm[n_hmm*max_states], hmm_tm[n_hmm], n_mix[n_gmm] */
int jd_am_get_topology(const jd_am *a, int32_t *hmm_nstates, int32_t *hmm_gmm, int32_t *hmm_tm, int32_t *n_mix);
/* Read back prepared parameters (host copies) - used by parity tests. */
int jd_am_get_flat(const jd_am *a, float *det, float *mean, float *ivar);
int jd_am_get_trans(const jd_am *a, float *trP, int16_t *se, float *tee);
void jd_am_destroy(jd_am *a);

/* ------------------------------------------------------------------ decoder */

typedef struct jd_stats {          /* WFSTDecoderLite.cpp:231-241 + build counters */
    int32_t n_frames;
    int64_t tot_active_emit_hyps;  /* totalActiveEmitHyps */
    int64_t tot_active_end_hyps;   /* totalActiveEndHyps  */
    int64_t tot_active_models;     /* totalActiveModels   */
    int64_t tot_proc_emit_hyps;    /* totalProcEmitHyps   */
    int64_t tot_proc_end_hyps;     /* totalProcEndHyps    */
    /* work the GPU path actually did: at most the reference's figures (only the best tokens
     * per destination state are expanded) and, with the inline closure, slightly run-dependent */
    int64_t tot_arcs_visited;      /* out-arcs visited by propagateToken (A)      */
    int64_t tot_paths;             /* Path records created (Wd)                    */
    int64_t tot_insts_in;          /* instances processed by internal propagation (Mdl) */
    int64_t ties;                  /* equal-score recombinations seen (oracle only; 0 on GPU) */
    /* what the kernels really touched - the figures above are the REFERENCE's counts (tot_insts_in includes candidates that
     * never become a record here, tot_arcs_visited the arcs a prefix walk accounts for without reading them); these price
     * the bytes the DESIGN moves (bench.py: roofline.design_bytes_per_launch): */
    int64_t tot_recs_read;         /* phase A: instance records read */
    int64_t tot_new_attached;      /* phase A: newly entered arcs taken up from the new-arc list (attachNetInst) */
    int64_t tot_recs_written;      /* phase A: records written to the next frame's list */
    int64_t tot_entry_items;       /* phase A: entry tokens pulled (the winning frontier item is gathered) */
    int64_t tot_items_expanded;    /* phase X: frontier items taken up (exit tokens, closure items, slices) */
    int64_t tot_arcs_walked;       /* phase X: arc records loaded */
    int64_t tot_closure_items;     /* phase X: closure items written (epsilon / tee arcs followed) */
    int64_t tot_bids_placed;       /* phase A: exit tokens that bid for their destination state - those of an arc that is the only
                                    * arc into its state recombine with nobody and place none (csrc/jd_search.h: REC_SOLE) */
} jd_stats;

/*
 * 1-best result: the DecHyp / DecHypHist chain of recognitionFinish()
 * (WFSTDecoderLite.cpp:262-308, DecHypHistPool.h:38-49,146-165) flattened to
 * arrays in CHAIN order: index 0 is hyp->hist (the newest word), index n-1 the
 * oldest.  label = DecHypHist::state (= output label = word id + 1), time =
 * DecHypHist::time.  Entry 0 and the totals carry the final-state weight.
 * n == -1 means "no token survived" (the reference returns NULL + warning).
 * Storage is owned by the decoder and valid until the next init of that stream.
 */
typedef struct jd_hyp {
    int32_t  n;
    const int32_t *label;
    const int32_t *time;
    const float   *score;
    const float   *ac;
    const float   *lm;
    float tot_score, tot_ac, tot_lm;
    jd_stats stats;
} jd_hyp;

/*
 * WFSTDecoderLite::WFSTDecoderLite(network, models, phoneStartPruneWin,
 * emitPruneWin, phoneEndPruneWin, wordPruneWin, maxEmitHyps)
 * (WFSTDecoderLite.h:81-89).  A window is enabled iff > 0; max_hyps == 0
 * disables the histogram (WFSTDecoderLite.cpp:76-82).  block_size mirrors
 * IModels::setBlockSize (1..20, juicer.cpp:255); it never changes results.
 * device = HIP device ordinal; max_streams = utterances decoded concurrently.
 * Like the reference's decoder (juicer.cpp:647-652) a decoder does not own its network or models: both must
 * outlive it - jd_dec_destroy still talks to the network (utterances of the streaming interface that were never
 * finished leave a lazily composed network there).
 */
int jd_dec_create(jd_dec **out, const jd_net *net, const jd_am *am,
                  float start_beam, float main_beam, float end_beam, float word_beam,
                  int32_t max_hyps, int32_t block_size, int32_t device, int32_t max_streams);
void jd_dec_destroy(jd_dec *d);

/* Per-stream arena capacities, in records (call before the first init; 0 keeps the default).
 * Defaults are taken from the free HBM when the arenas are first allocated: 70 % of it is split
 * over max_streams, each stream's share going 50/20/30 to instance records (80 B, two lists),
 * frontier items and Path records, never below 2^19 / 2^21 / 2^21 and never above what the
 * graph can need (one instance per arc).  An overflow is reported as JD_ENOMEM naming the arena. */
int jd_dec_set_capacity(jd_dec *d, int64_t max_slots, int64_t max_paths, int64_t max_items);
/* The stream arenas of a decoder are one slab of device memory; jd_dec_destroy keeps the largest one per device
 * for the next decoder created there (a decoder's set-up time is the driver clearing the memory it hands out:
 * seconds for the default sizes).  This gives the cached slab of `device` back to the driver (also: environment
 * JD_ARENA_CACHE=0 never keeps one). */
int jd_release_cached_memory(int32_t device);
/* WFSTDecoderLite::setMaxAllocModels (WFSTDecoderLite.cpp:807-820; environment variable
 * MaxAllocModels, default 10, :73-74), same argument convention: < 100 a percentage of the network's
 * transitions, 100..7999 a memory limit in MB (of 40 + 24 * maxNStates bytes per instance), otherwise
 * a number of instances.  A SOFT limit, as in the reference, which drops its cached NetInst objects
 * between utterances when more than this many are allocated (:164-169) and never fails a decode
 * because of it: here nothing is cached between utterances (one arena of records per stream, reused
 * wholesale), so the value can only RAISE that arena above its automatic size, never lower it - a
 * hard capacity is jd_dec_set_capacity's max_slots.  Before the first decode. */
int jd_dec_set_max_alloc_models(jd_dec *d, int32_t max_alloc_models);

/* IDecoder::init() (Decoder.h:26) for stream s. */
int jd_stream_init(jd_dec *d, int32_t s);
/* IDecoder::processFrame() x n_frames (Decoder.h:27): frames is n_frames x D
 * contiguous HOST floats; the look-ahead rows of the reference protocol
 * (DecoderSingleTest.cpp:267-295) carry no information beyond the frames
 * themselves, so the adapter folds them into pushes. */
int jd_stream_push(jd_dec *d, int32_t s, const float *frames, int32_t n_frames);
/* IDecoder::finish() (Decoder.h:28). */
int jd_stream_finish(jd_dec *d, int32_t s, jd_hyp *out);
/* jd_stream_push for several streams at once (each at most once per call): ONE scoring launch and ONE search launch
 * for all of them - the streams may sit at different frames.  Not with partial traces (jd_dec_set_partial_interval). */
int jd_streams_push(jd_dec *d, int32_t n, const int32_t *streams, const float *const *frames, const int32_t *n_frames);
/* max_streams / feature vector size of a decoder */
int jd_dec_info(const jd_dec *d, int32_t *max_streams, int32_t *vec_size);

/* ------------------------------------------------------------------ many IDecoder instances on one decoder
 *
 * The reference's harness decodes serially through ONE IDecoder (DecoderBatchTest.cpp:738-771 ->
 * DecoderSingleTest.cpp:259-324) and scales out as independent processes over split file lists
 * (doc/userman/juicer_userman.tex:584).  A broker lets N such serial callers - threads of one process - share the
 * streams of one decoder: each drives its own client with init / push / finish.  Up to 64 clients are served by a search
 * kernel that STAYS on the device while there is work (csrc/jd_resident.h): every client's stream has a cluster of
 * workgroups of its own there, the worker thread scores a client's frames as they come (one launch per round for all
 * the clients that brought some) and hands them to the cluster as soon as it is through with the chunk before - every
 * stream at its own pace; init, the Path collections and finish are small kernels beside it.  The kernel leaves the
 * device after 3 ms without work and comes back with the next request; while it is there, other search launches on the
 * device (this process: they wait for it; other processes: the file lock) and anything that synchronises the whole
 * device wait for that.  More than 64 clients, a lazily composed network, or JD_BROKER_RESIDENT=0: a worker that turns
 * what has been pushed since its last TICK into one scoring launch and one search launch over all the streams concerned
 * (jd_streams_push).  The decoder must not be used directly while a broker owns it; clients may be driven from
 * different threads, one thread per client at a time.  jd_hyp arrays stay valid until the client's next init.
 * Environment: JD_BROKER_TICK_FRAMES (frames of one client per chunk / tick, default 256 / 192), JD_BROKER_COALESCE_US
 * (ticks: how long one waits for the other open clients' frames, default 300). */
typedef struct jd_broker jd_broker;
typedef struct jd_broker_stats {
    int64_t ticks, frames, stream_ticks;           /* launches, frames, streams summed over ticks */
    int64_t us_idle, us_coalesce;                  /* worker thread: waiting for work; waiting for the other clients' frames */
    int64_t us_init, us_push, us_finish;           /* ... inside jd_stream_init / jd_streams_push / jd_stream_finish */
    int64_t us_search;                             /* of us_push: the search launches (device time) */
    int64_t resident;                              /* > 0: the worker drives the resident search kernel, clusters of this many workgroups.
                                                      A "tick" is then one stream's chunk, and: us_search = the time from its post to its
                                                      report, us_coalesce = the time its cluster spent on it (device clock), us_init = the
                                                      NUMBER of Path collections between chunks, us_idle = what the cluster waited for the
                                                      next chunk of the same utterance, us_push = the staging calls, us_finish = the result
                                                      fetches */
} jd_broker_stats;
int jd_broker_create(jd_broker **out, jd_dec *dec, int32_t n_clients);   /* n_clients <= the decoder's max_streams */
void jd_broker_destroy(jd_broker *b);                                    /* (the decoder is the caller's to destroy, afterwards) */
int jd_broker_open(jd_broker *b, int32_t *client);
int jd_broker_close(jd_broker *b, int32_t client);
int jd_broker_init(jd_broker *b, int32_t client);                                              /* IDecoder::init */
/* IDecoder::processFrame x n: returns once the frames are taken */
int jd_broker_push(jd_broker *b, int32_t client, const float *frames, int32_t n_frames);
int jd_broker_finish(jd_broker *b, int32_t client, jd_hyp *out);                               /* IDecoder::finish */
int jd_broker_get_stats(jd_broker *b, jd_broker_stats *out);

/*
 * PARTIAL_DECODING (WFSTDecoderLite.cpp:822-896, compiled in by src/CMakeLists.txt:5): the Path
 * records every open hypothesis of a stream has converged into - the part of the result that can
 * no longer change - collected in the decoder's partialPaths list (WFSTDecoderLite.h:199-205).
 *
 * jd_dec_set_partial_interval = setPartialDecodeOptions (:892-896; the reference takes the value
 * from the environment variable PartialTraceInterval, :116-119; 0 = off, the default).  With an
 * interval > 0 jd_stream_push traces on the reference's schedule - together with a path
 * collection, if f - lastPartialTraceFrame > interval (:362-368) - and jd_stream_finish completes
 * the list from the best token (:245-251).  A collection runs after frame f under the reference's
 * two triggers (:362): f - lastPathCollectFrame > 100, or nPath / nPathNew > 12 with nPath > 10000.
 * The first is exact.  The second counts Path objects: the reference creates one per labelled
 * propagateToken call (:497-509), i.e. for every exit token and every labelled epsilon / tee arc of the
 * closure behind it, whether or not the token then wins its entry state, and keeps what the tokens of its
 * active instances reach (:703-745).  With the end and word beams off (the CLI's default) those are static
 * properties of the graph, and the decoder counts them without writing the losers: the collections then run
 * after the reference's own frames (tests: equal to the oracle's frame by frame, counts included).  With
 * an end or word beam, or a lazily composed network, the rule runs on this build's own records - at most
 * the reference's, so it fires no earlier - an approximation.  The schedule decides when a trace is
 * taken, never what a trace at a given frame finds.
 * jd_stream_collect_info: collections of the stream's utterance so far and lastPathCollectFrame.
 * jd_stream_path_counts: nPath and nPathNew as the trigger reads them, *exact = 1 when they are the reference's.
 *
 * jd_stream_partial returns the stream's partialPaths - (output label, frame) of each record,
 * oldest first; *n is the full length, at most cap entries are written - after, if trace_now != 0,
 * running tracePartialPath (:824-868) on the frame the stream has reached (*found = its return
 * value).  Streaming API only: a batch returns whole results, and the list recognitionFinish
 * ends up with is the hypothesis itself (jd_hyp.label / .time, newest first).
 */
int jd_dec_set_partial_interval(jd_dec *d, int32_t interval);
int jd_stream_collect_info(jd_dec *d, int32_t s, int32_t *n_collections, int32_t *last_collect_frame);
int jd_stream_path_counts(jd_dec *d, int32_t s, int32_t *n_path, int32_t *n_path_new, int32_t *exact);
/* Diagnostics / tests (host only): per state, the Path objects WFSTDecoderLite::propagateToken creates behind ONE token that
 * arrives there with the end and word beams off - one per labelled epsilon arc and per labelled arc of a tee model that
 * leaves the state, plus what arrives behind each of those arcs, with multiplicity (:497-509 inside :533-541, :583-599);
 * saturated at 2^20.  out[n_states]; *acyclic = 0: the label-less part of the graph has a cycle, the counts are not used. */
int jd_debug_closure_path_counts(const jd_net *net, const jd_am *am, int32_t *out, int32_t *acyclic);
int jd_stream_partial(jd_dec *d, int32_t s, int32_t trace_now, int32_t cap, int32_t *n,
                      int32_t *labels, int32_t *times, int32_t *found);

/*
 * DecoderBatchTest::run() inner loop (DecoderBatchTest.cpp:738-771) for a
 * batch: decodes n_utts utterances (n_utts may exceed max_streams; they are
 * processed in waves of max_streams).  feats[u] = n_frames[u] x D host floats.
 */
int jd_decode_batch(jd_dec *d, int32_t n_utts, const float *const *feats,
                    const int32_t *n_frames, jd_hyp *out);

/*
 * Same, with features already resident in device memory (bench path: inputs
 * in HBM before the timed region).  d_feats is one device buffer holding all
 * utterances back to back; offs[u] is the frame offset of utterance u
 * (n_utts+1 entries).  hip_stream is a hipStream_t (or NULL for the default
 * stream).  Results are valid after the call returns (it synchronises).
 */
int jd_decode_batch_device(jd_dec *d, int32_t n_utts, const float *d_feats,
                           const int64_t *offs, void *hip_stream, jd_hyp *out);

/*
 * Scoring one batch ahead.  The reference scores and searches in turn (HTKFlatModels::calcOutput is
 * called from the search, src/HTKFlatModels.cpp:190-262; its two-thread organisation overlaps them on
 * two cores).  Here the batch path scores a whole batch's likelihood table and then searches it with a
 * persistent kernel whose clusters leave their CUs as their utterances end; announcing the NEXT batch
 * (same arguments as the jd_decode_batch_device call that will decode it; at most max_streams
 * utterances) before decoding the current one lets the scoring kernel of that next batch run on the
 * CUs the current search leaves idle, into a second table.  The announced decode then finds its table
 * scored.  The device buffer of the announced batch must stay valid and unchanged until that decode
 * (offs is copied); a decode with other arguments simply drops the table - results never depend on the
 * announcement.  n_utts = 0 drops whatever was announced or scored ahead.  Calling this is optional.
 * Announcements queue up, in the order of the decodes to come (at most three; one more is not taken), and
 * SEARCHING ahead follows from scoring ahead: a batch lasts as long as its longest utterance while the clusters of
 * its shorter ones are long through, so when the batch BEHIND the one being decoded has its table already - it was
 * announced two decodes ahead - and each of the two fills at most half of the decoder's streams, its utterances are
 * started beside the running batch, one workgroup each, on the other half of the streams; when its turn comes they
 * are hundreds of frames in (jd_timing.ahead_frames) and the call is that much shorter.  A stream of batches gets
 * this with two announcements before its first decode and one before every later one (jd_dec_set_pipeline(d, JD_FLOW_SERIAL, 0, 0) switches it off).
 * While the scoring runs the search launch is not re-planned under way when the scoring is a sizeable part
 * of the step (measured on the decoder's last batches: from a tenth of the search on), so that its blocks find
 * CUs; else it is slotted in at the re-planning cuts.
 */
int jd_dec_prefetch_scores(jd_dec *d, int32_t n_utts, const float *d_feats,
                           const int64_t *offs, void *hip_stream);

/*
 * The same loop sharded over the GPUs of ONE node from C++ (no Python, no torchrun): one decoder
 * and one host thread per device, utterances dealt by length (longest first, each to the device
 * with the fewest frames so far), no data-path collective, and ONE RCCL all-gather (over xGMI) of
 * padded 1-best records at the end - a record is as long as the batch's longest hypothesis; results
 * come back in the caller's utterance order.  An utterance that fails leaves an empty hypothesis,
 * the others are returned and the call reports the first error.  devices == NULL means devices
 * 0 .. n_devices-1.  RCCL is loaded with dlopen on first use.  jd_hyp storage is owned by the
 * jd_multi and valid until its next decode.
 */
typedef struct jd_multi jd_multi;
int jd_multi_create(jd_multi **out, const jd_net *net, const jd_am *am,
                    float start_beam, float main_beam, float end_beam, float word_beam,
                    int32_t max_hyps, int32_t block_size, int32_t n_devices, const int32_t *devices,
                    int32_t max_streams_per_device);
/* the same over lazily composed networks (jd_net_create_lazy), one per device, owned by the jd_multi: every
 * device expands the part of C.L o G its own utterances reach */
int jd_multi_create_lazy(jd_multi **out, const jd_net *cl, const jd_net *g, const jd_am *am,
                         int64_t max_states, int64_t max_arcs, int32_t pushing,
                         float start_beam, float main_beam, float end_beam, float word_beam,
                         int32_t max_hyps, int32_t block_size, int32_t n_devices, const int32_t *devices,
                         int32_t max_streams_per_device);
int jd_multi_decode_batch(jd_multi *m, int32_t n_utts, const float *const *feats,
                          const int32_t *n_frames, jd_hyp *out);
void jd_multi_destroy(jd_multi *m);

/* Timing of the most recent jd_decode_batch*() (HIP events on the decoder's own streams): total
 * duration of the GMM-kernel launches, of the k_search launches (the persistent search kernel:
 * one launch per chunk of frames, repeated when a stream had to stop for Path garbage
 * collection), wall time of the call. */
typedef struct jd_timing {
    double gmm_ms, search_ms, total_ms;
    double gmm_wait_ms;       /* host time spent waiting for scores the search needed next      */
    int32_t gmm_launches, search_launches;
    int32_t relaunches;       /* k_search launches repeated after an in-chunk garbage collection */
    int32_t cluster_wgs;      /* workgroups (1024 threads) per stream cluster in the last launch */
    int64_t gmm_frames;       /* stream-frames scored                           */
    int64_t gmm_states;       /* tied states scored per frame                    */
    int64_t search_frames;    /* stream-frames decoded                          */
    int32_t prefetched;       /* waves whose table had been scored ahead (jd_dec_prefetch_scores): their gmm_ms is the
                                 span of that scoring beside the previous search, and gmm_wait_ms what was left of it */
    int32_t ahead_frames;     /* frames of this batch that had been searched beside the batch before it ("two batches in
                                 flight": announcements two batches ahead, a batch on at most half of the streams) */
    int32_t slot_launches;    /* of search_launches: launches of the slot kernel (csrc/jd_slot.h: one workgroup per stream, two per
                                 CU - batches of more streams than the chip has CUs) */
    int32_t pad0;
} jd_timing;
int jd_dec_last_timing(const jd_dec *d, jd_timing *out);

/*
 * How batches that follow each other share the chip - the decoder's counterpart of the reference's constructor-time
 * configuration (WFSTDecoderLite.h:81-89; its harness decodes list entry after list entry, DecoderBatchTest.cpp:738-771).
 * Results never depend on it.  Call it any time between two decodes; whatever is announced or under way is dropped.
 *   JD_FLOW_SERIAL         one batch on the chip at a time (an announced batch is still SCORED beside the running search).
 *   JD_FLOW_TWO_IN_FLIGHT  (default) the batch behind the running one is started beside it when its table is there:
 *                          announcements two batches ahead, each batch on at most half of the decoder's streams.
 *   JD_FLOW_RESIDENT       announced batches go through a search kernel that STAYS on the device, utterance by utterance:
 *                          every stream is a slot of one workgroup that takes the next queued utterance the moment its
 *                          own is through; the scoring runs beside them.  depth = batches announced and not yet handed
 *                          back, at most (2..32; 0 = by the slots: 8, or slots / 32 + 2 - a likelihood table each); slots =
 *                          one-workgroup slots (1..max_streams; 0 = max_streams): a slot is half a CU and the slots are dealt
 *                          one per CU while there are CUs - the scoring kernel's workgroups take the other half of the
 *                          same CUs, so as many slots as the device has CUs is the count to ask for.
 *                          jd_decode_batch_device must be called for the batches in the order they
 *                          were announced (anything else drops what is under way and decodes the usual way); a batch
 *                          larger than the decoder's slots goes through them without any announcement.  See
 *                          jd_dec_quiesce for what the resident kernel means for the rest of the process.
 *                          Not with lazily composed networks, hybrid scoring or PARTIAL_DECODING (those decode the
 *                          usual way).
 */
#define JD_FLOW_SERIAL         0
#define JD_FLOW_TWO_IN_FLIGHT  1
#define JD_FLOW_RESIDENT       3
int jd_dec_set_pipeline(jd_dec *d, int32_t mode, int32_t depth, int32_t slots);

/*
 * How the likelihood tables are scored - an OPTION beside the default, for a caller that wants throughput over bit equality.
 *   JD_SCORE_EXACT  (default) HTKFlatModels::calcGMMOutput + logAdd with the reference's own roundings (src/HTKFlatModels.cpp:226-293: no
 *                   contraction, the host libm's expf, log(1 + e) in double): every log-likelihood equals the reference's bit for bit.
 *   JD_SCORE_FAST   fused multiply-add distance on pre-scaled parameters and an fp32 logAdd on the hardware's exp / log: a log-likelihood
 *                   moves by ~1e-5 of its magnitude; hypotheses keep their words and times on every fixture of the test suite and their
 *                   scores stay within 1e-4 relative (what BASELINE.json's north_star asks of the path; tests/test_gpu_fastscore.py).
 *                   The table costs 0.4 of the exact one (csrc/jd_gmm.h: jd_gmm_fast39).  39-dimensional GMM models only.
 * Call it between two decodes; whatever was scored or announced ahead is dropped.  jd_am_score_frames_mode: jd_am_score_frames with the option.
 */
#define JD_SCORE_EXACT 0
#define JD_SCORE_FAST  1
int jd_dec_set_scoring(jd_dec *d, int32_t mode);
int jd_am_score_frames_mode(const jd_am *a, int32_t device, int32_t mode, const float *frames, int32_t n_frames, float *out);

/* What JD_FLOW_RESIDENT has done so far (cumulative over the decoder's life; a bench reads it on either side of its
 * timed region: frames_searched is what the slots really advanced in between, whatever was announced or handed back). */
typedef struct jd_pipe_stats {
    int32_t mode, depth, slots;
    int32_t resident;            /* the kernel is on the device right now                                     */
    int32_t batches_announced;   /* announced and not yet handed back                                         */
    int32_t pad0;
    int64_t frames_searched;     /* stream-frames the slots have advanced (their own reports, command by command) */
    int64_t utts_through;        /* utterances whose result has been exported                                 */
    int64_t rows_scored;         /* likelihood rows whose scoring has been enqueued                           */
    int64_t batches_back;        /* batches handed back by jd_decode_batch_device                              */
    int64_t collections;         /* Path collections between commands                                         */
    double slot_busy_us;         /* sum over the slots of their own clocks on their commands                  */
    double on_us;                /* wall time the resident kernel has been on the device                      */
} jd_pipe_stats;
int jd_dec_pipeline_stats(const jd_dec *d, jd_pipe_stats *out);

/* The decoder's work on the device comes to rest: with JD_FLOW_RESIDENT (announced batches go through a search kernel that stays
 * on the device, utterance by utterance: jd_dec_prefetch_scores up to `depth` batches ahead) that kernel lets the
 * commands that are running run out (128 frames at most) and leaves; nothing announced or under way is lost - it comes back
 * with the next call.  What a caller needs before a device-wide synchronisation while batches are announced - and before
 * it puts work of its own on OTHER streams of the device: HIP maps streams onto a few hardware queues, and a kernel that
 * is queued behind the resident one waits until it leaves (tools/resident_alias_probe.py: one fresh stream in fourteen;
 * by itself the kernel leaves after 5 s without a command, and the decode that follows says so).  The same holds while a
 * broker's resident kernel has work (it leaves after 3 ms without). */
int jd_dec_quiesce(jd_dec *d);

/* Diagnostics: per-workgroup cycle accounting of k_search (100 MHz wall clock).  enable >= 0 with
 * fetch == NULL switches it on and clears it, enable < 0 switches it off; with fetch != NULL
 * (1024 x 16 int64) the sums so far are copied out: per workgroup of the grid (thread 0's timeline)
 * {work lists A, phase A, workgroup wait, cluster barrier 1, work lists X, phase X, workgroup wait,
 *  cluster barriers of X, frames, ...}. */
int jd_dec_debug_trace(jd_dec *d, int32_t enable, int64_t *fetch);
/* Diagnostics: what part of a batch's likelihood table does the search read?  The reference scores a tied state only when
 * a token that passed the emit threshold asks for it (WFSTDecoderLite.cpp:409-411); this build scores every state of
 * every frame.  enable != 0: from the next decode on every cell (frame, tied state) whose value is added to a token is
 * marked; enable == 0: *cells_read = the marks, *cells_total = frames x tied states of the last decode, marking stops. */
int jd_dec_debug_cells(jd_dec *d, int32_t enable, int64_t *cells_read, int64_t *cells_total);

/*
 * Companion kernel on its own: HTKFlatModels::calcGMMOutput
 * (HTKFlatModels.cpp:226-262) for every tied state of every frame.
 * frames: n_frames x D host floats; out: n_frames x n_gmm host floats.
 */
int jd_am_score_frames(const jd_am *a, int32_t device, const float *frames,
                       int32_t n_frames, float *out);

/* Test hook behind the bit-exactness of logAdd (HTKFlatModels.cpp:266-293 calls expf on a float):
 * the kernels' own expf for x[0..n), evaluated on HIP device `device`, or by its host twin
 * (same source) when device == -1.  tests/test_expf.py compares both with the host libm. */
int jd_debug_expf(int32_t device, const float *x, int64_t n, float *out);

const char *jd_last_error(void);
const char *jd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JUICER_AMD_H */
