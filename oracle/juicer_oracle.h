/*
 * juicer_oracle.h - CPU ORACLE for the juicer_amd parity tests.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under juicer_amd/ or include/ may link,
 * import or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (idiap/juicer) ships no tests, fixtures or
 * golden vectors for this path, and its sources cannot be compiled in this
 * image (they need the absent Torch3 and Tracter headers plus a bison/flex
 * generated parser; writing stand-ins for those is not a reference build -
 * tools/refbase does write them, for TIMING the reference's classes and for a
 * differential check of this restatement against them, which it passes bit for
 * bit on the bench workload; that pins nothing).
 * This file is therefore a line-by-line *restatement* of the reference
 * algorithm, each function citing the reference file:line it follows.
 * Third-party constants it depends on (Torch3 >= 3.1, configure.ac:42, not in
 * the tree): real = float, LOG_ZERO = -FLT_MAX, LOG_2_PI = 1.83787706640934548355.
 */
#ifndef JUICER_ORACLE_H
#define JUICER_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct jo_net jo_net;
typedef struct jo_am  jo_am;
typedef struct jo_dec jo_dec;

typedef struct jo_stats {
    int32_t n_frames;
    int64_t tot_active_emit_hyps, tot_active_end_hyps, tot_active_models;
    int64_t tot_proc_emit_hyps, tot_proc_end_hyps;
    int64_t tot_arcs_visited, tot_paths, tot_insts_in;
    int64_t ties;              /* equal-score recombinations whose winner depends on visiting order */
} jo_stats;

typedef struct jo_hyp {
    int32_t n;                 /* -1: no surviving token (reference returns NULL) */
    const int32_t *label, *time;
    const float *score, *ac, *lm;
    float tot_score, tot_ac, tot_lm;
    jo_stats stats;
} jo_hyp;

int jo_net_create_arcs(jo_net **out, int64_t n_arcs, const int32_t *from, const int32_t *to,
                       const int32_t *in, const int32_t *outl, const float *w_file,
                       int32_t n_final, const int32_t *fstate, const float *fweight_file,
                       float lm_scale, float ins_penalty);
int jo_net_create_csr(jo_net **out, int32_t n_states, int32_t init_state,
                      const int32_t *row_ptr, const int32_t *to, const float *w,
                      const int32_t *in, const int32_t *outl,
                      int32_t n_final, const int32_t *fstate, const float *fweight);
int64_t jo_net_num_arcs(const jo_net *n);
int32_t jo_net_num_states(const jo_net *n);
int32_t jo_net_init_state(const jo_net *n);
/* prepared arrays in file order (for loader parity tests) */
int jo_net_get(const jo_net *n, int32_t *first, int32_t *cnt, int32_t *to, float *w,
               int32_t *in, int32_t *outl, int32_t *final_ind, float *final_w);
void jo_net_destroy(jo_net *n);

int jo_am_create_htk(jo_am **out, int32_t D, int32_t n_gmm, int32_t max_mix,
                     const int32_t *n_mix, const float *weight, const float *mean, const float *var,
                     int32_t n_hmm, int32_t max_n, const int32_t *hmm_nstates,
                     const int32_t *hmm_gmm, const int32_t *hmm_tm,
                     int32_t n_tm, const int32_t *tm_nstates, const float *transp);
int jo_am_get_flat(const jo_am *a, float *det, float *mean, float *ivar);
int jo_am_get_trans(const jo_am *a, float *trP, int16_t *se, float *tee);
/* HTKFlatModels::calcGMMOutput for every tied state of every frame, no cache */
int jo_am_score_frames(const jo_am *a, const float *frames, int32_t n_frames, float *out);
/* HTKModels::Load(phonesListFName, priorsFName, statesPerModel), HTKModels.cpp:74-218: hybrid ANN / HMM models */
int jo_am_create_hybrid(jo_am **out, int32_t n_phones, const float *priors, int32_t states_per_model);
void jo_am_destroy(jo_am *a);

int jo_dec_create(jo_dec **out, const jo_net *net, const jo_am *am,
                  float start_beam, float main_beam, float end_beam, float word_beam,
                  int32_t max_hyps, int32_t block_size);
void jo_dec_destroy(jo_dec *d);
int jo_init(jo_dec *d);
/* rows[0] = frame `frame`, rows[1..n_avail-1] = look-ahead rows */
int jo_process_frame(jo_dec *d, const float *const *rows, int32_t frame, int32_t n_avail);
int jo_finish(jo_dec *d, jo_hyp *out);
/* DecoderSingleTest::decodeUtterance loop (DecoderSingleTest.cpp:259-298);
 * *cpu_seconds receives clock()-based CPU time like decodeTime (:299-300). */
int jo_decode_utt(jo_dec *d, const float *feats, int32_t n_frames, jo_hyp *out, double *cpu_seconds);
/* per-frame trace for debugging parity: bestEmitScore after each frame */
int jo_set_trace(jo_dec *d, float *best_emit_per_frame, int32_t cap);
int jo_set_cells(jo_dec *d, uint8_t *cells, int32_t frames);   /* diagnostics: the (frame, tied state) cells calcGMMOutput is asked for */

/* The reference's two-thread organisation (WFSTDecoderLiteThreading + HTKFlatModelsThreading): the
 * frame loop of jo_decode_utt over a search thread and a scoring thread.  Same results as
 * jo_decode_utt; wall_seconds is wall-clock time with two busy cores. */
int jo_decode_utt_threading(jo_dec *d, const float *feats, int32_t T, jo_hyp *out, double *wall_seconds);

/* PARTIAL_DECODING (WFSTDecoderLite.cpp:822-896): jo_set_partial_interval = setPartialDecodeOptions
 * (the reference reads it from the environment variable PartialTraceInterval, :116-119);
 * jo_trace_partial = tracePartialPath on the current state (the frame loop calls it on the
 * reference's schedule, :362-368); jo_partial_get = the partialPaths list, oldest first. */
int jo_set_partial_interval(jo_dec *d, int32_t interval);
int jo_trace_partial(jo_dec *d);
int jo_partial_get(jo_dec *d, int32_t *n, const int32_t **labels, const int32_t **times);
/* The reference's Path bookkeeping behind its collection schedule (WFSTDecoderLite.cpp:358-370): out[0] = collectPaths
 * calls so far in this utterance, out[1] = lastPathCollectFrame, out[2] = nPath (live Path objects as the reference's
 * allocator counts them), out[3] = nPathNew (their number right after the last collection). */
int jo_path_counts(const jo_dec *d, int64_t out[4]);

/* equal-score recombinations of the last utterance by kind: [0] bestFinalToken, [1] entry token,
 * [2] HMM-internal (lowest predecessor wins: not order dependent), [3] entry-token ties whose
 * two tokens differ in acoustic / LM score or history (the only ones whose winner matters) */
int jo_tie_breakdown(const jo_dec *d, int64_t out[4]);
/* test aid: 0 = reference rule (first visited token keeps an equal-score recombination),
 * 1 = last visited wins.  jo_stats.ties counts the order-dependent ties only ([0] + [3]). */
int jo_dec_set_tie_mode(jo_dec *d, int mode);

/* host libm expf, elementwise */
int jo_expf_array(const float *x, int64_t n, float *out);

const char *jo_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
