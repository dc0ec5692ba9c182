// jd_internal.h - shared between the host-side model/network preparation
// (jd_host.cpp) and the HIP decoder (jd_device.hip).  Not part of the C ABI.
#ifndef JD_INTERNAL_H
#define JD_INTERNAL_H

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "juicer_amd.h"

#define JD_MAXN 8          // max HMM states (incl. entry/exit) the search kernel handles

// 16-byte arc record: one coalesced load per visited arc (WFSTTransition,
// WFSTNetwork.h:41-52, minus id/hook).
struct JdArc { int32_t to; float w; int32_t in; int32_t out; };

struct jd_net {
    int32_t n_states = 0, init = 0, n_final = 0;
    int64_t n_arcs = 0;
    std::vector<int32_t> row_ptr;      // n_states+1 (CSR)
    std::vector<JdArc> arcs;           // sorted by source state, file order within a state
    std::vector<float> fin_w;          // per state; +inf when not final
    int32_t max_in = 0;
    // the scaling the arc weights carry (transWeightScalingFactor / insPenalty of
    // WFSTNetwork.h); jd_net_save_jwnt removes it again the way writeBinary does
    float lm_scale = 1.0f, ins_penalty = 0.0f;
    // search-driven composition (jd_net_create_lazy, jd_compose.hip): the graph lives on one device and grows
    // while decoders search it; n_states / n_arcs are then its CAPACITIES and row_ptr / arcs / fin_w are empty
    void *lazy_dev = nullptr;          // LazyDev (device copy)
    int lazy_device = -1;
    std::vector<void *> lazy_allocs;   // device allocations behind it
    void (*lazy_free)(jd_net *) = nullptr;
    void *lazy_tee = nullptr, *lazy_ok = nullptr;      // device: tee log-probabilities by HMM, the init kernel's answer
    uint32_t lazy_cf0 = 0; int32_t lazy_g0 = 0;        // the start pair
    // bounded look-ahead memory (jd_lazy_enter / jd_lazy_leave, jd_compose.hip): utterances inside the network, arena
    // generations so far, and the fill (of states or arcs) beyond which the arena starts again between utterances
    mutable std::mutex lazy_mu;
    mutable int lazy_busy = 0;
    mutable int64_t lazy_generation = 0;
    double lazy_high_water = 0.9;
};

// Utterances enter and leave a lazily composed network (no-ops on ordinary ones).  jd_lazy_enter starts a new arena
// GENERATION - everything expanded so far is dropped - when nobody is inside an utterance and the arena is past its
// high-water mark or has run out of room; *failed = the network is (still) out of room, i.e. others are inside it.
int jd_lazy_enter(const jd_net *n, int n_utts, bool *failed);
void jd_lazy_leave(const jd_net *n, int n_utts);

struct jd_am {
    int32_t D = 0, n_gmm = 0, max_mix = 0, n_hmm = 0, max_n = 0, n_tm = 0;
    std::vector<int32_t> n_mix;
    std::vector<float> det, mean, ivar;           // reference layout [g][m], [g][m][D]
    std::vector<int32_t> hmm_n, hmm_gmm, hmm_tm;  // [h], [h][max_n], [h]
    std::vector<float> hmm_tee;
    std::vector<int32_t> tm_n;
    std::vector<float> trP;                       // [tm][max_n][max_n]
    std::vector<int16_t> se;                      // [tm][max_n][2]
    // HTK-level values behind det / trP, kept for jd_am_save_jmbi (HTKModels::output(.., true)):
    std::vector<float> var, weight;               // [g][m][D], [g][m]
    std::vector<float> sum_log_var, log_weight;   // [g][m]  VarVec::sumLogVarPlusNObsLog2Pi, GMM::logCompWeights
    std::vector<float> transp;                    // [tm][max_n][max_n]  a_ij
    // hybrid (ANN / LNA) scoring, HTKModels::Load(phones, priors, statesPerModel) HTKModels.cpp:74-218:
    // the feature vector holds one log posterior per model, output = x[model] - log prior (:481-512)
    bool hybrid = false;
    std::vector<float> log_prior;                 // [n_hmm]
};

int jd_fail(int code, const char *fmt, ...);      // sets jd_last_error(), returns code

// Development knobs (planner constants, forced code paths for the toy-size tests, experiments) are read from the
// environment only when JD_DEV=1 is exported with them: nothing a caller NEEDS is selected by an environment variable -
// the header is the interface (jd_dec_set_pipeline, jd_dec_set_capacity, ...).  What stays readable without the gate is
// operational: JD_VERBOSE (diagnostics on stderr), JD_GPU_LOCK / JD_GPU_LOCK_DIR (the per-GPU file lock between
// processes), JD_ARENA_CACHE (keep a destroyed decoder's arena slab for the next one).
const char *jd_dev_env(const char *name);

// The resident search kernel of a broker (jd_device.hip, jd_resident.h) - internal: jd_broker.cpp drives it.
int jd_res_start(jd_dec *d, int n_streams, int rows_per_buf);
int jd_res_stop(jd_dec *d);
int jd_res_cluster(const jd_dec *d);
int jd_res_should_yield(const jd_dec *d);
int jd_res_yield(jd_dec *d);
long long jd_res_collections(const jd_dec *d);
long long jd_res_run_us(const jd_dec *d);          // (statistics) device time the clusters spent on their commands
int jd_res_init(jd_dec *d, int s);
int jd_res_stage_many(jd_dec *d, int n, const int *streams, const int *bufs, const float *const *frames, const int *n_frames);
int jd_res_post(jd_dec *d, int s, int buf, int n_frames);
int jd_res_poll(jd_dec *d, int s, int *idle, int *frame, int *error, int *stopped);
int jd_res_stream_error(jd_dec *d, int s, int dev_error, int frame);   // a poll's device-side error as code + jd_last_error()
int jd_res_collect(jd_dec *d, int s);
int jd_res_finish(jd_dec *d, int s, jd_hyp *out);
#endif
