/*
 * juicer_oracle.c - CPU ORACLE (test infrastructure only; see juicer_oracle.h).
 *
 * PARITY UNPINNED: no reference tests / golden vectors exist for this path, and the
 * reference cannot be built here in a way that pins anything - its hot-path sources DO
 * compile against stand-ins for the absent Torch3 / Tracter headers (tools/refbase does
 * that, in the build container, as a timing and differential aid: on the first 12
 * configs[1] utterances the reference's own WFSTDecoderLite and WFSTDecoderLiteThreading
 * give this file's words, times and scores bit for bit, profiles/cpu_reference_baseline.json),
 * but a build against stand-ins is not a reference build.  Plain C restatement of:
 *   WFSTNetwork text-load arithmetic      src/WFSTNetwork.cpp:403-560, 709-721
 *   HTKModels parameter preparation       src/HTKModels.cpp:581-593, 600-676, 835-870, 873-974, 2330-2390
 *   HTKFlatModels flatten + GMM + logAdd  src/HTKFlatModels.cpp:94-177, 226-306
 *   Histogram                             src/Histogram.cpp:23-56, 64-120, 134-158
 *   WFSTDecoderLite                       src/WFSTDecoderLite.cpp:139-228, 230-309, 311-605, 751-805, 822-896, 899-982
 *   DecoderSingleTest frame protocol      src/DecoderSingleTest.cpp:259-298
 * Compile with:  gcc -O2 -ffp-contract=off  (no -march=native, no -ffast-math)
 * so that every float operation rounds exactly once, in the reference's order.
 *
 * Deliberate omissions that do not change results: freeing Path records (collectPaths,
 * WFSTDecoderLite.cpp:699-747, only frees unreachable ones; WHEN it runs - both triggers of :362,
 * with the allocator's live count it reads - is kept for the PARTIAL_DECODING trace, see
 * jo_process_frame), LogFile.
 */
#include "juicer_oracle.h"

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define LZ (-FLT_MAX)                       /* Torch3 LOG_ZERO = -INF = -FLT_MAX */
#define LOG_2_PI 1.83787706640934548355     /* Torch3 log_add.h */
#define MINUS_LOG_THRESHOLD (-18.42)        /* HTKFlatModels.cpp:60 (float build) */

static __thread char g_err[512];
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
const char *jo_last_error(void) { return g_err; }

/* ======================================================================= net */

struct jo_net {
    int32_t n_states, init, n_final;
    int64_t n_arcs;
    int32_t *first, *cnt;          /* per state: first arc (file order) + count   */
    int32_t *to, *in, *out;        /* per arc                                     */
    float *w;
    int32_t *final_ind;            /* per state, -1 if not final                  */
    float *final_w;                /* per final entry                             */
};

void jo_net_destroy(jo_net *n)
{
    if (!n) return;
    free(n->first); free(n->cnt); free(n->to); free(n->in); free(n->out);
    free(n->w); free(n->final_ind); free(n->final_w); free(n);
}

static jo_net *net_alloc(int32_t n_states, int64_t n_arcs, int32_t n_final)
{
    jo_net *n = (jo_net *)calloc(1, sizeof *n);
    n->n_states = n_states; n->n_arcs = n_arcs; n->n_final = n_final;
    n->first = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_states > 0 ? n_states : 1));
    n->cnt = (int32_t *)calloc((size_t)(n_states > 0 ? n_states : 1), sizeof(int32_t));
    n->to = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_arcs > 0 ? n_arcs : 1));
    n->in = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_arcs > 0 ? n_arcs : 1));
    n->out = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_arcs > 0 ? n_arcs : 1));
    n->w = (float *)malloc(sizeof(float) * (size_t)(n_arcs > 0 ? n_arcs : 1));
    n->final_ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_states > 0 ? n_states : 1));
    n->final_w = (float *)malloc(sizeof(float) * (size_t)(n_final > 0 ? n_final : 1));
    for (int32_t i = 0; i < n_states; ++i) { n->first[i] = 0; n->final_ind[i] = -1; }
    return n;
}

/* WFSTNetwork::WFSTNetwork(text), WFSTNetwork.cpp:403-560 */
int jo_net_create_arcs(jo_net **out, int64_t n_arcs, const int32_t *from, const int32_t *to,
                       const int32_t *in, const int32_t *outl, const float *w_file,
                       int32_t n_final, const int32_t *fstate, const float *fweight_file,
                       float lm_scale, float ins_penalty)
{
    if (!out || n_arcs <= 0) return fail(-1, "jo_net_create_arcs: no arcs");
    int32_t max_state = -1;
    for (int64_t i = 0; i < n_arcs; ++i) {
        /* :452-453 "something < 0" */
        if (from[i] < 0 || to[i] < 0 || in[i] < 0 || outl[i] < 0)
            return fail(-1, "WFSTNetwork - something < 0. %d %d %d %d", from[i], to[i], in[i], outl[i]);
        if (from[i] > max_state) max_state = from[i];   /* :458-461 */
        if (to[i] > max_state) max_state = to[i];
    }
    jo_net *n = net_alloc(max_state + 1, n_arcs, n_final);
    n->init = from[0];                                   /* :455-456 init = source of first line */
    for (int64_t i = 0; i < n_arcs; ++i) {
        n->to[i] = to[i]; n->in[i] = in[i]; n->out[i] = outl[i];
        /* :481-486  weight = (real)(-weight * scale); if (out > 0) weight += insPenalty */
        float w = (float)(-w_file[i] * lm_scale);
        if (outl[i] > 0) w += ins_penalty;
        n->w[i] = w;
        int32_t s = from[i];
        if (n->cnt[s] == 0) n->first[s] = (int32_t)i;
        else if (n->first[s] + n->cnt[s] != i) {
            /* getTransitions (:709-721) returns transitions+trans[0] and nTrans: the
             * reference silently assumes contiguity; we refuse instead. */
            jo_net_destroy(n);
            return fail(-7, "arcs of state %d are not contiguous (arc %lld)", s, (long long)i);
        }
        n->cnt[s]++;
    }
    for (int32_t i = 0; i < n_final; ++i) {              /* :439-442, 539-548 */
        if (fstate[i] < 0 || fstate[i] > max_state) {
            jo_net_destroy(n);
            return fail(-1, "WFSTNetwork - finalState[%d].id out of range", i);
        }
        n->final_w[i] = (float)(-fweight_file[i] * lm_scale);
        n->final_ind[fstate[i]] = i;
    }
    *out = n;
    return 0;
}

int jo_net_create_csr(jo_net **out, int32_t n_states, int32_t init_state,
                      const int32_t *row_ptr, const int32_t *to, const float *w,
                      const int32_t *in, const int32_t *outl,
                      int32_t n_final, const int32_t *fstate, const float *fweight)
{
    if (!out || n_states <= 0 || init_state < 0 || init_state >= n_states)
        return fail(-1, "jo_net_create_csr: bad arguments");
    int64_t n_arcs = row_ptr[n_states];
    jo_net *n = net_alloc(n_states, n_arcs, n_final);
    n->init = init_state;
    for (int32_t s = 0; s < n_states; ++s) { n->first[s] = row_ptr[s]; n->cnt[s] = row_ptr[s + 1] - row_ptr[s]; }
    memcpy(n->to, to, sizeof(int32_t) * (size_t)n_arcs);
    memcpy(n->in, in, sizeof(int32_t) * (size_t)n_arcs);
    memcpy(n->out, outl, sizeof(int32_t) * (size_t)n_arcs);
    memcpy(n->w, w, sizeof(float) * (size_t)n_arcs);
    for (int32_t i = 0; i < n_final; ++i) {
        if (fstate[i] < 0 || fstate[i] >= n_states) { jo_net_destroy(n); return fail(-1, "final state out of range"); }
        n->final_w[i] = fweight[i];
        n->final_ind[fstate[i]] = i;
    }
    *out = n;
    return 0;
}

int64_t jo_net_num_arcs(const jo_net *n) { return n->n_arcs; }
int32_t jo_net_num_states(const jo_net *n) { return n->n_states; }
int32_t jo_net_init_state(const jo_net *n) { return n->init; }

int jo_net_get(const jo_net *n, int32_t *first, int32_t *cnt, int32_t *to, float *w,
               int32_t *in, int32_t *outl, int32_t *final_ind, float *final_w)
{
    if (first) memcpy(first, n->first, sizeof(int32_t) * (size_t)n->n_states);
    if (cnt) memcpy(cnt, n->cnt, sizeof(int32_t) * (size_t)n->n_states);
    if (to) memcpy(to, n->to, sizeof(int32_t) * (size_t)n->n_arcs);
    if (w) memcpy(w, n->w, sizeof(float) * (size_t)n->n_arcs);
    if (in) memcpy(in, n->in, sizeof(int32_t) * (size_t)n->n_arcs);
    if (outl) memcpy(outl, n->out, sizeof(int32_t) * (size_t)n->n_arcs);
    if (final_ind) memcpy(final_ind, n->final_ind, sizeof(int32_t) * (size_t)n->n_states);
    if (final_w) memcpy(final_w, n->final_w, sizeof(float) * (size_t)n->n_final);
    return 0;
}

/* ======================================================================== am */

struct jo_am {
    int32_t D, n_gmm, max_mix, n_hmm, max_n, n_tm;
    int32_t *n_mix;
    float *det, *mean, *ivar;          /* [g][m], [g][m][D], [g][m][D]              */
    int32_t *hmm_n, *hmm_gmm, *hmm_tm; /* [h], [h][max_n], [h]                      */
    float *hmm_tee;                    /* [h]                                       */
    int32_t *tm_n;
    float *trP;                        /* [tm][max_n][max_n]  trP[i][j]             */
    int16_t *se;                       /* [tm][max_n][2]  start,end for state j     */
    float *tm_tee;
    int hybrid; float *log_prior;      /* hybridMode, logPriors (HTKModels.cpp:65-66, 163-183) */
};

void jo_am_destroy(jo_am *a)
{
    if (!a) return;
    free(a->n_mix); free(a->det); free(a->mean); free(a->ivar); free(a->hmm_n); free(a->hmm_gmm);
    free(a->hmm_tm); free(a->hmm_tee); free(a->tm_n); free(a->trP); free(a->se); free(a->tm_tee); free(a->log_prior); free(a);
}

int jo_am_create_htk(jo_am **out, int32_t D, int32_t n_gmm, int32_t max_mix,
                     const int32_t *n_mix, const float *weight, const float *mean, const float *var,
                     int32_t n_hmm, int32_t max_n, const int32_t *hmm_nstates,
                     const int32_t *hmm_gmm, const int32_t *hmm_tm,
                     int32_t n_tm, const int32_t *tm_nstates, const float *transp)
{
    if (!out || D <= 0 || n_gmm <= 0 || max_mix <= 0 || n_hmm <= 0 || max_n < 3 || n_tm <= 0)
        return fail(-1, "jo_am_create_htk: bad sizes");
    jo_am *a = (jo_am *)calloc(1, sizeof *a);
    a->D = D; a->n_gmm = n_gmm; a->max_mix = max_mix; a->n_hmm = n_hmm; a->max_n = max_n; a->n_tm = n_tm;
    size_t gm = (size_t)n_gmm * max_mix;
    a->n_mix = (int32_t *)malloc(sizeof(int32_t) * n_gmm);
    a->det = (float *)malloc(sizeof(float) * gm);
    a->mean = (float *)malloc(sizeof(float) * gm * D);
    a->ivar = (float *)malloc(sizeof(float) * gm * D);
    a->hmm_n = (int32_t *)malloc(sizeof(int32_t) * n_hmm);
    a->hmm_gmm = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_hmm * max_n);
    a->hmm_tm = (int32_t *)malloc(sizeof(int32_t) * n_hmm);
    a->hmm_tee = (float *)malloc(sizeof(float) * n_hmm);
    a->tm_n = (int32_t *)malloc(sizeof(int32_t) * n_tm);
    a->trP = (float *)malloc(sizeof(float) * (size_t)n_tm * max_n * max_n);
    a->se = (int16_t *)calloc((size_t)n_tm * max_n * 2, sizeof(int16_t));
    a->tm_tee = (float *)malloc(sizeof(float) * n_tm);

    for (int32_t g = 0; g < n_gmm; ++g) {
        int32_t nm = n_mix[g];
        if (nm < 1 || nm > max_mix) { jo_am_destroy(a); return fail(-1, "n_mix[%d] out of range", g); }
        a->n_mix[g] = nm;
        for (int32_t m = 0; m < max_mix; ++m) {
            size_t gi = (size_t)g * max_mix + m;
            if (m >= nm) {
                a->det[gi] = LZ;
                for (int32_t k = 0; k < D; ++k) { a->mean[gi * D + k] = 0.0f; a->ivar[gi * D + k] = 0.0f; }
                continue;
            }
            /* HTKModels::addVarVec :857-866 */
            float acc = (float)(D * LOG_2_PI);
            for (int32_t k = 0; k < D; ++k) {
                float v = var[gi * D + k];
                acc += logf(v);                      /* log(float) -> logf overload */
                /* HTKFlatModels::init :159-160 */
                a->mean[gi * D + k] = mean[gi * D + k];
                a->ivar[gi * D + k] = (float)(1.0 / v);
            }
            acc *= -0.5;                             /* float *= double literal */
            /* HTKModels::addGMM :657-663 log weights, HTKFlatModels::init :169-176 */
            float wgt = weight[gi];
            float lw = (wgt > 0.0) ? logf(wgt) : LZ;
            a->det[gi] = acc + lw;
        }
        /* addGMM :665 */
        if (nm == 1 && weight[(size_t)g * max_mix] != 1.0f) {
            jo_am_destroy(a);
            return fail(-1, "HTKModels::addGMM - (n_mixes == 1) && (compWeights[0] != 1.0)");
        }
    }

    for (int32_t t = 0; t < n_tm; ++t) {
        int32_t n = tm_nstates[t];
        if (n < 3 || n > max_n) { jo_am_destroy(a); return fail(-1, "tm_nstates[%d] out of range", t); }
        a->tm_n[t] = n;
        float *trP = a->trP + (size_t)t * max_n * max_n;
        const float *tp = transp + (size_t)t * max_n * max_n;
        /* createTrPandSEIndex :2349-2364 (via addTransMatrix :941-951 logProbs = log(a_ij)) */
        for (int32_t i = 0; i < max_n; ++i)
            for (int32_t j = 0; j < max_n; ++j) trP[i * max_n + j] = LZ;
        for (int32_t i = 0; i < n; ++i)
            for (int32_t j = 0; j < n; ++j)
                if (tp[i * max_n + j] > 0.0) trP[i * max_n + j] = logf(tp[i * max_n + j]);
        /* SEIndex :2368-2387 */
        int16_t *se = a->se + (size_t)t * max_n * 2;
        for (int32_t j = 1; j < n; ++j) {
            int32_t mn, mx;
            for (mn = (j == n - 1 ? 1 : 0); mn < n - 1; ++mn)
                if (trP[mn * max_n + j] > LZ) break;
            for (mx = n - 1; mx >= 1; --mx)
                if (trP[mx * max_n + j] > LZ) break;
            se[j * 2] = (int16_t)mn;
            se[j * 2 + 1] = (int16_t)(mx + 1);
        }
        /* addHMM :581-593 tee: successor list of state 0 scanned from index 1 */
        float tee = LZ;
        int32_t suc_idx = 0;
        for (int32_t j = 0; j < n; ++j) {
            if (tp[0 * max_n + j] > 0.0) {
                if (suc_idx >= 1 && j == n - 1) tee = trP[0 * max_n + j];
                ++suc_idx;
            }
        }
        a->tm_tee[t] = tee;
    }

    for (int32_t h = 0; h < n_hmm; ++h) {
        int32_t n = hmm_nstates[h];
        int32_t t = hmm_tm[h];
        if (t < 0 || t >= n_tm || a->tm_n[t] != n) { jo_am_destroy(a); return fail(-1, "HTKModels::addHMM - nStates != transmat n_states (hmm %d)", h); }
        a->hmm_n[h] = n; a->hmm_tm[h] = t; a->hmm_tee[h] = a->tm_tee[t];
        for (int32_t j = 0; j < max_n; ++j) {
            int32_t g = (j >= 1 && j < n - 1) ? hmm_gmm[(size_t)h * max_n + j] : -1;
            if (j >= 1 && j < n - 1 && (g < 0 || g >= n_gmm)) { jo_am_destroy(a); return fail(-1, "hmm %d state %d: bad gmm", h, j); }
            a->hmm_gmm[(size_t)h * max_n + j] = g;
        }
    }
    *out = a;
    return 0;
}

/* HTKModels::Load(phonesListFName, priorsFName, statesPerModel), HTKModels.cpp:74-218 (hybrid ANN / HMM) */
int jo_am_create_hybrid(jo_am **out, int32_t n_phones, const float *priors, int32_t states_per_model)
{
    if (!out || !priors || n_phones <= 0) return fail(-1, "jo_am_create_hybrid: bad argument");
    if (states_per_model <= 2) return fail(-1, "HTKModels::Models(3) - statesPerModel <= 2 (ie. no emitting states)");
    const int32_t P = n_phones, N = states_per_model;
    int32_t *n_mix = (int32_t *)malloc(sizeof(int32_t) * P), *hmm_n = (int32_t *)malloc(sizeof(int32_t) * P);
    int32_t *hmm_gmm = (int32_t *)malloc(sizeof(int32_t) * (size_t)P * N), *hmm_tm = (int32_t *)calloc(P, sizeof(int32_t));
    float *weight = (float *)malloc(sizeof(float) * P), *mean = (float *)calloc((size_t)P * P, sizeof(float));
    float *var = (float *)malloc(sizeof(float) * (size_t)P * P), *transp = (float *)calloc((size_t)N * N, sizeof(float));
    for (int32_t h = 0; h < P; ++h) {
        n_mix[h] = 1; hmm_n[h] = N; weight[h] = 1.0f;
        for (int32_t j = 0; j < N; ++j) hmm_gmm[(size_t)h * N + j] = (j >= 1 && j < N - 1) ? h : -1;   /* :176-181 */
    }
    for (size_t i = 0; i < (size_t)P * P; ++i) var[i] = 1.0f;          /* (placeholders: hybrid mode reads no Gaussian) */
    transp[1] = 1.0f;                                                   /* :196-199 */
    for (int32_t i = 1; i < N - 1; ++i) { transp[i * N + i] = 0.5f; transp[i * N + i + 1] = 0.5f; }   /* :200-204 */
    int32_t tm_n = N;
    int rc = jo_am_create_htk(out, P, P, 1, n_mix, weight, mean, var, P, N, hmm_n, hmm_gmm, hmm_tm, 1, &tm_n, transp);
    free(n_mix); free(hmm_n); free(hmm_gmm); free(hmm_tm); free(weight); free(mean); free(var); free(transp);
    if (rc) return rc;
    (*out)->hybrid = 1;
    (*out)->log_prior = (float *)malloc(sizeof(float) * P);
    for (int32_t h = 0; h < P; ++h) (*out)->log_prior[h] = logf(priors[h]);   /* :183 log(float) */
    return 0;
}

int jo_am_get_flat(const jo_am *a, float *det, float *mean, float *ivar)
{
    size_t gm = (size_t)a->n_gmm * a->max_mix;
    if (det) memcpy(det, a->det, sizeof(float) * gm);
    if (mean) memcpy(mean, a->mean, sizeof(float) * gm * a->D);
    if (ivar) memcpy(ivar, a->ivar, sizeof(float) * gm * a->D);
    return 0;
}

int jo_am_get_trans(const jo_am *a, float *trP, int16_t *se, float *tee)
{
    if (trP) memcpy(trP, a->trP, sizeof(float) * (size_t)a->n_tm * a->max_n * a->max_n);
    if (se) memcpy(se, a->se, sizeof(int16_t) * (size_t)a->n_tm * a->max_n * 2);
    if (tee) memcpy(tee, a->hmm_tee, sizeof(float) * (size_t)a->n_hmm);
    return 0;
}

/* HTKFlatModels::logAdd, HTKFlatModels.cpp:266-293 (float build, no FAST_* options) */
static float log_add(float x, float y)
{
    if (x < y) { float t = x; x = y; y = t; }
    float diff = y - x;
    if (diff < MINUS_LOG_THRESHOLD) return x;
    /* "x + log(1.0 + exp(diff))": exp on a float argument binds to the float
     * overload (expf); 1.0 + float -> double; log in double; sum in double;
     * converted to real on return. */
    return (float)((double)x + log(1.0 + (double)expf(diff)));
}

/* inner loops of HTKFlatModels::calcGMMOutput :239-256 for one (gmm, frame) */
static float gmm_one(const jo_am *a, int32_t g, const float *x)
{
    if (a->hybrid) return x[g] - a->log_prior[g];                   /* calcOutput, HTKFlatModels.cpp:190-197, 216-220 */
    int32_t D = a->D, nMix = a->n_mix[g];
    const float *means = a->mean + (size_t)g * a->max_mix * D;
    const float *vars = a->ivar + (size_t)g * a->max_mix * D;
    const float *dets = a->det + (size_t)g * a->max_mix;
    float logProb = LZ;
    for (int32_t i = 0; i < nMix; ++i) {
        float sumxmu = 0.0f;
        for (int32_t j = 0; j < D; ++j) {
            float xmu = x[j] - means[j];
            sumxmu += xmu * xmu * vars[j];
        }
        means += D; vars += D;
        logProb = log_add(logProb, (float)(-0.5 * sumxmu + dets[i]));
    }
    return logProb;
}

int jo_am_score_frames(const jo_am *a, const float *frames, int32_t n_frames, float *out)
{
    for (int32_t t = 0; t < n_frames; ++t)
        for (int32_t g = 0; g < a->n_gmm; ++g)
            out[(size_t)t * a->n_gmm + g] = gmm_one(a, g, frames + (size_t)t * a->D);
    return 0;
}

/* ================================================================= histogram */

typedef struct {
    int nBins, minScore, maxScore, count;
    int *cnt;
} Hist;

/* Histogram::Histogram(1, minScore_, maxScore_) Histogram.cpp:23-56 (binWidth 1) */
static Hist *hist_new(float minScore_, float maxScore_)
{
    Hist *h = (Hist *)calloc(1, sizeof *h);
    h->minScore = (int)(minScore_ - 1.0);
    h->maxScore = (int)(maxScore_ + 1.0);
    h->nBins = h->maxScore - h->minScore + 1;
    h->cnt = (int *)calloc((size_t)h->nBins, sizeof(int));
    h->count = 0;
    return h;
}
static void hist_reset(Hist *h) { h->count = 0; memset(h->cnt, 0, sizeof(int) * (size_t)h->nBins); }
/* Histogram::addScore :64-100 ; returns -5 on the reference's fatal error */
static int hist_add(Hist *h, float score)
{
    int sc;
    if (score < 0.0) sc = (int)(score - 0.5); else sc = (int)(score + 0.5);
    if (sc > h->maxScore) return -5;
    if (sc < h->minScore) return 0;
    h->cnt[sc - h->minScore]++;
    h->count++;
    return 0;
}
/* Histogram::calcThresh :134-158 */
static float hist_thresh(const Hist *h, int maxN)
{
    int total = 0;
    if (h->count <= maxN) return (float)((float)(h->minScore) - 0.5);
    for (int i = h->nBins - 1; i >= 0; --i) {
        total += h->cnt[i];
        if (total >= maxN) return (float)((float)(i + h->minScore) - 0.5);
    }
    return (float)h->minScore;
}

/* =================================================================== decoder */

typedef struct { float score, ac, lm; int32_t path; } Tok;     /* Token, WFSTDecoderLite.h:58-63 */
static const Tok NULLTOK = {LZ, LZ, LZ, -1};                    /* nullToken, .cpp:35 */
typedef struct { int32_t next, hmm, n, nact, arc; float tee; } Inst;   /* NetInst, .h:66-75 */
typedef struct { int32_t prev, frame, label; float score, ac, lm; } PathRec; /* Path, .h:39-56 */

struct jo_dec {
    const jo_net *net; const jo_am *am;
    float startWin, emitWin, endWin, wordWin; int maxHyps, fnBlock;
    Hist *hist;
    int32_t *hook;                        /* WFSTTransition::hook, per arc        */
    Inst *insts; Tok *toks; int32_t n_insts, cap_insts, maxN;
    PathRec *paths; int64_t n_paths, cap_paths;
    /* live Path objects as the reference's allocator counts them (nPath: ++ in createNewNoRefPath :618, -- in
     * destroyPath :626, i.e. in collectPaths only) and their number right after the last collection (nPathNew :745) */
    int64_t nPath, nPathNew, paths_at_collect;
    unsigned char *pmark; int64_t cap_pmark;
    int32_t n_collections;
    int32_t active, newActive, newActiveLast;
    Tok *tokenBuf; Tok bestFinal;
    float normaliseScore, bestEmitScore;
    float startTh, endTh, wordTh, emitTh;
    int32_t currFrame, nActiveInsts, nActiveEmitHyps, nActiveEndHyps, nEmitProc, nEndProc;
    jo_stats st;
    int tie_mode;                         /* 0 = reference (first visited wins), 1 = last visited wins (test aid) */
    /* PARTIAL_DECODING (WFSTDecoderLite.h:199-205) */
    int32_t partialTraceInterval, lastPartialTraceFrame, lastPathCollectFrame;
    int32_t *partialPaths; int32_t n_partial, cap_partial;   /* vector<Path*> partialPaths, as Path indices */
    int32_t *jointCount; int64_t cap_joint;                  /* Path::jointCount */
    int32_t *p_label, *p_time;                               /* jo_partial_get view */
    int64_t tie_kind[4];                  /* bestFinal, entry token, HMM-internal, entry ties whose tokens differ */
    /* HTKFlatModels cache state */
    int32_t *cacheT; float *cache; const float *const *currInput; int32_t currInputLen, amFrame;
    int err, started;
    /* result */
    int32_t *r_label, *r_time; float *r_score, *r_ac, *r_lm; int32_t r_cap;
    float *trace; int32_t trace_cap;
    uint8_t *cells; int32_t cells_frames;   /* jo_set_cells: cells[frame * n_gmm + g] = 1 for every calcGMMOutput(g) call of that frame */
};

void jo_dec_destroy(jo_dec *d)
{
    if (!d) return;
    if (d->hist) { free(d->hist->cnt); free(d->hist); }
    free(d->hook); free(d->insts); free(d->toks); free(d->paths); free(d->tokenBuf); free(d->pmark);
    free(d->cacheT); free(d->cache);
    free(d->r_label); free(d->r_time); free(d->r_score); free(d->r_ac); free(d->r_lm);
    free(d->partialPaths); free(d->jointCount); free(d->p_label); free(d->p_time);
    free(d);
}

/* WFSTDecoderLite::WFSTDecoderLite, WFSTDecoderLite.cpp:38-120 */
int jo_dec_create(jo_dec **out, const jo_net *net, const jo_am *am,
                  float start_beam, float main_beam, float end_beam, float word_beam,
                  int32_t max_hyps, int32_t block_size)
{
    if (!out || !net || !am) return fail(-1, "jo_dec_create: null argument");
    if (block_size < 1 || block_size > 20)   /* HTKFlatModels::setBlockSize :311-312 */
        return fail(-1, "HTKFlatModels::setBlockSize fnBlock should be in [1, 20]");
    for (int64_t i = 0; i < net->n_arcs; ++i)
        if (net->in[i] > am->n_hmm) return fail(-1, "arc %lld: inLabel %d > number of HMMs", (long long)i, net->in[i]);
    jo_dec *d = (jo_dec *)calloc(1, sizeof *d);
    d->net = net; d->am = am;
    d->startWin = start_beam; d->emitWin = main_beam; d->endWin = end_beam; d->wordWin = word_beam;
    d->maxHyps = max_hyps; d->fnBlock = block_size;
    if (max_hyps > 0) {                       /* :76-82 */
        if (main_beam > 0.0) d->hist = hist_new((float)(-main_beam - 800.0), 200.0f);
        else d->hist = hist_new(-1000.0f, 200.0f);
    }
    d->maxN = am->max_n;
    d->hook = (int32_t *)malloc(sizeof(int32_t) * (size_t)net->n_arcs);
    for (int64_t i = 0; i < net->n_arcs; ++i) d->hook[i] = -1;
    d->cap_insts = 4096;
    d->insts = (Inst *)malloc(sizeof(Inst) * (size_t)d->cap_insts);
    d->toks = (Tok *)malloc(sizeof(Tok) * (size_t)d->cap_insts * d->maxN);
    d->cap_paths = 1 << 16;
    d->paths = (PathRec *)malloc(sizeof(PathRec) * (size_t)d->cap_paths);
    d->tokenBuf = (Tok *)malloc(sizeof(Tok) * (size_t)d->maxN);
    d->tokenBuf[0] = NULLTOK;                /* :107-108 */
    d->active = d->newActive = d->newActiveLast = -1;
    d->cacheT = (int32_t *)malloc(sizeof(int32_t) * (size_t)am->n_gmm);
    d->cache = (float *)malloc(sizeof(float) * (size_t)am->n_gmm * block_size);
    d->amFrame = -1;
    *out = d;
    return 0;
}

/* diagnostics (tools/demand_stats.py): which (frame, tied state) cells does the search ask calcGMMOutput for?  cells: frames x n_gmm
 * bytes, the caller's (or NULL: off); marked whether the value comes out of the block cache or is computed */
int jo_set_cells(jo_dec *d, uint8_t *cells, int32_t frames)
{
    if (!d) return fail(-1, "jo_set_cells: null");
    d->cells = cells; d->cells_frames = cells ? frames : 0;
    return 0;
}

int jo_set_trace(jo_dec *d, float *best_emit_per_frame, int32_t cap)
{
    d->trace = best_emit_per_frame; d->trace_cap = cap;
    return 0;
}

/* HTKFlatModels::newFrame, HTKFlatModels.cpp:295-306 */
static int am_new_frame(jo_dec *d, int32_t frame, const float *const *input, int32_t nData)
{
    if (frame > 0 && frame != d->amFrame + 1) return fail(-6, "HTKFlatModels::newFrame - invalid frame");
    d->amFrame = frame; d->currInput = input; d->currInputLen = nData;
    if (frame == 0)
        for (int32_t i = 0; i < d->am->n_gmm; ++i) d->cacheT[i] = -1000;
    return 0;
}

/* HTKFlatModels::calcGMMOutput, HTKFlatModels.cpp:226-262 (block cache) */
static float am_calc_gmm(jo_dec *d, int32_t g)
{
    if (d->cells && d->amFrame >= 0 && d->amFrame < d->cells_frames) d->cells[(size_t)d->amFrame * d->am->n_gmm + g] = 1;
    int32_t n = d->amFrame - d->cacheT[g];
    if (n < d->fnBlock) return d->cache[(size_t)g * d->fnBlock + n];
    int32_t m = d->currInputLen < d->fnBlock ? d->currInputLen : d->fnBlock;
    for (int32_t k = 0; k < m; ++k)
        d->cache[(size_t)g * d->fnBlock + k] = gmm_one(d->am, g, d->currInput[k]);
    d->cacheT[g] = d->amFrame;
    return d->cache[(size_t)g * d->fnBlock];
}

/* attachNetInst, WFSTDecoderLite.cpp:751-774 (pool = growable arrays here) */
static int32_t attach_inst(jo_dec *d, int32_t arc)
{
    if (d->n_insts == d->cap_insts) {
        d->cap_insts *= 2;
        d->insts = (Inst *)realloc(d->insts, sizeof(Inst) * (size_t)d->cap_insts);
        d->toks = (Tok *)realloc(d->toks, sizeof(Tok) * (size_t)d->cap_insts * d->maxN);
    }
    int32_t id = d->n_insts++;
    Inst *inst = &d->insts[id];
    int32_t hmm = d->net->in[arc] - 1;              /* :754 */
    inst->hmm = hmm; inst->n = d->am->hmm_n[hmm];
    for (int32_t i = 0; i < d->maxN; ++i) d->toks[(size_t)id * d->maxN + i] = NULLTOK;
    d->hook[arc] = id; inst->arc = arc;
    inst->tee = d->am->hmm_tee[hmm];                /* :765 */
    inst->nact = 0;
    inst->next = d->newActive; d->newActive = id;   /* :767-770 prepend */
    if (d->newActiveLast < 0) d->newActiveLast = id;
    ++d->nActiveInsts;
    return id;
}

/* returnNetInst, :777-797; returns the next instance index */
static int32_t return_inst(jo_dec *d, int32_t inst, int32_t prev)
{
    int32_t nxt = d->insts[inst].next;
    if (prev < 0) d->active = nxt; else d->insts[prev].next = nxt;
    for (int32_t i = 0; i < d->insts[inst].n; ++i) d->toks[(size_t)inst * d->maxN + i] = NULLTOK;
    --d->nActiveInsts;
    return nxt;
}

/* joinNewActiveInstList, :799-805 */
static void join_new(jo_dec *d)
{
    if (d->newActive < 0) return;
    d->insts[d->newActiveLast].next = d->active;
    d->active = d->newActive;
    d->newActive = d->newActiveLast = -1;
}

static int32_t new_path(jo_dec *d)
{
    if (d->n_paths == d->cap_paths) {
        d->cap_paths *= 2;
        d->paths = (PathRec *)realloc(d->paths, sizeof(PathRec) * (size_t)d->cap_paths);
    }
    ++d->st.tot_paths;
    ++d->nPath;                                                     /* createNewNoRefPath :618 */
    return (int32_t)d->n_paths++;
}

/* propagateToken, WFSTDecoderLite.cpp:491-605 ; arc < 0 means trans == NULL */
static void propagate(jo_dec *d, Tok *tok, int32_t arc)
{
    const jo_net *net = d->net;
    int32_t state;
    if (arc >= 0) {
        if (net->out[arc] != 0) {                                   /* :497-509 */
            int32_t p = new_path(d);
            PathRec *pr = &d->paths[p];
            pr->frame = d->currFrame; pr->score = tok->score; pr->lm = tok->lm; pr->ac = tok->ac;
            pr->label = net->out[arc]; pr->prev = tok->path;
            tok->path = p;
        }
        int32_t fi = net->final_ind[net->to[arc]];                  /* :513-520 */
        if (fi >= 0) {
            float weight = net->final_w[fi];
            const int ftie = tok->score + weight == d->bestFinal.score && d->bestFinal.score > LZ;
            if (ftie) { ++d->st.ties; ++d->tie_kind[0]; }
            if (tok->score + weight > d->bestFinal.score || (ftie && d->tie_mode)) {
                d->bestFinal = *tok;
                d->bestFinal.score += weight;
                d->bestFinal.lm += weight;
            }
        }
        state = net->to[arc];
    } else state = net->init;

    int32_t nTrans = net->cnt[state], first = net->first[state];    /* :527 */
    for (int32_t it = 0; it < nTrans; ++it) {
        int32_t b = first + it;
        ++d->st.tot_arcs_visited;
        if (net->in[b] == 0) {                                      /* :533-540 */
            Tok tmp = *tok;
            tmp.score += net->w[b];
            tmp.lm += net->w[b];
            if (tmp.score > d->endTh) propagate(d, &tmp, b);
        } else {
            int32_t inst = d->hook[b];                              /* :544-557 */
            if (inst < 0) inst = attach_inst(d, b);
            else if (d->insts[inst].nact == 0) {
                d->insts[inst].next = d->newActive; d->newActive = inst;
                if (d->newActiveLast < 0) d->newActiveLast = inst;
                ++d->nActiveInsts;
            }
            Tok *res = &d->toks[(size_t)inst * d->maxN];            /* :560-582 */
            float newScore = tok->score + net->w[b];
            int etie = 0;
            if (newScore == res->score && newScore > LZ) {
                ++d->tie_kind[1];
                /* order dependent only if the two tokens differ in what they carry */
                if (tok->path != res->path || tok->ac != res->ac || tok->lm + net->w[b] != res->lm) {
                    ++d->tie_kind[3]; ++d->st.ties; etie = 1;
                }
            }
            if (newScore > res->score || (etie && d->tie_mode)) {
                if (res->score <= LZ) ++d->insts[inst].nact;
                *res = *tok;
                res->score = newScore;
                res->lm += net->w[b];
                if (newScore > d->bestEmitScore) d->bestEmitScore = newScore;
            }
            float tee = d->insts[inst].tee;                         /* :584-600 */
            if (tee > LZ) {
                newScore += tee;
                Tok tmp = *tok;
                tmp.score = newScore;
                tmp.ac += tee;
                tmp.lm += net->w[b];
                if (net->out[b] != 0) { if (newScore > d->wordTh) propagate(d, &tmp, b); }
                else { if (newScore > d->endTh) propagate(d, &tmp, b); }
            }
        }
    }
}

/* recognitionStart, WFSTDecoderLite.cpp:139-228 */
int jo_init(jo_dec *d)
{
    d->currFrame = 0;
    d->bestFinal = NULLTOK;
    /* <<Free per-utterance memory>> :148-183.  Instances are dropped wholesale
     * (the reference keeps hooks across utterances unless over maxAllocModels;
     * either way a re-touched instance is PREPENDED to the new-active list, so
     * list order - the only thing hooks could influence - is identical). */
    for (int32_t i = 0; i < d->n_insts; ++i) d->hook[d->insts[i].arc] = -1;
    d->n_insts = 0; d->active = d->newActive = d->newActiveLast = -1;
    d->n_paths = 0;
    d->nPath = d->nPathNew = d->paths_at_collect = 0; d->n_collections = 0;   /* resetPathLists :695 */
    if (d->hist) hist_reset(d->hist);                               /* :186-187 */
    d->normaliseScore = 0.0f; d->bestEmitScore = LZ;                /* :189-191 */
    d->startTh = d->endTh = d->wordTh = d->emitTh = LZ;             /* :197-200 */
    memset(&d->st, 0, sizeof d->st);
    memset(d->tie_kind, 0, sizeof d->tie_kind);
    d->nActiveInsts = d->nActiveEmitHyps = d->nActiveEndHyps = d->nEmitProc = d->nEndProc = 0;
    d->err = 0; d->started = 1; d->amFrame = -1;
    d->n_partial = 0;                                               /* :179-181 */
    d->lastPathCollectFrame = -1; d->lastPartialTraceFrame = -1;    /* :202-206 */
    Tok tmp = {0.0f, 0.0f, 0.0f, -1};                               /* :221-227 */
    propagate(d, &tmp, -1);
    join_new(d);
    return 0;
}

/* HMMInternalPropagation, WFSTDecoderLite.cpp:376-484 */
static void hmm_internal(jo_dec *d, int32_t ii)
{
    Inst *inst = &d->insts[ii];
    Tok *states = &d->toks[(size_t)ii * d->maxN];
    const jo_am *am = d->am;
    int32_t maxN = am->max_n, N_1 = inst->n - 1;
    int32_t tm = am->hmm_tm[inst->hmm];
    const float *trP = am->trP + (size_t)tm * maxN * maxN;
    const int16_t *se = am->se + (size_t)tm * maxN * 2;
    Tok *res = d->tokenBuf + 1;
    for (int32_t j = 1; j < N_1; ++j, ++res) {                      /* :387-424 */
        int32_t i = se[j * 2], endi = se[j * 2 + 1];
        const Tok *cur = &states[i];
        *res = *cur;
        res->score += trP[i * maxN + j];
        res->ac += trP[i * maxN + j];
        for (++i, ++cur; i < endi; ++i, ++cur) {
            float tmpScore = cur->score + trP[i * maxN + j];
            if (tmpScore > res->score) {
                *res = *cur;
                res->score = tmpScore;
                res->ac += trP[i * maxN + j];
            } else if (tmpScore == res->score && tmpScore > LZ) ++d->tie_kind[2];
        }
        res->score -= d->normaliseScore;
        if (res->score > d->emitTh) {
            ++d->nEmitProc;
            float outp = am_calc_gmm(d, am->hmm_gmm[(size_t)inst->hmm * maxN + j]);
            res->score += outp;
            res->ac += outp;
            if (d->hist && hist_add(d->hist, res->score) != 0) d->err = -5;
            if (res->score > d->bestEmitScore) d->bestEmitScore = res->score;
        } else *res = NULLTOK;
    }
    inst->nact = 0;                                                 /* :428-436 */
    for (int32_t i = 0; i < N_1; ++i) {
        if (d->tokenBuf[i].score > LZ) ++inst->nact;
        states[i] = d->tokenBuf[i];
    }
    d->nActiveEmitHyps += inst->nact;
    {                                                               /* :443-483 exit state */
        int32_t i = se[N_1 * 2], endi = se[N_1 * 2 + 1];
        Tok *r = &states[N_1];
        const Tok *cur = &states[i];
        *r = *cur;
        r->score += trP[i * maxN + N_1];
        r->ac += trP[i * maxN + N_1];
        for (++i, ++cur; i < endi; ++i, ++cur) {
            float tmpScore = cur->score + trP[i * maxN + N_1];
            if (tmpScore > r->score) {
                *r = *cur;
                r->score = tmpScore;
                r->ac += trP[i * maxN + N_1];
            } else if (tmpScore == r->score && tmpScore > LZ) ++d->tie_kind[2];
        }
        if (r->score <= LZ) *r = NULLTOK;
        else { ++inst->nact; ++d->nActiveEndHyps; }
    }
}

/* doHMMInternalPropagation, :899-935 */
static void do_internal(jo_dec *d)
{
    d->nActiveEmitHyps = d->nActiveEndHyps = d->nEmitProc = d->nEndProc = 0;
    d->bestEmitScore = LZ;
    int32_t prev = -1, inst = d->active;
    while (inst >= 0) {
        Tok *entry = &d->toks[(size_t)inst * d->maxN];
        if (entry->score > LZ && entry->score < d->startTh) {       /* :915-918 */
            *entry = NULLTOK;
            --d->insts[inst].nact;
        }
        ++d->st.tot_insts_in;
        hmm_internal(d, inst);
        if (d->insts[inst].nact == 0) inst = return_inst(d, inst, prev);
        else { prev = inst; inst = d->insts[inst].next; }
    }
    d->st.tot_active_emit_hyps += d->nActiveEmitHyps;
    d->st.tot_active_end_hyps += d->nActiveEndHyps;
    d->st.tot_proc_emit_hyps += d->nEmitProc;
}

/* doHMMExternalPropagation, :937-982 */
static void do_external(jo_dec *d)
{
    d->nEndProc = 0;
    int32_t prev = -1, inst = d->active;
    while (inst >= 0) {
        int32_t arc = d->insts[inst].arc;
        int32_t n = d->insts[inst].n;
        Tok exit_tok = d->toks[(size_t)inst * d->maxN + n - 1];     /* copy: pools may move */
        if (exit_tok.score > LZ) {
            if (d->net->out[arc] == 0) {
                if (exit_tok.score > d->endTh) { ++d->nEndProc; propagate(d, &exit_tok, arc); }
            } else {
                if (exit_tok.score > d->wordTh) { ++d->nEndProc; propagate(d, &exit_tok, arc); }
            }
            d->toks[(size_t)inst * d->maxN + n - 1] = NULLTOK;      /* :964 */
            if (--d->insts[inst].nact == 0) inst = return_inst(d, inst, prev);
            else { prev = inst; inst = d->insts[inst].next; }
        } else { prev = inst; inst = d->insts[inst].next; }
    }
    d->st.tot_proc_end_hyps += d->nEndProc;
    join_new(d);
    d->st.tot_active_models += d->nActiveInsts;
}

/* traceWinningPaths, WFSTDecoderLite.cpp:874-890: the Path records from the last traced one
 * (exclusive) down to |path| (inclusive) are appended to partialPaths, oldest first */
static void trace_winning(jo_dec *d, int32_t path)
{
    if (path < 0) return;                                           /* assert(path) :875 */
    int32_t last = d->n_partial ? d->partialPaths[d->n_partial - 1] : -1;
    if (path == last) return;                                       /* :880 */
    int32_t n = 1;
    for (int32_t q = path; d->paths[q].prev != last && d->paths[q].prev >= 0; q = d->paths[q].prev) ++n;   /* :882-886 */
    if (d->n_partial + n > d->cap_partial) {
        d->cap_partial = (d->n_partial + n) * 2 + 64;
        d->partialPaths = (int32_t *)realloc(d->partialPaths, sizeof(int32_t) * (size_t)d->cap_partial);
    }
    int32_t q = path;
    for (int32_t k = n - 1; k >= 0; --k) { d->partialPaths[d->n_partial + k] = q; q = d->paths[q].prev; }   /* :887-889 */
    d->n_partial += n;
}

/* tracePartialPath, WFSTDecoderLite.cpp:824-868.  Returns 1 if a Path record all open
 * hypotheses converge into was found (and partialPaths extended), else 0. */
int jo_trace_partial(jo_dec *d)
{
    if (!d || !d->started) return fail(-6, "tracePartialPath before init");
    int found = 0;
    int32_t lastTraced = d->n_partial ? d->partialPaths[d->n_partial - 1] : -1;
    int32_t lastTracedFrame = lastTraced >= 0 ? d->paths[lastTraced].frame : -1;
    /* step 1 (:832-842): reset jointCount of every Path newer than the last traced one */
    if (d->n_paths > d->cap_joint) {
        d->cap_joint = d->n_paths * 2 + 64;
        d->jointCount = (int32_t *)realloc(d->jointCount, sizeof(int32_t) * (size_t)d->cap_joint);
    }
    for (int64_t q = 0; q < d->n_paths; ++q)
        if (d->paths[q].frame > lastTracedFrame) d->jointCount[q] = 0;
    /* step 2 (:844-865): from every instance's first token that has a path, count the visits */
    for (int32_t inst = d->active; !found && inst >= 0; inst = d->insts[inst].next) {
        const Tok *tok = &d->toks[(size_t)inst * d->maxN];
        int32_t path = -1;
        /* :850-854 walks tok upwards until a path is found; it is bounded by the instance here
         * (an instance none of whose tokens has a path yet contributes nothing, so nothing is
         * found; the reference would read past the instance) */
        for (int32_t i = 0; i < d->insts[inst].n && path < 0; ++i) path = tok[i].path;
        while (path >= 0) {                                         /* :857-866 */
            if (d->paths[path].frame > lastTracedFrame && ++d->jointCount[path] == d->nActiveInsts) {
                found = 1;
                trace_winning(d, path);
                break;
            }
            path = d->paths[path].prev;
        }
    }
    d->lastPartialTraceFrame = d->currFrame;                        /* :867 */
    return found;
}

/* setPartialDecodeOptions, :892-896 */
int jo_path_counts(const jo_dec *d, int64_t out[4])
{
    if (!d || !out) return fail(-1, "jo_path_counts: null");
    out[0] = d->n_collections; out[1] = d->lastPathCollectFrame; out[2] = d->nPath; out[3] = d->nPathNew;
    return 0;
}

int jo_set_partial_interval(jo_dec *d, int32_t interval)
{
    if (!d || interval < 0) return fail(-1, "setPartialDecodeOptions: traceInterval >= 0");
    d->partialTraceInterval = interval;
    return 0;
}

/* partialPaths as (label, frame) of its Path records, oldest first (:252-256 prints the frames) */
int jo_partial_get(jo_dec *d, int32_t *n, const int32_t **labels, const int32_t **times)
{
    if (!d || !n) return fail(-1, "jo_partial_get: null argument");
    d->p_label = (int32_t *)realloc(d->p_label, sizeof(int32_t) * (size_t)(d->n_partial + 1));
    d->p_time = (int32_t *)realloc(d->p_time, sizeof(int32_t) * (size_t)(d->n_partial + 1));
    for (int32_t k = 0; k < d->n_partial; ++k) {
        d->p_label[k] = d->paths[d->partialPaths[k]].label;
        d->p_time[k] = d->paths[d->partialPaths[k]].frame;
    }
    *n = d->n_partial;
    if (labels) *labels = d->p_label;
    if (times) *times = d->p_time;
    return 0;
}

/* processFrame, WFSTDecoderLite.cpp:311-372 */
int jo_process_frame(jo_dec *d, const float *const *rows, int32_t frame, int32_t n_avail)
{
    if (!d->started) return fail(-6, "processFrame before init");
    d->currFrame = frame;
    int rc = am_new_frame(d, frame, rows, n_avail);                 /* :315 */
    if (rc) return rc;
    d->bestFinal = NULLTOK;                                         /* :316 */
    d->normaliseScore = (d->bestEmitScore > LZ ? d->bestEmitScore : 0.0f);   /* :321 */
    if (d->hist) {                                                  /* :322-329 */
        d->emitTh = hist_thresh(d->hist, d->maxHyps);
        d->emitTh -= d->normaliseScore;
        if (d->emitWin > 0.0 && d->emitTh < -d->emitWin) d->emitTh = -d->emitWin;
        hist_reset(d->hist);
    } else d->emitTh = (d->emitWin > 0.0 ? -d->emitWin : LZ);      /* :331 */
    d->startTh = (d->startWin > 0.0 ? (d->bestEmitScore - d->startWin) : LZ);   /* :337 */
    do_internal(d);                                                 /* :341 */
    d->endTh = (d->endWin > 0.0 ? (d->bestEmitScore - d->endWin) : LZ);        /* :349 */
    d->wordTh = (d->wordWin > 0.0 ? (d->bestEmitScore - d->wordWin) : LZ);     /* :350 */
    do_external(d);                                                 /* :353 */
    /* path collection :355-370.  collectPaths (:699-747) frees the Path objects no token of an active instance
     * reaches, directly or through prev links; records are never freed here (that changes no result), but what the
     * reference's allocator would hold is COUNTED, because both of its triggers are modelled and the second one
     * reads that count:  (nPath / nPathNew > 12 and nPath > 10000)  or  (currFrame - lastPathCollectFrame > 100).
     * nPathNew is 0 until the first collection: nPath / 0 is +inf for nPath > 0 (IEEE float division, :360), so
     * before it the first trigger is nPath > 10000 alone; 0 / 0 is NaN and compares false. */
    {
        const float pathRatio = (float)d->nPath / (float)d->nPathNew;
        if ((pathRatio > 12. && d->nPath > 10000) || (d->currFrame - d->lastPathCollectFrame > 100)) {
            /* survivors of collectPaths: the Paths reachable from the tokens of the active instances */
            if (d->n_paths > d->cap_pmark) {
                d->cap_pmark = d->n_paths * 2 + 64;
                d->pmark = (unsigned char *)realloc(d->pmark, (size_t)d->cap_pmark);
            }
            memset(d->pmark, 0, (size_t)d->n_paths);
            int64_t live = 0;
            for (int32_t inst = d->active; inst >= 0; inst = d->insts[inst].next)
                for (int32_t i = 0; i < d->insts[inst].n; ++i)
                    for (int32_t q = d->toks[(size_t)inst * d->maxN + i].path; q >= 0 && !d->pmark[q]; q = d->paths[q].prev) {
                        d->pmark[q] = 1;
                        ++live;
                    }
            d->nPath = d->nPathNew = live;                          /* :745 */
            ++d->n_collections;
            d->lastPathCollectFrame = d->currFrame;                 /* :746 */
            if (d->partialTraceInterval > 0 && (d->currFrame - d->lastPartialTraceFrame > d->partialTraceInterval))
                jo_trace_partial(d);                                /* :365-368 */
        }
    }
    if (d->trace && frame < d->trace_cap) d->trace[frame] = d->bestEmitScore;
    if (d->err == -5) return fail(-5, "Histogram::addScore - score > maxScore");
    return 0;
}

/* recognitionFinish, WFSTDecoderLite.cpp:230-309 */
int jo_finish(jo_dec *d, jo_hyp *out)
{
    memset(out, 0, sizeof *out);
    d->st.n_frames = d->currFrame + 1;
    out->stats = d->st;
    Tok best = d->bestFinal;
    if (d->partialTraceInterval > 0 && best.score > LZ)             /* :245-251 one more trace from the best token */
        trace_winning(d, best.path);
    if (best.score == LZ) { out->n = -1; return 0; }                /* :264-267 */
    int32_t n = 0;
    for (int32_t p = best.path; p >= 0; p = d->paths[p].prev) ++n;
    if (n > d->r_cap) {
        d->r_cap = n + 64;
        d->r_label = (int32_t *)realloc(d->r_label, sizeof(int32_t) * (size_t)d->r_cap);
        d->r_time = (int32_t *)realloc(d->r_time, sizeof(int32_t) * (size_t)d->r_cap);
        d->r_score = (float *)realloc(d->r_score, sizeof(float) * (size_t)d->r_cap);
        d->r_ac = (float *)realloc(d->r_ac, sizeof(float) * (size_t)d->r_cap);
        d->r_lm = (float *)realloc(d->r_lm, sizeof(float) * (size_t)d->r_cap);
    }
    int32_t k = 0;
    for (int32_t p = best.path; p >= 0; p = d->paths[p].prev, ++k) {   /* :273-305 */
        d->r_label[k] = d->paths[p].label;
        d->r_time[k] = d->paths[p].frame;
        d->r_score[k] = d->paths[p].score;
        d->r_ac[k] = d->paths[p].ac;
        d->r_lm[k] = d->paths[p].lm;
        if (k == 0) {                                               /* :293-300 */
            d->r_lm[0] = best.lm; d->r_ac[0] = best.ac; d->r_score[0] = best.score;
        }
    }
    out->n = n;
    out->label = d->r_label; out->time = d->r_time;
    out->score = d->r_score; out->ac = d->r_ac; out->lm = d->r_lm;
    /* bestDecHyp totals are only assigned inside the loop (:297-300): with an
     * empty history the DecHyp keeps its constructor values (LOG_ZERO). */
    if (n > 0) { out->tot_score = best.score; out->tot_ac = best.ac; out->tot_lm = best.lm; }
    else { out->tot_score = LZ; out->tot_ac = LZ; out->tot_lm = LZ; }
    return 0;
}

/* DecoderSingleTest::decodeUtterance, DecoderSingleTest.cpp:259-300 */
int jo_decode_utt(jo_dec *d, const float *feats, int32_t T, jo_hyp *out, double *cpu_seconds)
{
    int32_t D = d->am->D;
    const float **rows = (const float **)malloc(sizeof(float *) * (size_t)(T > 0 ? T : 1));
    for (int32_t t = 0; t < T; ++t) rows[t] = feats + (size_t)t * D;
    clock_t t0 = clock();
    int rc = jo_init(d);
    int32_t nFrames = 0, preRead = 20, nData = 0;
    while (nData < preRead && nData < T) ++nData;                   /* :267-277 */
    while (rc == 0 && nData > 0) {                                  /* :280-295 */
        rc = jo_process_frame(d, &rows[nFrames], nFrames, nData);
        ++nFrames;
        if (nFrames + nData - 1 < T) { /* one more frame fetched, window slides */ }
        else --nData;
    }
    if (rc == 0) rc = jo_finish(d, out);
    clock_t t1 = clock();
    if (cpu_seconds) *cpu_seconds = (double)(t1 - t0) / CLOCKS_PER_SEC;
    free(rows);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * WFSTDecoderLiteThreading + HTKFlatModelsThreading (src/WFSTDecoderLiteThreading.cpp:78-287,
 * src/HTKFlatModelsThreading.cpp:40-131): the reference's two-thread organisation.  The search
 * thread's internal propagation runs in two passes: pass 1 (:178-286) does the transitions of every
 * instance and, for every emitting state above the threshold whose output is not in the block
 * cache, queues its GMM for the scoring thread; pass 2 (:120-170) adds the outputs in queue order
 * as they become ready.  The scoring thread (calcStates, :110-129) computes the block of every
 * queued GMM.  Results equal the single-thread core's (tests assert it); only the wall time differs.
 * What is NOT taken over is the reference's queue, a list threaded through plain ints that both
 * threads write without any synchronisation (its own comment calls the consequence "a very rare
 * sync problem", :66-69): here a single-producer / single-consumer ring with C11 atomics carries
 * the same GMM ids in the same order.  Like the reference (:57-60) it refuses models with more than
 * one transition into the exit state.
 */
#include <pthread.h>
#include <stdatomic.h>

typedef struct {
    jo_dec *d;
    int32_t *ring;                          /* queued GMM ids (capacity n_gmm: a GMM is queued once per frame) */
    atomic_long pushed, done;               /* totals since the utterance began */
    atomic_int running;
} ThreadQ;

static void *scoring_thread(void *arg)      /* HTKFlatModelsThreading::calcStates, :110-129 */
{
    ThreadQ *q = (ThreadQ *)arg;
    jo_dec *d = q->d;
    long n = 0;
    while (atomic_load_explicit(&q->running, memory_order_acquire)) {
        if (atomic_load_explicit(&q->pushed, memory_order_acquire) > n) {
            int32_t g = q->ring[n % d->am->n_gmm];
            int32_t m = d->currInputLen < d->fnBlock ? d->currInputLen : d->fnBlock;   /* calcGMMOutput's block, HTKFlatModels.cpp:232-258 */
            for (int32_t k = 0; k < m; ++k)
                d->cache[(size_t)g * d->fnBlock + k] = gmm_one(d->am, g, d->currInput[k]);
            d->cacheT[g] = d->amFrame;
            ++n;
            atomic_store_explicit(&q->done, n, memory_order_release);
        }
    }
    return NULL;
}

typedef struct { int32_t inst, state, next; } WaitState;      /* WFSTDecoderLiteThreading.h WaitState + list link */

/* WFSTDecoderLiteThreading::doHMMInternalPropagation, :78-176 (with HMMInternalPropagationPass1, :178-286) */
static int do_internal_threading(jo_dec *d, ThreadQ *q, int32_t *wait_head, int32_t *wait_tail, int32_t *wait_gmms,
                                 WaitState **pool, int32_t *pool_cap)
{
    const jo_am *am = d->am;
    const int32_t maxN = am->max_n;
    d->nActiveEmitHyps = d->nActiveEndHyps = d->nEmitProc = d->nEndProc = 0;
    d->bestEmitScore = LZ;
    int32_t waiting = -1, n_pool = 0;
    const long base = atomic_load_explicit(&q->pushed, memory_order_relaxed);
    int32_t prev = -1, inst = d->active;
    while (inst >= 0) {
        Tok *states = &d->toks[(size_t)inst * maxN];
        Inst *I = &d->insts[inst];
        if (states[0].score > LZ && states[0].score < d->startTh) {  /* :97-101 language model pruning */
            states[0] = NULLTOK;
            --I->nact;
        }
        ++d->st.tot_insts_in;
        if (I->nact > 0) {                                           /* Pass1, :178-286 */
            const int32_t N_1 = I->n - 1, tm = am->hmm_tm[I->hmm];
            const float *trP = am->trP + (size_t)tm * maxN * maxN;
            const int16_t *se = am->se + (size_t)tm * maxN * 2;
            if (se[N_1 * 2 + 1] - se[N_1 * 2] != 1 || se[N_1 * 2] != N_1 - 1)
                return fail(-1, "WFSTDecoderLiteThreading can not deal with HMMs with more than one to-exit transition");
            I->nact = 0;
            Tok *res = d->tokenBuf + 1;
            for (int32_t j = 1; j < N_1; ++j, ++res) {
                int32_t i = se[j * 2], endi = se[j * 2 + 1];
                const Tok *cur = &states[i];
                *res = *cur;
                res->score += trP[i * maxN + j];
                res->ac += trP[i * maxN + j];
                for (++i, ++cur; i < endi; ++i, ++cur) {
                    float tmpScore = cur->score + trP[i * maxN + j];
                    if (tmpScore > res->score) { *res = *cur; res->score = tmpScore; res->ac += trP[i * maxN + j]; }
                }
                res->score -= d->normaliseScore;
                if (res->score > d->emitTh) {
                    ++d->nEmitProc;
                    const int32_t g = am->hmm_gmm[(size_t)I->hmm * maxN + j];
                    /* (queued already this frame: join the wait list without looking at a cache entry the other
                     * thread may be writing; the reference reads it regardless, :234) */
                    if (wait_head[g] < 0 && d->amFrame - d->cacheT[g] < d->fnBlock) {   /* cachedOutput, :52-60 */
                        const float outp = d->cache[(size_t)g * d->fnBlock + (d->amFrame - d->cacheT[g])];
                        res->score += outp;
                        res->ac += outp;
                        if (d->hist && hist_add(d->hist, res->score) != 0) d->err = -5;
                        if (res->score > d->bestEmitScore) d->bestEmitScore = res->score;
                        if (j == N_1 - 1) {                          /* :244-260 pass to exit state */
                            Tok *ex = &states[N_1];
                            *ex = *res;
                            ex->score += trP[j * maxN + N_1];
                            ex->ac += trP[j * maxN + N_1];
                            ++I->nact;
                            ++d->nActiveEndHyps;
                        }
                    } else {                                         /* :261-274 not calculated yet: queue */
                        if (n_pool == *pool_cap) { *pool_cap *= 2; *pool = (WaitState *)realloc(*pool, sizeof(WaitState) * (size_t)*pool_cap); }
                        if (wait_head[g] < 0) {
                            ++waiting;
                            wait_gmms[waiting] = g;
                            q->ring[(base + waiting) % am->n_gmm] = g;                               /* addQueue, :102-108 */
                            atomic_store_explicit(&q->pushed, base + waiting + 1, memory_order_release);
                            wait_head[g] = n_pool;
                        } else (*pool)[wait_tail[g]].next = n_pool;
                        wait_tail[g] = n_pool;
                        (*pool)[n_pool] = (WaitState){inst, j, -1};
                        ++n_pool;
                    }
                } else *res = NULLTOK;
            }
            if ((res - 1)->score <= LZ) states[N_1] = NULLTOK;       /* :277-279 */
            for (int32_t i = 0; i < N_1; ++i) {                      /* :283-291 */
                if (d->tokenBuf[i].score > LZ) { ++I->nact; ++d->nActiveEmitHyps; }
                states[i] = d->tokenBuf[i];
            }
        }
        if (I->nact == 0) inst = return_inst(d, inst, prev);
        else { prev = inst; inst = I->next; }
    }
    /* pass 2 (:120-170): the queued states, in queue order, as their GMMs become ready */
    for (int32_t i = 0; i <= waiting; ++i) {
        while (atomic_load_explicit(&q->done, memory_order_acquire) < base + i + 1) { /* nReadyStates() spin, :124-126 */ }
        const int32_t g = wait_gmms[i];
        const float outp = d->cache[(size_t)g * d->fnBlock + (d->amFrame - d->cacheT[g])];   /* readOutput, :62-73 */
        for (int32_t k = wait_head[g]; k >= 0; k = (*pool)[k].next) {
            const WaitState ws = (*pool)[k];
            Inst *I = &d->insts[ws.inst];
            Tok *res = &d->toks[(size_t)ws.inst * maxN + ws.state];
            res->score += outp;
            res->ac += outp;
            if (d->hist && hist_add(d->hist, res->score) != 0) d->err = -5;
            if (res->score > d->bestEmitScore) d->bestEmitScore = res->score;
            const int32_t N_1 = I->n - 1;
            if (ws.state == N_1 - 1) {                               /* :143-162 exit state */
                const float *trP = am->trP + (size_t)am->hmm_tm[I->hmm] * maxN * maxN;
                Tok *ex = res + 1;
                *ex = *res;
                ex->score += trP[ws.state * maxN + N_1];
                ex->ac += trP[ws.state * maxN + N_1];
                ++I->nact;
                ++d->nActiveEndHyps;
            }
        }
        wait_head[g] = wait_tail[g] = -1;
    }
    d->st.tot_active_emit_hyps += d->nActiveEmitHyps;
    d->st.tot_active_end_hyps += d->nActiveEndHyps;
    d->st.tot_proc_emit_hyps += d->nEmitProc;
    return 0;
}

/* DecoderSingleTest's frame loop (as jo_decode_utt) over the two-thread core; wall_seconds is wall-clock
 * time (two cores are busy: the scoring thread spins like the reference's) */
int jo_decode_utt_threading(jo_dec *d, const float *feats, int32_t T, jo_hyp *out, double *wall_seconds)
{
    const int32_t D = d->am->D, G = d->am->n_gmm;
    const float **rows = (const float **)malloc(sizeof(float *) * (size_t)(T > 0 ? T : 1));
    for (int32_t t = 0; t < T; ++t) rows[t] = feats + (size_t)t * D;
    ThreadQ q;
    q.d = d; q.ring = (int32_t *)malloc(sizeof(int32_t) * (size_t)G);
    atomic_init(&q.pushed, 0); atomic_init(&q.done, 0); atomic_init(&q.running, 1);
    int32_t *wait_head = (int32_t *)malloc(sizeof(int32_t) * (size_t)G), *wait_tail = (int32_t *)malloc(sizeof(int32_t) * (size_t)G);
    int32_t *wait_gmms = (int32_t *)malloc(sizeof(int32_t) * (size_t)G);
    for (int32_t g = 0; g < G; ++g) wait_head[g] = wait_tail[g] = -1;
    int32_t pool_cap = 4096;
    WaitState *pool = (WaitState *)malloc(sizeof(WaitState) * (size_t)pool_cap);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int rc = jo_init(d);
    pthread_t th;
    if (rc == 0 && pthread_create(&th, NULL, scoring_thread, &q) != 0) rc = fail(-7, "pthread_create failed");
    const int started = rc == 0;
    int32_t nFrames = 0, nData = T < 20 ? T : 20;                   /* DecoderSingleTest.cpp:267-295 */
    while (rc == 0 && nData > 0) {
        /* processFrame (WFSTDecoderLite.cpp:311-372) with the threading core's internal propagation */
        d->currFrame = nFrames;
        rc = am_new_frame(d, nFrames, &rows[nFrames], nData);
        if (rc) break;
        d->bestFinal = NULLTOK;
        d->normaliseScore = (d->bestEmitScore > LZ ? d->bestEmitScore : 0.0f);
        if (d->hist) {
            d->emitTh = hist_thresh(d->hist, d->maxHyps);
            d->emitTh -= d->normaliseScore;
            if (d->emitWin > 0.0 && d->emitTh < -d->emitWin) d->emitTh = -d->emitWin;
            hist_reset(d->hist);
        } else d->emitTh = (d->emitWin > 0.0 ? -d->emitWin : LZ);
        d->startTh = (d->startWin > 0.0 ? (d->bestEmitScore - d->startWin) : LZ);
        rc = do_internal_threading(d, &q, wait_head, wait_tail, wait_gmms, &pool, &pool_cap);
        if (rc) break;
        d->endTh = (d->endWin > 0.0 ? (d->bestEmitScore - d->endWin) : LZ);
        d->wordTh = (d->wordWin > 0.0 ? (d->bestEmitScore - d->wordWin) : LZ);
        do_external(d);
        if (d->err == -5) { rc = fail(-5, "Histogram::addScore - score > maxScore"); break; }
        ++nFrames;
        if (!(nFrames + nData - 1 < T)) --nData;
    }
    if (started) {
        atomic_store_explicit(&q.running, 0, memory_order_release);
        pthread_join(th, NULL);
    }
    if (rc == 0) rc = jo_finish(d, out);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (wall_seconds) *wall_seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    free(rows); free(q.ring); free(wait_head); free(wait_tail); free(wait_gmms); free(pool);
    return rc;
}

/* equal-score recombinations by kind (see jo_stats.ties): bestFinalToken (:513-520), entry
 * token (:560-582), HMM-internal max over predecessors (:393-406, :459: lowest index wins, an
 * order every implementation shares), and the entry-token ties whose two tokens actually differ */
int jo_tie_breakdown(const jo_dec *d, int64_t out[4])
{
    if (!d || !out) return -1;
    for (int i = 0; i < 4; ++i) out[i] = d->tie_kind[i];
    return 0;
}

/* Test aid: mode 1 lets the LAST visited token win equal-score recombinations of entry tokens and
 * bestFinalToken (the reference keeps the first, strict > at :516, :563).  A result that is the
 * same under both rules does not depend on visiting order at all. */
int jo_dec_set_tie_mode(jo_dec *d, int mode)
{
    if (!d) return -1;
    d->tie_mode = mode ? 1 : 0;
    return 0;
}

/* the host libm's expf, elementwise (what HTKFlatModels::logAdd calls on a float, :266-293) */
int jo_expf_array(const float *x, int64_t n, float *out)
{
    if (!x || !out || n < 0) return -1;
    for (int64_t i = 0; i < n; ++i) out[i] = expf(x[i]);
    return 0;
}
