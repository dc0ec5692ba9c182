// jd_host.cpp - host-side network / acoustic-model preparation for juicer_amd.
//
// Re-expresses the reference's static data (WFSTNetwork, HTKModels,
// HTKFlatModels) as flat arrays ready for upload to HBM.  The load-time float
// arithmetic follows the reference exactly (same libm, same operation order),
// so prepared parameters are bit-identical to what Juicer would hold in memory:
//   WFSTNetwork::WFSTNetwork(text)      src/WFSTNetwork.cpp:403-560
//   HTKModels::addVarVec / addGMM       src/HTKModels.cpp:835-870, 600-676
//   HTKModels::addTransMatrix / addHMM  src/HTKModels.cpp:873-974, 581-593
//   HTKModels::createTrPandSEIndex      src/HTKModels.cpp:2330-2390
//   HTKFlatModels::init                 src/HTKFlatModels.cpp:94-177
// Compiled with -ffp-contract=off.
#include <atomic>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

#include "jd_internal.h"

static thread_local char g_jd_err[1024] = "";

int jd_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_jd_err, sizeof g_jd_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *jd_dev_env(const char *name)
{
    const char *on = getenv("JD_DEV");
    if (!on || on[0] == '\0' || (on[0] == '0' && on[1] == '\0')) {
        // a knob that is set but not read changes nothing - said once per process, so that a deployment which exported
        // JD_PIPELINE=0 or JD_BROKER_RESIDENT=0 for an earlier build hears that it no longer does anything
        // (the interface: jd_dec_set_pipeline, jd_broker_create's arguments, include/juicer_amd.h)
        static std::atomic<bool> said{false};
        if (getenv(name) && !said.exchange(true))
            fprintf(stderr, "juicer_amd: %s is set but ignored - JD_* development knobs are read only with JD_DEV=1 (the interface is "
                            "include/juicer_amd.h: jd_dec_set_pipeline, jd_dec_set_capacity, jd_broker_create); further such knobs are not reported\n", name);
        return nullptr;
    }
    return getenv(name);
}

extern "C" const char *jd_last_error(void) { return g_jd_err; }
extern "C" const char *jd_version(void) { return "juicer_amd 0.1 (gfx950)"; }

static const float LZ = -FLT_MAX;                       // Torch3 LOG_ZERO
static const double LOG_2_PI = 1.83787706640934548355;  // Torch3 log_add.h

// ----------------------------------------------------------------------- network

static int finish_net(jd_net *n, int32_t n_final, const int32_t *fstate, const float *fw, bool scale,
                      float lm_scale)
{
    n->fin_w.assign((size_t)n->n_states, std::numeric_limits<float>::infinity());
    n->n_final = n_final;
    for (int32_t i = 0; i < n_final; ++i) {
        if (fstate[i] < 0 || fstate[i] >= n->n_states)
            return jd_fail(JD_EINVAL, "WFSTNetwork - finalState[%d].id out of range", i);
        // WFSTNetwork.cpp:441  finalStates[].weight = (real)(-weight * transWeightScalingFactor)
        n->fin_w[(size_t)fstate[i]] = scale ? (float)(-fw[i] * lm_scale) : fw[i];
    }
    n->max_in = 0;
    for (const JdArc &a : n->arcs) n->max_in = std::max(n->max_in, a.in);
    return JD_OK;
}

extern "C" int jd_net_create_arcs(jd_net **out, int64_t n_arcs, const int32_t *from, const int32_t *to,
                                  const int32_t *in, const int32_t *outl, const float *w_file,
                                  int32_t n_final, const int32_t *fstate, const float *fweight_file,
                                  float lm_scale, float ins_penalty)
{
    if (!out || n_arcs <= 0 || !from || !to || !in || !outl || !w_file)
        return jd_fail(JD_EINVAL, "jd_net_create_arcs: no arcs / null argument");
    if (n_arcs > 0x7fffffff) return jd_fail(JD_EINVAL, "jd_net_create_arcs: more than 2^31-1 arcs");
    int32_t max_state = -1;
    for (int64_t i = 0; i < n_arcs; ++i) {
        if (from[i] < 0 || to[i] < 0 || in[i] < 0 || outl[i] < 0)     // WFSTNetwork.cpp:449-450
            return jd_fail(JD_EINVAL, "WFSTNetwork::WFSTNetwork - something < 0. %d %d %d %d", from[i], to[i],
                           in[i], outl[i]);
        max_state = std::max(max_state, std::max(from[i], to[i]));
    }
    jd_net *n = new jd_net();
    n->n_states = max_state + 1;
    n->n_arcs = n_arcs;
    n->init = from[0];                                                // :452-453
    // contiguity check (getTransitions returns first arc + count, :709-721)
    std::vector<int32_t> cnt((size_t)n->n_states, 0), first((size_t)n->n_states, 0);
    for (int64_t i = 0; i < n_arcs; ++i) {
        int32_t s = from[i];
        if (cnt[s] == 0) first[s] = (int32_t)i;
        else if ((int64_t)first[s] + cnt[s] != i) {
            delete n;
            return jd_fail(JD_EFORMAT, "arcs of state %d are not contiguous (arc %lld)", s, (long long)i);
        }
        ++cnt[s];
    }
    n->row_ptr.assign((size_t)n->n_states + 1, 0);
    for (int32_t s = 0; s < n->n_states; ++s) n->row_ptr[s + 1] = n->row_ptr[s] + cnt[s];
    n->arcs.resize((size_t)n_arcs);
    for (int32_t s = 0; s < n->n_states; ++s) {
        for (int32_t k = 0; k < cnt[s]; ++k) {
            int64_t i = (int64_t)first[s] + k;
            float w = (float)(-w_file[i] * lm_scale);                 // :481
            if (outl[i] > 0) w += ins_penalty;                        // :485-486
            n->arcs[(size_t)n->row_ptr[s] + k] = JdArc{to[i], w, in[i], outl[i]};
        }
    }
    int rc = finish_net(n, n_final, fstate, fweight_file, true, lm_scale);
    if (rc) { delete n; return rc; }
    n->lm_scale = lm_scale; n->ins_penalty = ins_penalty;
    *out = n;
    return JD_OK;
}

extern "C" int jd_net_create_csr(jd_net **out, int32_t n_states, int32_t init_state, const int32_t *row_ptr,
                                 const int32_t *to, const float *w, const int32_t *in, const int32_t *outl,
                                 int32_t n_final, const int32_t *fstate, const float *fweight)
{
    if (!out || n_states <= 0 || init_state < 0 || init_state >= n_states || !row_ptr || !to || !w || !in || !outl)
        return jd_fail(JD_EINVAL, "jd_net_create_csr: bad arguments");
    if (row_ptr[0] != 0) return jd_fail(JD_EINVAL, "jd_net_create_csr: row_ptr[0] must be 0");
    for (int32_t s = 0; s < n_states; ++s)
        if (row_ptr[s + 1] < row_ptr[s])
            return jd_fail(JD_EINVAL, "jd_net_create_csr: row_ptr decreases at state %d", s);
    if (row_ptr[n_states] <= 0) return jd_fail(JD_EINVAL, "jd_net_create_csr: no arcs");
    jd_net *n = new jd_net();
    n->n_states = n_states;
    n->init = init_state;
    n->n_arcs = row_ptr[n_states];
    n->row_ptr.assign(row_ptr, row_ptr + n_states + 1);
    n->arcs.resize((size_t)n->n_arcs);
    for (int64_t i = 0; i < n->n_arcs; ++i) {
        if (to[i] < 0 || to[i] >= n_states || in[i] < 0 || outl[i] < 0) {
            delete n;
            return jd_fail(JD_EINVAL, "jd_net_create_csr: arc %lld out of range", (long long)i);
        }
        n->arcs[(size_t)i] = JdArc{to[i], w[i], in[i], outl[i]};
    }
    int rc = finish_net(n, n_final, fstate, fweight, false, 1.0f);
    if (rc) { delete n; return rc; }
    *out = n;
    return JD_OK;
}

// WFSTAlphabet::WFSTAlphabet (WFSTNetwork.cpp:48-112): only maxLabel and the
// presence of '#' auxiliary symbols matter to this path.
static int read_syms(const char *path, int *max_label, bool *has_aux)
{
    FILE *fd = fopen(path, "rb");
    if (!fd) return jd_fail(JD_EFORMAT, "WFSTAlphabet::WFSTAlphabet - error opening symbols filename %s", path);
    char line[10000], sym[10000];
    int id;
    *max_label = -1;
    *has_aux = false;
    while (fgets(line, sizeof line, fd)) {
        if (sscanf(line, "%s %d", sym, &id) != 2) continue;
        if (id > *max_label) *max_label = id;
        if (sym[0] == '#') *has_aux = true;
    }
    fclose(fd);
    return JD_OK;
}

extern "C" int jd_net_load_fsm(jd_net **out, const char *fsm_path, const char *insyms_path,
                               const char *outsyms_path, float lm_scale, float ins_penalty)
{
    if (!out || !fsm_path) return jd_fail(JD_EINVAL, "jd_net_load_fsm: null argument");
    FILE *fd = fopen(fsm_path, "rb");
    if (!fd) return jd_fail(JD_EFORMAT, "WFSTNetwork::WFSTNetwork - error opening wfstFilename %s", fsm_path);
    std::vector<int32_t> from, to, in, outl, fstate;
    std::vector<float> w, fw;
    char line[10000];
    int f, t, i, o, fin;
    float weight;
    // sscanf cascade of WFSTNetwork.cpp:414-447
    while (fgets(line, sizeof line, fd)) {
        if (sscanf(line, "%d %d %d %d %f", &f, &t, &i, &o, &weight) != 5) {
            if (sscanf(line, "%d %d %d %d", &f, &t, &i, &o) != 4) {
                if (sscanf(line, "%d %f", &fin, &weight) != 2) {
                    if (sscanf(line, "%d", &fin) != 1) continue;
                    weight = 0.0f;
                }
                fstate.push_back(fin);
                fw.push_back(weight);
                continue;
            }
            weight = 0.0f;
        }
        from.push_back(f); to.push_back(t); in.push_back(i); outl.push_back(o); w.push_back(weight);
    }
    fclose(fd);
    if (from.empty()) return jd_fail(JD_EFORMAT, "%s: no arcs", fsm_path);
    int max_in = *std::max_element(in.begin(), in.end());
    int max_out = *std::max_element(outl.begin(), outl.end());
    if (insyms_path) {
        int ml; bool aux;
        int rc = read_syms(insyms_path, &ml, &aux);
        if (rc) return rc;
        if (max_in > ml)                                              // :566-568
            return jd_fail(JD_EFORMAT, "WFSTNetwork::WFSTNetwork - maxInLab > inputAlphabet->getMaxLabel()");
        if (aux) return jd_fail(JD_EFORMAT, "%s holds auxiliary (#) symbols: remove them first "
                                "(the Lite core does not understand aux symbols)", insyms_path);
    }
    if (outsyms_path) {
        int ml; bool aux;
        int rc = read_syms(outsyms_path, &ml, &aux);
        if (rc) return rc;
        if (max_out > ml)                                             // :573-577
            return jd_fail(JD_EFORMAT, "WFSTNetwork::WFSTNetwork - maxOutLab=%d > outputAlphabet->getMaxLabel()=%d",
                           max_out, ml);
        if (aux) return jd_fail(JD_EFORMAT, "%s holds auxiliary (#) symbols: remove them first", outsyms_path);
    }
    return jd_net_create_arcs(out, (int64_t)from.size(), from.data(), to.data(), in.data(), outl.data(), w.data(),
                              (int32_t)fstate.size(), fstate.data(), fw.data(), lm_scale, ins_penalty);
}

extern "C" int jd_net_get_csr(const jd_net *n, int32_t *row_ptr, int32_t *to, float *w, int32_t *in, int32_t *outl,
                              float *fin_w)
{
    if (!n) return jd_fail(JD_EINVAL, "jd_net_get_csr: null");
    if (n->lazy_dev) return jd_fail(JD_ESTATE, "jd_net_get_csr: a lazily composed network has no finished arc table");
    if (row_ptr) memcpy(row_ptr, n->row_ptr.data(), n->row_ptr.size() * sizeof(int32_t));
    for (int64_t i = 0; i < n->n_arcs; ++i) {
        const JdArc &a = n->arcs[(size_t)i];
        if (to) to[i] = a.to;
        if (w) w[i] = a.w;
        if (in) in[i] = a.in;
        if (outl) outl[i] = a.out;
    }
    if (fin_w) memcpy(fin_w, n->fin_w.data(), n->fin_w.size() * sizeof(float));
    return JD_OK;
}

extern "C" int64_t jd_net_num_arcs(const jd_net *n) { return n ? n->n_arcs : 0; }
extern "C" int32_t jd_net_num_states(const jd_net *n) { return n ? n->n_states : 0; }
extern "C" int32_t jd_net_init_state(const jd_net *n) { return n ? n->init : -1; }
extern "C" void jd_net_destroy(jd_net *n) { if (n && n->lazy_free) n->lazy_free(n); delete n; }

// ------------------------------------------------------------- acoustic models

extern "C" int jd_am_create_htk(jd_am **out, int32_t D, int32_t n_gmm, int32_t max_mix, const int32_t *n_mix,
                                const float *weight, const float *mean, const float *var, int32_t n_hmm,
                                int32_t max_n, const int32_t *hmm_nstates, const int32_t *hmm_gmm,
                                const int32_t *hmm_tm, int32_t n_tm, const int32_t *tm_nstates,
                                const float *transp)
{
    if (!out || D <= 0 || n_gmm <= 0 || max_mix <= 0 || n_hmm <= 0 || max_n < 3 || n_tm <= 0)
        return jd_fail(JD_EINVAL, "jd_am_create_htk: bad sizes");
    if (max_n > JD_MAXN)
        return jd_fail(JD_EINVAL, "jd_am_create_htk: HMMs with more than %d states are not supported", JD_MAXN);
    jd_am *a = new jd_am();
    a->D = D; a->n_gmm = n_gmm; a->max_mix = max_mix; a->n_hmm = n_hmm; a->max_n = max_n; a->n_tm = n_tm;
    size_t gm = (size_t)n_gmm * max_mix;
    a->n_mix.assign(n_mix, n_mix + n_gmm);
    a->det.assign(gm, LZ);
    a->mean.assign(gm * D, 0.0f);
    a->ivar.assign(gm * D, 0.0f);
    a->var.assign(var, var + gm * D);
    a->weight.assign(weight, weight + gm);
    a->sum_log_var.assign(gm, 0.0f);
    a->log_weight.assign(gm, LZ);
    a->transp.assign(transp, transp + (size_t)n_tm * max_n * max_n);
    for (int32_t g = 0; g < n_gmm; ++g) {
        int32_t nm = n_mix[g];
        if (nm < 1 || nm > max_mix) { delete a; return jd_fail(JD_EINVAL, "n_mix[%d] out of range", g); }
        for (int32_t m = 0; m < nm; ++m) {
            size_t gi = (size_t)g * max_mix + m;
            float acc = (float)(D * LOG_2_PI);                        // HTKModels.cpp:859
            for (int32_t k = 0; k < D; ++k) {
                float v = var[gi * D + k];
                acc += std::log(v);                                   // :864 log(float) -> logf
                a->mean[gi * D + k] = mean[gi * D + k];               // HTKFlatModels.cpp:159
                a->ivar[gi * D + k] = (float)(1.0 / v);               // :160
            }
            acc *= -0.5;                                              // HTKModels.cpp:866
            float wgt = weight[gi];
            float lw = (wgt > 0.0) ? std::log(wgt) : LZ;              // :657-663
            a->sum_log_var[gi] = acc; a->log_weight[gi] = lw;
            a->det[gi] = acc + lw;                                    // HTKFlatModels.cpp:174
        }
        if (nm == 1 && weight[(size_t)g * max_mix] != 1.0f) {         // HTKModels.cpp:665
            delete a;
            return jd_fail(JD_EINVAL, "HTKModels::addGMM - (n_mixes == 1) && (compWeights[0] != 1.0)");
        }
    }
    a->tm_n.assign(tm_nstates, tm_nstates + n_tm);
    a->trP.assign((size_t)n_tm * max_n * max_n, LZ);
    a->se.assign((size_t)n_tm * max_n * 2, 0);
    std::vector<float> tm_tee((size_t)n_tm, LZ);
    for (int32_t t = 0; t < n_tm; ++t) {
        int32_t n = tm_nstates[t];
        if (n < 3 || n > max_n) { delete a; return jd_fail(JD_EINVAL, "tm_nstates[%d] out of range", t); }
        float *trP = a->trP.data() + (size_t)t * max_n * max_n;
        const float *tp = transp + (size_t)t * max_n * max_n;
        for (int32_t i = 0; i < n; ++i)                               // :941-951, :2349-2364
            for (int32_t j = 0; j < n; ++j)
                if (tp[i * max_n + j] > 0.0) trP[i * max_n + j] = std::log(tp[i * max_n + j]);
        int16_t *se = a->se.data() + (size_t)t * max_n * 2;
        for (int32_t j = 1; j < n; ++j) {                             // :2376-2386
            int32_t mn, mx;
            for (mn = (j == n - 1 ? 1 : 0); mn < n - 1; ++mn)
                if (trP[mn * max_n + j] > LZ) break;
            for (mx = n - 1; mx >= 1; --mx)
                if (trP[mx * max_n + j] > LZ) break;
            se[j * 2] = (int16_t)mn;
            se[j * 2 + 1] = (int16_t)(mx + 1);
        }
        int32_t suc = 0;                                              // :581-593
        for (int32_t j = 0; j < n; ++j)
            if (tp[j] > 0.0) {
                if (suc >= 1 && j == n - 1) tm_tee[t] = trP[j];
                ++suc;
            }
    }
    a->hmm_n.assign(hmm_nstates, hmm_nstates + n_hmm);
    a->hmm_tm.assign(hmm_tm, hmm_tm + n_hmm);
    a->hmm_tee.assign((size_t)n_hmm, LZ);
    a->hmm_gmm.assign((size_t)n_hmm * max_n, -1);
    for (int32_t h = 0; h < n_hmm; ++h) {
        int32_t n = hmm_nstates[h], t = hmm_tm[h];
        if (t < 0 || t >= n_tm || tm_nstates[t] != n) {
            delete a;
            return jd_fail(JD_EINVAL, "HTKModels::addHMM - curr->nStates != hmm->transmat->n_states (hmm %d)", h);
        }
        a->hmm_tee[h] = tm_tee[t];
        for (int32_t j = 1; j < n - 1; ++j) {
            int32_t g = hmm_gmm[(size_t)h * max_n + j];
            if (g < 0 || g >= n_gmm) { delete a; return jd_fail(JD_EINVAL, "hmm %d state %d: bad gmm index", h, j); }
            a->hmm_gmm[(size_t)h * max_n + j] = g;
        }
    }
    *out = a;
    return JD_OK;
}

// Models from the PREPARED arrays a loaded HTKFlatModels holds (HTKFlatModels.cpp:94-177: fDets, fMeans, fVars =
// inverse variances; HTKModels.cpp:2330-2390: trP, SEIndex; :581-593 teeWeight): nothing is recomputed, so the
// values are the reference's own, bit for bit.  What include/juicer_amd_decoder.hpp's exact-signature
// constructor hands over after walking an IModels*.
extern "C" int jd_am_create_flat(jd_am **out, int32_t D, int32_t n_gmm, int32_t max_mix, const int32_t *n_mix,
                                 const float *det, const float *mean, const float *ivar, int32_t n_hmm, int32_t max_n,
                                 const int32_t *hmm_nstates, const int32_t *hmm_gmm, const int32_t *hmm_tm,
                                 const float *hmm_tee, int32_t n_tm, const int32_t *tm_nstates, const float *trP,
                                 const int16_t *se)
{
    if (!out || !n_mix || !det || !mean || !ivar || !hmm_nstates || !hmm_gmm || !hmm_tm || !hmm_tee || !tm_nstates || !trP || !se ||
        D <= 0 || n_gmm <= 0 || max_mix <= 0 || n_hmm <= 0 || max_n < 3 || n_tm <= 0)
        return jd_fail(JD_EINVAL, "jd_am_create_flat: bad argument");
    if (max_n > JD_MAXN)
        return jd_fail(JD_EINVAL, "jd_am_create_flat: HMMs with more than %d states are not supported", JD_MAXN);
    jd_am *a = new jd_am();
    a->D = D; a->n_gmm = n_gmm; a->max_mix = max_mix; a->n_hmm = n_hmm; a->max_n = max_n; a->n_tm = n_tm;
    const size_t gm = (size_t)n_gmm * max_mix;
    a->n_mix.assign(n_mix, n_mix + n_gmm);
    a->det.assign(gm, LZ); a->mean.assign(gm * D, 0.0f); a->ivar.assign(gm * D, 0.0f);
    for (int32_t g = 0; g < n_gmm; ++g) {
        if (n_mix[g] < 1 || n_mix[g] > max_mix) { delete a; return jd_fail(JD_EINVAL, "n_mix[%d] out of range", g); }
        for (int32_t m = 0; m < n_mix[g]; ++m) {                      // (components beyond n_mix keep the padding values)
            const size_t gi = (size_t)g * max_mix + m;
            a->det[gi] = det[gi];
            memcpy(&a->mean[gi * D], &mean[gi * D], (size_t)D * sizeof(float));
            memcpy(&a->ivar[gi * D], &ivar[gi * D], (size_t)D * sizeof(float));
        }
    }
    a->tm_n.assign(tm_nstates, tm_nstates + n_tm);
    for (int32_t t = 0; t < n_tm; ++t)
        if (tm_nstates[t] < 3 || tm_nstates[t] > max_n) { delete a; return jd_fail(JD_EINVAL, "tm_nstates[%d] out of range", t); }
    a->trP.assign(trP, trP + (size_t)n_tm * max_n * max_n);
    a->se.assign(se, se + (size_t)n_tm * max_n * 2);
    a->hmm_n.assign(hmm_nstates, hmm_nstates + n_hmm);
    a->hmm_tm.assign(hmm_tm, hmm_tm + n_hmm);
    a->hmm_tee.assign(hmm_tee, hmm_tee + n_hmm);
    a->hmm_gmm.assign((size_t)n_hmm * max_n, -1);
    for (int32_t h = 0; h < n_hmm; ++h) {
        const int32_t n = hmm_nstates[h], t = hmm_tm[h];
        if (t < 0 || t >= n_tm || tm_nstates[t] != n) {
            delete a;
            return jd_fail(JD_EINVAL, "HTKModels::addHMM - curr->nStates != hmm->transmat->n_states (hmm %d)", h);
        }
        for (int32_t j = 1; j < n - 1; ++j) {
            const int32_t g = hmm_gmm[(size_t)h * max_n + j];
            if (g < 0 || g >= n_gmm) { delete a; return jd_fail(JD_EINVAL, "hmm %d state %d: bad gmm index", h, j); }
            a->hmm_gmm[(size_t)h * max_n + j] = g;
        }
    }
    *out = a;
    return JD_OK;
}

// HTKModels::Load(phonesListFName, priorsFName, statesPerModel), HTKModels.cpp:74-218: hybrid ANN / HMM
// models.  One HMM per phone with statesPerModel states whose emitting states all score
// x[phone] - log(prior[phone]) (calcOutput, :481-512 / HTKFlatModels.cpp:190-222); ONE transition matrix:
// entry -> first emitting state 1.0, every emitting state 0.5 self loop / 0.5 forward (:188-207).
extern "C" int jd_am_create_hybrid(jd_am **out, int32_t n_phones, const float *priors, int32_t states_per_model)
{
    if (!out || !priors || n_phones <= 0) return jd_fail(JD_EINVAL, "jd_am_create_hybrid: bad argument");
    if (states_per_model <= 2) return jd_fail(JD_EINVAL, "HTKModels::Models(3) - statesPerModel <= 2 (ie. no emitting states)");
    if (states_per_model > JD_MAXN) return jd_fail(JD_EINVAL, "jd_am_create_hybrid: HMMs with more than %d states are not supported", JD_MAXN);
    const int32_t P = n_phones, N = states_per_model;
    std::vector<int32_t> n_mix((size_t)P, 1), hmm_n((size_t)P, N), hmm_gmm((size_t)P * N, -1), hmm_tm((size_t)P, 0), tm_n(1, N);
    std::vector<float> weight((size_t)P, 1.0f), mean((size_t)P * P, 0.0f), var((size_t)P * P, 1.0f), transp((size_t)N * N, 0.0f);
    for (int32_t h = 0; h < P; ++h)
        for (int32_t j = 1; j < N - 1; ++j) hmm_gmm[(size_t)h * N + j] = h;     // :176-179 gmmInds[i] = nHMMs
    transp[1] = 1.0f;                                                            // :196-199
    for (int32_t i = 1; i < N - 1; ++i) { transp[(size_t)i * N + i] = 0.5f; transp[(size_t)i * N + i + 1] = 0.5f; }   // :200-204
    jd_am *a = nullptr;
    // (the Gaussian tables are placeholders: in hybrid mode nothing reads them)
    int rc = jd_am_create_htk(&a, P, P, 1, n_mix.data(), weight.data(), mean.data(), var.data(), P, N, hmm_n.data(), hmm_gmm.data(),
                              hmm_tm.data(), 1, tm_n.data(), transp.data());
    if (rc) return rc;
    a->hybrid = true;
    a->log_prior.resize((size_t)P);
    for (int32_t h = 0; h < P; ++h) {
        if (!(priors[h] > 0.0f)) { delete a; return jd_fail(JD_EINVAL, "jd_am_create_hybrid: prior %d is not positive", h); }
        a->log_prior[(size_t)h] = std::log(priors[h]);                            // :183 log(float)
    }
    *out = a;
    return JD_OK;
}

extern "C" int32_t jd_am_num_hmms(const jd_am *a) { return a ? a->n_hmm : 0; }
extern "C" int32_t jd_am_num_gmms(const jd_am *a) { return a ? a->n_gmm : 0; }
extern "C" int32_t jd_am_vec_size(const jd_am *a) { return a ? a->D : 0; }
extern "C" int32_t jd_am_max_states(const jd_am *a) { return a ? a->max_n : 0; }
extern "C" int32_t jd_am_max_mix(const jd_am *a) { return a ? a->max_mix : 0; }
extern "C" int32_t jd_am_num_transmats(const jd_am *a) { return a ? a->n_tm : 0; }
extern "C" int jd_am_get_topology(const jd_am *a, int32_t *hmm_nstates, int32_t *hmm_gmm, int32_t *hmm_tm, int32_t *n_mix)
{
    if (!a) return jd_fail(JD_EINVAL, "jd_am_get_topology: null");
    if (hmm_nstates) memcpy(hmm_nstates, a->hmm_n.data(), a->hmm_n.size() * sizeof(int32_t));
    if (hmm_gmm) memcpy(hmm_gmm, a->hmm_gmm.data(), a->hmm_gmm.size() * sizeof(int32_t));
    if (hmm_tm) memcpy(hmm_tm, a->hmm_tm.data(), a->hmm_tm.size() * sizeof(int32_t));
    if (n_mix) memcpy(n_mix, a->n_mix.data(), a->n_mix.size() * sizeof(int32_t));
    return JD_OK;
}

extern "C" int jd_am_get_flat(const jd_am *a, float *det, float *mean, float *ivar)
{
    if (!a) return jd_fail(JD_EINVAL, "jd_am_get_flat: null");
    if (det) memcpy(det, a->det.data(), a->det.size() * sizeof(float));
    if (mean) memcpy(mean, a->mean.data(), a->mean.size() * sizeof(float));
    if (ivar) memcpy(ivar, a->ivar.data(), a->ivar.size() * sizeof(float));
    return JD_OK;
}

extern "C" int jd_am_get_trans(const jd_am *a, float *trP, int16_t *se, float *tee)
{
    if (!a) return jd_fail(JD_EINVAL, "jd_am_get_trans: null");
    if (trP) memcpy(trP, a->trP.data(), a->trP.size() * sizeof(float));
    if (se) memcpy(se, a->se.data(), a->se.size() * sizeof(int16_t));
    if (tee) memcpy(tee, a->hmm_tee.data(), a->hmm_tee.size() * sizeof(float));
    return JD_OK;
}

extern "C" void jd_am_destroy(jd_am *a) { delete a; }

// ------------------------------------------------------------ HTK MMF text loader
//
// The subset of HTK's MMF format that the reference's flex/bison front-end accepts
// (src/htkparse.l.lpp:21-268, src/htkparse.y.ypp:113-147, 414-685): global options (~o with
// <STREAMINFO>/<VECSIZE>/parameter-kind tags), ~v (ignored with a warning, like the reference),
// shared states ~s, shared transition matrices ~t, HMM definitions ~h with shared (~s/~t) or
// inline states / <TRANSP>, <NUMMIXES>/<MIXTURE> or the implicit single-mixture form, optional
// <GCONST> (parsed, ignored: HTKModels.cpp:835-870 recomputes it).  Numbers go through
// (float)atof exactly as htkparse.l.lpp:38-41.  Tied-mixture pools (~m / <TMIX>) are rejected:
// HTKFlatModels assumes mixtureInd == gmmInd (HTKFlatModels.h:61-63).
// Index order follows HTKModels::initFromHTKParseResult: shared transition matrices and shared
// states in file order first, then per HMM (file order = in-label order, WFSTDecoderLite.cpp:754)
// its inline states / matrix.
namespace {
struct MmfTok { std::string s; };
struct MmfLexer {
    std::vector<std::string> t; size_t p = 0;
    bool load(const char *path)
    {
        FILE *f = fopen(path, "rb");
        if (!f) return false;
        std::string cur; int c; bool inq = false, intag = false;
        auto flush = [&]() { if (!cur.empty()) { t.push_back(cur); cur.clear(); } };
        while ((c = fgetc(f)) != EOF) {
            if (inq) { cur.push_back((char)c); if (c == '"') { inq = false; flush(); } continue; }
            if (intag) { cur.push_back((char)c); if (c == '>') { intag = false; flush(); } continue; }
            if (c == '"') { flush(); cur.push_back('"'); inq = true; continue; }
            if (c == '<') { flush(); cur.push_back('<'); intag = true; continue; }
            if (c == ' ' || c == '\t' || c == '\r' || c == '\n') { flush(); continue; }
            cur.push_back((char)c);
        }
        flush();
        fclose(f);
        return true;
    }
    bool end() const { return p >= t.size(); }
    const std::string &peek() const { static const std::string e; return end() ? e : t[p]; }
    std::string next() { return end() ? std::string() : t[p++]; }
};
std::string upper(std::string s) { for (auto &ch : s) ch = (char)toupper((unsigned char)ch); return s; }
std::string unquote(const std::string &s) { return (s.size() >= 2 && s[0] == '"') ? s.substr(1, s.size() - 2) : s; }
bool is_num(const std::string &s)
{
    if (s.empty()) return false;
    char *e = nullptr;
    (void)strtod(s.c_str(), &e);
    return e && *e == 0;
}
struct MmfState { std::vector<float> w, mu, var; int n_mix = 0; };
struct MmfTm { int n = 0; std::vector<float> a; };
}  // namespace

extern "C" int jd_am_load_mmf(jd_am **out, const char *mmf_path)
{
    if (!out || !mmf_path) return jd_fail(JD_EINVAL, "jd_am_load_mmf: null argument");
    MmfLexer L;
    if (!L.load(mmf_path)) return jd_fail(JD_EFORMAT, "HTKModels::Load - error opening %s", mmf_path);
    int D = -1;
    std::vector<MmfState> states;                    // GMMs in index order
    std::vector<std::string> sh_state_names;
    std::vector<MmfTm> tms;
    std::vector<std::string> sh_tm_names;
    struct Hmm { int n; std::vector<int> gmm; int tm; };
    std::vector<Hmm> hmms;
#define MMF_FAIL(...) return jd_fail(JD_EFORMAT, __VA_ARGS__)
    auto tag_int = [&](const std::string &tagname, int *v) -> bool {   // "<TAG> int" (htkparse.l: tag + INT is one token)
        if (upper(L.peek()) != tagname) return false;
        L.next();
        if (!is_num(L.peek())) return false;
        *v = atoi(L.next().c_str());
        return true;
    };
    auto rvector = [&](int n, std::vector<float> &dst) -> bool {
        for (int i = 0; i < n; ++i) {
            if (!is_num(L.peek())) return false;
            dst.push_back((float)atof(L.next().c_str()));              // htkparse.l.lpp:38-41
        }
        return true;
    };
    // mixpdf: <MEAN> n v.. <VARIANCE> n v.. [<GCONST> x]   (htkparse.y.ypp mixpdf/meanvec/variancevec/gconst)
    auto mixpdf = [&](MmfState &st) -> int {
        int n = 0;
        if (!tag_int("<MEAN>", &n)) return jd_fail(JD_EFORMAT, "MMF: <MEAN> expected near token %zu ('%s')", L.p, L.peek().c_str());
        if (n != D) return jd_fail(JD_EFORMAT, "HTKPARSE:meanvec - MEAN value did not match global vec size");
        if (!rvector(n, st.mu)) return jd_fail(JD_EFORMAT, "HTKPARSE:meanvec - n_elems did not match MEAN value");
        if (!tag_int("<VARIANCE>", &n)) return jd_fail(JD_EFORMAT, "MMF: <VARIANCE> expected (only diagonal covariances are supported)");
        if (n != D) return jd_fail(JD_EFORMAT, "HTKPARSE:variancevec - VARIANCE value did not match global vec size");
        if (!rvector(n, st.var)) return jd_fail(JD_EFORMAT, "HTKPARSE:variancevec - n_elems did not match VARIANCE value");
        if (upper(L.peek()) == "<GCONST>") { L.next(); if (is_num(L.peek())) L.next(); }
        return JD_OK;
    };
    // state body after <STATE> i / ~s "name": [<NUMMIXES> n] (<MIXTURE> i w mixpdf)+ | mixpdf
    auto state_body = [&](MmfState &st) -> int {
        int nm = 0;
        (void)tag_int("<NUMMIXES>", &nm);
        if (upper(L.peek()) == "<TMIX>") return jd_fail(JD_EFORMAT, "MMF: tied-mixture (<TMIX>) states are not supported by the flat models");
        if (upper(L.peek()) == "<MIXTURE>") {
            while (upper(L.peek()) == "<MIXTURE>") {
                L.next();
                if (!is_num(L.peek())) return jd_fail(JD_EFORMAT, "MMF: <MIXTURE> index expected");
                L.next();
                if (!is_num(L.peek())) return jd_fail(JD_EFORMAT, "MMF: <MIXTURE> weight expected");
                st.w.push_back((float)atof(L.next().c_str()));
                int rc = mixpdf(st);
                if (rc) return rc;
                ++st.n_mix;
            }
        } else {                                       // implicit single mixture, weight 1.0 (mixturedef: mixpdf)
            st.w.push_back(1.0f);
            int rc = mixpdf(st);
            if (rc) return rc;
            st.n_mix = 1;
        }
        return JD_OK;
    };
    auto transp = [&](MmfTm &tm) -> int {
        int n = 0;
        if (!tag_int("<TRANSP>", &n)) return jd_fail(JD_EFORMAT, "MMF: <TRANSP> expected near '%s'", L.peek().c_str());
        tm.n = n;
        if (!rvector(n * n, tm.a)) return jd_fail(JD_EFORMAT, "HTKPARSE:transp - vec n_elems did not match TRANSP value");
        return JD_OK;
    };
    auto find = [](const std::vector<std::string> &v, const std::string &s) {
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == s) return (int)i;
        return -1;
    };

    while (!L.end()) {
        std::string m = L.next();
        if (m == "~o" || m == "~O") {
            while (!L.end() && L.peek()[0] != '~') {
                std::string tg = upper(L.next());
                if (tg == "<VECSIZE>") { if (!is_num(L.peek())) MMF_FAIL("MMF: <VECSIZE> value expected"); D = atoi(L.next().c_str()); }
                else if (tg == "<STREAMINFO>") {
                    if (!is_num(L.peek())) MMF_FAIL("MMF: <STREAMINFO> count expected");
                    int ns = atoi(L.next().c_str());
                    if (ns != 1) MMF_FAIL("MMF: only single-stream models are supported");
                    if (is_num(L.peek())) { int w = atoi(L.next().c_str()); if (D < 0) D = w; }
                } else if (tg == "<HMMSETID>") { L.next(); }
                // parameter kind / covariance kind / duration kind tags carry no values
            }
        } else if (m == "~v" || m == "~V") {
            fprintf(stderr, "htkparse: ~v macros not supported - ignoring ~v %s definition\n", L.peek().c_str());
            L.next();
            int n = 0;
            if (!tag_int("<VARIANCE>", &n)) MMF_FAIL("MMF: ~v without <VARIANCE>");
            std::vector<float> tmp;
            if (!rvector(n, tmp)) MMF_FAIL("MMF: ~v vector too short");
        } else if (m == "~s" || m == "~S") {
            if (D <= 0) MMF_FAIL("MMF: ~s before the global <VECSIZE>");
            sh_state_names.push_back(unquote(L.next()));
            states.emplace_back();
            int rc = state_body(states.back());
            if (rc) return rc;
        } else if (m == "~t" || m == "~T") {
            sh_tm_names.push_back(unquote(L.next()));
            tms.emplace_back();
            int rc = transp(tms.back());
            if (rc) return rc;
        } else if (m == "~m" || m == "~M") {
            MMF_FAIL("MMF: ~m tied-mixture pools are not supported by the flat models (HTKFlatModels.h:61-63)");
        } else if (m == "~h" || m == "~H") {
            if (D <= 0) MMF_FAIL("MMF: ~h before the global <VECSIZE>");
            L.next();                                  // name (HMM index = order of appearance)
            if (upper(L.next()) != "<BEGINHMM>") MMF_FAIL("MMF: <BEGINHMM> expected");
            Hmm h; h.n = 0; h.tm = -1;
            if (!tag_int("<NUMSTATES>", &h.n) || h.n < 3) MMF_FAIL("MMF: <NUMSTATES> expected");
            h.gmm.assign((size_t)h.n, -1);
            while (upper(L.peek()) == "<VECSIZE>" || (L.peek().size() > 1 && L.peek()[0] == '<' && upper(L.peek()) != "<STATE>" &&
                   upper(L.peek()) != "<TRANSP>")) {   // optglobopts inside the HMM
                std::string tg = upper(L.next());
                if (tg == "<VECSIZE>" || tg == "<STREAMINFO>") while (is_num(L.peek())) L.next();
            }
            int n_emit = 0;
            while (upper(L.peek()) == "<STATE>") {
                L.next();
                if (!is_num(L.peek())) MMF_FAIL("MMF: <STATE> index expected");
                int si = atoi(L.next().c_str());
                if (si < 2 || si > h.n - 1) MMF_FAIL("MMF: <STATE> %d out of range", si);
                if (L.peek() == "~s" || L.peek() == "~S") {
                    L.next();
                    int g = find(sh_state_names, unquote(L.next()));
                    if (g < 0) MMF_FAIL("HTKPARSE:statedef - SMACRO string not found in htk_def");
                    h.gmm[(size_t)si - 1] = g;
                } else {
                    states.emplace_back();
                    sh_state_names.push_back(std::string());
                    int rc = state_body(states.back());
                    if (rc) return rc;
                    h.gmm[(size_t)si - 1] = (int)states.size() - 1;
                }
                ++n_emit;
            }
            if (n_emit != h.n - 2) MMF_FAIL("HTKPARSE:hmmdef - hmmstatelist n_elems did not match n_states");
            if (L.peek() == "~t" || L.peek() == "~T") {
                L.next();
                h.tm = find(sh_tm_names, unquote(L.next()));
                if (h.tm < 0) MMF_FAIL("HTKPARSE:transmatdef - SMACRO string not found in htk_def");
            } else {
                tms.emplace_back();
                sh_tm_names.push_back(std::string());
                int rc = transp(tms.back());
                if (rc) return rc;
                h.tm = (int)tms.size() - 1;
            }
            if (tms[(size_t)h.tm].n != h.n) MMF_FAIL("HTKModels::addHMM - curr->nStates != hmm->transmat->n_states");
            if (upper(L.next()) != "<ENDHMM>") MMF_FAIL("MMF: <ENDHMM> expected");
            hmms.push_back(h);
        } else {
            MMF_FAIL("MMF: unexpected token '%s'", m.c_str());
        }
    }
#undef MMF_FAIL
    if (hmms.empty() || states.empty() || D <= 0) return jd_fail(JD_EFORMAT, "%s: no HMM definitions", mmf_path);
    int max_mix = 0, max_n = 0;
    for (auto &s : states) max_mix = std::max(max_mix, s.n_mix);
    for (auto &t : tms) max_n = std::max(max_n, t.n);
    const int G = (int)states.size(), H = (int)hmms.size(), NT = (int)tms.size();
    std::vector<int32_t> n_mix((size_t)G), hn((size_t)H), hg((size_t)H * max_n, -1), ht((size_t)H), tn((size_t)NT);
    std::vector<float> wt((size_t)G * max_mix, 0.0f), mu((size_t)G * max_mix * D, 0.0f), var((size_t)G * max_mix * D, 1.0f);
    std::vector<float> tp((size_t)NT * max_n * max_n, 0.0f);
    for (int g = 0; g < G; ++g) {
        n_mix[(size_t)g] = states[(size_t)g].n_mix;
        for (int m = 0; m < states[(size_t)g].n_mix; ++m) {
            wt[(size_t)g * max_mix + m] = states[(size_t)g].w[(size_t)m];
            memcpy(&mu[((size_t)g * max_mix + m) * D], &states[(size_t)g].mu[(size_t)m * D], sizeof(float) * D);
            memcpy(&var[((size_t)g * max_mix + m) * D], &states[(size_t)g].var[(size_t)m * D], sizeof(float) * D);
        }
    }
    for (int t = 0; t < NT; ++t) {
        tn[(size_t)t] = tms[(size_t)t].n;
        for (int i = 0; i < tms[(size_t)t].n; ++i)
            for (int j = 0; j < tms[(size_t)t].n; ++j)
                tp[((size_t)t * max_n + i) * max_n + j] = tms[(size_t)t].a[(size_t)i * tms[(size_t)t].n + j];
    }
    for (int h = 0; h < H; ++h) {
        hn[(size_t)h] = hmms[(size_t)h].n; ht[(size_t)h] = hmms[(size_t)h].tm;
        for (int j = 0; j < hmms[(size_t)h].n; ++j) hg[(size_t)h * max_n + j] = hmms[(size_t)h].gmm[(size_t)j];
    }
    return jd_am_create_htk(out, D, G, max_mix, n_mix.data(), wt.data(), mu.data(), var.data(), H, max_n, hn.data(),
                            hg.data(), ht.data(), NT, tn.data(), tp.data());
}

// ------------------------------------------------- Juicer binary caches: JWNT / JMBI
//
// "<file>.bin" caches that juicer.cpp:854-866 / :778-784 prefer over the text files when they
// exist.  Layouts restated from WFSTNetwork::writeBinary/readBinary (WFSTNetwork.cpp:1106-1365),
// WFSTAlphabet::writeBinary/readBinary (:213-297) and HTKModels::output(.., true) / readBinary
// and the per-record readers (HTKModels.cpp:1044-1105, 1110-1233, 1309-1372, 1440-1492,
// 1580-1631, 1697-1740, 1803-1851, 1947-2040).  Native byte order, sizeof(int) == sizeof(real)
// == 4, sizeof(bool) == 1, as the reference writes them.
namespace {
struct BinReader {
    FILE *f = nullptr; const char *what = "";
    bool ok = true;
    template <typename T> T get() { T v{}; if (ok && fread(&v, sizeof(T), 1, f) != 1) ok = false; return v; }
    template <typename T> bool get_n(T *dst, size_t n) { if (ok && n && fread(dst, sizeof(T), n, f) != n) ok = false; return ok; }
    bool id(const char *tag) { char b[4] = {0, 0, 0, 0}; get_n(b, 4); return ok && memcmp(b, tag, 4) == 0; }
    bool skip_name()                              // int len (incl. NUL) + len bytes
    {
        int len = get<int>();
        if (!ok || len < 0 || len > (1 << 20)) return ok = false;
        if (len > 0) { std::vector<char> nm((size_t)len); get_n(nm.data(), (size_t)len); }
        return ok;
    }
};
struct BinWriter {
    FILE *f = nullptr;
    template <typename T> void put(T v) { fwrite(&v, sizeof(T), 1, f); }
    template <typename T> void put_n(const T *p, size_t n) { if (n) fwrite(p, sizeof(T), n, f); }
    void id(const char *tag) { fwrite(tag, 1, 4, f); }
};
// WFSTAlphabet::readBinary (WFSTNetwork.cpp:250-297): contents are not needed by this path
bool skip_alphabet(BinReader &r, bool *has_aux)
{
    if (!r.id("JWAL")) return false;
    int max_label = r.get<int>();
    (void)r.get<int>();                           // nLabels
    if (!r.ok) return false;
    if (max_label >= 0) {
        for (int i = 0; i <= max_label && r.ok; ++i) r.skip_name();
        int n_aux = r.get<int>();
        std::vector<char> is_aux((size_t)max_label + 1);
        r.get_n(is_aux.data(), is_aux.size());
        if (r.ok && n_aux > 0) *has_aux = true;
    }
    return r.ok;
}
}  // namespace

extern "C" int jd_net_load_jwnt(jd_net **out, const char *path, float lm_scale, float ins_penalty)
{
    if (!out || !path) return jd_fail(JD_EINVAL, "jd_net_load_jwnt: null argument");
    BinReader r;
    r.f = fopen(path, "rb");
    if (!r.f) return jd_fail(JD_EFORMAT, "WFSTNetwork::readBinary - error opening input file %s", path);
#define JW_FAIL(...) do { fclose(r.f); return jd_fail(JD_EFORMAT, __VA_ARGS__); } while (0)
    if (!r.id("JWNT")) JW_FAIL("WFSTNetwork::readBinary - invalid ID");
    const int init = r.get<int>(), max_state = r.get<int>();
    (void)r.get<int>();                           // nStates (labelled states)
    (void)r.get<int>();                           // maxOutTransitions
    (void)r.get<int>(); (void)r.get<int>(); (void)r.get<int>();   // wordEndMarker, silMarker, spMarker
    if (!r.ok || max_state < 0 || init < 0 || init > max_state) JW_FAIL("WFSTNetwork::readBinary - bad header");
    const int S = max_state + 1;
    std::vector<int> first((size_t)S, 0), cnt((size_t)S, 0), final_ind((size_t)S, -1);
    std::vector<int> tl;
    for (int i = 0; i < S; ++i) {
        (void)r.get<int>();                       // label
        final_ind[(size_t)i] = r.get<int>();
        const int nt = r.get<int>();
        if (!r.ok || nt < 0) JW_FAIL("WFSTNetwork::readBinary - error reading states[%d]", i);
        cnt[(size_t)i] = nt;
        if (nt > 0) {
            tl.resize((size_t)nt);
            if (!r.get_n(tl.data(), (size_t)nt)) JW_FAIL("WFSTNetwork::readBinary - error reading states[%d].trans array", i);
            first[(size_t)i] = tl[0];
            // the Lite core takes "first transition + count" (getTransitions, WFSTNetwork.cpp:709-721):
            // a state whose list is not a contiguous run would silently decode other arcs there
            for (int k = 1; k < nt; ++k)
                if (tl[(size_t)k] != tl[0] + k) JW_FAIL("transitions of state %d are not contiguous", i);
        }
    }
    const int n_final = r.get<int>();
    if (!r.ok || n_final < 0) JW_FAIL("WFSTNetwork::readBinary - error reading nFinalStates");
    std::vector<float> fweight((size_t)n_final);
    for (int i = 0; i < n_final; ++i) { (void)r.get<int>(); fweight[(size_t)i] = r.get<float>(); }
    const int n_trans = r.get<int>();
    if (!r.ok || n_trans <= 0) JW_FAIL("WFSTNetwork::readBinary - error reading nTransitions");
    struct T5 { int id, to; float w; int in, out; };
    std::vector<T5> tr((size_t)n_trans);
    if (!r.get_n(tr.data(), (size_t)n_trans)) JW_FAIL("WFSTNetwork::readBinary - error reading transitions");
    bool aux = false;
    if (r.get<char>() && !skip_alphabet(r, &aux)) JW_FAIL("WFSTAlphabet::readBinary - bad input alphabet");
    if (r.get<char>() && !skip_alphabet(r, &aux)) JW_FAIL("WFSTAlphabet::readBinary - bad output alphabet");
    if (!r.id("JWNT")) JW_FAIL("WFSTNetwork::readBinary - invalid ID (2)");
    fclose(r.f);
#undef JW_FAIL
    if (aux) return jd_fail(JD_EFORMAT, "%s holds auxiliary (#) symbols: remove them first "
                            "(the Lite core does not understand aux symbols)", path);
    jd_net *n = new jd_net();
    n->n_states = S; n->init = init; n->n_arcs = n_trans;
    n->row_ptr.assign((size_t)S + 1, 0);
    for (int s = 0; s < S; ++s) n->row_ptr[(size_t)s + 1] = n->row_ptr[(size_t)s] + cnt[(size_t)s];
    if (n->row_ptr[(size_t)S] != n_trans) { delete n; return jd_fail(JD_EFORMAT, "%s: state lists cover %d of %d transitions", path, n->row_ptr[(size_t)S], n_trans); }
    n->arcs.resize((size_t)n_trans);
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < cnt[(size_t)s]; ++k) {
            const int64_t i = (int64_t)first[(size_t)s] + k;
            if (i < 0 || i >= n_trans) { delete n; return jd_fail(JD_EFORMAT, "%s: state %d lists transition %lld", path, s, (long long)i); }
            const T5 &t = tr[(size_t)i];
            if (t.to < 0 || t.to >= S || t.in < 0 || t.out < 0) { delete n; return jd_fail(JD_EFORMAT, "%s: transition %lld out of range", path, (long long)i); }
            float w = t.w;                                            // stored without scale and penalty
            if (lm_scale != 1.0f) w *= lm_scale;                      // WFSTNetwork.cpp:1343-1349
            if (ins_penalty != 0.0f && t.out > 0) w += ins_penalty;   // :1351-1357
            n->arcs[(size_t)n->row_ptr[(size_t)s] + k] = JdArc{t.to, w, t.in, t.out};
        }
    // final weights are stored as the writer scaled them and are not rescaled (:1290-1304)
    n->fin_w.assign((size_t)S, std::numeric_limits<float>::infinity());
    n->n_final = n_final;
    for (int s = 0; s < S; ++s) {
        const int fi = final_ind[(size_t)s];
        if (fi < 0) continue;
        if (fi >= n_final) { delete n; return jd_fail(JD_EFORMAT, "%s: states[%d].finalInd out of range", path, s); }
        n->fin_w[(size_t)s] = fweight[(size_t)fi];
    }
    n->max_in = 0;
    for (const JdArc &a : n->arcs) n->max_in = std::max(n->max_in, a.in);
    n->lm_scale = lm_scale; n->ins_penalty = ins_penalty;
    *out = n;
    return JD_OK;
}

// WFSTNetwork::writeBinary (WFSTNetwork.cpp:1106-1225): the penalty, then the scale, are taken
// off the arc weights (float arithmetic, in that order); final weights are written as held.
extern "C" int jd_net_save_jwnt(const jd_net *n, const char *path)
{
    if (!n || !path) return jd_fail(JD_EINVAL, "jd_net_save_jwnt: null argument");
    if (n->lazy_dev) return jd_fail(JD_ESTATE, "jd_net_save_jwnt: a lazily composed network has no finished arc table");
    BinWriter w;
    w.f = fopen(path, "wb");
    if (!w.f) return jd_fail(JD_EFORMAT, "WFSTNetwork::writeBinary - error opening output file %s", path);
    const int S = n->n_states;
    std::vector<char> used((size_t)S, 0);
    int max_out_tr = 0, max_lab = 0, n_used = 0;
    for (int s = 0; s < S; ++s) {
        const int c = n->row_ptr[(size_t)s + 1] - n->row_ptr[(size_t)s];
        if (c > 0) used[(size_t)s] = 1;
        max_out_tr = std::max(max_out_tr, c);
        if (n->fin_w[(size_t)s] < std::numeric_limits<float>::infinity()) used[(size_t)s] = 1;
    }
    for (const JdArc &a : n->arcs) { used[(size_t)a.to] = 1; max_lab = std::max(max_lab, std::max(a.in, a.out)); }
    for (int s = 0; s < S; ++s) n_used += used[(size_t)s];
    w.id("JWNT");
    w.put<int>(n->init); w.put<int>(S - 1); w.put<int>(n_used); w.put<int>(max_out_tr);
    w.put<int>(max_lab + 1); w.put<int>(-1); w.put<int>(-1);          // wordEndMarker, silMarker, spMarker
    int fi = 0;
    std::vector<int> idx;
    for (int s = 0; s < S; ++s) {
        const bool fin = n->fin_w[(size_t)s] < std::numeric_limits<float>::infinity();
        const int b = n->row_ptr[(size_t)s], c = n->row_ptr[(size_t)s + 1] - b;
        w.put<int>(used[(size_t)s] ? s : -1);
        w.put<int>(fin ? fi++ : -1);
        w.put<int>(c);
        idx.resize((size_t)c);
        for (int k = 0; k < c; ++k) idx[(size_t)k] = b + k;
        w.put_n(idx.data(), (size_t)c);
    }
    w.put<int>(fi);
    for (int s = 0; s < S; ++s)
        if (n->fin_w[(size_t)s] < std::numeric_limits<float>::infinity()) { w.put<int>(s); w.put<float>(n->fin_w[(size_t)s]); }
    w.put<int>((int)n->n_arcs);
    for (int64_t i = 0; i < n->n_arcs; ++i) {
        const JdArc &a = n->arcs[(size_t)i];
        float x = a.w;
        if (n->ins_penalty != 0.0f && a.out > 0) x -= n->ins_penalty;  // :1108-1115
        if (n->lm_scale != 1.0f) x /= n->lm_scale;                     // :1120-1126
        w.put<int>((int)i); w.put<int>(a.to); w.put<float>(x); w.put<int>(a.in); w.put<int>(a.out);
    }
    w.put<char>(0); w.put<char>(0);                                    // no alphabets held
    w.id("JWNT");
    const bool bad = ferror(w.f) != 0;
    if (fclose(w.f) != 0 || bad) return jd_fail(JD_EFORMAT, "%s: write error", path);
    return JD_OK;
}

static void build_se_index(const float *trP, int n, int max_n, int16_t *se)     // HTKModels.cpp:2376-2386
{
    for (int j = 1; j < n; ++j) {
        int mn, mx;
        for (mn = (j == n - 1 ? 1 : 0); mn < n - 1; ++mn)
            if (trP[mn * max_n + j] > LZ) break;
        for (mx = n - 1; mx >= 1; --mx)
            if (trP[mx * max_n + j] > LZ) break;
        se[j * 2] = (int16_t)mn;
        se[j * 2 + 1] = (int16_t)(mx + 1);
    }
}

// HTKModels::readBinary (HTKModels.cpp:1110-1233) followed by HTKFlatModels::init
// (HTKFlatModels.cpp:148-176).  The derived values held by the file (sumLogVarPlusNObsLog2Pi,
// logCompWeights, transition logProbs) are used as stored, exactly like the reference; only
// 1.0/var is recomputed.
extern "C" int jd_am_load_jmbi(jd_am **out, const char *path)
{
    if (!out || !path) return jd_fail(JD_EINVAL, "jd_am_load_jmbi: null argument");
    BinReader r;
    r.f = fopen(path, "rb");
    if (!r.f) return jd_fail(JD_EFORMAT, "HTKModels::readBinary - error opening file %s", path);
#define JM_FAIL(...) do { fclose(r.f); return jd_fail(JD_EFORMAT, __VA_ARGS__); } while (0)
    if (!r.id("JMBI")) JM_FAIL("HTKModels::readBinary - invalid ID");
    const int D = r.get<int>(), n_mean = r.get<int>(), n_var = r.get<int>(), n_mixt = r.get<int>();
    const int n_gmm = r.get<int>(), n_tm = r.get<int>(), n_hmm = r.get<int>();
    const int LIM = 1 << 26;
    if (!r.ok || D <= 0 || D > 4096 || n_mean <= 0 || n_var <= 0 || n_mixt <= 0 || n_gmm <= 0 || n_tm <= 0 || n_hmm <= 0 ||
        n_mean > LIM || n_var > LIM || n_mixt > LIM || n_gmm > LIM || n_tm > LIM || n_hmm > LIM)
        JM_FAIL("HTKModels::readBinary - bad header");
    std::vector<float> means((size_t)n_mean * D), vars((size_t)n_var * D), slv((size_t)n_var);
    std::vector<float> tmpv((size_t)D);
    for (int i = 0; i < n_mean; ++i) {
        if (!r.id("JMMN") || !r.skip_name() || !r.get_n(&means[(size_t)i * D], (size_t)D)) JM_FAIL("HTKModels::readBinaryMeanVec - error (vector %d)", i);
    }
    for (int i = 0; i < n_var; ++i) {
        if (!r.id("JMVR") || !r.skip_name() || !r.get_n(&vars[(size_t)i * D], (size_t)D) || !r.get_n(tmpv.data(), (size_t)D))
            JM_FAIL("HTKModels::readBinaryVarVec - error (vector %d)", i);
        slv[(size_t)i] = r.get<float>();
    }
    std::vector<std::vector<int>> mix_mean((size_t)n_mixt), mix_var((size_t)n_mixt);
    int max_mix = 0;
    for (int i = 0; i < n_mixt; ++i) {
        if (!r.id("JMMX") || !r.skip_name()) JM_FAIL("HTKModels::readBinaryMixture - error (mixture %d)", i);
        const int nc = r.get<int>();
        if (!r.ok || nc <= 0 || nc > 65536) JM_FAIL("HTKModels::readBinaryMixture - error reading nComps");
        mix_mean[(size_t)i].resize((size_t)nc); mix_var[(size_t)i].resize((size_t)nc);
        r.get_n(mix_mean[(size_t)i].data(), (size_t)nc); r.get_n(mix_var[(size_t)i].data(), (size_t)nc);
        for (int j = 0; j < nc && r.ok; ++j)
            if (mix_mean[(size_t)i][(size_t)j] < 0 || mix_mean[(size_t)i][(size_t)j] >= n_mean ||
                mix_var[(size_t)i][(size_t)j] < 0 || mix_var[(size_t)i][(size_t)j] >= n_var) r.ok = false;
        if (!r.ok) JM_FAIL("HTKModels::readBinaryMixture - error reading meanVecInds+varVecInds");
        max_mix = std::max(max_mix, nc);
    }
    // HTKFlatModels indexes its flat parameters by GMM but fills them per mixture
    // (HTKFlatModels.h:61-63, .cpp:148-176): only valid when mixtureInd == gmmInd
    if (n_mixt != n_gmm) JM_FAIL("HTKFlatModels: %d mixtures for %d GMMs (shared mixture pools are not supported)", n_mixt, n_gmm);
    std::vector<std::vector<float>> gw((size_t)n_gmm), glw((size_t)n_gmm);
    for (int g = 0; g < n_gmm; ++g) {
        if (!r.id("JMGM") || !r.skip_name()) JM_FAIL("HTKModels::readBinaryGMM - error (GMM %d)", g);
        const int mi = r.get<int>(), nc = r.get<int>();
        if (!r.ok || mi != g) JM_FAIL("HTKFlatModels: GMM %d uses mixture %d (mixtureInd != gmmInd is not supported)", g, mi);
        if (nc != (int)mix_mean[(size_t)g].size()) JM_FAIL("HTKModels::readBinaryGMM - component count mismatch (GMM %d)", g);
        gw[(size_t)g].resize((size_t)nc); glw[(size_t)g].resize((size_t)nc);
        r.get_n(gw[(size_t)g].data(), (size_t)nc); r.get_n(glw[(size_t)g].data(), (size_t)nc);
        if (!r.ok) JM_FAIL("HTKModels::readBinaryGMM - error reading compWeights+nCompWeights");
    }
    struct Tm { int n; std::vector<int> nsuc; std::vector<std::vector<int>> suc; std::vector<std::vector<float>> p, lp; };
    std::vector<Tm> tms((size_t)n_tm);
    int max_n = 0;
    for (int t = 0; t < n_tm; ++t) {
        Tm &m = tms[(size_t)t];
        if (!r.id("JMTM") || !r.skip_name()) JM_FAIL("HTKModels::readBinaryTransMat - error (matrix %d)", t);
        m.n = r.get<int>();
        if (!r.ok || m.n < 3 || m.n > JD_MAXN) JM_FAIL("transition matrix %d: %d states (3..%d supported)", t, m.n, JD_MAXN);
        m.nsuc.resize((size_t)m.n); r.get_n(m.nsuc.data(), (size_t)m.n);
        m.suc.resize((size_t)m.n); m.p.resize((size_t)m.n); m.lp.resize((size_t)m.n);
        for (int i = 0; i < m.n && r.ok; ++i) if (m.nsuc[(size_t)i] < 0 || m.nsuc[(size_t)i] > m.n) r.ok = false;
        for (int i = 0; i < m.n && r.ok; ++i) { m.suc[(size_t)i].resize((size_t)m.nsuc[(size_t)i]); r.get_n(m.suc[(size_t)i].data(), m.suc[(size_t)i].size()); }
        for (int i = 0; i < m.n && r.ok; ++i) { m.p[(size_t)i].resize((size_t)m.nsuc[(size_t)i]); r.get_n(m.p[(size_t)i].data(), m.p[(size_t)i].size()); }
        for (int i = 0; i < m.n && r.ok; ++i) { m.lp[(size_t)i].resize((size_t)m.nsuc[(size_t)i]); r.get_n(m.lp[(size_t)i].data(), m.lp[(size_t)i].size()); }
        for (int i = 0; i < m.n && r.ok; ++i) for (int sj : m.suc[(size_t)i]) if (sj < 0 || sj >= m.n) r.ok = false;
        if (!r.ok) JM_FAIL("HTKModels::readBinaryTransMat - error reading matrix %d", t);
        max_n = std::max(max_n, m.n);
    }
    struct Hm { int n; std::vector<int> g; int tm; };
    std::vector<Hm> hmms((size_t)n_hmm);
    for (int h = 0; h < n_hmm; ++h) {
        Hm &m = hmms[(size_t)h];
        if (!r.id("JMHM") || !r.skip_name()) JM_FAIL("HTKModels::readBinaryHMM - error (HMM %d)", h);
        m.n = r.get<int>();
        if (!r.ok || m.n < 3 || m.n > JD_MAXN) JM_FAIL("HMM %d: %d states (3..%d supported)", h, m.n, JD_MAXN);
        m.g.resize((size_t)m.n); r.get_n(m.g.data(), (size_t)m.n);
        m.tm = r.get<int>();
        if (!r.ok || m.tm < 0 || m.tm >= n_tm || tms[(size_t)m.tm].n != m.n) JM_FAIL("HTKModels::readBinaryHMM - error reading HMM %d", h);
    }
    const bool hybrid = r.get<char>() != 0;
    if (!r.ok) JM_FAIL("HTKModels::readBinary - error reading hybridMode");
    fclose(r.f);
#undef JM_FAIL
    if (hybrid) return jd_fail(JD_EFORMAT, "%s: hybrid (ANN posterior) models are outside this path", path);

    jd_am *a = new jd_am();
    a->D = D; a->n_gmm = n_gmm; a->max_mix = max_mix; a->n_hmm = n_hmm; a->max_n = max_n; a->n_tm = n_tm;
    const size_t gm = (size_t)n_gmm * max_mix;
    a->n_mix.resize((size_t)n_gmm);
    a->det.assign(gm, LZ); a->mean.assign(gm * D, 0.0f); a->ivar.assign(gm * D, 0.0f);
    a->var.assign(gm * D, 1.0f); a->weight.assign(gm, 0.0f); a->sum_log_var.assign(gm, 0.0f); a->log_weight.assign(gm, LZ);
    for (int g = 0; g < n_gmm; ++g) {
        const int nc = (int)mix_mean[(size_t)g].size();
        a->n_mix[(size_t)g] = nc;
        for (int j = 0; j < nc; ++j) {
            const size_t gi = (size_t)g * max_mix + j;
            const float *mu = &means[(size_t)mix_mean[(size_t)g][(size_t)j] * D];
            const float *vv = &vars[(size_t)mix_var[(size_t)g][(size_t)j] * D];
            for (int k = 0; k < D; ++k) {
                a->mean[gi * D + k] = mu[k];                          // HTKFlatModels.cpp:159
                a->ivar[gi * D + k] = (float)(1.0 / vv[k]);           // :160
                a->var[gi * D + k] = vv[k];
            }
            float det = slv[(size_t)mix_var[(size_t)g][(size_t)j]];   // :163
            det += glw[(size_t)g][(size_t)j];                         // :174
            a->det[gi] = det;
            a->sum_log_var[gi] = slv[(size_t)mix_var[(size_t)g][(size_t)j]];
            a->log_weight[gi] = glw[(size_t)g][(size_t)j];
            a->weight[gi] = gw[(size_t)g][(size_t)j];
        }
    }
    a->tm_n.resize((size_t)n_tm);
    a->trP.assign((size_t)n_tm * max_n * max_n, LZ);
    a->transp.assign((size_t)n_tm * max_n * max_n, 0.0f);
    a->se.assign((size_t)n_tm * max_n * 2, 0);
    std::vector<float> tm_tee((size_t)n_tm, LZ);
    for (int t = 0; t < n_tm; ++t) {
        const Tm &m = tms[(size_t)t];
        a->tm_n[(size_t)t] = m.n;
        float *trP = a->trP.data() + (size_t)t * max_n * max_n;
        float *tp = a->transp.data() + (size_t)t * max_n * max_n;
        for (int i = 0; i < m.n; ++i)                                 // HTKModels.cpp:2357-2361
            for (size_t k = 0; k < m.suc[(size_t)i].size(); ++k) {
                trP[i * max_n + m.suc[(size_t)i][k]] = m.lp[(size_t)i][k];
                tp[i * max_n + m.suc[(size_t)i][k]] = m.p[(size_t)i][k];
            }
        build_se_index(trP, m.n, max_n, a->se.data() + (size_t)t * max_n * 2);
        for (size_t k = 1; k < m.suc[0].size(); ++k)                  // :1359-1369 tee weight
            if (m.suc[0][k] == m.n - 1) {
                if (tm_tee[(size_t)t] != LZ) { delete a; return jd_fail(JD_EFORMAT, "HTKModels::addHMM more then one tee transition found (matrix %d)", t); }
                tm_tee[(size_t)t] = m.lp[0][k];
            }
    }
    a->hmm_n.resize((size_t)n_hmm); a->hmm_tm.resize((size_t)n_hmm);
    a->hmm_tee.assign((size_t)n_hmm, LZ);
    a->hmm_gmm.assign((size_t)n_hmm * max_n, -1);
    for (int h = 0; h < n_hmm; ++h) {
        const Hm &m = hmms[(size_t)h];
        a->hmm_n[(size_t)h] = m.n; a->hmm_tm[(size_t)h] = m.tm; a->hmm_tee[(size_t)h] = tm_tee[(size_t)m.tm];
        for (int j = 1; j < m.n - 1; ++j) {
            if (m.g[(size_t)j] < 0 || m.g[(size_t)j] >= n_gmm) { delete a; return jd_fail(JD_EFORMAT, "HMM %d state %d: bad gmm index", h, j); }
            a->hmm_gmm[(size_t)h * max_n + j] = m.g[(size_t)j];
        }
    }
    *out = a;
    return JD_OK;
}

// HTKModels::output(fName, true) (HTKModels.cpp:1044-1105): one unnamed mean / variance vector
// per Gaussian, mixture i belongs to GMM i.
extern "C" int jd_am_save_jmbi(const jd_am *a, const char *path)
{
    if (!a || !path) return jd_fail(JD_EINVAL, "jd_am_save_jmbi: null argument");
    if (a->var.empty() || a->transp.empty()) return jd_fail(JD_ESTATE, "jd_am_save_jmbi: models hold no HTK-level parameters");
    if (a->hybrid) return jd_fail(JD_ESTATE, "jd_am_save_jmbi: hybrid models are not written (create them from the priors)");
    BinWriter w;
    w.f = fopen(path, "wb");
    if (!w.f) return jd_fail(JD_EFORMAT, "HTKModels::output - error opening file %s", path);
    const int D = a->D, G = a->n_gmm, MM = a->max_mix, MN = a->max_n;
    int n_gauss = 0;
    for (int g = 0; g < G; ++g) n_gauss += a->n_mix[(size_t)g];
    w.id("JMBI");
    w.put<int>(D); w.put<int>(n_gauss); w.put<int>(n_gauss); w.put<int>(G); w.put<int>(G); w.put<int>(a->n_tm); w.put<int>(a->n_hmm);
    for (int g = 0; g < G; ++g)
        for (int m = 0; m < a->n_mix[(size_t)g]; ++m) { w.id("JMMN"); w.put<int>(0); w.put_n(&a->mean[((size_t)g * MM + m) * D], (size_t)D); }
    std::vector<float> mh((size_t)D);
    for (int g = 0; g < G; ++g)
        for (int m = 0; m < a->n_mix[(size_t)g]; ++m) {
            const size_t gi = (size_t)g * MM + m;
            w.id("JMVR"); w.put<int>(0);
            w.put_n(&a->var[gi * D], (size_t)D);
            for (int k = 0; k < D; ++k) mh[(size_t)k] = (float)-0.5 / a->var[gi * D + k];     // HTKModels.cpp:863
            w.put_n(mh.data(), (size_t)D);
            w.put<float>(a->sum_log_var[gi]);
        }
    int gi0 = 0;
    std::vector<int> idx;
    for (int g = 0; g < G; ++g) {
        const int nc = a->n_mix[(size_t)g];
        w.id("JMMX"); w.put<int>(0); w.put<int>(nc);
        idx.resize((size_t)nc);
        for (int m = 0; m < nc; ++m) idx[(size_t)m] = gi0 + m;
        w.put_n(idx.data(), (size_t)nc); w.put_n(idx.data(), (size_t)nc);
        gi0 += nc;
    }
    for (int g = 0; g < G; ++g) {
        const int nc = a->n_mix[(size_t)g];
        w.id("JMGM"); w.put<int>(0); w.put<int>(g); w.put<int>(nc);
        w.put_n(&a->weight[(size_t)g * MM], (size_t)nc); w.put_n(&a->log_weight[(size_t)g * MM], (size_t)nc);
    }
    for (int t = 0; t < a->n_tm; ++t) {
        const int n = a->tm_n[(size_t)t];
        const float *tp = &a->transp[(size_t)t * MN * MN], *lp = &a->trP[(size_t)t * MN * MN];
        w.id("JMTM"); w.put<int>(0); w.put<int>(n);
        std::vector<int> nsuc((size_t)n, 0);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (tp[i * MN + j] > 0.0f) ++nsuc[(size_t)i];
        w.put_n(nsuc.data(), (size_t)n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (tp[i * MN + j] > 0.0f) w.put<int>(j);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (tp[i * MN + j] > 0.0f) w.put<float>(tp[i * MN + j]);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (tp[i * MN + j] > 0.0f) w.put<float>(lp[i * MN + j]);
    }
    for (int h = 0; h < a->n_hmm; ++h) {
        const int n = a->hmm_n[(size_t)h];
        w.id("JMHM"); w.put<int>(0); w.put<int>(n);
        w.put_n(&a->hmm_gmm[(size_t)h * MN], (size_t)n);
        w.put<int>(a->hmm_tm[(size_t)h]);
    }
    w.put<char>(0);                                                    // hybridMode
    const bool bad = ferror(w.f) != 0;
    if (fclose(w.f) != 0 || bad) return jd_fail(JD_EFORMAT, "%s: write error", path);
    return JD_OK;
}
