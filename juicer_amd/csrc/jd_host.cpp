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
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
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
    *out = n;
    return JD_OK;
}

extern "C" int jd_net_create_csr(jd_net **out, int32_t n_states, int32_t init_state, const int32_t *row_ptr,
                                 const int32_t *to, const float *w, const int32_t *in, const int32_t *outl,
                                 int32_t n_final, const int32_t *fstate, const float *fweight)
{
    if (!out || n_states <= 0 || init_state < 0 || init_state >= n_states || !row_ptr)
        return jd_fail(JD_EINVAL, "jd_net_create_csr: bad arguments");
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
extern "C" void jd_net_destroy(jd_net *n) { delete n; }

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
