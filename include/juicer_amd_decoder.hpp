// juicer_amd_decoder.hpp - header-only C++ adapter presenting the C ABI of
// juicer_amd.h through Juicer's decoder plugin interface.
//
//   class IDecoder        src/Decoder.h:18-30
//   DecHyp / DecHypHist   src/DecHypHistPool.h:38-49, 146-165
//   WFSTDecoderLite ctor  src/WFSTDecoderLite.h:81-89
//
// Inside the Juicer tree include Juicer's own "Decoder.h" BEFORE this header:
// GpuWFSTDecoder then derives from Juicer::IDecoder and returns Juicer::DecHyp.
// Stand-alone, equivalent mirror types are declared here.
#ifndef JUICER_AMD_DECODER_HPP
#define JUICER_AMD_DECODER_HPP

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "juicer_amd.h"

#ifdef DECODER_H            // Juicer's src/Decoder.h include guard
namespace JuicerAmd {
using Juicer::DecHyp;
using Juicer::DecHypHist;
using Juicer::IDecoder;
typedef Juicer::WFSTLattice LatticeT;     // src/WFSTLattice.h:52 (namespace Juicer, :20)
}
#else
namespace JuicerAmd {
#ifndef DHHTYPE
#define DHHTYPE 1                     // DecHypHistPool.h:106
#endif
struct DecHypHist {                  // DecHypHistPool.h:38-49
    unsigned char type;
    int nConnect;
    DecHypHist *prev;
    int state;                       // output label (word id + 1)
    int time;
    float score, acousticScore, lmScore;
};
struct DecHyp {                      // DecHypHistPool.h:146-165 (fields used by the Lite core)
    DecHypHist *hist;
    int state;
    float score, acousticScore, lmScore;
    DecHyp() : hist(0), state(-1), score(JD_LOG_ZERO), acousticScore(JD_LOG_ZERO), lmScore(JD_LOG_ZERO) {}
};
typedef void LatticeT;
class IDecoder {                     // Decoder.h:18-30
public:
    virtual ~IDecoder() {}
    virtual bool modelLevelOutput() = 0;
    virtual LatticeT *getLattice() = 0;
    virtual void init() = 0;
    virtual void processFrame(float **inputVec, int currFrame_, int nFrames) = 0;
    virtual DecHyp *finish() = 0;
};
}
#endif

#if defined(DECODER_H) && defined(WFST_NETWORK_INC) && defined(_HTKFLATMODELS_H)
// ---- Inside the Juicer tree, with "WFSTNetwork.h" and "HTKFlatModels.h" included as well (juicer.cpp includes
// both): the objects juicer.cpp has ALREADY built (setupNetworks / setupModels, juicer.cpp:664-900) are walked
// into the C ABI's arrays, so that GpuWFSTDecoder has the reference constructor's own signature
//     WFSTDecoderLite(WFSTNetwork*, IModels*, real, real, real, real, int)      src/WFSTDecoderLite.h:81-89
// Nothing is loaded a second time and nothing is recomputed: arc weights carry the scale and penalty the
// network was loaded with, the Gaussian tables are HTKFlatModels::init()'s.
#define JUICER_AMD_HAVE_BRIDGE 1
#include <limits>
namespace JuicerAmd {

// WFSTNetwork -> CSR through its public getters (src/WFSTNetwork.h:129-167)
inline jd_net *netFromJuicer(Juicer::WFSTNetwork *network)
{
    const int nStates = network->getNumStates();
    std::vector<int32_t> row((size_t)nStates + 1, 0), to, in, outl, fstate;
    std::vector<float> w, fweight;
    to.reserve((size_t)network->getNumTransitions()); in.reserve(to.capacity()); outl.reserve(to.capacity()); w.reserve(to.capacity());
    for (int s = 0; s < nStates; ++s) {
        const int n = network->getNumTransitionsOfOneState(s);
        for (int k = 0; k < n; ++k) {                                   // the state's arcs in their stored order (:709-721)
            const Juicer::WFSTTransition *t = network->getOneTransition(network->getTransID(s, k));
            to.push_back(t->toState); w.push_back(t->weight); in.push_back(t->inLabel); outl.push_back(t->outLabel);
        }
        row[(size_t)s + 1] = (int32_t)to.size();
        if (network->isFinalState(s)) { fstate.push_back(s); fweight.push_back(network->getFinalStateWeight(s)); }
    }
    jd_net *net = 0;
    if (jd_net_create_csr(&net, nStates, network->getInitState(), &row[0], to.empty() ? 0 : &to[0], w.empty() ? 0 : &w[0],
                          in.empty() ? 0 : &in[0], outl.empty() ? 0 : &outl[0], (int32_t)fstate.size(),
                          fstate.empty() ? 0 : &fstate[0], fweight.empty() ? 0 : &fweight[0]) != JD_OK) {
        fprintf(stderr, "juicer_amd: %s\n", jd_last_error());
        exit(1);
    }
    return net;
}

// HTKFlatModels keeps its tables protected and IModels (src/Models.h:29-67) has no getter for them: they are read
// through pointers to members formed in a derived class (well-defined: &Derived::member of a protected base member
// has the base's member-pointer type and applies to any object of the base)
struct FlatModelsView_ : public Juicer::HTKFlatModels {
    template <typename T, typename C> static T get(C *m, T C::*p) { return m->*p; }
    static jd_am *make(Juicer::HTKFlatModels *m)
    {
        const int D = m->getInputVecSize(), nHMM = m->getNumHMMs();
        const int nGMM = get<int, Juicer::HTKModels>(m, &FlatModelsView_::nGMMs);
        const int nTM = get<int, Juicer::HTKModels>(m, &FlatModelsView_::nTransMats);
        if (get<bool, Juicer::HTKModels>(m, &FlatModelsView_::hybridMode)) {
            fprintf(stderr, "juicer_amd: hybrid (ANN posterior) models: build them with jd_am_create_hybrid\n");
            exit(1);
        }
        const Juicer::GMM *gmms = get<Juicer::GMM *, Juicer::HTKModels>(m, &FlatModelsView_::gMMs);
        const Juicer::HMM *hmms = get<Juicer::HMM *, Juicer::HTKModels>(m, &FlatModelsView_::hMMs);
        const Juicer::TransMatrix *tms = get<Juicer::TransMatrix *, Juicer::HTKModels>(m, &FlatModelsView_::transMats);
        const Juicer::FMixture *fmix = get<Juicer::FMixture *, Juicer::HTKFlatModels>(m, &FlatModelsView_::fMixtures);
        const float *fdets = get<real *, Juicer::HTKFlatModels>(m, &FlatModelsView_::fDets);
        const float *fmeans = get<real *, Juicer::HTKFlatModels>(m, &FlatModelsView_::fMeans);
        const float *fvars = get<real *, Juicer::HTKFlatModels>(m, &FlatModelsView_::fVars);
        const int stride = get<int, Juicer::HTKFlatModels>(m, &FlatModelsView_::fvecSize4);
        int maxMix = 1, maxN = 3;
        for (int g = 0; g < nGMM; ++g) {
            if (gmms[g].mixtureInd != g) {                              // HTKFlatModels indexes its tables by GMM (HTKFlatModels.h:61-63)
                fprintf(stderr, "juicer_amd: GMM %d shares mixture %d (HTKFlatModels assumes mixtureInd == gmmInd)\n", g, gmms[g].mixtureInd);
                exit(1);
            }
            if (fmix[g].compNum > maxMix) maxMix = fmix[g].compNum;
        }
        for (int t = 0; t < nTM; ++t) if (tms[t].nStates > maxN) maxN = tms[t].nStates;
        std::vector<int32_t> nMix((size_t)nGMM), hmmN((size_t)nHMM), hmmTm((size_t)nHMM), hmmGmm((size_t)nHMM * maxN, -1), tmN((size_t)nTM);
        std::vector<float> det((size_t)nGMM * maxMix, JD_LOG_ZERO), mean((size_t)nGMM * maxMix * D, 0.0f), ivar((size_t)nGMM * maxMix * D, 0.0f);
        std::vector<float> tee((size_t)nHMM), trP((size_t)nTM * maxN * maxN, JD_LOG_ZERO);
        std::vector<int16_t> se((size_t)nTM * maxN * 2, 0);
        for (int g = 0; g < nGMM; ++g) {                                // fDet / fMean / fVar of HTKFlatModels.h:67-69
            nMix[(size_t)g] = fmix[g].compNum;
            for (int c = 0; c < fmix[g].compNum; ++c) {
                const size_t src = (size_t)fmix[g].compInd + c, dst = (size_t)g * maxMix + c;
                det[dst] = fdets[src];
                for (int k = 0; k < D; ++k) { mean[dst * D + k] = fmeans[src * stride + k]; ivar[dst * D + k] = fvars[src * stride + k]; }
            }
        }
        for (int t = 0; t < nTM; ++t) {                                 // createTrPandSEIndex, HTKModels.cpp:2330-2390
            const int n = tms[t].nStates;
            tmN[(size_t)t] = n;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) trP[((size_t)t * maxN + i) * maxN + j] = tms[t].trP[i][j];
            for (int j = 1; j < n; ++j) { se[((size_t)t * maxN + j) * 2] = tms[t].seIndexes[j].start; se[((size_t)t * maxN + j) * 2 + 1] = tms[t].seIndexes[j].end; }
        }
        for (int h = 0; h < nHMM; ++h) {
            hmmN[(size_t)h] = hmms[h].nStates; hmmTm[(size_t)h] = hmms[h].transMatrixInd; tee[(size_t)h] = m->getTeeLogProb(h);
            for (int j = 1; j < hmms[h].nStates - 1; ++j) hmmGmm[(size_t)h * maxN + j] = hmms[h].gmmInds[j];
        }
        jd_am *am = 0;
        if (jd_am_create_flat(&am, D, nGMM, maxMix, &nMix[0], &det[0], &mean[0], &ivar[0], nHMM, maxN, &hmmN[0], &hmmGmm[0], &hmmTm[0],
                              &tee[0], nTM, &tmN[0], &trP[0], &se[0]) != JD_OK) {
            fprintf(stderr, "juicer_amd: %s\n", jd_last_error());
            exit(1);
        }
        return am;
    }
};
inline jd_am *modelsFromJuicer(Juicer::IModels *models)
{
    Juicer::HTKFlatModels *flat = dynamic_cast<Juicer::HTKFlatModels *>(models);
    if (!flat) {                                                        // (juicer.cpp:762-771 builds HTKFlatModels under OPT_FLATMODEL)
        fprintf(stderr, "juicer_amd: GpuWFSTDecoder needs the models as HTKFlatModels (OPT_FLATMODEL)\n");
        exit(1);
    }
    return FlatModelsView_::make(flat);
}
struct BridgedInputs_ {                  // owns what was walked out of the reference's objects (a base: built before the decoder)
    jd_net *bridgedNet_; jd_am *bridgedAm_;
    BridgedInputs_() : bridgedNet_(0), bridgedAm_(0) {}
    BridgedInputs_(Juicer::WFSTNetwork *n, Juicer::IModels *m) : bridgedNet_(netFromJuicer(n)), bridgedAm_(modelsFromJuicer(m)) {}
    ~BridgedInputs_() { jd_net_destroy(bridgedNet_); jd_am_destroy(bridgedAm_); }
};
}
#else
namespace JuicerAmd { struct BridgedInputs_ { jd_net *bridgedNet_; jd_am *bridgedAm_; BridgedInputs_() : bridgedNet_(0), bridgedAm_(0) {} }; }
#endif

namespace JuicerAmd {

// Drop-in for `new WFSTDecoderLite(network, models, phoneStartBeam, mainBeam,
// phoneEndBeam, wordEmitBeam, maxHyps)` (juicer.cpp:577-586).  Like the reference it
// does not own the network / models.  Errors follow Torch3 error(): message + exit.
class GpuWFSTDecoder : private BridgedInputs_, public IDecoder {
public:
    GpuWFSTDecoder(const jd_net *network, const jd_am *models, float phoneStartPruneWin, float emitPruneWin,
                   float phoneEndPruneWin, float wordPruneWin, int maxEmitHyps, int device = 0,
                   int blockSize = 5, int flushFrames = 64)
        : dec_(0), vecSize_(jd_am_vec_size(models)), nextFrame_(0), flush_(flushFrames)
    {
        create(network, models, phoneStartPruneWin, emitPruneWin, phoneEndPruneWin, wordPruneWin, maxEmitHyps, device, blockSize);
    }
#ifdef JUICER_AMD_HAVE_BRIDGE
    // The reference constructor's own signature (src/WFSTDecoderLite.h:81-89; call site juicer.cpp:582-586):
    //     decoder = new GpuWFSTDecoder(network, models, phoneStartBeam, mainBeam, phoneEndBeam, wordEmitBeam, maxHyps);
    // Like the reference it does not own network / models; it keeps its own copies of what it read from them.
    GpuWFSTDecoder(Juicer::WFSTNetwork *network_, Juicer::IModels *models_, real phoneStartPruneWin_, real emitPruneWin_,
                   real phoneEndPruneWin_, real wordPruneWin_, int maxEmitHyps_)
        : BridgedInputs_(network_, models_), dec_(0), vecSize_(jd_am_vec_size(bridgedAm_)), nextFrame_(0), flush_(64)
    {
        int device = 0, blockSize = 5;
        if (const char *e = getenv("JUICER_AMD_DEVICE")) device = atoi(e);
        if (const char *e = getenv("JUICER_AMD_BLOCKSIZE")) blockSize = atoi(e);   // (-blockSize reaches the models only, juicer.cpp:255)
        create(bridgedNet_, bridgedAm_, phoneStartPruneWin_, emitPruneWin_, phoneEndPruneWin_, wordPruneWin_, maxEmitHyps_, device, blockSize);
    }
#endif
    virtual ~GpuWFSTDecoder() { jd_dec_destroy(dec_); }

    bool modelLevelOutput() { return false; }      // WFSTDecoderLite.h:106
    LatticeT *getLattice() { return 0; }           // WFSTDecoderLite.h:107

    void init()                                     // recognitionStart
    {
        check(jd_stream_init(dec_, 0));
        pending_.clear();
        nextFrame_ = 0;
    }

    // inputVec[0] is frame currFrame_; inputVec[1..nFrames-1] are look-ahead rows that will be
    // presented again by later calls (DecoderSingleTest.cpp:267-295), so only row 0 is consumed.
    void processFrame(float **inputVec, int currFrame_, int nFrames)
    {
        (void)nFrames;
        if (currFrame_ != nextFrame_) {             // HTKFlatModels::newFrame, HTKFlatModels.cpp:296-297
            fprintf(stderr, "HTKFlatModels::newFrame - invalid frame\n");
            exit(1);
        }
        pending_.insert(pending_.end(), inputVec[0], inputVec[0] + vecSize_);
        ++nextFrame_;
        if ((int)(pending_.size() / vecSize_) >= flush_) flush();
    }

    DecHyp *finish()                                // recognitionFinish
    {
        flush();
        jd_hyp h;
        check(jd_stream_finish(dec_, 0, &h));
        stats_ = h.stats;
        if (h.n < 0) {
            fprintf(stderr, "WARNING: no token survived at the end of decoding\n");   // WFSTDecoderLite.cpp:266
            return 0;
        }
        hist_.assign(h.n > 0 ? h.n : 0, DecHypHist());
        for (int k = 0; k < h.n; ++k) {             // chain order: hist_[0] is hyp->hist (newest word)
            DecHypHist &d = hist_[k];
            d.type = DHHTYPE; d.nConnect = 1; d.prev = (k + 1 < h.n) ? &hist_[k + 1] : 0;
            d.state = h.label[k]; d.time = h.time[k];
            d.score = h.score[k]; d.acousticScore = h.ac[k]; d.lmScore = h.lm[k];
        }
        hyp_ = DecHyp();
        if (h.n > 0) {
            hyp_.hist = &hist_[0];
            hyp_.score = h.tot_score; hyp_.acousticScore = h.tot_ac; hyp_.lmScore = h.tot_lm;
        }
        return &hyp_;                               // valid until the next init(), like the reference
    }

    const jd_stats &statistics() const { return stats_; }   // WFSTDecoderLite.cpp:231-241

    // WFSTDecoderLite::setMaxAllocModels (WFSTDecoderLite.h:101, .cpp:807-820): before the first init()
    void setMaxAllocModels(int maxAllocModels) { check(jd_dec_set_max_alloc_models(dec_, maxAllocModels)); }
    // PARTIAL_DECODING: setPartialDecodeOptions (.cpp:892-896) and the partialPaths list (.h:199-205)
    // as (output label, frame) pairs, oldest first; traceNow runs tracePartialPath (.cpp:824-868) on
    // the frames processed so far (buffered frames are flushed first)
    void setPartialDecodeOptions(int traceInterval) { check(jd_dec_set_partial_interval(dec_, traceInterval)); }
    bool partialPaths(std::vector<int> &labels, std::vector<int> &frames, bool traceNow = false)
    {
        if (traceNow) flush();
        int n = 0, found = 0;
        check(jd_stream_partial(dec_, 0, traceNow ? 1 : 0, 0, &n, 0, 0, &found));
        labels.assign((size_t)n, 0); frames.assign((size_t)n, 0);
        int f2 = 0;
        if (n > 0) check(jd_stream_partial(dec_, 0, 0, n, &n, &labels[0], &frames[0], &f2));
        return found != 0;
    }

private:
    void create(const jd_net *network, const jd_am *models, float startWin, float emitWin, float endWin, float wordWin,
                int maxEmitHyps, int device, int blockSize)
    {
        check(jd_dec_create(&dec_, network, models, startWin, emitWin, endWin, wordWin, maxEmitHyps, blockSize, device, 1));
        if (const char *e = getenv("PartialTraceInterval"))            // WFSTDecoderLite.cpp:116-119 (GetEnv, default 0)
            setPartialDecodeOptions(atoi(e));
    }
    void flush()
    {
        if (pending_.empty()) return;
        check(jd_stream_push(dec_, 0, &pending_[0], (int)(pending_.size() / vecSize_)));
        pending_.clear();
    }
    static void check(int rc)
    {
        if (rc != JD_OK) { fprintf(stderr, "juicer_amd: %s\n", jd_last_error()); exit(1); }
    }
    jd_dec *dec_;
    int vecSize_, nextFrame_, flush_;
    std::vector<float> pending_;
    std::vector<DecHypHist> hist_;
    DecHyp hyp_;
    jd_stats stats_;
};

// Drop-in for `new WFSTOnTheFlyDecoder(clNetwork, gNetwork, models, mainBeam, phoneEndBeam, maxHyps,
// modelLevelOutput, latticeGeneration, doLabelAndWeightPushing, true)` (juicer.cpp:594-598): C.L and G stay
// apart and are composed where the search goes (jd_net_create_lazy).  The composed network is owned here and
// keeps what has been expanded from one utterance to the next.  modelLevelOutput / latticeGeneration are not
// offered (as in GpuWFSTDecoder); doPushing: JD_PUSH_WEIGHTS | JD_PUSH_LABELS is the reference's doLabelAndWeightPushing
// = true; maxStates / maxArcs: the room the network may grow into (0 = defaults; when it fills up the arena starts
// again between utterances, jd_net_lazy_set_high_water).
struct LazyNetHolder_ {
    jd_net *lazyNet_;
    LazyNetHolder_(const jd_net *cl, const jd_net *g, const jd_am *models, int device, long long maxStates, long long maxArcs, int pushing)
        : lazyNet_(0)
    {
        if (pushing < 0 || pushing > (JD_PUSH_WEIGHTS | JD_PUSH_LABELS)) {
            fprintf(stderr, "juicer_amd: doPushing is a mask of JD_PUSH_WEIGHTS | JD_PUSH_LABELS (got %d)\n", pushing);
            exit(1);
        }
        if (jd_net_create_lazy(&lazyNet_, cl, g, models, device, maxStates, maxArcs, pushing) != JD_OK) {
            fprintf(stderr, "juicer_amd: %s\n", jd_last_error());
            exit(1);
        }
    }
    ~LazyNetHolder_() { jd_net_destroy(lazyNet_); }
};
class GpuWFSTOnTheFlyDecoder : private LazyNetHolder_, public GpuWFSTDecoder {
public:
    GpuWFSTOnTheFlyDecoder(const jd_net *clNetwork, const jd_net *gNetwork, const jd_am *models, float emitPruneWin,
                           float phoneEndPruneWin, int maxEmitHyps, int doPushing = 0, int device = 0,
                           long long maxStates = 0, long long maxArcs = 0, int blockSize = 5, int flushFrames = 64)
        : LazyNetHolder_(clNetwork, gNetwork, models, device, maxStates, maxArcs, doPushing),
          GpuWFSTDecoder(lazyNet_, models, 0.0f, emitPruneWin, phoneEndPruneWin, 0.0f, maxEmitHyps, device, blockSize, flushFrames) {}
    // the reference's own argument type: `true` is doLabelAndWeightPushing, i.e. BOTH halves (a bool passed to the int
    // mask above would silently mean JD_PUSH_WEIGHTS alone)
    GpuWFSTOnTheFlyDecoder(const jd_net *clNetwork, const jd_net *gNetwork, const jd_am *models, float emitPruneWin,
                           float phoneEndPruneWin, int maxEmitHyps, bool doLabelAndWeightPushing, int device = 0,
                           long long maxStates = 0, long long maxArcs = 0, int blockSize = 5, int flushFrames = 64)
        : LazyNetHolder_(clNetwork, gNetwork, models, device, maxStates, maxArcs,
                         doLabelAndWeightPushing ? (JD_PUSH_WEIGHTS | JD_PUSH_LABELS) : 0),
          GpuWFSTDecoder(lazyNet_, models, 0.0f, emitPruneWin, phoneEndPruneWin, 0.0f, maxEmitHyps, device, blockSize, flushFrames) {}
    // composed states / arcs materialised so far
    void composedSize(long long &states, long long &arcs) const
    {
        int64_t s = 0, a = 0;
        if (jd_net_lazy_size(lazyNet_, &s, &a) != JD_OK) { fprintf(stderr, "juicer_amd: %s\n", jd_last_error()); exit(1); }
        states = s; arcs = a;
    }
};

// ---- several IDecoder instances on ONE decoder (jd_broker_*, juicer_amd.h).
//
// The reference scales out by running juicer several times over split file lists (doc/userman/juicer_userman.tex:584);
// a GPU wants the utterances of all those runs side by side on one chip.  A pool owns one decoder with a stream per
// caller and the broker in front of it; every harness thread gets a GpuWFSTPooledDecoder - an IDecoder like any other,
// driven serially with init / processFrame / finish - and the broker's worker thread turns what the callers have pushed
// since its last tick into one scoring launch and one search launch:
//     JuicerAmd::GpuDecoderPool pool(network, models, startBeam, mainBeam, endBeam, wordBeam, maxHyps, nThreads);
//     ... in thread t:   JuicerAmd::GpuWFSTPooledDecoder decoder(pool);   // then exactly as with WFSTDecoderLite
class GpuDecoderPool {
public:
    GpuDecoderPool(const jd_net *network, const jd_am *models, float phoneStartPruneWin, float emitPruneWin, float phoneEndPruneWin,
                   float wordPruneWin, int maxEmitHyps, int nCallers, int device = 0, int blockSize = 5)
        : dec_(0), broker_(0), vecSize_(jd_am_vec_size(models))
    {
        check(jd_dec_create(&dec_, network, models, phoneStartPruneWin, emitPruneWin, phoneEndPruneWin, wordPruneWin, maxEmitHyps, blockSize,
                            device, nCallers));
        check(jd_broker_create(&broker_, dec_, nCallers));
    }
    ~GpuDecoderPool() { jd_broker_destroy(broker_); jd_dec_destroy(dec_); }
    jd_broker *broker() const { return broker_; }
    int vecSize() const { return vecSize_; }
    static void check(int rc)
    {
        if (rc != JD_OK) { fprintf(stderr, "juicer_amd: %s\n", jd_last_error()); exit(1); }
    }
private:
    GpuDecoderPool(const GpuDecoderPool &);
    GpuDecoderPool &operator=(const GpuDecoderPool &);
    jd_dec *dec_;
    jd_broker *broker_;
    int vecSize_;
};

class GpuWFSTPooledDecoder : public IDecoder {
public:
    explicit GpuWFSTPooledDecoder(GpuDecoderPool &pool, int flushFrames = 64)
        : b_(pool.broker()), client_(-1), vecSize_(pool.vecSize()), nextFrame_(0), flush_(flushFrames)
    {
        GpuDecoderPool::check(jd_broker_open(b_, &client_));
    }
    virtual ~GpuWFSTPooledDecoder() { if (client_ >= 0) (void)jd_broker_close(b_, client_); }
    bool modelLevelOutput() { return false; }      // WFSTDecoderLite.h:106
    LatticeT *getLattice() { return 0; }           // WFSTDecoderLite.h:107
    void init()
    {
        GpuDecoderPool::check(jd_broker_init(b_, client_));
        pending_.clear();
        nextFrame_ = 0;
    }
    void processFrame(float **inputVec, int currFrame_, int nFrames)
    {
        (void)nFrames;
        if (currFrame_ != nextFrame_) {             // HTKFlatModels::newFrame, HTKFlatModels.cpp:296-297
            fprintf(stderr, "HTKFlatModels::newFrame - invalid frame\n");
            exit(1);
        }
        pending_.insert(pending_.end(), inputVec[0], inputVec[0] + vecSize_);
        ++nextFrame_;
        if ((int)(pending_.size() / vecSize_) >= flush_) flush();
    }
    DecHyp *finish()
    {
        flush();
        jd_hyp h;
        GpuDecoderPool::check(jd_broker_finish(b_, client_, &h));
        stats_ = h.stats;
        if (h.n < 0) {
            fprintf(stderr, "WARNING: no token survived at the end of decoding\n");   // WFSTDecoderLite.cpp:266
            return 0;
        }
        hist_.assign(h.n > 0 ? h.n : 0, DecHypHist());
        for (int k = 0; k < h.n; ++k) {             // chain order: hist_[0] is hyp->hist (newest word)
            DecHypHist &d = hist_[k];
            d.type = DHHTYPE; d.nConnect = 1; d.prev = (k + 1 < h.n) ? &hist_[k + 1] : 0;
            d.state = h.label[k]; d.time = h.time[k];
            d.score = h.score[k]; d.acousticScore = h.ac[k]; d.lmScore = h.lm[k];
        }
        hyp_ = DecHyp();
        if (h.n > 0) {
            hyp_.hist = &hist_[0];
            hyp_.score = h.tot_score; hyp_.acousticScore = h.tot_ac; hyp_.lmScore = h.tot_lm;
        }
        return &hyp_;                               // valid until the next init(), like the reference
    }
    const jd_stats &statistics() const { return stats_; }
private:
    void flush()
    {
        if (pending_.empty()) return;
        GpuDecoderPool::check(jd_broker_push(b_, client_, &pending_[0], (int)(pending_.size() / vecSize_)));
        pending_.clear();
    }
    jd_broker *b_;
    int client_, vecSize_, nextFrame_, flush_;
    std::vector<float> pending_;
    std::vector<DecHypHist> hist_;
    DecHyp hyp_;
    jd_stats stats_;
};

}  // namespace JuicerAmd
#endif
