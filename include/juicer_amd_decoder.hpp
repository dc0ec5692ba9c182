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

namespace JuicerAmd {

// Drop-in for `new WFSTDecoderLite(network, models, phoneStartBeam, mainBeam,
// phoneEndBeam, wordEmitBeam, maxHyps)` (juicer.cpp:577-586).  Like the reference it
// does not own the network / models.  Errors follow Torch3 error(): message + exit.
class GpuWFSTDecoder : public IDecoder {
public:
    GpuWFSTDecoder(const jd_net *network, const jd_am *models, float phoneStartPruneWin, float emitPruneWin,
                   float phoneEndPruneWin, float wordPruneWin, int maxEmitHyps, int device = 0,
                   int blockSize = 5, int flushFrames = 64)
        : dec_(0), vecSize_(jd_am_vec_size(models)), nextFrame_(0), flush_(flushFrames)
    {
        check(jd_dec_create(&dec_, network, models, phoneStartPruneWin, emitPruneWin, phoneEndPruneWin,
                            wordPruneWin, maxEmitHyps, blockSize, device, 1));
        if (const char *e = getenv("PartialTraceInterval"))            // WFSTDecoderLite.cpp:116-119 (GetEnv, default 0)
            setPartialDecodeOptions(atoi(e));
    }
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
// offered (as in GpuWFSTDecoder); maxStates / maxArcs: the room the network may grow into (0 = defaults).
struct LazyNetHolder_ {
    jd_net *lazyNet_;
    LazyNetHolder_(const jd_net *cl, const jd_net *g, const jd_am *models, int device, long long maxStates, long long maxArcs, bool pushing)
        : lazyNet_(0)
    {
        if (jd_net_create_lazy(&lazyNet_, cl, g, models, device, maxStates, maxArcs, pushing ? 1 : 0) != JD_OK) {
            fprintf(stderr, "juicer_amd: %s\n", jd_last_error());
            exit(1);
        }
    }
    ~LazyNetHolder_() { jd_net_destroy(lazyNet_); }
};
class GpuWFSTOnTheFlyDecoder : private LazyNetHolder_, public GpuWFSTDecoder {
public:
    GpuWFSTOnTheFlyDecoder(const jd_net *clNetwork, const jd_net *gNetwork, const jd_am *models, float emitPruneWin,
                           float phoneEndPruneWin, int maxEmitHyps, bool doPushing = false, int device = 0,
                           long long maxStates = 0, long long maxArcs = 0, int blockSize = 5, int flushFrames = 64)
        : LazyNetHolder_(clNetwork, gNetwork, models, device, maxStates, maxArcs, doPushing),
          GpuWFSTDecoder(lazyNet_, models, 0.0f, emitPruneWin, phoneEndPruneWin, 0.0f, maxEmitHyps, device, blockSize, flushFrames) {}
    // composed states / arcs materialised so far
    void composedSize(long long &states, long long &arcs) const
    {
        int64_t s = 0, a = 0;
        if (jd_net_lazy_size(lazyNet_, &s, &a) != JD_OK) { fprintf(stderr, "juicer_amd: %s\n", jd_last_error()); exit(1); }
        states = s; arcs = a;
    }
};

}  // namespace JuicerAmd
#endif
