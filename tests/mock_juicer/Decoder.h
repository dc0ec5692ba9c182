// Own-written stand-in for the *declarations* of Juicer's src/Decoder.h (:8, :13-31),
// src/WFSTLattice.h (:20, :52) and src/DecHypHistPool.h (:38-49, :106, :146-165): just enough
// for a compile-only check that include/juicer_amd_decoder.hpp's in-tree branch (#ifdef
// DECODER_H) builds against types living in namespace Juicer.  Test infrastructure only.
#ifndef DECODER_H
#define DECODER_H
#include <cfloat>
#include <cstddef>
#define real float
#define LOG_ZERO (-FLT_MAX)
#define DHHTYPE 1
namespace Juicer {
class WFSTLattice;
struct DecHypHist {
    unsigned char type; int nConnect; DecHypHist *prev;
    int state; int time; real score; real acousticScore; real lmScore;
};
class DecHyp {
public:
    DecHypHist *hist; int state; real score, acousticScore, lmScore;
    char nLabelsNR; int labelsNR[2];
    DecHyp() : hist(NULL), state(-1), score(LOG_ZERO), acousticScore(LOG_ZERO), lmScore(LOG_ZERO), nLabelsNR(0) {}
    virtual ~DecHyp() {}
};
class IDecoder {
public:
    virtual ~IDecoder() {}
    virtual bool modelLevelOutput() = 0;
    virtual WFSTLattice *getLattice() = 0;
    virtual void init() = 0;
    virtual void processFrame(float **inputVec, int currFrame_, int nFrames) = 0;
    virtual DecHyp *finish() = 0;
};
}
#endif
