// Own-written stand-in for the *declarations* of Juicer's src/Models.h (:18-67), src/HTKModels.h (:25-200) and
// src/HTKFlatModels.h (:24-71) that the exact-signature bridge of include/juicer_amd_decoder.hpp reads: the
// public IModels getters and the PROTECTED tables of HTKModels / HTKFlatModels, under their own names and types.
// A mock-only constructor fills them from arrays so that the bridge can be RUN in a test.  Test infrastructure only.
#ifndef _HTKFLATMODELS_H
#define _HTKFLATMODELS_H
#ifndef MODELS_H
#define MODELS_H
#endif
#include <vector>
#include "Decoder.h"
namespace Juicer {
typedef struct SEIndex_ { short start; short end; } SEIndex;
class IModels {
public:
    virtual ~IModels() {}
    virtual int getNumHMMs() = 0;
    virtual int getInputVecSize() = 0;
    virtual int getNumStates(int hmmInd) = 0;
    virtual real getTeeLogProb(int hmmInd) = 0;
    virtual real **getTransMat(int) { return NULL; }
    virtual SEIndex *getSEIndex(int) { return NULL; }
};
struct TransMatrix { char *name; int nStates; int *nSucs; int **sucs; real **probs; real **logProbs; SEIndex *seIndexes; real **trP; };
struct Mixture { char *name; int nComps; int *meanVecInds; int *varVecInds; real *currCompOutputs; bool currCompOutputsValid; };
struct GMM { char *name; int mixtureInd; real *compWeights; real *logCompWeights; };
struct HMM { char *name; int nStates; int *gmmInds; int transMatrixInd; real teeWeight; };
class HTKModels : public IModels {
public:
    int getNumHMMs() { return nHMMs; }
    int getInputVecSize() { return vecSize; }
    int getNumStates(int hmmInd) { return hMMs[hmmInd].nStates; }
    real getTeeLogProb(int hmmInd) { return hMMs[hmmInd].teeWeight; }
    real **getTransMat(int hmmInd) { return transMats[hMMs[hmmInd].transMatrixInd].trP; }
    SEIndex *getSEIndex(int hmmInd) { return transMats[hMMs[hmmInd].transMatrixInd].seIndexes; }
protected:
    int vecSize;
    int nTransMats; TransMatrix *transMats;
    int nMixtures; Mixture *mixtures;
    int nGMMs; GMM *gMMs;
    int nHMMs; HMM *hMMs;
    bool hybridMode;
};
typedef struct { int compNum; int compInd; } FMixture;
class HTKFlatModels : public HTKModels {
public:
    // (mock-only constructor; the flat tables are laid out as HTKFlatModels::init leaves them, HTKFlatModels.cpp:94-177:
    // every GMM owns maxMix consecutive component slots, vectors are fvecSize4 apart)
    HTKFlatModels(int D, int nGmm, int maxMix, const int *nMix, const float *det, const float *mean, const float *ivar,
                  int nHmm, int maxN, const int *hmmN, const int *hmmGmm, const int *hmmTm, const float *tee,
                  int nTm, const int *tmN, const float *trP, const short *se, int pad = 0)
    {
        vecSize = D; fvecSize4 = D + pad; hybridMode = false;
        nGMMs = nMixtures = nGmm; nHMMs = nHmm; nTransMats = nTm;
        fm.resize((size_t)nGmm); g.resize((size_t)nGmm);
        fd.assign((size_t)nGmm * maxMix, 0.0f); fmu.assign((size_t)nGmm * maxMix * fvecSize4, 0.0f); fiv = fmu;
        for (int i = 0; i < nGmm; ++i) {
            fm[(size_t)i].compNum = nMix[i]; fm[(size_t)i].compInd = i * maxMix;
            g[(size_t)i].name = 0; g[(size_t)i].mixtureInd = i; g[(size_t)i].compWeights = g[(size_t)i].logCompWeights = 0;
            for (int c = 0; c < nMix[i]; ++c) {
                fd[(size_t)i * maxMix + c] = det[(size_t)i * maxMix + c];
                for (int k = 0; k < D; ++k) {
                    fmu[((size_t)i * maxMix + c) * fvecSize4 + k] = mean[((size_t)i * maxMix + c) * D + k];
                    fiv[((size_t)i * maxMix + c) * fvecSize4 + k] = ivar[((size_t)i * maxMix + c) * D + k];
                }
            }
        }
        tm.resize((size_t)nTm); rows.resize((size_t)nTm); rowp.resize((size_t)nTm); ses.resize((size_t)nTm);
        for (int t = 0; t < nTm; ++t) {
            const int n = tmN[t];
            rows[(size_t)t].assign((size_t)n * n, 0.0f); rowp[(size_t)t].resize((size_t)n); ses[(size_t)t].resize((size_t)n);
            for (int i = 0; i < n; ++i) {
                rowp[(size_t)t][(size_t)i] = &rows[(size_t)t][(size_t)i * n];
                for (int j = 0; j < n; ++j) rows[(size_t)t][(size_t)i * n + j] = trP[((size_t)t * maxN + i) * maxN + j];
                ses[(size_t)t][(size_t)i].start = se[((size_t)t * maxN + i) * 2]; ses[(size_t)t][(size_t)i].end = se[((size_t)t * maxN + i) * 2 + 1];
            }
            TransMatrix &x = tm[(size_t)t];
            x.name = 0; x.nStates = n; x.nSucs = 0; x.sucs = 0; x.probs = x.logProbs = 0; x.seIndexes = &ses[(size_t)t][0]; x.trP = &rowp[(size_t)t][0];
        }
        h.resize((size_t)nHmm); gi.resize((size_t)nHmm);
        for (int i = 0; i < nHmm; ++i) {
            gi[(size_t)i].assign(hmmGmm + (size_t)i * maxN, hmmGmm + (size_t)i * maxN + hmmN[i]);
            h[(size_t)i].name = 0; h[(size_t)i].nStates = hmmN[i]; h[(size_t)i].gmmInds = &gi[(size_t)i][0];
            h[(size_t)i].transMatrixInd = hmmTm[i]; h[(size_t)i].teeWeight = tee[i];
        }
        transMats = &tm[0]; mixtures = 0; gMMs = &g[0]; hMMs = &h[0];
        fMixtures = &fm[0]; fDets = &fd[0]; fMeans = &fmu[0]; fVars = &fiv[0];
    }
protected:
    int fvecSize4;
    FMixture *fMixtures;
    real *fDets, *fMeans, *fVars;
private:
    std::vector<FMixture> fm; std::vector<GMM> g; std::vector<HMM> h; std::vector<TransMatrix> tm;
    std::vector<float> fd, fmu, fiv;
    std::vector<std::vector<float> > rows; std::vector<std::vector<real *> > rowp; std::vector<std::vector<SEIndex> > ses;
    std::vector<std::vector<int> > gi;
};
}
#endif
