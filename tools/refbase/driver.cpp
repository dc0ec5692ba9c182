// tools/refbase/driver.cpp - drives the REFERENCE's own WFSTDecoderLite / WFSTDecoderLiteThreading on this build's synthetic
// workloads: BASELINE.md B1 / B2 (timing) and the differential check of the CPU oracle (tests/test_refdiff_cpu.py,
// tools/refbase/run_refbase.py).  This file is this build's code; the decoder, network and model classes it drives are
// compiled from the sources where they lie under /root/reference/src, against the stand-ins of tools/refbase/standins for the
// third-party headers the image lacks (Torch3 general.h / log_add.h, TracterObject.h).  Such a build is NOT a reference build
// and pins nothing; it is a timing and differential aid, run in the build container only.
//
// The loop around the decoder is DecoderSingleTest::decodeUtterance's (src/DecoderSingleTest.cpp:259-307): clock() around
// init .. finish, 20 frames of look-ahead handed to processFrame, CPU time halved for the two-thread decoder.
//
//   refbase_driver key=value ...
//     models=<file.jmbi>                       HTKModels::readBinary (the MMF text parser is generated code the image cannot make), or
//     phones=<list> priors=<file> spm=<n>      hybrid ANN / HMM models: HTKModels::Load(phonesList, priors, statesPerModel)
//     net=<file.jwnt>                          WFSTNetwork::readBinary (src/WFSTNetwork.cpp:1228-1365), or
//     fsm=<file.fsm> insyms=<file> outsyms=<file>   the TEXT constructor (src/WFSTNetwork.cpp:371-616)
//     feats=<file>                             {n_utts, D} then per utterance {T, T x D floats}
//     threading=0|1  main= start= end= word= maxhyps= lmscale= inspen=
//     pti=<frames>                             PARTIAL_DECODING: setPartialDecodeOptions (src/WFSTDecoderLite.cpp:892-896)
//     dumpmodels=<file>                        the reference's prepared model tables (without feats=: nothing is decoded)
//     dumpnet=<file>                           the network as the reference's loader left it (without feats=: nothing is decoded)
//     dumpll=<file> llframes=<n>               the reference's log-likelihoods of every tied state, first frames of the first utterance
//   One JSON line per utterance: the DecHyp chain, the reference's five statistics (its protected totals, read through a
//   subclass - src/WFSTDecoderLite.h:150-154), the frames of the partial paths it recovered.
#include <cassert>
#include <pthread.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "HTKFlatModels.h"
#include "HTKFlatModelsThreading.h"
#include "LogFile.h"
#include "WFSTDecoderLite.h"
#include "WFSTDecoderLiteThreading.h"
#include "WFSTNetwork.h"

using namespace Juicer;

// the reference keeps its statistics and the partial paths protected: a subclass may look
template <class Base> struct Probe : public Base {
    Probe(WFSTNetwork *n, IModels *m, real s, real e, real pe, real w, int mh) : Base(n, m, s, e, pe, w, mh) {}
    void print_extras()
    {
        printf("\"stats\": {\"n_frames\": %d, \"tot_active_emit_hyps\": %d, \"tot_active_end_hyps\": %d, \"tot_active_models\": %d, "
               "\"tot_proc_emit_hyps\": %d, \"tot_proc_end_hyps\": %d}, ", this->currFrame + 1, this->totalActiveEmitHyps, this->totalActiveEndHyps,
               this->totalActiveModels, this->totalProcEmitHyps, this->totalProcEndHyps);
        printf("\"partial\": [");
        const char *sep = "";
        for (size_t i = 0; i < this->partialPaths.size(); ++i) { printf("%s[%d, %d]", sep, this->partialPaths[i]->label, this->partialPaths[i]->frame); sep = ", "; }
        printf("], ");
    }
};

// ... and HTKFlatModels keeps its flat parameter tables protected (src/HTKFlatModels.h:42-58): dumpmodels=<file> writes what the
// REFERENCE's own init() made of the loaded models (src/HTKFlatModels.cpp:94-177: fDets, fMeans, fVars = INVERSE variances) and what its
// IModels interface says about every HMM (states, tee log-probability, log transition matrix, SEIndex: src/Models.h:57-64)
struct FlatProbe : public HTKFlatModels {
    int nTiedStates() { return nGMMs; }
    int dump(const char *fn)
    {
        FILE *f = fopen(fn, "wb");
        if (!f) return 1;
        int nmax = fnMixtures4 > 0 ? fnGaussians / nMixtures : 0;
        int hdr[4] = {nMixtures, nmax, vecSize, getNumHMMs()};
        fwrite(hdr, 4, 4, f);
        for (int g = 0; g < nMixtures; ++g) {
            int nc = fMixtures[g].compNum;
            fwrite(&nc, 4, 1, f);
            fwrite(fDet(g), sizeof(real), (size_t)nmax, f);
            fwrite(fMean(g), sizeof(real), (size_t)nmax * fvecSize4, f);
            fwrite(fVar(g), sizeof(real), (size_t)nmax * fvecSize4, f);
        }
        for (int h = 0; h < getNumHMMs(); ++h) {
            int n = getNumStates(h);
            real tee = getTeeLogProb(h);
            fwrite(&n, 4, 1, f); fwrite(&tee, sizeof(real), 1, f);
            real **tm = getTransMat(h);
            SEIndex *se = getSEIndex(h);
            for (int i = 0; i < n; ++i) fwrite(tm[i], sizeof(real), (size_t)n, f);
            for (int i = 0; i < n; ++i) { short v[2] = {se[i].start, se[i].end}; fwrite(v, 2, 2, f); }
        }
        fclose(f);
        return 0;
    }
};

static void *gmm_thread(void *arg)
{
    ((HTKFlatModelsThreading *)arg)->calcStates();                     // (src/juicer.cpp:79-85)
    return NULL;
}

int main(int argc, char **argv)
{
    std::map<std::string, std::string> kv;
    for (int i = 1; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "refbase_driver: argument '%s' is not key=value\n", argv[i]); return 2; }
        kv[std::string(argv[i], eq - argv[i])] = eq + 1;
    }
    auto S = [&](const char *k, const char *d) { return kv.count(k) ? kv[k] : std::string(d); };
    auto F = [&](const char *k, double d) { return kv.count(k) ? atof(kv[k].c_str()) : d; };
    if ((!kv.count("models") && !kv.count("phones")) || (!kv.count("dumpmodels") && !kv.count("dumpll") && !kv.count("dumpnet") && (!kv.count("feats") || (!kv.count("net") && !kv.count("fsm")))) ||
        (kv.count("dumpll") && !kv.count("feats")) || (kv.count("dumpnet") && !kv.count("net") && !kv.count("fsm"))) {
        fprintf(stderr, "usage: refbase_driver models=.. (net=.. | fsm=.. insyms=.. outsyms=..) feats=.. [threading= main= start= end= word= maxhyps= lmscale= inspen= pti=]\n");
        return 2;
    }
    const int threading = (int)F("threading", 0);
    const float mainBeam = F("main", 0), startBeam = F("start", 0), endBeam = F("end", 0), wordBeam = F("word", 0);
    const int maxHyps = (int)F("maxhyps", 0), pti = (int)F("pti", 0);
    const float lmScale = F("lmscale", 1), insPen = F("inspen", 0);
    const bool hybrid = kv.count("phones") != 0;
    FlatProbe *probe = (threading || hybrid) ? NULL : new FlatProbe();
    IModels *models;
    if (hybrid) {
        // hybrid ANN / HMM models: HTKModels::Load(phonesList, priors, statesPerModel), src/HTKModels.cpp:74-218.  Through the plain HTKModels
        // class: HTKFlatModels::newFrame never sets the `currInput` its own hybrid calcOutput reads (src/HTKFlatModels.cpp:196, 295-306)
        HTKModels *hm = new HTKModels();
        hm->Load(kv["phones"].c_str(), S("priors", "").c_str(), (int)F("spm", 5));
        models = hm;
    } else {
        HTKFlatModels *fm = threading ? (HTKFlatModels *)new HTKFlatModelsThreading() : (HTKFlatModels *)probe;
        fm->setBlockSize(5);                                           // (before the models are there: HTKFlatModels.cpp:308-313)
        fm->readBinary(S("models", "").c_str());
        models = fm;
    }
    if (kv.count("dumpmodels")) {
        if (!probe || probe->dump(kv["dumpmodels"].c_str())) { fprintf(stderr, "refbase_driver: dumpmodels failed\n"); return 1; }
        if (!kv.count("feats")) { fflush(stdout); _exit(0); }
    }
    if (kv.count("dumpll")) {
        // the reference's own log-likelihoods: HTKFlatModels::newFrame + calcOutput(g) (src/HTKFlatModels.cpp:202-293: calcGMMOutput's
        // five-frame blocks, logAdd) for every tied state of the first `llframes` frames of the first utterance -> float32 [frames][G]
        FILE *ff = fopen(S("feats", "").c_str(), "rb"), *fo = fopen(kv["dumpll"].c_str(), "wb");
        int nu = 0, D = 0, T = 0;
        if (!ff || !fo || fread(&nu, 4, 1, ff) != 1 || fread(&D, 4, 1, ff) != 1 || fread(&T, 4, 1, ff) != 1) return 1;
        std::vector<float> x((size_t)T * D);
        if (fread(x.data(), 4, x.size(), ff) != x.size()) return 1;
        std::vector<float *> rows((size_t)T);
        for (int t = 0; t < T; ++t) rows[(size_t)t] = x.data() + (size_t)t * D;
        const int nf = std::min(T, (int)F("llframes", 32)), G = probe ? probe->nTiedStates() : 0;
        int hdr[2] = {nf, G};
        fwrite(hdr, 4, 2, fo);
        for (int t = 0; t < nf; ++t) {
            models->newFrame(t, &rows[(size_t)t], std::min(20, T - t));
            for (int g = 0; g < G; ++g) { const float v = models->calcOutput(g); fwrite(&v, 4, 1, fo); }
        }
        fclose(fo); fclose(ff);
        fflush(stdout); _exit(0);
    }
    pthread_t th;
    if (threading && pthread_create(&th, NULL, gmm_thread, (HTKFlatModelsThreading *)models)) { fprintf(stderr, "pthread_create failed\n"); return 1; }
    WFSTNetwork *net;
    if (kv.count("fsm"))                                               // the text constructor: scales and negates the weights itself (:371-616)
        net = new WFSTNetwork(kv["fsm"].c_str(), S("insyms", "").c_str(), S("outsyms", "").c_str(), lmScale, insPen, REMOVEBOTH);
    else {
        net = new WFSTNetwork(lmScale, insPen);
        net->readBinary(kv["net"].c_str());
    }
    if (kv.count("dumpnet")) {
        // what the REFERENCE's loader made of the network (text constructor or readBinary): per state its transitions in the order
        // getTransitions walks them - {to, in, out, weight} - and the final weight (NaN: not final); src/WFSTNetwork.h:126-165
        FILE *fo = fopen(kv["dumpnet"].c_str(), "wb");
        if (!fo) return 1;
        const int ns = net->getNumStates();
        int hdr[3] = {ns, net->getNumTransitions(), net->getInitState()};
        fwrite(hdr, 4, 3, fo);
        for (int q = 0; q < ns; ++q) {
            const int nt = net->getNumTransitionsOfOneState(q);
            float fw = net->isFinalState(q) ? net->getFinalStateWeight(q) : __builtin_nanf("");
            fwrite(&nt, 4, 1, fo); fwrite(&fw, 4, 1, fo);
            for (int k = 0; k < nt; ++k) {
                const WFSTTransition *t = net->getOneTransition(net->getTransID(q, k));
                int v[3] = {t->toState, t->inLabel, t->outLabel};
                float w = t->weight;
                fwrite(v, 4, 3, fo); fwrite(&w, 4, 1, fo);
            }
        }
        fclose(fo);
        if (!kv.count("feats")) { fflush(stdout); _exit(0); }
    }
    Probe<WFSTDecoderLite> *d1 = threading ? NULL : new Probe<WFSTDecoderLite>(net, models, startBeam, mainBeam, endBeam, wordBeam, maxHyps);
    Probe<WFSTDecoderLiteThreading> *d2 = threading ? new Probe<WFSTDecoderLiteThreading>(net, models, startBeam, mainBeam, endBeam, wordBeam, maxHyps) : NULL;
    WFSTDecoderLite *dec = threading ? (WFSTDecoderLite *)d2 : (WFSTDecoderLite *)d1;
    if (pti > 0) dec->setPartialDecodeOptions(pti);
    FILE *f = fopen(S("feats", "").c_str(), "rb");
    if (!f) { perror("feats"); return 1; }
    int n_utts = 0, D = 0;
    if (fread(&n_utts, 4, 1, f) != 1 || fread(&D, 4, 1, f) != 1) return 1;
    for (int u = 0; u < n_utts; ++u) {
        int T = 0;
        if (fread(&T, 4, 1, f) != 1) return 1;
        std::vector<float> x((size_t)T * D);
        if (T && fread(x.data(), 4, x.size(), f) != x.size()) return 1;
        std::vector<float *> rows((size_t)T);
        for (int t = 0; t < T; ++t) rows[(size_t)t] = x.data() + (size_t)t * D;
        const clock_t t0 = clock();
        dec->init();
        int nFrames = 0, nData = T < 20 ? T : 20;                      // preRead = 20 (DecoderSingleTest.cpp:267-277)
        while (nData > 0) {                                            // :280-295
            dec->processFrame(&rows[(size_t)nFrames], nFrames, nData);
            ++nFrames;
            if (!(nFrames + nData - 1 < T)) --nData;
        }
        DecHyp *hyp = dec->finish();
        double secs = (double)(clock() - t0) / CLOCKS_PER_SEC;
        if (threading) secs /= 2;                                      // :303-307
        printf("{\"u\": %d, \"T\": %d, \"cpu_s\": %.6f, ", u, T, secs);
        if (d1) d1->print_extras(); else d2->print_extras();
        if (!hyp) { printf("\"n\": -1}\n"); fflush(stdout); continue; }
        int n = 0;
        for (DecHypHist *h = hyp->hist; h; h = h->prev) ++n;
        printf("\"n\": %d, \"tot\": [%.9g, %.9g, %.9g], \"label\": [", n, hyp->score, hyp->acousticScore, hyp->lmScore);
        const char *sep = "";
        for (DecHypHist *h = hyp->hist; h; h = h->prev) { printf("%s%d", sep, h->state); sep = ", "; }
        printf("], \"time\": ["); sep = "";
        for (DecHypHist *h = hyp->hist; h; h = h->prev) { printf("%s%d", sep, h->time); sep = ", "; }
        printf("], \"score\": ["); sep = "";
        for (DecHypHist *h = hyp->hist; h; h = h->prev) { printf("%s%.9g", sep, h->score); sep = ", "; }
        printf("], \"ac\": ["); sep = "";
        for (DecHypHist *h = hyp->hist; h; h = h->prev) { printf("%s%.9g", sep, h->acousticScore); sep = ", "; }
        printf("], \"lm\": ["); sep = "";
        for (DecHypHist *h = hyp->hist; h; h = h->prev) { printf("%s%.9g", sep, h->lmScore); sep = ", "; }
        printf("]}\n");
        fflush(stdout);
    }
    // (the scoring thread spins on a plain bool in HTKFlatModelsThreading::calcStates - stop() is not seen by optimised code: the process just ends)
    fflush(stdout);
    _exit(0);
}
