// tools/refbase/driver.cpp - times the REFERENCE's own WFSTDecoderLite / WFSTDecoderLiteThreading on this build's synthetic
// workloads.  This file is this build's code; the decoder, network and model classes it drives are
// compiled from the sources where they lie under /root/reference/src, against the stand-ins of tools/refbase/standins for the
// third-party headers the image lacks (Torch3 general.h / log_add.h, TracterObject.h).  Such a build is NOT a reference build
// and pins nothing; it is a timing and differential aid, run in the build container only (tools/refbase/run_refbase.py).
//
// The loop around the decoder is DecoderSingleTest::decodeUtterance's (src/DecoderSingleTest.cpp:259-307): clock() around
// init .. finish, 20 frames of look-ahead handed to processFrame, CPU time halved for the two-thread decoder.
#include <cassert>
#include <pthread.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "HTKFlatModels.h"
#include "HTKFlatModelsThreading.h"
#include "LogFile.h"
#include "WFSTDecoderLite.h"
#include "WFSTDecoderLiteThreading.h"
#include "WFSTNetwork.h"

using namespace Juicer;

static void *gmm_thread(void *arg)
{
    ((HTKFlatModelsThreading *)arg)->calcStates();                     // (src/juicer.cpp:79-85)
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 12) {
        fprintf(stderr, "usage: driver models.jmbi net.jwnt feats.bin threading mainBeam startBeam endBeam wordBeam maxHyps lmScale insPenalty\n");
        return 2;
    }
    const char *jmbi = argv[1], *jwnt = argv[2], *featf = argv[3];
    const int threading = atoi(argv[4]);
    const float mainBeam = atof(argv[5]), startBeam = atof(argv[6]), endBeam = atof(argv[7]), wordBeam = atof(argv[8]);
    const int maxHyps = atoi(argv[9]);
    const float lmScale = atof(argv[10]), insPen = atof(argv[11]);
    HTKFlatModels *models = threading ? new HTKFlatModelsThreading() : new HTKFlatModels();
    models->setBlockSize(5);                                           // (before the models are there: HTKFlatModels.cpp:308-313)
    models->readBinary(jmbi);
    pthread_t th;
    if (threading && pthread_create(&th, NULL, gmm_thread, models)) { fprintf(stderr, "pthread_create failed\n"); return 1; }
    WFSTNetwork *net = new WFSTNetwork(lmScale, insPen);
    net->readBinary(jwnt);
    WFSTDecoderLite *dec = threading ? new WFSTDecoderLiteThreading(net, models, startBeam, mainBeam, endBeam, wordBeam, maxHyps)
                                     : new WFSTDecoderLite(net, models, startBeam, mainBeam, endBeam, wordBeam, maxHyps);
    FILE *f = fopen(featf, "rb");
    if (!f) { perror(featf); return 1; }
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
        if (!hyp) { printf("\"n\": -1}\n"); continue; }
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
