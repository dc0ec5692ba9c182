// The exact-signature seam of include/juicer_amd_decoder.hpp, RUN: a network and a model set are read from a small
// binary file (written by the test from its synthetic case), put into the mock Juicer::WFSTNetwork /
// Juicer::HTKFlatModels objects, walked through netFromJuicer / modelsFromJuicer, and what the C ABI then holds is
// written back for the test to compare.  With "decode" as the third argument the utterance of the file is decoded
// through `new GpuWFSTDecoder(network, models, startBeam, mainBeam, endBeam, wordBeam, maxHyps)` (needs a GPU).
#include "Decoder.h"
#include "WFSTNetwork.h"
#include "HTKFlatModels.h"
#include "juicer_amd_decoder.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
static_assert(std::is_constructible<JuicerAmd::GpuWFSTDecoder, Juicer::WFSTNetwork *, Juicer::IModels *, real, real, real, real, int>::value,
              "WFSTDecoderLite's constructor signature (WFSTDecoderLite.h:81-89)");
template <typename T> static std::vector<T> rd(FILE *f)
{
    int n = 0;
    if (fread(&n, 4, 1, f) != 1) { fprintf(stderr, "short file\n"); exit(2); }
    std::vector<T> v((size_t)n);
    if (n && fread(&v[0], sizeof(T), (size_t)n, f) != (size_t)n) { fprintf(stderr, "short file\n"); exit(2); }
    return v;
}
template <typename T> static void wr(FILE *f, const std::vector<T> &v) { const int n = (int)v.size(); fwrite(&n, 4, 1, f); if (n) fwrite(&v[0], sizeof(T), v.size(), f); }
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<int> hdr = rd<int>(f);        // nStates, init, D, nGmm, maxMix, nHmm, maxN, nTm, pad
    std::vector<int> from = rd<int>(f), to = rd<int>(f), in = rd<int>(f), outl = rd<int>(f);
    std::vector<float> w = rd<float>(f);
    std::vector<int> fstate = rd<int>(f);
    std::vector<float> fweight = rd<float>(f);
    std::vector<int> nMix = rd<int>(f);
    std::vector<float> det = rd<float>(f), mean = rd<float>(f), ivar = rd<float>(f);
    std::vector<int> hmmN = rd<int>(f), hmmGmm = rd<int>(f), hmmTm = rd<int>(f);
    std::vector<float> tee = rd<float>(f);
    std::vector<int> tmN = rd<int>(f);
    std::vector<float> trP = rd<float>(f);
    std::vector<short> se = rd<short>(f);
    std::vector<float> feats = rd<float>(f);
    std::vector<float> beams = rd<float>(f);  // start, main, end, word, maxHyps
    fclose(f);
    std::vector<Juicer::WFSTTransition> tr(to.size());
    for (size_t a = 0; a < to.size(); ++a) { tr[a].id = (int)a; tr[a].toState = to[a]; tr[a].weight = w[a]; tr[a].inLabel = in[a]; tr[a].outLabel = outl[a]; tr[a].hook = 0; }
    Juicer::WFSTNetwork network(hdr[0], hdr[1], from, tr, fstate, fweight);
    Juicer::HTKFlatModels flat(hdr[2], hdr[3], hdr[4], &nMix[0], &det[0], &mean[0], &ivar[0], hdr[5], hdr[6], &hmmN[0], &hmmGmm[0], &hmmTm[0],
                               &tee[0], hdr[7], &tmN[0], &trP[0], &se[0], hdr[8]);
    Juicer::IModels *models = &flat;
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    if (argc > 3) {                                                    // the drop-in line of juicer.cpp:582-586
        Juicer::IDecoder *decoder = new JuicerAmd::GpuWFSTDecoder(&network, models, beams[0], beams[1], beams[2], beams[3], (int)beams[4]);
        const int D = hdr[2], T = (int)(feats.size() / (size_t)D);
        decoder->init();
        std::vector<float *> rows((size_t)T);
        for (int t = 0; t < T; ++t) rows[(size_t)t] = &feats[(size_t)t * D];
        for (int t = 0; t < T; ++t) decoder->processFrame(&rows[(size_t)t], t, T - t < 20 ? T - t : 20);   // DecoderSingleTest.cpp:267-295
        Juicer::DecHyp *hyp = decoder->finish();
        std::vector<int> lab, tim; std::vector<float> sc;
        for (Juicer::DecHypHist *h = hyp ? hyp->hist : 0; h; h = h->prev) { lab.push_back(h->state); tim.push_back(h->time); sc.push_back(h->score); }
        std::vector<float> tot; if (hyp) { tot.push_back(hyp->score); tot.push_back(hyp->acousticScore); tot.push_back(hyp->lmScore); }
        wr(o, lab); wr(o, tim); wr(o, sc); wr(o, tot);
        delete decoder;
    } else {
        jd_net *n = JuicerAmd::netFromJuicer(&network);
        jd_am *a = JuicerAmd::modelsFromJuicer(models);
        const int S = jd_net_num_states(n), A = (int)jd_net_num_arcs(n);
        std::vector<int> row((size_t)S + 1), t2((size_t)A), i2((size_t)A), o2((size_t)A); std::vector<float> w2((size_t)A), fin((size_t)S);
        if (jd_net_get_csr(n, &row[0], &t2[0], &w2[0], &i2[0], &o2[0], &fin[0]) != JD_OK) return 3;
        std::vector<int> meta; meta.push_back(S); meta.push_back(jd_net_init_state(n)); meta.push_back(jd_am_num_gmms(a)); meta.push_back(jd_am_max_mix(a));
        meta.push_back(jd_am_num_hmms(a)); meta.push_back(jd_am_max_states(a)); meta.push_back(jd_am_num_transmats(a)); meta.push_back(jd_am_vec_size(a));
        const size_t gm = (size_t)meta[2] * meta[3], D = (size_t)meta[7];
        std::vector<float> d2(gm), m2(gm * D), v2(gm * D), trP2((size_t)meta[6] * meta[5] * meta[5]), tee2((size_t)meta[4]);
        std::vector<short> se2((size_t)meta[6] * meta[5] * 2);
        std::vector<int> hn((size_t)meta[4]), hg((size_t)meta[4] * meta[5]), ht((size_t)meta[4]), nm((size_t)meta[2]);
        if (jd_am_get_flat(a, &d2[0], &m2[0], &v2[0]) != JD_OK || jd_am_get_trans(a, &trP2[0], &se2[0], &tee2[0]) != JD_OK ||
            jd_am_get_topology(a, &hn[0], &hg[0], &ht[0], &nm[0]) != JD_OK) return 3;
        wr(o, meta); wr(o, row); wr(o, t2); wr(o, w2); wr(o, i2); wr(o, o2); wr(o, fin);
        wr(o, nm); wr(o, d2); wr(o, m2); wr(o, v2); wr(o, hn); wr(o, hg); wr(o, ht); wr(o, tee2); wr(o, trP2); wr(o, se2);
        jd_net_destroy(n); jd_am_destroy(a);
    }
    fclose(o);
    return 0;
}
