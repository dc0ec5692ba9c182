// jd_batch_test - counterpart of Juicer's DecoderBatchTest harness for the GPU decoder.
//
//   DecoderBatchTest::configureTests  list file, one path per line, '#'/blank skipped  (DecoderBatchTest.cpp:822-844)
//   DecoderBatchTest::run             decode, output, "CPU time .. speech time .. RT factor" (:738-777)
//   DecoderBatchTest::outputResult    ref / trans / mlf / xmlf / verbose formats           (:339-430)
//   -refFName expected results (MLF or one line per file), "Expected :" lines, insertion /
//   deletion / substitution totals at HTK costs 7/7/10                                      (:145-201, :804-939)
//   DecoderSingleTest::extractResultsFromHypWordMode  label-1, start/end frames       (DecoderSingleTest.cpp:403-468)
//
// Networks / models: the text FSM and HTK MMF, or - preferred when present, like
// juicer.cpp:854-866 / :778-784 - Juicer's binary caches "<fsm>.bin" (JWNT) / "<mmf>.bin" (JMBI);
// -writeBinaryFiles writes them after a text load (juicer.cpp:879-882, :791-795).
// Feature files: uncompressed HTK parameter files (12-byte big-endian header + big-endian
// float32 vectors; Tracter's HTKSource is not in the tree, the format is HTK's published one) or
// the build's own .jdf container (int32 T, int32 D, then T*D float32).  Models can also come
// from a .jdam dump (int32 magic 'JDAM', D,n_gmm,max_mix,n_hmm,max_n,n_tm + the jd_am_create_htk
// arrays).  Words are printed through the output symbol table when given, else as integer ids
// (outLabel-1, what vocab->words[] is indexed by).
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "juicer_amd.h"
#include "juicer_amd_decoder.hpp"

static void die(const char *what) { fprintf(stderr, "jd_batch_test: %s: %s\n", what, jd_last_error()); exit(1); }

template <typename T> static std::vector<T> rd(FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "jd_batch_test: short read\n"); exit(1); }
    return v;
}

static jd_am *load_jdam(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "jd_batch_test: cannot open %s\n", path); exit(1); }
    std::vector<int32_t> h = rd<int32_t>(f, 7);
    if (h[0] != 0x4d41444a) { fprintf(stderr, "jd_batch_test: %s is not a .jdam file\n", path); exit(1); }
    const int D = h[1], G = h[2], M = h[3], H = h[4], MN = h[5], NT = h[6];
    std::vector<int32_t> n_mix = rd<int32_t>(f, G);
    std::vector<float> wt = rd<float>(f, (size_t)G * M), mu = rd<float>(f, (size_t)G * M * D), var = rd<float>(f, (size_t)G * M * D);
    std::vector<int32_t> hn = rd<int32_t>(f, H), hg = rd<int32_t>(f, (size_t)H * MN), ht = rd<int32_t>(f, H), tn = rd<int32_t>(f, NT);
    std::vector<float> tp = rd<float>(f, (size_t)NT * MN * MN);
    fclose(f);
    jd_am *am = 0;
    if (jd_am_create_htk(&am, D, G, M, n_mix.data(), wt.data(), mu.data(), var.data(), H, MN, hn.data(), hg.data(),
                         ht.data(), NT, tn.data(), tp.data()))
        die("jd_am_create_htk");
    return am;
}

static bool file_exists(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != 0; }

// One utterance: .jdf, or an HTK parameter file (big-endian: int32 nSamples, int32 sampPeriod,
// int16 sampSize, int16 parmKind; plain float samples or _C compressed 16-bit ones, a _K checksum is ignored).
static bool load_features(const char *path, int D, std::vector<float> &x, int32_t &T)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "jd_batch_test: cannot open %s\n", path); return false; }
    unsigned char h[12];
    if (fread(h, 1, 12, f) != 12) { fprintf(stderr, "jd_batch_test: %s: short header\n", path); fclose(f); return false; }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    auto be32 = [&](int o) { return (int32_t)((uint32_t)h[o] << 24 | (uint32_t)h[o + 1] << 16 | (uint32_t)h[o + 2] << 8 | h[o + 3]); };
    auto be16 = [&](int o) { return (int)(h[o] << 8 | h[o + 1]); };
    int32_t le[2];
    memcpy(le, h, 8);
    const int32_t nS = be32(0);
    const int sampSize = be16(8), parmKind = be16(10);
    if (nS >= 0 && sampSize == D * 4 && size >= 12 + (long)nS * sampSize && !(parmKind & 0x400)) {       // HTK
        T = nS;
        x.resize((size_t)T * D);
        fseek(f, 12, SEEK_SET);
        std::vector<unsigned char> raw((size_t)T * D * 4);
        if (!raw.empty() && fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); return false; }
        for (size_t i = 0; i < x.size(); ++i) {
            const uint32_t v = (uint32_t)raw[4 * i] << 24 | (uint32_t)raw[4 * i + 1] << 16 | (uint32_t)raw[4 * i + 2] << 8 | raw[4 * i + 3];
            memcpy(&x[i], &v, 4);
        }
    } else if (nS >= 4 && (parmKind & 0x400) && sampSize == D * 2 && size >= 12 + 8L * D + (long)(nS - 4) * sampSize) {
        // HTK _C: 16-bit samples, value = (sample + B[k]) / A[k]; the A and B vectors (big-endian floats) sit in front of
        // the data and count as 4 of the header's nSamples (HTK Book, "Storage of parameter files: compression")
        T = nS - 4;
        x.resize((size_t)T * D);
        fseek(f, 12, SEEK_SET);
        std::vector<unsigned char> ab((size_t)D * 8), raw((size_t)T * D * 2);
        if (fread(ab.data(), 1, ab.size(), f) != ab.size() || (!raw.empty() && fread(raw.data(), 1, raw.size(), f) != raw.size())) { fclose(f); return false; }
        std::vector<float> A((size_t)D), B((size_t)D);
        for (int k = 0; k < 2 * D; ++k) {
            const uint32_t v = (uint32_t)ab[4 * k] << 24 | (uint32_t)ab[4 * k + 1] << 16 | (uint32_t)ab[4 * k + 2] << 8 | ab[4 * k + 3];
            memcpy(k < D ? &A[(size_t)k] : &B[(size_t)(k - D)], &v, 4);
        }
        for (size_t i = 0; i < x.size(); ++i) {
            const int16_t sv = (int16_t)((uint16_t)raw[2 * i] << 8 | raw[2 * i + 1]);
            x[i] = ((float)sv + B[i % (size_t)D]) / A[i % (size_t)D];
        }
    } else if (le[0] >= 0 && le[1] == D && size == 8 + (long)le[0] * D * 4) {                             // .jdf
        T = le[0];
        x.resize((size_t)T * D);
        fseek(f, 8, SEEK_SET);
        if (!x.empty() && fread(x.data(), 4, x.size(), f) != x.size()) { fclose(f); return false; }
    } else {
        fprintf(stderr, "jd_batch_test: %s is neither an HTK parameter file (plain or _C) nor a .jdf file of vecSize %d\n", path, D);
        fclose(f);
        return false;
    }
    fclose(f);
    return true;
}

int main(int argc, char **argv)
{
    const char *fsm = 0, *insyms = 0, *outsyms = 0, *amf = 0, *mmf = 0, *list = 0;
    const char *gramFsm = 0, *gramInSyms = 0, *gramOutSyms = 0;              // juicer.cpp:128-130: separate C.L and G
    float mainBeam = 0, startBeam = 0, endBeam = 0, wordBeam = 0, lmScale = 1.0f, insPen = 0.0f;
    int maxHyps = 0, framesPerSec = 100, device = 0, batch = 64, useAdapter = 0, writeBinaryFiles = 0, nDevices = 0, pushing = 0, lazy = 0, nThreads = 0;
    std::string outputFormat = "ref";          // -outputFormat ref|trans|mlf|xmlf|verbose (juicer.cpp:263-264)
    const char *refFName = 0;                  // -refFName: expected results, MLF or one line per file (juicer.cpp:267)
    const char *outputFName = 0;               // -outputFName: "", "stdout", "stderr" or a file (DecoderBatchTest.cpp:216-230)
    int removeSentMarks = 0;                   // -removeSentMarks (juicer.cpp:273)
    std::string sentStartWord, sentEndWord;    // -sentStartWord / -sentEndWord (DecVocabulary)
    int residentSlots = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto nxt = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "-fsmFName") fsm = nxt(); else if (a == "-inSymsFName") insyms = nxt();
        else if (a == "-gramFsmFName") gramFsm = nxt(); else if (a == "-gramInSymsFName") gramInSyms = nxt();
        // juicer.cpp:240 doLabelAndWeightPushing
        else if (a == "-gramOutSymsFName") gramOutSyms = nxt(); else if (a == "-pushing") pushing = JD_PUSH_WEIGHTS | JD_PUSH_LABELS;
        else if (a == "-weightPushing") pushing = JD_PUSH_WEIGHTS;
        else if (a == "-lazy") lazy = 1;                                         // compose where the search goes (jd_net_create_lazy)
        else if (a == "-outSymsFName") outsyms = nxt(); else if (a == "-modelsFName") amf = nxt();
        else if (a == "-htkModelsFName") mmf = nxt();
        else if (a == "-inputFName") list = nxt(); else if (a == "-mainBeam") mainBeam = (float)atof(nxt());
        else if (a == "-phoneStartBeam") startBeam = (float)atof(nxt()); else if (a == "-phoneEndBeam") endBeam = (float)atof(nxt());
        else if (a == "-wordEmitBeam") wordBeam = (float)atof(nxt()); else if (a == "-maxHyps") maxHyps = atoi(nxt());
        else if (a == "-lmScaleFactor") lmScale = (float)atof(nxt()); else if (a == "-insPenalty") insPen = (float)atof(nxt());
        else if (a == "-framesPerSec") framesPerSec = atoi(nxt()); else if (a == "-device") device = atoi(nxt());
        else if (a == "-batch") batch = atoi(nxt()); else if (a == "-perFrameAdapter") useAdapter = 1;
        else if (a == "-devices") nDevices = atoi(nxt());
        else if (a == "-threads") nThreads = atoi(nxt());
        else if (a == "-residentSlots") residentSlots = atoi(nxt());              // jd_dec_set_pipeline(JD_FLOW_RESIDENT): the list through -batch
                                                                                  // one-workgroup slots of a search kernel that stays
        else if (a == "-outputFormat") outputFormat = nxt();
        else if (a == "-writeBinaryFiles") writeBinaryFiles = 1;
        else if (a == "-refFName") refFName = nxt(); else if (a == "-removeSentMarks") removeSentMarks = 1;
        else if (a == "-outputFName") outputFName = nxt();
        else if (a == "-sentStartWord") sentStartWord = nxt(); else if (a == "-sentEndWord") sentEndWord = nxt();
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (!fsm || (!amf && !mmf) || !list) {
        fprintf(stderr, "usage: jd_batch_test -fsmFName F (-htkModelsFName M.mmf | -modelsFName M.jdam) -inputFName LIST [-mainBeam b] [-phoneStartBeam b]\n"
                        "       [-phoneEndBeam b] [-wordEmitBeam b] [-maxHyps n] [-lmScaleFactor s] [-insPenalty p] [-batch n] [-perFrameAdapter]\n"
                        "       [-outputFormat ref|trans|mlf|xmlf|verbose] [-writeBinaryFiles] [-refFName REF] [-removeSentMarks]\n"
                        "       [-sentStartWord W] [-sentEndWord W] [-outSymsFName SYMS] [-outputFName stdout|stderr|FILE]\n"
                        "       [-device d | -devices N   (N GPUs of this node: utterances sharded, one RCCL gather of the 1-best)]\n"
                        "       [-threads N   (N serial harness threads - the reference's loop each - through one decoder: the broker)]\n"
                        "       [-residentSlots N   (a list longer than N goes through N one-workgroup slots of a search kernel that stays: a slot takes the\n"
                        "        next utterance the moment its own is through - jd_dec_set_pipeline, JD_FLOW_RESIDENT; -batch is then at least N)]\n"
                        "       [-gramFsmFName G [-gramInSymsFName S] [-gramOutSymsFName S] [-pushing | -weightPushing] [-lazy]   (-fsmFName is then C.L: composed with G on the\n"
                        "        device, as a whole before the search or - with -lazy - by the search, where it goes)]\n");
        return 2;
    }
    if (outputFName && outputFName[0] && strcmp(outputFName, "stdout") != 0) {     // DecoderBatchTest::openOutputFile
        if (strcmp(outputFName, "stderr") == 0) { if (dup2(2, 1) < 0) { perror("dup2"); return 1; } }
        else if (!freopen(outputFName, "wb", stdout)) { fprintf(stderr, "DecoderBatchTest::setupOutputFile - error opening output file\n"); return 1; }
    }
    jd_net *net = 0, *lazy_cl = 0, *lazy_g = 0;
    const std::string netBin = std::string(fsm) + ".bin";                    // juicer.cpp:854-882
    if (gramFsm) {
        // -gramFsmFName switches to on-the-fly composition (juicer.cpp:332-333): -fsmFName is C.L, loaded with
        // scale 1.0 (:933-940), the grammar with lmScaleFactor (:968-970); here the two are composed on the
        // device and decoded by the static core (jd_net_compose)
        jd_net *cl = 0, *g = 0;
        if (jd_net_load_fsm(&cl, fsm, insyms, outsyms, 1.0f, 0.0f)) die("jd_net_load_fsm (C.L)");
        if (jd_net_load_fsm(&g, gramFsm, gramInSyms, gramOutSyms, lmScale, 0.0f)) die("jd_net_load_fsm (G)");
        if (lazy) {
            // the reference's mode proper: nothing is composed before the search starts (needs the models: tee HMMs)
            lazy_cl = cl; lazy_g = g;
        } else {
            if (jd_net_compose(&net, cl, g, device, 0, 0, pushing)) die("jd_net_compose");
            fprintf(stderr, "C.L (%lld arcs) o G (%lld arcs) composed on device %d: %d states, %lld arcs\n", (long long)jd_net_num_arcs(cl),
                    (long long)jd_net_num_arcs(g), device, (int)jd_net_num_states(net), (long long)jd_net_num_arcs(net));
            jd_net_destroy(cl); jd_net_destroy(g);
        }
    } else if (file_exists(netBin)) {
        fprintf(stderr, "network from pre-existing binary file %s\n", netBin.c_str());
        if (jd_net_load_jwnt(&net, netBin.c_str(), lmScale, insPen)) die("jd_net_load_jwnt");
    } else {
        if (jd_net_load_fsm(&net, fsm, insyms, outsyms, lmScale, insPen)) die("jd_net_load_fsm");
        if (writeBinaryFiles && jd_net_save_jwnt(net, netBin.c_str())) die("jd_net_save_jwnt");
    }
    jd_am *am = 0;
    if (mmf) {                                                               // -htkModelsFName, juicer.cpp:762-797
        const std::string amBin = std::string(mmf) + ".bin";
        if (file_exists(amBin)) {
            fprintf(stderr, "models from pre-existing binary file %s\n", amBin.c_str());
            if (jd_am_load_jmbi(&am, amBin.c_str())) die("jd_am_load_jmbi");
        } else {
            if (jd_am_load_mmf(&am, mmf)) die("jd_am_load_mmf");
            if (writeBinaryFiles && jd_am_save_jmbi(am, amBin.c_str())) die("jd_am_save_jmbi");
        }
    } else am = load_jdam(amf);
    const int D = jd_am_vec_size(am);
    if (lazy_cl && (useAdapter || nDevices > 0)) {
        // the decoder object composes for itself (GpuWFSTOnTheFlyDecoder, the mirror of juicer.cpp:594-598), resp.
        // every device gets a lazily composed network of its own (jd_multi_create_lazy)
    } else if (lazy_cl) {
        if (jd_net_create_lazy(&net, lazy_cl, lazy_g, am, device, 0, 0, pushing)) die("jd_net_create_lazy");
        fprintf(stderr, "C.L (%lld arcs) o G (%lld arcs): composed by the search on device %d\n", (long long)jd_net_num_arcs(lazy_cl),
                (long long)jd_net_num_arcs(lazy_g), device);
        jd_net_destroy(lazy_cl); jd_net_destroy(lazy_g);
    }

    // configureTests: list of input files
    std::vector<std::string> files;
    {
        FILE *f = fopen(list, "rb");
        if (!f) { fprintf(stderr, "DecoderBatchTest::configureTests - error opening input file\n"); return 1; }
        char line[100000];
        while (fgets(line, sizeof line, f)) {
            if (line[0] == '#' || line[0] == '\n' || line[0] == '\r' || line[0] == ' ' || line[0] == '\t' || !line[0]) continue;
            line[strcspn(line, "\r\n")] = 0;
            files.push_back(line);
        }
        fclose(f);
    }
    std::vector<std::vector<float>> feats(files.size());
    std::vector<int32_t> nfr(files.size());
    for (size_t u = 0; u < files.size(); ++u) {
        if (!load_features(files[u].c_str(), D, feats[u], nfr[u])) return 1;
    }

    // word strings: vocab->words[label-1] in the reference (DecoderSingleTest.cpp:443); here the output
    // symbol table of the transducer (symbol id == output label), integers when it is not given
    std::vector<std::string> syms;
    if (outsyms) {
        FILE *f = fopen(outsyms, "rb");
        char line[10000], sym[10000]; int id;
        while (f && fgets(line, sizeof line, f))
            if (sscanf(line, "%s %d", sym, &id) == 2 && id >= 0) { if ((size_t)id >= syms.size()) syms.resize(id + 1); syms[id] = sym; }
        if (f) fclose(f);
    }
    auto word = [&](int label) -> std::string {
        if (label >= 0 && (size_t)label < syms.size() && !syms[label].empty()) return syms[label];
        return std::to_string(label - 1);
    };
    // vocab->getIndex(word): word id (= output label - 1) or -1
    auto word_index = [&](const char *w) -> int {
        if (syms.empty()) {
            char *e = 0;
            const long v = strtol(w, &e, 10);
            return (e && *e == 0 && e != w && v >= 0) ? (int)v : -1;
        }
        for (size_t id = 1; id < syms.size(); ++id) if (syms[id] == w) return (int)id - 1;
        return -1;
    };
    // expected results (DecoderBatchTest::configureTests, DecoderBatchTest.cpp:804-939): an HTK MLF
    // (first line holds "MLF"; entries are matched to input files by base name) or one line of
    // words per input file, in order.
    const bool haveExpResults = refFName && refFName[0];
    std::vector<std::vector<int>> expected(files.size());
    if (haveExpResults) {
        FILE *f = fopen(refFName, "rb");
        if (!f) { fprintf(stderr, "DecoderBatchTest::configureTests - error opening results file\n"); return 1; }
        std::vector<char> buf(100000), nm(100000), rw(1000);
        char *line = buf.data(), *fname = nm.data();
        std::vector<char> have(files.size(), 0);
        bool haveMLF = false;
        if (fgets(line, 1000, f) && strstr(line, "MLF")) haveMLF = true; else fseek(f, 0, SEEK_SET);
        size_t testIndex = 0;
        while (fgets(line, 100000, f)) {
            if (haveMLF) {
                if (sscanf(line, "\"%[^\"]", fname) != 1) continue;
                char *ptr;
                if ((ptr = strrchr(fname, '/')) != 0) memmove(fname, ptr + 1, strlen(ptr) + 1);
                if ((ptr = strrchr(fname, '.')) != 0) *(ptr + 1) = 0;       // keep the '.' (:861)
                for (size_t i = 0; i < files.size(); ++i) {
                    if (!strstr(files[i].c_str(), fname)) continue;
                    if (have[i]) { fprintf(stderr, "DecoderBatchTest::configureTests - duplicate reference transcript %s\n", fname); return 1; }
                    while (fgets(line, 100000, f) && line[0] != '.') {
                        if (sscanf(line, "%999s", rw.data()) != 1) continue;
                        const int id = word_index(rw.data());
                        if (id >= 0) expected[i].push_back(id);
                        else fprintf(stderr, "WARNING: Unknown word in ground truth for %s\n", files[i].c_str());
                    }
                    have[i] = 1;
                    break;
                }
            } else {
                if (testIndex >= files.size()) { fprintf(stderr, "DBT::configureTests - testIndex out of range\n"); return 1; }
                for (char *ptr = strtok(line, " \r\n\t"); ptr; ptr = strtok(0, " \r\n\t")) {
                    const int id = word_index(ptr);
                    if (id < 0) printf("DBT::cfgTests - result word %s not in vocab for test %d\n", ptr, (int)testIndex + 1);
                    expected[testIndex].push_back(id);
                }
                have[testIndex++] = 1;
            }
        }
        fclose(f);
        for (size_t i = 0; i < files.size(); ++i)
            if (!have[i]) { fprintf(stderr, "DBT::configureTests - ref transcription not found for file %s\n", files[i].c_str()); return 1; }
    }
    std::vector<std::vector<int>> actual(files.size());                            // word ids, for the statistics
    if (outputFormat == "mlf" || outputFormat == "xmlf") printf("#!MLF!#\n");     // DecoderBatchTest::openOutputFile
    double decodeTime = 0.0, speechTime = 0.0;
    // DecoderBatchTest::outputResult (DecoderBatchTest.cpp:339-430) on the word list that
    // DecoderSingleTest::extractResultsFromHypWordMode (DecoderSingleTest.cpp:403-468) derives from the
    // DecHyp chain: chain is newest first; start time = previous end time; per-word score deltas.
    auto print_utt = [&](size_t u, int n, const int32_t *label, const int32_t *time, const float *ac, const float *lm,
                         double decTime) {
        fprintf(stderr, "File: %s\n", files[u].c_str());
        // chain entries kept: all, or all but <s> / </s> with -removeSentMarks (DecoderSingleTest.cpp:412-416)
        std::vector<int> keep;
        for (int k = n - 1; k >= 0; --k) {
            if (removeSentMarks) {
                const std::string ws = word(label[k]);
                if ((!sentStartWord.empty() && ws == sentStartWord) || (!sentEndWord.empty() && ws == sentEndWord)) continue;
            }
            keep.push_back(k);
        }
        n = (int)keep.size();
        std::vector<int> lab(n), st(n), et(n);
        std::vector<float> wac(n), wlm(n);
        for (int w = 0; w < n; ++w) {
            const int k = keep[w];                     // chain index of word w
            lab[w] = label[k]; et[w] = time[k];
            st[w] = (w == 0) ? 0 : et[w - 1];
            wac[w] = ac ? ac[k] - ((w > 0) ? ac[keep[w - 1]] : 0.0f) : 0.0f;
            wlm[w] = lm ? lm[k] - ((w > 0) ? lm[keep[w - 1]] : 0.0f) : 0.0f;
            actual[u].push_back(lab[w] - 1);
        }
        if (outputFormat == "ref") {
            for (int w = 0; w < n; ++w) printf("%s ", word(lab[w]).c_str());
            printf("\n");
        } else if (outputFormat == "trans") {
            for (int w = 0; w < n; ++w) printf("%s ", word(lab[w]).c_str());
            printf("(trans-%d)\n", n);
        } else if (outputFormat == "mlf" || outputFormat == "xmlf") {
            std::string base = files[u];
            size_t sl = base.rfind('/'); if (sl != std::string::npos) base = base.substr(sl + 1);
            size_t dot = base.rfind('.'); if (dot != std::string::npos) base = base.substr(0, dot);
            printf("\"*/%s.rec\"\n", base.c_str());
            for (int w = 0; w < n; ++w) {
                if (outputFormat == "mlf") printf("%s\n", word(lab[w]).c_str());
                else {                                 // HTK 100 ns units (:381-403)
                    double s0 = (float)1.0e7 / (float)framesPerSec * (float)st[w];
                    if (s0 > 0) s0 += (float)1.0e7 / (float)framesPerSec;
                    double e0 = (float)1.0e7 / (float)framesPerSec * (float)et[w];
                    if (e0 > 0) e0 += (float)1.0e7 / (float)framesPerSec;
                    printf("%.0f %.0f %s %f\n", s0, e0, word(lab[w]).c_str(), wac[w] + wlm[w]);
                }
            }
            printf(".\n");
        } else {                                       // verbose
            printf("%s\n", files[u].c_str());
            if (haveExpResults) {                      // :311-322
                printf("\tExpected :  ");
                for (int id : expected[u]) { if (id < 0) printf("<OOV> "); else printf("%s ", word(id + 1).c_str()); }
                printf("\n");
            }
            printf("\tActual :    ");
            for (int w = 0; w < n; ++w) printf("%s ", word(lab[w]).c_str());
            printf("  [ ");
            for (int w = 0; w < n; ++w) printf("%d ", et[w] + 1);
            printf("(%d) ]\n", nfr[u]);
        }
        const double uttTime = (double)nfr[u] / framesPerSec;
        fprintf(stderr, "CPU time %.3f  speech time %.3f  RT factor %.3f\n", decTime, uttTime,
                uttTime > 0 ? decTime / uttTime : 0.0);
        decodeTime += decTime; speechTime += uttTime;
    };

    if (useAdapter) {
        // the reference's serial protocol: one IDecoder, frame by frame with 20-row look-ahead
        JuicerAmd::GpuWFSTDecoder *decp = lazy_cl
            ? new JuicerAmd::GpuWFSTOnTheFlyDecoder(lazy_cl, lazy_g, am, mainBeam, endBeam, maxHyps, pushing, device)
            : new JuicerAmd::GpuWFSTDecoder(net, am, startBeam, mainBeam, endBeam, wordBeam, maxHyps, device);
        JuicerAmd::GpuWFSTDecoder &dec = *decp;
        if (lazy_cl) fprintf(stderr, "C.L (%lld arcs) o G (%lld arcs): composed by the search on device %d\n", (long long)jd_net_num_arcs(lazy_cl),
                             (long long)jd_net_num_arcs(lazy_g), device);
        for (size_t u = 0; u < files.size(); ++u) {
            auto t0 = std::chrono::steady_clock::now();
            dec.init();
            std::vector<float *> rows(nfr[u]);
            for (int t = 0; t < nfr[u]; ++t) rows[t] = feats[u].data() + (size_t)t * D;
            int nFrames = 0, nData = nfr[u] < 20 ? nfr[u] : 20;          // DecoderSingleTest.cpp:267-295
            while (nData > 0) {
                dec.processFrame(&rows[nFrames], nFrames, nData);
                ++nFrames;
                if (nFrames + nData - 1 >= nfr[u]) --nData;
            }
            JuicerAmd::DecHyp *hyp = dec.finish();
            if (getenv("PartialTraceInterval") && atoi(getenv("PartialTraceInterval")) > 0) {   // WFSTDecoderLite.cpp:245-258
                std::vector<int> pl, pf;
                dec.partialPaths(pl, pf);
                fprintf(stderr, "Partial paths recovered at frames: ");
                for (size_t k = 0; k < pf.size(); ++k) fprintf(stderr, "%03d ", pf[k]);
                fprintf(stderr, "\n");
            }
            std::vector<int32_t> lab, tim;
            std::vector<float> hac, hlm;
            for (JuicerAmd::DecHypHist *h = hyp ? hyp->hist : 0; h; h = h->prev) {
                lab.push_back(h->state); tim.push_back(h->time); hac.push_back(h->acousticScore); hlm.push_back(h->lmScore);
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            print_utt(u, (int)lab.size(), lab.data(), tim.data(), hac.data(), hlm.data(), dt);
        }
        delete decp;
        if (lazy_cl) { jd_net_destroy(lazy_cl); jd_net_destroy(lazy_g); }
    } else if (nThreads > 0) {
        // -threads N: N harness threads, each the reference's serial loop (init / processFrame x T / finish,
        // DecoderSingleTest.cpp:259-324) over its share of the list - what N juicer processes over split file lists do
        // (doc/userman/juicer_userman.tex:584) - through ONE decoder: a GpuWFSTPooledDecoder per thread, the broker behind
        // them (juicer_amd_decoder.hpp)
        if (lazy_cl) { fprintf(stderr, "jd_batch_test: -threads works on a composed network\n"); return 1; }
        JuicerAmd::GpuDecoderPool pool(net, am, startBeam, mainBeam, endBeam, wordBeam, maxHyps, nThreads, device);
        struct Res { std::vector<int32_t> lab, tim; std::vector<float> ac, lm; double dt = 0.0; };
        std::vector<Res> res(files.size());
        std::vector<std::thread> th;
        const auto t_all = std::chrono::steady_clock::now();
        for (int t = 0; t < nThreads; ++t)
            th.emplace_back([&, t]() {
                JuicerAmd::GpuWFSTPooledDecoder dec(pool);
                for (size_t u = (size_t)t; u < files.size(); u += (size_t)nThreads) {
                    auto t0 = std::chrono::steady_clock::now();
                    dec.init();
                    std::vector<float *> rows(nfr[u]);
                    for (int f = 0; f < nfr[u]; ++f) rows[f] = feats[u].data() + (size_t)f * D;
                    int nFrames = 0, nData = nfr[u] < 20 ? nfr[u] : 20;      // DecoderSingleTest.cpp:267-295
                    while (nData > 0) {
                        dec.processFrame(&rows[nFrames], nFrames, nData);
                        ++nFrames;
                        if (nFrames + nData - 1 >= nfr[u]) --nData;
                    }
                    JuicerAmd::DecHyp *hyp = dec.finish();
                    Res &r = res[u];
                    for (JuicerAmd::DecHypHist *h = hyp ? hyp->hist : 0; h; h = h->prev) {
                        r.lab.push_back(h->state); r.tim.push_back(h->time); r.ac.push_back(h->acousticScore); r.lm.push_back(h->lmScore);
                    }
                    r.dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                }
            });
        for (std::thread &t : th) t.join();
        {
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
            long tot = 0;
            for (size_t u = 0; u < files.size(); ++u) tot += nfr[u];
            fprintf(stderr, "%d harness threads: %ld frames in %.3f s = %.0f frames/s\n", nThreads, tot, wall, wall > 0 ? tot / wall : 0.0);
        }
        for (size_t u = 0; u < files.size(); ++u)
            print_utt(u, (int)res[u].lab.size(), res[u].lab.data(), res[u].tim.data(), res[u].ac.data(), res[u].lm.data(), res[u].dt);
    } else if (nDevices > 0) {
        // -devices N: the utterance loop sharded over N GPUs of this node, one RCCL gather of the 1-best
        jd_multi *mg = 0;
        if (lazy_cl) {
            if (jd_multi_create_lazy(&mg, lazy_cl, lazy_g, am, 0, 0, pushing, startBeam, mainBeam, endBeam, wordBeam, maxHyps, 5, nDevices, 0, batch))
                die("jd_multi_create_lazy");
            fprintf(stderr, "C.L (%lld arcs) o G (%lld arcs): composed by the search on each of %d devices\n", (long long)jd_net_num_arcs(lazy_cl),
                    (long long)jd_net_num_arcs(lazy_g), nDevices);
            jd_net_destroy(lazy_cl); jd_net_destroy(lazy_g);
        } else if (jd_multi_create(&mg, net, am, startBeam, mainBeam, endBeam, wordBeam, maxHyps, 5, nDevices, 0, batch)) die("jd_multi_create");
        std::vector<const float *> ptr(files.size());
        for (size_t u = 0; u < files.size(); ++u) ptr[u] = feats[u].data();
        std::vector<jd_hyp> hyps(files.size());
        auto t0 = std::chrono::steady_clock::now();
        if (jd_multi_decode_batch(mg, (int)files.size(), ptr.data(), nfr.data(), hyps.data())) die("jd_multi_decode_batch");
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long tot = 0;
        for (size_t u = 0; u < files.size(); ++u) tot += nfr[u];
        for (size_t u = 0; u < files.size(); ++u) {
            if (hyps[u].n < 0) fprintf(stderr, "WARNING: no token survived at the end of decoding\n");
            print_utt(u, hyps[u].n > 0 ? hyps[u].n : 0, hyps[u].label, hyps[u].time, hyps[u].ac, hyps[u].lm,
                      tot ? dt * nfr[u] / tot : 0.0);
        }
        jd_multi_destroy(mg);
    } else {
        jd_dec *dec = 0;
        if (residentSlots > batch) batch = residentSlots;
        if (jd_dec_create(&dec, net, am, startBeam, mainBeam, endBeam, wordBeam, maxHyps, 5, device, batch)) die("jd_dec_create");
        if (residentSlots > 0 && jd_dec_set_pipeline(dec, JD_FLOW_RESIDENT, 2, residentSlots)) die("jd_dec_set_pipeline");
        std::vector<const float *> ptr(files.size());
        for (size_t u = 0; u < files.size(); ++u) ptr[u] = feats[u].data();
        std::vector<jd_hyp> hyps(files.size());
        auto t0 = std::chrono::steady_clock::now();
        if (jd_decode_batch(dec, (int)files.size(), ptr.data(), nfr.data(), hyps.data())) die("jd_decode_batch");
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long tot = 0;
        for (size_t u = 0; u < files.size(); ++u) tot += nfr[u];
        for (size_t u = 0; u < files.size(); ++u) {
            if (hyps[u].n < 0) fprintf(stderr, "WARNING: no token survived at the end of decoding\n");
            print_utt(u, hyps[u].n > 0 ? hyps[u].n : 0, hyps[u].label, hyps[u].time, hyps[u].ac, hyps[u].lm,
                      tot ? dt * nfr[u] / tot : 0.0);
        }
        jd_dec_destroy(dec);
    }
    fprintf(stderr, "\n\nTotal CPU time %.3f  Total speech time %.3f  Avg. RT factor %.3f\n", decodeTime, speechTime,
            speechTime > 0 ? decodeTime / speechTime : 0.0);
    // DecoderBatchTest::closeOutputFile -> printStatistics(7, 7, 10) (DecoderBatchTest.cpp:243-253, 145-201):
    // verbose output with expected results ends with the decode-time lines and the edit-distance
    // totals at HTK's insertion / deletion / substitution costs.  The alignment itself is Torch3's
    // EditDistance (not in the tree): restated here as the usual minimum-cost alignment; the layout of
    // the two totals lines is this build's own (parity unpinned).
    if (outputFormat == "verbose" && haveExpResults) {
        const int ci = 7, cd = 7, cs = 10;
        long nIns = 0, nDel = 0, nSub = 0, nRef = 0, nSeq = 0, nSeqOk = 0;
        for (size_t u = 0; u < files.size(); ++u) {
            const std::vector<int> &a = actual[u], &e = expected[u];
            const size_t A = a.size(), E = e.size();
            std::vector<int> cost((A + 1) * (E + 1)), op((A + 1) * (E + 1));     // op: 0 ok, 1 sub, 2 ins, 3 del
            for (size_t i = 0; i <= A; ++i)
                for (size_t j = 0; j <= E; ++j) {
                    int &c = cost[i * (E + 1) + j], &o = op[i * (E + 1) + j];
                    if (!i && !j) { c = 0; o = 0; continue; }
                    c = 0x7fffffff;
                    if (i && j) { const bool eq = a[i - 1] == e[j - 1]; c = cost[(i - 1) * (E + 1) + j - 1] + (eq ? 0 : cs); o = eq ? 0 : 1; }
                    if (i && cost[(i - 1) * (E + 1) + j] + ci < c) { c = cost[(i - 1) * (E + 1) + j] + ci; o = 2; }
                    if (j && cost[i * (E + 1) + j - 1] + cd < c) { c = cost[i * (E + 1) + j - 1] + cd; o = 3; }
                }
            long ui = 0, ud = 0, us = 0;
            for (size_t i = A, j = E; i || j;) {
                const int o = op[i * (E + 1) + j];
                if (o == 2) { ++ui; --i; } else if (o == 3) { ++ud; --j; } else { us += o; --i; --j; }
            }
            nIns += ui; nDel += ud; nSub += us; nRef += (long)E; ++nSeq; nSeqOk += (ui + ud + us) == 0;
        }
        printf("\nTotal time spent decoding = %.2f secs\n", decodeTime);
        printf("Total amount of speech    = %.2f secs\n", speechTime);
        printf("Real-time (RT) factor     = %.2f\n", speechTime > 0 ? decodeTime / speechTime : 0.0);
        printf("total %ld: insert %ld / delete %ld / subst %ld / n_seq %ld / n_seq_correct %ld\n", nRef, nIns, nDel, nSub, nSeq, nSeqOk);
        printf("accuracy %.2f  (insert %.2f%% / delete %.2f%% / subst %.2f%%)\n\n",
               nRef ? 100.0 * (nRef - nIns - nDel - nSub) / nRef : 0.0, nRef ? 100.0 * nIns / nRef : 0.0,
               nRef ? 100.0 * nDel / nRef : 0.0, nRef ? 100.0 * nSub / nRef : 0.0);
    }
    jd_am_destroy(am);
    jd_net_destroy(net);
    return 0;
}
