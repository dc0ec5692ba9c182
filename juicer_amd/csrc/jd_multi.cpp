// jd_multi.cpp - the utterance loop of DecoderBatchTest::run (src/DecoderBatchTest.cpp:738-771)
// sharded over the GPUs of one node from C++: one jd_dec per device, one host thread per device,
// utterances dealt by length (longest first, each to the device with the fewest frames so far: they
// are independent - the reference decodes them serially), no data-path collective - and ONE RCCL
// all-gather of padded 1-best records at the end, after which every device holds every hypothesis
// (the host reads them from the first one).  A record is as long as the batch's longest hypothesis.
//
// RCCL is loaded with dlopen when the first multi-device decoder is created: libjuicer_amd.so has
// no link-time dependency on it (a process that already carries PyTorch's own copy must not get a
// second one through us).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <string>
#include <thread>
#include <vector>

#include "jd_internal.h"

#define JM_HDR 6                               // record = n, n_frames, tot[3], utterance, label[L], time[L], score[L], ac[L], lm[L]

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_rccl.h) return JD_OK;
    // RCCL writes its version banner (and, with NCCL_DEBUG set, its log) to stdout when the first communicator is
    // created; stdout is where the harness writes its results (DecoderBatchTest.cpp:216-230).  RCCL's own switch for
    // that is NCCL_DEBUG_FILE: its output goes to stderr unless the caller has chosen a file - set once, before the
    // library is loaded and reads its environment.  RCCL honours the file from level WARN up only: NCCL_DEBUG=VERSION
    // (what the GPU boxes of this project export; tools/rccl_banner_probe.py) prints the banner on stdout whatever
    // the file says, so that one level is dropped - at every other level, and with no level, stdout stays clean.
    // (Round 3 parked file descriptor 1 around ncclCommInitAll with dup2: process-wide, and it swallowed whatever
    // another thread printed meanwhile.)
    (void)setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
    if (const char *lv = getenv("NCCL_DEBUG")) if (strcasecmp(lv, "VERSION") == 0) (void)unsetenv("NCCL_DEBUG");
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return jd_fail(JD_EHIP, "cannot load librccl.so: %s", dlerror());
#define SYM(field, name) do { *(void **)&g_rccl.field = dlsym(h, name); if (!g_rccl.field) return jd_fail(JD_EHIP, "librccl.so lacks %s", name); } while (0)
    SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(AllGather, "ncclAllGather");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return JD_OK;
}
}  // namespace

struct jd_multi {
    int n_dev = 0;
    std::vector<int> devices;
    std::vector<jd_dec *> dec;
    std::vector<jd_net *> own_net;                 // jd_multi_create_lazy: one lazily composed network per device
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<int32_t *> d_send, d_recv;
    size_t cap_per_dev = 0;                        // 32-bit words per device the gather buffers hold
    // storage behind the jd_hyp pointers handed out (valid until the next decode)
    std::vector<std::vector<int32_t>> label, time;
    std::vector<std::vector<float>> score, ac, lm;
};

extern "C" void jd_multi_destroy(jd_multi *m)
{
    if (!m) return;
    for (int d = 0; d < m->n_dev; ++d) {
        (void)hipSetDevice(m->devices[(size_t)d]);
        if ((size_t)d < m->d_send.size() && m->d_send[(size_t)d]) (void)hipFree(m->d_send[(size_t)d]);
        if ((size_t)d < m->d_recv.size() && m->d_recv[(size_t)d]) (void)hipFree(m->d_recv[(size_t)d]);
        if ((size_t)d < m->stream.size() && m->stream[(size_t)d]) (void)hipStreamDestroy(m->stream[(size_t)d]);
        if ((size_t)d < m->comm.size() && m->comm[(size_t)d] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[(size_t)d]);
        if ((size_t)d < m->dec.size()) jd_dec_destroy(m->dec[(size_t)d]);
        if ((size_t)d < m->own_net.size()) jd_net_destroy(m->own_net[(size_t)d]);
    }
    delete m;
}

// net: one network for every device, or (lazy_cl, lazy_g): a lazily composed network per device
static int multi_create(jd_multi **out, const jd_net *net, const jd_net *lazy_cl, const jd_net *lazy_g, int64_t lazy_states,
                        int64_t lazy_arcs, int32_t pushing, const jd_am *am, float start_beam, float main_beam,
                        float end_beam, float word_beam, int32_t max_hyps, int32_t block_size, int32_t n_devices,
                        const int32_t *devices, int32_t max_streams_per_device)
{
    if (!out || (!net && !(lazy_cl && lazy_g)) || !am || n_devices < 1) return jd_fail(JD_EINVAL, "jd_multi_create: bad argument");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0)
        return jd_fail(JD_ENODEV, "no HIP device available; juicer_amd has no CPU fallback");
    jd_multi *m = new jd_multi();
    m->n_dev = n_devices;
    for (int d = 0; d < n_devices; ++d) {
        const int dev = devices ? devices[d] : d;
        if (dev < 0 || dev >= have || std::find(m->devices.begin(), m->devices.end(), dev) != m->devices.end()) {
            delete m;
            return jd_fail(JD_ENODEV, "jd_multi_create: device %d of %d requested, %d visible (each device once)", dev, n_devices, have);
        }
        m->devices.push_back(dev);
    }
    int rc = load_rccl();
    if (rc) { delete m; return rc; }
    m->dec.assign((size_t)n_devices, nullptr);
    m->comm.assign((size_t)n_devices, nullptr);
    m->stream.assign((size_t)n_devices, nullptr);
    m->d_send.assign((size_t)n_devices, nullptr);
    m->d_recv.assign((size_t)n_devices, nullptr);
    if (!net) m->own_net.assign((size_t)n_devices, nullptr);
    for (int d = 0; d < n_devices; ++d) {
        if (!net) {
            rc = jd_net_create_lazy(&m->own_net[(size_t)d], lazy_cl, lazy_g, am, m->devices[(size_t)d], lazy_states, lazy_arcs, pushing);
            if (rc) { jd_multi_destroy(m); return rc; }
        }
        rc = jd_dec_create(&m->dec[(size_t)d], net ? net : m->own_net[(size_t)d], am, start_beam, main_beam, end_beam, word_beam, max_hyps, block_size,
                           m->devices[(size_t)d], max_streams_per_device);
        if (rc) { jd_multi_destroy(m); return rc; }
        if (hipSetDevice(m->devices[(size_t)d]) != hipSuccess || hipStreamCreate(&m->stream[(size_t)d]) != hipSuccess) {
            jd_multi_destroy(m);
            return jd_fail(JD_EHIP, "jd_multi_create: cannot create a stream on device %d", m->devices[(size_t)d]);
        }
    }
    const ncclResult_t nr = g_rccl.CommInitAll(m->comm.data(), n_devices, m->devices.data());   // (its banner: load_rccl)
    if (nr != ncclSuccess) {
        const char *why = g_rccl.GetErrorString(nr);
        jd_multi_destroy(m);
        return jd_fail(JD_EHIP, "ncclCommInitAll over %d devices failed: %s", n_devices, why);
    }
    *out = m;
    return JD_OK;
}

extern "C" int jd_multi_create(jd_multi **out, const jd_net *net, const jd_am *am, float start_beam, float main_beam,
                               float end_beam, float word_beam, int32_t max_hyps, int32_t block_size, int32_t n_devices,
                               const int32_t *devices, int32_t max_streams_per_device)
{
    if (!net) return jd_fail(JD_EINVAL, "jd_multi_create: bad argument");
    return multi_create(out, net, nullptr, nullptr, 0, 0, 0, am, start_beam, main_beam, end_beam, word_beam, max_hyps, block_size,
                        n_devices, devices, max_streams_per_device);
}

extern "C" int jd_multi_create_lazy(jd_multi **out, const jd_net *cl, const jd_net *g, const jd_am *am, int64_t max_states,
                                    int64_t max_arcs, int32_t pushing, float start_beam, float main_beam, float end_beam,
                                    float word_beam, int32_t max_hyps, int32_t block_size, int32_t n_devices,
                                    const int32_t *devices, int32_t max_streams_per_device)
{
    if (!cl || !g) return jd_fail(JD_EINVAL, "jd_multi_create_lazy: bad argument");
    return multi_create(out, nullptr, cl, g, max_states, max_arcs, pushing, am, start_beam, main_beam, end_beam, word_beam, max_hyps,
                        block_size, n_devices, devices, max_streams_per_device);
}

// Longest-processing-time-first: utterances by decreasing length, each to the shard with the fewest frames
// so far.  A batch's duration is its slowest device's; contiguous shards of a list sorted by
// anything else than length can differ by the length of their longest utterances.
static void lpt_shards(int32_t n_utts, const int32_t *n_frames, int N, std::vector<std::vector<int>> &idx)
{
    idx.assign((size_t)N, {});
    std::vector<int> order((size_t)n_utts);
    for (int u = 0; u < n_utts; ++u) order[(size_t)u] = u;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_frames[a] > n_frames[b]; });
    std::vector<long long> load((size_t)N, 0);
    for (int u : order) {
        int best = 0;
        for (int d = 1; d < N; ++d)
            if (load[(size_t)d] < load[(size_t)best] || (load[(size_t)d] == load[(size_t)best] && idx[(size_t)d].size() < idx[(size_t)best].size())) best = d;
        idx[(size_t)best].push_back(u);
        load[(size_t)best] += std::max(n_frames[u], 0) + 1;               // (+1: empty utterances are spread as well)
    }
}

static void pack(const jd_hyp &h, int utt, int32_t *r, size_t L)
{
    const size_t rec = JM_HDR + 5 * L;
    memset(r, 0, rec * sizeof(int32_t));
    const size_t k = (size_t)std::max(0, (int)h.n);
    r[0] = h.n; r[1] = h.stats.n_frames; r[5] = utt;
    memcpy(r + 2, &h.tot_score, 4); memcpy(r + 3, &h.tot_ac, 4); memcpy(r + 4, &h.tot_lm, 4);
    if (k) {
        memcpy(r + JM_HDR, h.label, k * 4); memcpy(r + JM_HDR + L, h.time, k * 4);
        memcpy(r + JM_HDR + 2 * L, h.score, k * 4); memcpy(r + JM_HDR + 3 * L, h.ac, k * 4);
        memcpy(r + JM_HDR + 4 * L, h.lm, k * 4);
    }
}

extern "C" int jd_multi_decode_batch(jd_multi *m, int32_t n_utts, const float *const *feats, const int32_t *n_frames, jd_hyp *out)
{
    if (!m || !feats || !n_frames || !out || n_utts < 0) return jd_fail(JD_EINVAL, "jd_multi_decode_batch: bad argument");
    const int N = m->n_dev;
    // ---- shards balanced by frames, one host thread per device
    std::vector<std::vector<int>> idx;
    lpt_shards(n_utts, n_frames, N, idx);
    size_t per = 0;                                                          // records every device contributes (padded)
    for (int d = 0; d < N; ++d) per = std::max(per, idx[(size_t)d].size());
    std::vector<int> rcs((size_t)N, JD_OK);
    std::vector<std::string> errs((size_t)N);
    std::vector<std::vector<jd_hyp>> local((size_t)N);
    std::vector<std::thread> th;
    for (int d = 0; d < N; ++d)
        th.emplace_back([&, d]() {
            const std::vector<int> &mine = idx[(size_t)d];
            const int n = (int)mine.size();
            local[(size_t)d].resize((size_t)n);
            if (n == 0) return;
            std::vector<const float *> f((size_t)n);
            std::vector<int32_t> nf((size_t)n);
            for (int i = 0; i < n; ++i) { f[(size_t)i] = feats[mine[(size_t)i]]; nf[(size_t)i] = n_frames[mine[(size_t)i]]; }
            // (an utterance that fails leaves an empty hypothesis and the error; the others of the shard are decoded)
            rcs[(size_t)d] = jd_decode_batch(m->dec[(size_t)d], n, f.data(), nf.data(), local[(size_t)d].data());
            if (rcs[(size_t)d]) errs[(size_t)d] = jd_last_error();           // (thread-local in the library)
        });
    for (auto &t : th) t.join();
    int first_err = JD_OK;
    std::string first_msg;
    for (int d = 0; d < N; ++d)
        if (rcs[(size_t)d] && first_err == JD_OK) { first_err = rcs[(size_t)d]; first_msg = "device " + std::to_string(m->devices[(size_t)d]) + ": " + errs[(size_t)d]; }
    // ---- the one collective: all-gather of the padded records (RCCL over xGMI between the GPUs); a record holds
    // the longest hypothesis of the batch
    size_t L = 1;
    for (int d = 0; d < N; ++d)
        for (const jd_hyp &h : local[(size_t)d]) L = std::max(L, (size_t)std::max(0, (int)h.n));
    const size_t rec = JM_HDR + 5 * L, words = per * rec;
    std::vector<std::vector<int32_t>> send((size_t)N);
    for (int d = 0; d < N; ++d) {
        send[(size_t)d].assign(words, 0);
        for (size_t i = 0; i < per; ++i) {
            int32_t *r = send[(size_t)d].data() + i * rec;
            if (i < idx[(size_t)d].size()) pack(local[(size_t)d][i], idx[(size_t)d][i], r, L);
            else r[0] = -2;                                                  // padding
        }
    }
    if (words > m->cap_per_dev) {
        for (int d = 0; d < N; ++d) {
            if (hipSetDevice(m->devices[(size_t)d]) != hipSuccess) return jd_fail(JD_EHIP, "hipSetDevice failed");
            if (m->d_send[(size_t)d]) (void)hipFree(m->d_send[(size_t)d]);
            if (m->d_recv[(size_t)d]) (void)hipFree(m->d_recv[(size_t)d]);
            m->d_send[(size_t)d] = m->d_recv[(size_t)d] = nullptr;
            if (hipMalloc(&m->d_send[(size_t)d], words * sizeof(int32_t)) != hipSuccess ||
                hipMalloc(&m->d_recv[(size_t)d], words * sizeof(int32_t) * (size_t)N) != hipSuccess) {
                m->cap_per_dev = 0;
                return jd_fail(JD_EHIP, "jd_multi_decode_batch: hipMalloc of the gather buffers failed");
            }
        }
        m->cap_per_dev = words;
    }
    std::vector<int32_t> all(words * (size_t)N);
    if (per > 0) {
        for (int d = 0; d < N; ++d) {
            if (hipSetDevice(m->devices[(size_t)d]) != hipSuccess ||
                hipMemcpyAsync(m->d_send[(size_t)d], send[(size_t)d].data(), words * sizeof(int32_t), hipMemcpyHostToDevice,
                               m->stream[(size_t)d]) != hipSuccess)
                return jd_fail(JD_EHIP, "jd_multi_decode_batch: upload of the records failed");
        }
        ncclResult_t nr = g_rccl.GroupStart();
        for (int d = 0; d < N && nr == ncclSuccess; ++d)
            nr = g_rccl.AllGather(m->d_send[(size_t)d], m->d_recv[(size_t)d], words, ncclInt32, m->comm[(size_t)d], m->stream[(size_t)d]);
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess)
            return jd_fail(JD_EHIP, "ncclAllGather failed: %s", g_rccl.GetErrorString(nr != ncclSuccess ? nr : ne));
        for (int d = 0; d < N; ++d)
            if (hipSetDevice(m->devices[(size_t)d]) != hipSuccess || hipStreamSynchronize(m->stream[(size_t)d]) != hipSuccess)
                return jd_fail(JD_EHIP, "jd_multi_decode_batch: the gather did not complete on device %d", m->devices[(size_t)d]);
        if (hipSetDevice(m->devices[0]) != hipSuccess ||
            hipMemcpy(all.data(), m->d_recv[0], all.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
            return jd_fail(JD_EHIP, "jd_multi_decode_batch: download of the gathered records failed");
    }
    // ---- unpack into the caller's order (a record names its utterance; the padding records carry n = -2)
    m->label.assign((size_t)n_utts, {}); m->time.assign((size_t)n_utts, {});
    m->score.assign((size_t)n_utts, {}); m->ac.assign((size_t)n_utts, {}); m->lm.assign((size_t)n_utts, {});
    int seen = 0;
    for (int d = 0; d < N; ++d)
        for (size_t i = 0; i < per; ++i) {
            const int32_t *r = all.data() + ((size_t)d * per + i) * rec;
            if (r[0] == -2) continue;
            const int u = r[5];
            if (u < 0 || u >= n_utts || i >= idx[(size_t)d].size() || idx[(size_t)d][i] != u)
                return jd_fail(JD_EHIP, "jd_multi_decode_batch: gathered record %zu of device %d names utterance %d", i, d, u);
            jd_hyp &H = out[u];
            H = local[(size_t)d][i];                                         // statistics stay local; the rest comes from the gather
            const size_t k = (size_t)std::max(0, (int)r[0]);
            H.n = r[0];
            memcpy(&H.tot_score, r + 2, 4); memcpy(&H.tot_ac, r + 3, 4); memcpy(&H.tot_lm, r + 4, 4);
            m->label[(size_t)u].assign(r + JM_HDR, r + JM_HDR + k); m->time[(size_t)u].assign(r + JM_HDR + L, r + JM_HDR + L + k);
            m->score[(size_t)u].resize(k); m->ac[(size_t)u].resize(k); m->lm[(size_t)u].resize(k);
            memcpy(m->score[(size_t)u].data(), r + JM_HDR + 2 * L, k * 4);
            memcpy(m->ac[(size_t)u].data(), r + JM_HDR + 3 * L, k * 4);
            memcpy(m->lm[(size_t)u].data(), r + JM_HDR + 4 * L, k * 4);
            H.label = m->label[(size_t)u].data(); H.time = m->time[(size_t)u].data();
            H.score = m->score[(size_t)u].data(); H.ac = m->ac[(size_t)u].data(); H.lm = m->lm[(size_t)u].data();
            ++seen;
        }
    if (seen != n_utts) return jd_fail(JD_EHIP, "jd_multi_decode_batch: %d records gathered for %d utterances", seen, n_utts);
    if (first_err) return jd_fail(first_err, "%s", first_msg.c_str());
    return JD_OK;
}
