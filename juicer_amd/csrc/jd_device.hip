// jd_device.hip - gfx950 (MI355X) kernels + decoder runtime of juicer_amd.
//
// Two kernels make up the hot path (reference: WFSTDecoderLite::processFrame,
// src/WFSTDecoderLite.cpp:311-372, and HTKFlatModels::calcGMMOutput,
// src/HTKFlatModels.cpp:226-262):
//
//  jd_gmm_kernel     companion kernel.  Diagonal-GMM log-likelihood of every
//                    tied state for a tile of 64 stream-frames.  One lane owns
//                    one frame (its 39-dim vector staged through LDS into
//                    registers); the tied state is wave-uniform, so its
//                    mean/inverse-variance stream arrives through the scalar
//                    cache and the per-lane work is pure VALU in the
//                    reference's operation order (no FMA contraction, no MFMA:
//                    elementwise + reduction).  The log-sum-exp over mixtures is
//                    the reference's sequential logAdd chain, evaluated per lane.
//
//  k_search          persistent token-passing search (jd_search.h): every utterance
//                    stream is served by a cluster of workgroups that runs all
//                    frames of a chunk without returning to the host.  Per frame:
//                    (A) HMM-internal propagation over the active arc instances
//                    with beam / histogram pruning and ballot compaction into
//                    wave-owned list segments, (X) frontier expansion over the CSR
//                    arc table iterated to epsilon/tee closure, with 64-bit
//                    atomic-max Viterbi recombination into per-arc keys and
//                    word-boundary Path records appended to a device arena.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cerrno>
#include <numeric>
#include <queue>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <type_traits>
#include <vector>

#include "jd_internal.h"

#define LZ (-3.402823466e+38f)
#define GMM_ROWS 64             // stream-frames per GMM tile (one per lane)
#define GMM_GT 64               // tied states per GMM workgroup (16 per wave)
#define GMM_GT_SMALL 16         // ... of a launch with few rows (jd_gmm_kernel39 only)
#define HIST_MAX_BINS 2048

struct __align__(16) Tok { float score, ac, lm; int path; };
struct __align__(16) PathRec { int prev, frame, label, pad0; float score, ac, lm, pad1; };

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return jd_fail(JD_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                           __FILE__, __LINE__);                                              \
    } while (0)

// ------------------------------------------------------------------ device utils

__device__ __forceinline__ unsigned f2o(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o)
{
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int rank_in(unsigned long long bal)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
}

// ------------------------------------------------------------------- scoring kernels
#include "jd_gmm.h"

// --------------------------------------------------------------- search kernels
#include "jd_search.h"

// ------------------------------------------- Path collection, partial trace, finish, launch helpers
#include "jd_gc.h"
#include "jd_resident.h"
#include "jd_slot.h"

// --------------------------------------------------------------- host runtime

#include "jd_host_scoring.h"

#define JD_MAX_DEVICES 64
static std::mutex g_search_mu[JD_MAX_DEVICES];     // one persistent search launch at a time per device (launch_search)
static std::atomic<long long> g_search_turn[JD_MAX_DEVICES];  // counts the holders of the lock: who gives it up for a waiter sees the waiter take it
static std::atomic<int> g_search_waiters[JD_MAX_DEVICES];   // ... and who waits for it (a resident kernel makes room: jd_res_should_yield)

// ... and across PROCESSES: the reference's way of using several cores is several processes over split file lists
// (doc/userman/juicer_userman.tex:584), and two of them pointed at one GPU would each get part of the CUs for their
// persistent launch and spin at their cluster barriers until the 30 s time-out.  An advisory file lock per device -
// named after its PCI bus id, so that every process means the same GPU whatever its visible-device order - is held
// from the dispatch of a search launch to its completion: the processes take turns launch by launch (a launch is
// milliseconds) and both finish.  JD_GPU_LOCK=0 switches it off, JD_GPU_LOCK_DIR moves the files (default /tmp); a
// lock file that cannot be opened leaves the launch unguarded, as before.
struct GpuFileLock {
    int fd = -2;                                   // -2: not tried yet, -1: unavailable
    std::mutex mu;
};
static GpuFileLock g_gpu_lock[JD_MAX_DEVICES];
static int gpu_lock_fd(int device)
{
    GpuFileLock &L = g_gpu_lock[(size_t)std::min(std::max(device, 0), JD_MAX_DEVICES - 1)];
    std::lock_guard<std::mutex> lk(L.mu);
    if (L.fd != -2) return L.fd;
    L.fd = -1;
    const char *off = getenv("JD_GPU_LOCK");
    if (off && atoi(off) == 0) return -1;
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *p = bus; *p; ++p) if (*p == ':' || *p == '/') *p = '_';
    const char *dir = getenv("JD_GPU_LOCK_DIR");
    char path[512];
    snprintf(path, sizeof path, "%s/juicer_amd.gpu-%s.lock", dir && *dir ? dir : "/tmp", bus);
    L.fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0666);
    return L.fd;
}
struct GpuLockGuard {
    int fd;
    explicit GpuLockGuard(int device) : fd(gpu_lock_fd(device)) { if (fd >= 0) while (flock(fd, LOCK_EX) != 0 && errno == EINTR) { } }
    ~GpuLockGuard() { if (fd >= 0) (void)flock(fd, LOCK_UN); }
};

// The stream arenas of a decoder are ONE slab, and the slab of a destroyed decoder is kept (one per device, the
// largest) for the next decoder on that device: what a decoder's set-up costs is the driver clearing the bytes it
// hands out (25-60 GB/s, tools/alloc_probe.py - seconds for the default 70 % of a 288 GB device, every time a
// decoder follows another one), and nothing in the arenas needs to be clear except what ensure_arenas clears itself.
// JD_ARENA_CACHE=0 switches it off; jd_release_cached_memory gives the slab back.
struct ArenaSlab { void *p = nullptr; size_t bytes = 0; };
static std::mutex g_slab_mu;
static ArenaSlab g_slab[JD_MAX_DEVICES];
static size_t slab_cached_bytes(int device)
{
    std::lock_guard<std::mutex> lk(g_slab_mu);
    return (device >= 0 && device < JD_MAX_DEVICES) ? g_slab[device].bytes : 0;
}
static int slab_take(int device, size_t bytes, ArenaSlab *out)
{
    {
        std::lock_guard<std::mutex> lk(g_slab_mu);
        if (device >= 0 && device < JD_MAX_DEVICES && g_slab[device].p) {
            ArenaSlab &c = g_slab[device];
            if (c.bytes >= bytes) { *out = c; c = ArenaSlab(); return JD_OK; }
            (void)hipFree(c.p);                                        // too small: its bytes go towards the new one
            c = ArenaSlab();
        }
    }
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return jd_fail(e == hipErrorOutOfMemory ? JD_ENOMEM : JD_EHIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    out->p = q; out->bytes = std::max<size_t>(bytes, 256);
    return JD_OK;
}
static void slab_give(int device, ArenaSlab sl)
{
    if (!sl.p) return;
    const char *e = getenv("JD_ARENA_CACHE");
    const bool keep = !(e && atoi(e) == 0) && device >= 0 && device < JD_MAX_DEVICES;
    std::lock_guard<std::mutex> lk(g_slab_mu);
    if (keep && g_slab[device].bytes < sl.bytes) std::swap(g_slab[device], sl);
    if (sl.p) (void)hipFree(sl.p);
}
extern "C" int jd_release_cached_memory(int32_t device)
{
    if (device < 0 || device >= JD_MAX_DEVICES) return jd_fail(JD_EINVAL, "jd_release_cached_memory: bad device");
    ArenaSlab sl;
    { std::lock_guard<std::mutex> lk(g_slab_mu); std::swap(sl, g_slab[device]); }
    if (sl.p) { if (hipSetDevice(device) != hipSuccess) return jd_fail(JD_EHIP, "hipSetDevice(%d) failed", device); (void)hipFree(sl.p); }
    return JD_OK;
}

struct HostResult {
    std::vector<int32_t> label, time;
    std::vector<float> score, ac, lm;
};

// How one wave of utterances is laid out in likelihood tables (plan_wave): frames per chunk, the rows of every chunk
// (stream after stream, packed) and the source frame of every row.
struct WavePlan {
    int nb = 0, Fc = 128, n_chunks = 1, maxT = 0;
    std::vector<int> T;                        // frames per stream
    std::vector<size_t> chunk_row0;            // first entry of chunk c in row_src
    // [chunk][stream]: first row of the stream in the chunk's table; rows per chunk (a multiple of the scoring tile)
    std::vector<int> row_off, chunk_rows;
    std::vector<int> row_src;                  // row -> frame of d_feats (-1: unused)
    size_t max_rows = 0, n_rows_all = 0;
};
// A batch announced with jd_dec_prefetch_scores: scored on the scoring stream WHILE the batch before it is searched
// (the blocks of the scoring kernel take the CUs that clusters of the persistent search leave as their streams end).
struct Prefetch {
    int state = 0;                             // 0 none, 1 announced, 2 scoring enqueued (table d_ll[buf], event ev1)
    const float *feats = nullptr; int nb = 0;
    std::vector<int64_t> ustart, ulen;
    WavePlan plan;
    int buf = 0;
    int bank = -1;                             // >= 0: its utterances have been started, on this bank of streams ("two batches in flight")
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void drop_events() { if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); ev0 = ev1 = nullptr; }
    void drop() { drop_events(); state = 0; feats = nullptr; bank = -1; }
};

struct jd_dec {
    const jd_net *net = nullptr;
    const jd_am *am = nullptr;
    int device = 0, max_streams = 0, block_size = 5;
    DecConst C{};
    AmDevBuf amb;
    // device copies of static data
    int *d_row_ptr = nullptr; JdArc *d_arcs = nullptr; float *d_fin_w = nullptr; int *d_aux = nullptr;
    XState *d_xst = nullptr;              // per state: the decoder's arc order (DecConst::xst)
    int *d_se32 = nullptr;
    float *d_hmm_tee = nullptr, *d_trP = nullptr, *d_hmm_tmax0 = nullptr, *d_lrt = nullptr;
    int *d_pcount = nullptr;               // per state: Path objects of the reference per arriving token (DecConst::pcount)
    std::vector<int> state_new;           // the decoder's own numbering of the states: state_new[network state] (empty: the network's)
    // per-stream state
    StreamDev *d_streams = nullptr;
    StreamCtl *d_ctl = nullptr;
    std::vector<StreamDev> h_streams;          // host mirror (arena pointers)
    std::vector<void *> allocs;
    ArenaSlab slab;                       // the stream arenas (ensure_arenas)
    bool arenas_ready = false;
    int64_t cap_slots = 0, cap_paths = 0, cap_items = 0, cap_new = 0;
    int64_t slots_hint = 0;               // jd_dec_set_max_alloc_models
    int res_cap = 8192;
    int *d_res = nullptr;                 // result arena, see ensure_arenas
    int n_cus = 256;
    // search launches: one 512-thread workgroup per CU, a cluster of them per stream
    int max_cw = MAXCW;                   // upper bound of workgroups per stream cluster (JD_CW overrides)
    int weighted = 1;                     // size the clusters by the work ahead of each stream (JD_WEIGHTED=0: uniform)
    int rebalance = 1;                    // cut a launch short and plan the rest anew when part of the grid idles (JD_REBALANCE=0: off)
    double rebalance_frac = 0.2;          // ... this part (JD_REBALANCE_FRAC)
    double rebalance_min_us = 4000.0;     // ... and only launches predicted to last this long (JD_REBALANCE_MIN_US; 0 in tests)
    double model_a_us = 10.0, model_b_us = 360.0;   // cost model of a stream-frame: a + b / workgroups (launch_search), as fitted in round 2
    double pf_gmm_weight = 1.35;          // scoring beside a search is priced at this times its CU-time on its own (the CUs come late,
                                          // and piecemeal; JD_PF_GMM_WEIGHT)
    int plan_min_cw = 2;                  // greedy plan: workgroups every stream starts with (JD_PLAN_MIN_CW)
    int plan_mode = 0;                    // 0: round 2's constants + bisection; 1: measured curve (model2_*) + greedy whole workgroups (JD_PLAN)
    double model2_a_us = 24.6, model2_b_us = 61.4;   // ... as measured in round 3 (JD_MODEL2_A / JD_MODEL2_B)
    // b was fitted at configs[1]'s load (23.7 k instances + arcs per stream-frame); it scales with the load,
    // which a decoder learns from the batches it has decoded (first batch: as fitted)
    double load_scale = 1.0, load_sum = 0.0, load_frames = 0.0;
    int4 *d_work = nullptr; int work_cap = 0;
    unsigned *d_cells = nullptr; size_t cells_words = 0;   // diagnostics (jd_dec_debug_cells): a bit per cell of the likelihood slab
    int *d_status = nullptr; int *h_status = nullptr;
    bool xl_ok = true;                    // XCD-local launches allowed (JD_XCD_LOCAL=0 or one failed placement check switch them off)
    double xl_slack = 1.04;               // ... when the packed plan is predicted to end no later than this times the unpacked one (JD_XL_SLACK)
    long long *d_dbg = nullptr, *d_dbg_buf = nullptr;   // in-kernel cycle accounting (jd_dec_debug_trace): in use / allocated
    // chunked pipeline
    int Fc = 128;                         // frames per scoring chunk of the streaming API (jd_stream_push)
    int Fw_env = 0;                       // JD_FC: frames per chunk of the batch path (0 = as long as the longest utterance)
    float *d_ll_slab = nullptr;           // the three likelihood tables (ensure_slab)
    float *d_ll[3] = {nullptr, nullptr, nullptr};
    size_t ll_cap = 0;                    // floats per table
    int *d_row_src[3] = {nullptr, nullptr, nullptr}; size_t row_src_cap[3] = {0, 0, 0};   // row -> source frame tables, one per likelihood table
    int *d_T = nullptr;
    hipStream_t s_gmm = nullptr, s_search = nullptr;
    unsigned *h_park = nullptr; int *d_park = nullptr;   // jd_park_kernel's words (host-mapped state; arrival counts and quotas by XCD)
    // scoring one batch ahead (jd_dec_prefetch_scores): the table of the NEXT batch is scored while this one is searched
    std::deque<Prefetch> pf_q;            // the batches ahead, in the order announced: scored, started ("two batches in flight"), or neither yet
    int fg_buf = -1;                      // the table(s) the wave being decoded uses (-1: none; -2: tables 0 and 1, in chunks)
    int fg_bank = -1;                     // >= 0: the wave being decoded runs on this bank of streams and the batch behind it may be
                                          // started beside it
    int pipeline = 1;                     // two batches in flight (JD_PIPELINE=0: off)
    bool bg_ran = false;                  // the last launch_search advanced streams of the batch behind, too
    double bg_wait_us = 1000.0;           // how long the start of a launch waits for the table of the batch behind (JD_BG_WAIT_US)
    double last_wave_ms = 0.0;            // wall time of the last wave decoded (what a scoring beside the next one has to fit)
    int score_reserve = -1;               // CUs left to the scoring beside a launch with two batches in flight: -1 by its cost (JD_SCORE_RESERVE)
    int reserve_now = 0;                  // ... of the launch under way
    int bg_rebalance = 0;                 // re-plan a launch that runs two batches (JD_BG_REBALANCE)
    double bg_max_load = 4.0;             // two batches in flight up to this load_scale (JD_BG_MAX_LOAD)
    double bg_weight = 0.5;               // the plan counts this part of the frames a stream of the batch behind has ahead (JD_BG_WEIGHT)
    int fg_cw_cap = 8, bg_cw_cap = 4;     // two batches in flight: largest cluster of the running batch / of the batch behind (JD_FG_CW, JD_BG_CW)
    bool pf_armed = false;                // decode_wave: the coming launch_search may start the announced scorings
    bool pf_retry = false;                // decode_wave: this wave is being decoded again - what was scored beside its first attempt stays
    int pf_rebalance = -1;                // re-planning a launch beside which a table is scored: -1 by the measured ratio (launch_search),
                                          // JD_PF_REBALANCE=0: never while the scoring runs, =1: like any other launch
    double gmm_ms_per_row = 0.0, search_ms_per_frame = 0.0;   // measured on this decoder's last waves (scoring on its own / search)
    int *h_resident = nullptr, *d_resident = nullptr; int launch_seq = 0;   // host-mapped word: k_search's last workgroup has started
    // streaming API state
    std::vector<int> stream_T;                 // frames pushed so far
    std::vector<int> stream_started;
    // PARTIAL_DECODING (WFSTDecoderLite.h:199-205), streaming API
    int partial_interval = 0;                  // partialTraceInterval
    std::vector<int> last_collect, last_trace; // lastPathCollectFrame, lastPartialTraceFrame
    std::vector<int> n_collect_host;           // collections of the stream's utterance so far (jd_stream_collect_info)
    std::vector<std::vector<int32_t>> partial_label, partial_time;   // partialPaths, oldest first
    int *d_partial_out = nullptr;
    bool return_on_collect = false, collected_now = false;   // jd_stream_push: launch_search comes back after a collection by the count rule
    float *d_push = nullptr; size_t push_cap = 0;
    char *h_stage = nullptr; size_t stage_cap = 0;     // pinned staging of jd_streams_push
    bool xch_forced = false;               // JD_XCH given (development)
    struct Resident *res = nullptr;        // the resident search kernel of a broker (jd_res_*), or null
    std::recursive_mutex res_mu;           // jd_res_stop / jd_res_start (the broker's worker) against jd_res_finish (its finisher thread)
    long long res_yield_turn = -1;             // the lock's turn count when the resident kernel last made room for a waiter
    std::vector<hipStream_t> res_old_streams;   // search streams the resident kernel gave up (jd_res_start)
    struct Pipe *pipe = nullptr;           // batches through the resident kernel, utterance by utterance (jd_pipe_*; JD_PIPELINE=3)
    bool pipe_mode = false;
    bool pipe_on = false;                  // Pipe::on (for the code in front of the struct)
    int pipe_depth = 8;                    // likelihood tables = batches announced and not handed back, at most (jd_dec_set_pipeline)
    int pipe_slots = 0;                    // one-workgroup slots of the resident kernel (0: max_streams)
    // jd_dec_pipeline_stats: cumulative over the decoder's life
    long long pipe_frames_searched = 0;    // stream-frames the slots have advanced, as reported command by command
    long long pipe_utts_through = 0, pipe_rows_scored = 0, pipe_batches_back = 0, pipe_collections = 0;
    long long pipe_busy_ticks = 0;         // the slots' own clocks on their commands (100 MHz)
    double pipe_on_us = 0.0;               // wall time the resident kernel has been on the device for the pipeline
    const float *res_ll = nullptr;         // the likelihood slab the resident kernel reads (null: the broker's stream buffers)
    // results
    std::vector<HostResult> results;
    jd_timing timing{};
    // lazily composed networks (jd_lazy_enter / jd_lazy_leave): a stream of the last fetch ran out of graph room;
    // streams of the streaming API that are inside an utterance
    bool lazy_failed = false;
    std::vector<char> lazy_in;
    // a stream whose last launch ended in an error, or never reported back (a wave that returned early, a push that
    // failed and was not followed by a finish): its per-state and per-arc words may be anything - wiped before its next init
    std::vector<char> stream_dirty;
    bool occupancy_ok = false;            // k_search fits a CU the way launch_search's grid assumes (checked at the first launch)
};

template <typename T>
static int dmalloc(jd_dec *d, T **p, size_t n)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return jd_fail(e == hipErrorOutOfMemory ? JD_ENOMEM : JD_EHIP, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    }
    d->allocs.push_back(q);
    *p = (T *)q;
    return JD_OK;
}

template <typename T>
static int dupload(jd_dec *d, T **p, const T *src, size_t n)
{
    int rc = dmalloc(d, p, n);
    if (rc) return rc;
    HIPCHK(hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return JD_OK;
}

static void res_free_fwd(jd_dec *d);
static void pipe_free_fwd(jd_dec *d);
static void pipe_drain(jd_dec *d);
static int pipe_announce(jd_dec *d, int n_utts, const float *d_feats, const int64_t *offs, int *taken);
static int pipe_decode(jd_dec *d, int n_utts, const float *d_feats, const int64_t *offs, jd_hyp *out, int *handled);
extern "C" void jd_dec_destroy(jd_dec *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    pipe_drain(d);
    pipe_free_fwd(d);
    if (d->res) { (void)jd_res_stop(d); res_free_fwd(d); }
    (void)hipDeviceSynchronize();
    for (char in : d->lazy_in) if (in) jd_lazy_leave(d->net, 1);      // (utterances the caller never finished)
    for (void *p : d->allocs) (void)hipFree(p);
    slab_give(d->device, d->slab);
    d->slab = ArenaSlab();
    free_am_gmm(d->amb);
    if (d->d_ll_slab) (void)hipFree(d->d_ll_slab);
    if (d->d_cells) (void)hipFree(d->d_cells);
    for (int i = 0; i < 3; ++i)
        if (d->d_row_src[i]) (void)hipFree(d->d_row_src[i]);
    for (Prefetch &F : d->pf_q) F.drop();
    d->pf_q.clear();
    if (d->h_resident) (void)hipHostFree(d->h_resident);
    if (d->d_push) (void)hipFree(d->d_push);
    if (d->h_stage) (void)hipHostFree(d->h_stage);
    if (d->d_work) (void)hipFree(d->d_work);
    if (d->h_status) (void)hipHostFree(d->h_status);
    if (d->s_gmm) (void)hipStreamDestroy(d->s_gmm);
    if (d->s_search) (void)hipStreamDestroy(d->s_search);
    if (d->h_park) (void)hipHostFree(d->h_park);
    if (d->d_park) (void)hipFree(d->d_park);
    for (hipStream_t st : d->res_old_streams) (void)hipStreamDestroy(st);
    delete d;
}

extern "C" int jd_dec_create(jd_dec **out, const jd_net *net, const jd_am *am, float start_beam, float main_beam,
                             float end_beam, float word_beam, int32_t max_hyps, int32_t block_size,
                             int32_t device, int32_t max_streams)
{
    if (!out || !net || !am) return jd_fail(JD_EINVAL, "jd_dec_create: null argument");
    if (block_size < 1 || block_size > 20)      // HTKFlatModels::setBlockSize, HTKFlatModels.cpp:311-312
        return jd_fail(JD_EINVAL, "HTKFlatModels::setBlockSize fnBlock should be in [1, 20]");
    if (max_streams < 1) return jd_fail(JD_EINVAL, "jd_dec_create: max_streams < 1");
    if (net->max_in > am->n_hmm)
        return jd_fail(JD_EINVAL, "network input label %d exceeds the number of HMMs %d", net->max_in, am->n_hmm);
    if (am->max_n > JD_MAXN) return jd_fail(JD_EINVAL, "HMMs with more than %d states unsupported", JD_MAXN);
    if (am->n_hmm >= SOLE_FLAG || am->n_tm > 0x1fffff)                    // (the flag bits of a device arc's in-label and of a record's header)
        return jd_fail(JD_EINVAL, "more than %d HMMs or %d transition matrices unsupported", SOLE_FLAG - 1, 0x1fffff);
    if ((int64_t)net->n_states * (int64_t)sizeof(StateRec) > 0xf0000000LL)   // (32-bit byte offsets of the buffer descriptor)
        return jd_fail(JD_EINVAL, "networks with more than %lld states unsupported", (long long)(0xf0000000LL / (int64_t)sizeof(StateRec)));
    int rc = check_device(device);
    if (rc) return rc;
    jd_dec *d = new jd_dec();
    d->net = net; d->am = am; d->device = device; d->max_streams = max_streams; d->block_size = block_size;
    // development: frames per chunk of a batch
    if (const char *e = jd_dev_env("JD_FC")) { const int v = atoi(e); if (v >= 16 && v <= 65536) d->Fw_env = v; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) d->n_cus = prop.multiProcessorCount;
    }
    DecConst &C = d->C;
    C.start_win = start_beam; C.emit_win = main_beam; C.end_win = end_beam; C.word_win = word_beam;
    C.max_hyps = max_hyps;
    C.x_chunks = 2;
    if (const char *e = jd_dev_env("JD_XCH")) { const int v = atoi(e); if (v >= 1 && v <= 16) { C.x_chunks = v; d->xch_forced = true; } }   // development
    C.exp = 0; C.path_rule = 0; C.pcount = nullptr;
    if (const char *e = jd_dev_env("JD_EXP")) C.exp = atoi(e);                                                    // development
    C.hist_min = 0; C.hist_max = 0; C.hist_nbins = 0;
    if (max_hyps > 0) {                          // WFSTDecoderLite.cpp:76-82, Histogram.cpp:29-37
        float mn = (main_beam > 0.0) ? (float)(-main_beam - 800.0) : -1000.0f;
        C.hist_min = (int)(mn - 1.0);
        C.hist_max = (int)(200.0f + 1.0);
        C.hist_nbins = C.hist_max - C.hist_min + 1;
        if (C.hist_nbins > HIST_MAX_BINS) {
            delete d;
            return jd_fail(JD_EINVAL, "mainBeam %.1f needs %d histogram bins (> %d supported)", main_beam,
                           C.hist_nbins, HIST_MAX_BINS);
        }
    }
#define TRY(x) do { rc = (x); if (rc) { jd_dec_destroy(d); return rc; } } while (0)
    const bool lazy = net->lazy_dev != nullptr;
    if (lazy && net->lazy_device != device) {
        jd_dec_destroy(d);
        return jd_fail(JD_EINVAL, "jd_dec_create: the lazily composed network lives on device %d, not %d", net->lazy_device, device);
    }
    // The decoder's OWN numbering of the states.  A stream's per-state words (jd_search.h: StateRec) are gathered by state number, eight
    // arrival keys to a 64-byte line, so which states are neighbours in NUMBER decides how many lines a frame fetches - and tokens
    // move along chains.  The numbering: a state, then, arc by arc, the chain of one-arc states behind each of its arcs (the phones of
    // a word, one state after the other; the chains that leave one state side by side, as their instances are attached in the same
    // frame); the states the chains END in - the ones with a choice to make - get their number there and take their turn first come,
    // first served.  That is how a lexicon written chain after chain is laid out already, and such a network keeps its numbering; one
    // numbered by its composition (jd_net_compose: canonical, breadth first) gets this one.  State numbers never leave the device and
    // nothing breaks a tie by them (the frontier item's number does): results are bit-identical.  Measured on the composed configs[4]
    // graph (k frames/s, tools/r6_run21-24.sh): the network's numbers 36.7, along the chains of first model arcs 36.7 (round 6's first
    // attempt), depth first 41.3, blocks of 16 filled breadth first 43.5, this 45.0; on the bench's generated graphs it equals the
    // generator's own order (129.8 / 129.1 k, 5.60 / 5.65 k), every other order loses 1-4 % to it.
    // (JD_RENUMBER, development: 1 / 0 - always / never.)
    std::vector<int> rp_own;                                           // row_ptr in the decoder's numbering (empty: the network's)
    std::vector<JdArc> arcs_own;
    int64_t n_next_net = 0;                                            // arcs of the network that lead to the next state number
    if (!lazy)
        for (int q = 0; q < net->n_states; ++q)
            for (int b = net->row_ptr[(size_t)q]; b < net->row_ptr[(size_t)q + 1]; ++b) n_next_net += net->arcs[(size_t)b].to == q + 1;
    bool renumber = !lazy && net->n_states > 0 && 4 * n_next_net < (int64_t)net->n_arcs;
    if (const char *e = jd_dev_env("JD_RENUMBER")) renumber = !lazy && net->n_states > 0 && atoi(e) != 0;
    if (renumber) {
        const int ns = net->n_states;
        std::vector<int> new_of((size_t)ns, -1), old_of((size_t)ns);
        int next = 0;
        auto take = [&](int q) { new_of[(size_t)q] = next; old_of[(size_t)next] = q; ++next; };
        std::vector<int> pend;                                         // numbered states whose arcs are still to be followed, in the order they were met
        for (int pass = 0; pass < 2; ++pass)                           // (from the initial state; then whatever it does not reach, in the network's order)
            for (int s0 = pass == 0 ? net->init : 0; s0 < (pass == 0 ? net->init + 1 : ns); ++s0) {
                if (new_of[(size_t)s0] >= 0) continue;
                take(s0);
                pend.clear(); pend.push_back(s0);
                for (size_t ph = 0; ph < pend.size(); ++ph) {
                    const int q = pend[ph];
                    for (int b = net->row_ptr[(size_t)q]; b < net->row_ptr[(size_t)q + 1]; ++b) {
                        int t = net->arcs[(size_t)b].to;
                        while (new_of[(size_t)t] < 0 && net->row_ptr[(size_t)t + 1] - net->row_ptr[(size_t)t] == 1) {
                            take(t);
                            t = net->arcs[(size_t)net->row_ptr[(size_t)t]].to;
                        }
                        if (new_of[(size_t)t] < 0) { take(t); pend.push_back(t); }
                    }
                }
            }
        bool same = true;
        for (int q = 0; q < ns && same; ++q) same = new_of[(size_t)q] == q;
        if (!same) {
            rp_own.assign((size_t)ns + 1, 0);
            for (int n = 0; n < ns; ++n) rp_own[(size_t)n + 1] = rp_own[(size_t)n] + (net->row_ptr[(size_t)old_of[(size_t)n] + 1] - net->row_ptr[(size_t)old_of[(size_t)n]]);
            arcs_own.resize(net->arcs.size());
            for (int n = 0; n < ns; ++n) {
                const int q = old_of[(size_t)n], r0 = net->row_ptr[(size_t)q], r1 = net->row_ptr[(size_t)q + 1];
                for (int b = r0; b < r1; ++b) {
                    JdArc a = net->arcs[(size_t)b];
                    a.to = new_of[(size_t)a.to];
                    arcs_own[(size_t)rp_own[(size_t)n] + (size_t)(b - r0)] = a;
                }
            }
            d->state_new.swap(new_of);
        }
        if (getenv("JD_VERBOSE")) fprintf(stderr, "state numbers: %s\n", same ? "the network's (already along its chains)" : "the decoder's own (a state, the chains behind its arcs side by side)");
    }
    const std::vector<int> &row_ptr_h = rp_own.empty() ? net->row_ptr : rp_own;
    const std::vector<JdArc> &arcs_h = arcs_own.empty() ? net->arcs : arcs_own;
    if (!lazy) TRY(dupload(d, &d->d_row_ptr, row_ptr_h.data(), row_ptr_h.size()));
    std::vector<float> tmax0((size_t)am->n_hmm, LZ);   // largest log transition probability out of the entry state of every HMM
    for (int h = 0; h < am->n_hmm; ++h) {
        const float *t0 = am->trP.data() + (size_t)am->hmm_tm[(size_t)h] * am->max_n * am->max_n;
        for (int j = 0; j < am->hmm_n[(size_t)h]; ++j) tmax0[(size_t)h] = std::max(tmax0[(size_t)h], t0[j]);
    }
    if (!lazy) {   // device arc table: bit 30 of the in-label marks arcs whose HMM is a tee model
        std::vector<JdArc> darcs(arcs_h);
        for (JdArc &a : darcs)
            if (a.in > 0 && am->hmm_tee[(size_t)a.in - 1] > LZ) a.in |= TEE_FLAG;
        // The decoder's OWN order of a state's arcs (XState, jd_search.h): what every arrival walks first, then the arcs that
        // enter a model by descending w + tmax - phase X of the slot kernel walks a prefix of those.  Arc numbers never leave
        // the device (results carry labels, times and scores), and every arc has an instance of its own, so the order changes no
        // score; it can change which of two EQUAL-scored tokens a state keeps (the frontier item's number breaks the tie), which
        // the reference's own traversal order decides no better (tests: decode_certified).  JD_NO_XSORT (development): the file's order.
        const bool xsort = jd_dev_env("JD_NO_XSORT") == nullptr;
        std::vector<XState> xst((size_t)net->n_states);
        std::vector<std::pair<float, JdArc>> ent;
        for (int q = 0; q < net->n_states; ++q) {
            const int r0 = row_ptr_h[(size_t)q], r1 = row_ptr_h[(size_t)q + 1];
            XState &X = xst[(size_t)q];
            X.n_always = 0; X.n_entry = 0; X.wmax = LZ; X.n_model = 0;
            for (int i = 0; i < XNCAND; ++i) X.k[i] = LZ;
            ent.clear();
            // (only rows the cut can apply to - up to 57 arcs, jd_slot.h: the flags of a longer row do not fit an item's loads -
            // change their order: the long rows of trigram-shaped graphs keep the file's, which the searches of such graphs
            // are 2-3 % faster on; measured on the north-star graph)
            const bool sort_row = xsort && r1 - r0 <= 57;
            int at = r0;
            for (int b = r0; b < r1; ++b) {
                const JdArc a = darcs[(size_t)b];
                const int inl = a.in & ~TEE_FLAG;
                if (inl != 0) { ++X.n_model; X.wmax = std::max(X.wmax, a.w); }
                if (sort_row && inl != 0 && !(a.in & TEE_FLAG)) ent.push_back({a.w + tmax0[(size_t)inl - 1], a});
                else darcs[(size_t)at++] = a;                          // (in place: `at` never passes b)
            }
            X.n_always = at - r0;
            std::stable_sort(ent.begin(), ent.end(), [](const std::pair<float, JdArc> &x, const std::pair<float, JdArc> &y) { return x.first > y.first; });
            X.n_entry = (int)ent.size();
            for (size_t i = 0; i < ent.size(); ++i) darcs[(size_t)at + i] = ent[i].second;
            for (int i = 0; i < XNCAND; ++i) if (xcand(i) < X.n_entry) X.k[i] = ent[(size_t)xcand(i)].first;
        }
        {   // SOLE_FLAG (jd_search.h: REC_SOLE): the arc that enters a model and is the only arc of the network that leads to its
            // destination - its exit tokens recombine with nobody.  (JD_NO_SOLE, development: off.)
            std::vector<int> indeg((size_t)net->n_states, 0);
            for (const JdArc &a : darcs) ++indeg[(size_t)a.to];
            int64_t n_sole = 0, n_model = 0;
            const bool sole_on = jd_dev_env("JD_NO_SOLE") == nullptr;
            for (JdArc &a : darcs)
                if ((a.in & ~TEE_FLAG) != 0) { ++n_model; if (sole_on && indeg[(size_t)a.to] == 1 && !(a.in & TEE_FLAG)) { a.in |= SOLE_FLAG; ++n_sole; } }   // (a tee model's pass-through arrives beside its exit token)
            if (getenv("JD_VERBOSE")) fprintf(stderr, "exit tokens that recombine with nobody: those of %lld of %lld model arcs (the only arc into their state)\n", (long long)n_sole, (long long)n_model);
        }
        TRY(dupload(d, &d->d_arcs, darcs.data(), darcs.size()));
        TRY(dupload(d, &d->d_xst, xst.data(), xst.size()));
        // (k_search takes the cut where it pays - graphs whose rows are short throughout, like configs[1]'s: 99.7 % of its model arcs
        // sit in sorted rows, two batches in flight gain 10 % - and not where a few long rows carry the traffic: trigram-shaped
        // graphs have 85-87 % of their arcs in short rows, yet configs[3] loses 4 % to the item stage's extra loads and the
        // north-star graph gains nothing)
        int64_t n_sorted = 0, n_model_all = 0;
        for (const XState &X : xst) { n_sorted += X.n_entry; n_model_all += X.n_model; }
        C.xcut = (xsort && n_model_all > 0 && 20 * n_sorted >= 19 * n_model_all) ? 1 : 0;
        if (const char *e = jd_dev_env("JD_XCUT")) C.xcut = (atoi(e) != 0 && xsort) ? 1 : 0;
        if (getenv("JD_VERBOSE")) fprintf(stderr, "arc order: %lld of %lld model arcs in sorted rows (<= 57 arcs), %d states; k_search cuts walks: %d\n",
                                          (long long)n_sorted, (long long)n_model_all, net->n_states, C.xcut);
    }
    {   // The layout of a stream's per-state words (jd_search.h: StateRec): split - the arrival keys of all states in an array of their own,
        // four states to a 64-byte line - where the graph's numbering puts the states of a chain side by side (an arc to the NEXT state
        // number: the lexicon chains of a composed C.L.G written state after state - 42 % of the arcs of the bench graphs), joint where it
        // does not (a graph numbered by its composition: what neighbours in number have in common is nothing, and every exit token would
        // pay a second line).  (JD_SREC_SPLIT, development: 1 / 0.)
        int64_t n_next = 0;
        if (!lazy)
            for (int q = 0; q < net->n_states; ++q)
                for (int b = row_ptr_h[(size_t)q]; b < row_ptr_h[(size_t)q + 1]; ++b) n_next += arcs_h[(size_t)b].to == q + 1;
        int split = (!lazy && net->n_arcs > 0 && 4 * n_next >= (int64_t)net->n_arcs) ? 2 : 0;   // 0 joint, 1 split (both parities of a state together), 2 split by parity
        if (const char *e = jd_dev_env("JD_SREC_SPLIT")) split = std::max(0, std::min(2, atoi(e)));
        const unsigned ns = (unsigned)net->n_states;
        C.srec_stride = split ? 16u : 32u;
        C.srec_arr = split ? 16u * ns : 16u;
        C.srec_estride = split == 2 ? 8u : (split ? 16u : 32u);
        C.srec_par = split == 2 ? 8u * ns : 8u;
        if (getenv("JD_VERBOSE")) fprintf(stderr, "per-state words: %s (%lld of %lld arcs lead to the next state number)\n",
                                          split == 2 ? "split (bids | arrival keys of either frame parity)" : split ? "split (bids | arrival keys)" : "joint records",
                                          (long long)n_next, (long long)net->n_arcs);
    }
    if (!lazy) {
        if (d->state_new.empty()) TRY(dupload(d, &d->d_fin_w, net->fin_w.data(), net->fin_w.size()));
        else {
            std::vector<float> fw(net->fin_w.size());
            for (size_t q = 0; q < fw.size(); ++q) fw[(size_t)d->state_new[q]] = net->fin_w[q];
            TRY(dupload(d, &d->d_fin_w, fw.data(), fw.size()));
        }
    }
    TRY(dupload(d, &d->d_hmm_tee, am->hmm_tee.data(), am->hmm_tee.size()));
    TRY(dupload(d, &d->d_hmm_tmax0, tmax0.data(), tmax0.size()));     // (phase X, hopeless candidates)
    TRY(dupload(d, &d->d_trP, am->trP.data(), am->trP.size()));
    std::vector<int> se32((size_t)am->n_tm * am->max_n);
    for (size_t i = 0; i < se32.size(); ++i)
        se32[i] = ((int)am->se[i * 2] & 0xffff) | ((int)am->se[i * 2 + 1] << 16);
    TRY(dupload(d, &d->d_se32, se32.data(), se32.size()));
    TRY(upload_am_gmm(am, d->amb));
    C.row_ptr = d->d_row_ptr; C.arcs = d->d_arcs; C.fin_w = d->d_fin_w; C.init_state = d->state_new.empty() ? net->init : d->state_new[(size_t)net->init]; C.n_states = net->n_states;
    C.xst = d->d_xst;
    C.G = am->n_gmm; C.max_n = am->max_n; C.n_tm = am->n_tm;
    C.hmm_tee = d->d_hmm_tee; C.n_hmm = am->n_hmm; C.hmm_tmax0 = d->d_hmm_tmax0;
    C.lazy = (const LazyDev *)net->lazy_dev; C.aux_h = nullptr;
    {   // instance template: what phase A needs to attach an instance (attachNetInst :751-774), by HMM -
        // {nStates | transMat << 8, g0, g1, g2} (+ {g3, g4, g5, 0}): one hop behind the arc record's label, but the
        // table is a few tens of KB (L2 hits) where a per-arc copy was a second random 64-byte sector per new
        // instance and 16-32 B per arc of HBM (measured: same speed at configs[1], +0.5 % in the heavy legs)
        const int AI = (am->max_n <= 5) ? 4 : 8;
        std::vector<int> aux((size_t)am->n_hmm * AI, 0);
        for (int hm = 0; hm < am->n_hmm; ++hm) {
            const int n = am->hmm_n[(size_t)hm];
            int *a = aux.data() + (size_t)hm * AI;
            a[0] = n | (am->hmm_tm[(size_t)hm] << 8);
            for (int j = 1; j < n - 1 && j <= (AI == 4 ? 3 : 6); ++j)
                a[j] = am->hmm_gmm[(size_t)hm * am->max_n + j];
        }
        TRY(dupload(d, &d->d_aux, aux.data(), aux.size()));
        C.aux_h = d->d_aux;
    }
    C.trP = d->d_trP; C.se32 = d->d_se32;
    {   // Plain left-to-right topologies (every emitting state entered from its predecessor and itself,
        // the exit state from the last emitting state - createTrPandSEIndex, HTKModels.cpp:2330-2390,
        // gives SEIndex[j] = {j-1, j+1}): phase A then needs a_k = log P(k-1 -> k), s_k = log P(k -> k) only
        const int MNn = am->max_n, NEn = (MNn <= 5) ? 3 : 6, LRW = (NEn == 3) ? 8 : 16;
        bool all_lr = (size_t)am->n_tm * LRW <= TRP_LDS_MAX;
        for (int t = 0; t < am->n_tm && all_lr; ++t) {
            const int n = am->tm_n[(size_t)t];
            if (n < 3) all_lr = false;
            for (int j = 1; j < n && all_lr; ++j) {
                const int st = am->se[((size_t)t * MNn + j) * 2], en = am->se[((size_t)t * MNn + j) * 2 + 1];
                if (j < n - 1 ? (st != j - 1 || en != j + 1) : (st != n - 2 || en != n - 1)) all_lr = false;
            }
        }
        for (int h = 0; h < am->n_hmm && all_lr; ++h)
            if (am->hmm_n[(size_t)h] != am->tm_n[(size_t)am->hmm_tm[(size_t)h]]) all_lr = false;
        if (jd_dev_env("JD_NO_LR")) all_lr = false;                                  // development: force the general path
        C.lrt = nullptr;
        if (all_lr) {
            std::vector<float> lrt((size_t)am->n_tm * LRW, LZ);
            for (int t = 0; t < am->n_tm; ++t) {
                const float *tp = am->trP.data() + (size_t)t * MNn * MNn;
                const int n = am->tm_n[(size_t)t];
                for (int k = 1; k <= n - 1; ++k) lrt[(size_t)t * LRW + k - 1] = tp[(k - 1) * MNn + k];          // a_k
                for (int k = 1; k <= n - 2; ++k) lrt[(size_t)t * LRW + NEn + k] = tp[k * MNn + k];              // s_k
            }
            TRY(dupload(d, &d->d_lrt, lrt.data(), lrt.size()));
            C.lrt = d->d_lrt;
        }
    }
    if (const char *e = jd_dev_env("JD_CW")) { const int v = atoi(e); if (v >= 1 && v <= MAXCW) d->max_cw = v; }   // development
    if (const char *e = jd_dev_env("JD_WEIGHTED")) d->weighted = atoi(e) != 0;
    if (const char *e = jd_dev_env("JD_REBALANCE")) d->rebalance = atoi(e) != 0;
    if (const char *e = jd_dev_env("JD_REBALANCE_FRAC")) { const double v = atof(e); if (v > 0.0 && v < 1.0) d->rebalance_frac = v; }
    if (const char *e = jd_dev_env("JD_REBALANCE_MIN_US")) d->rebalance_min_us = atof(e);
    if (const char *e = jd_dev_env("JD_MODEL_A")) d->model_a_us = atof(e);
    if (const char *e = jd_dev_env("JD_MODEL_B")) d->model_b_us = atof(e);
    if (const char *e = jd_dev_env("JD_PLAN")) d->plan_mode = atoi(e) != 0;
    if (const char *e = jd_dev_env("JD_PLAN_MIN_CW")) d->plan_min_cw = std::max(1, atoi(e));
    if (const char *e = jd_dev_env("JD_PF_GMM_WEIGHT")) d->pf_gmm_weight = atof(e);
    if (const char *e = jd_dev_env("JD_MODEL2_A")) d->model2_a_us = atof(e);
    if (const char *e = jd_dev_env("JD_MODEL2_B")) d->model2_b_us = atof(e);
    if (const char *e = jd_dev_env("JD_XCD_LOCAL")) d->xl_ok = atoi(e) != 0;                      // development
    if (const char *e = jd_dev_env("JD_XL_SLACK")) { const double v = atof(e); if (v >= 1.0 && v <= 10.0) d->xl_slack = v; }
    // arena capacities: 0 = sized from the free HBM when the arenas are allocated (ensure_arenas)
    d->cap_slots = d->cap_items = d->cap_paths = d->cap_new = 0;
    if (const char *e = jd_dev_env("JD_PF_REBALANCE")) d->pf_rebalance = atoi(e) != 0;                // development
    if (const char *e = jd_dev_env("JD_PIPELINE")) { d->pipeline = atoi(e) != 0; d->pipe_mode = atoi(e) == 3; }
    if (const char *e = jd_dev_env("JD_PIPE_DEPTH")) { const int v = atoi(e); if (v >= 2 && v <= 32) d->pipe_depth = v; }
    if (const char *e = jd_dev_env("JD_BG_WAIT_US")) d->bg_wait_us = atof(e);
    if (const char *e = jd_dev_env("JD_SCORE_RESERVE")) d->score_reserve = atoi(e);
    if (const char *e = jd_dev_env("JD_BG_REBALANCE")) d->bg_rebalance = atoi(e) != 0;
    if (const char *e = jd_dev_env("JD_BG_WEIGHT")) d->bg_weight = atof(e);
    if (const char *e = jd_dev_env("JD_BG_MAX_LOAD")) d->bg_max_load = atof(e);
    if (const char *e = jd_dev_env("JD_FG_CW")) d->fg_cw_cap = std::max(1, atoi(e));
    if (const char *e = jd_dev_env("JD_BG_CW")) d->bg_cw_cap = std::max(1, atoi(e));
    hipError_t e;
    // the search stream has the highest priority, the scoring stream the lowest: when a table is scored while a search
    // runs (jd_dec_prefetch_scores) a search launch that needs CUs gets them before further scoring blocks do
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if ((e = hipStreamCreateWithPriority(&d->s_gmm, hipStreamNonBlocking, prio_lo)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&d->s_search, hipStreamNonBlocking, prio_hi)) != hipSuccess) {
        jd_dec_destroy(d);
        return jd_fail(JD_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    if ((e = hipHostMalloc((void **)&d->h_resident, 64, hipHostMallocMapped)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void **)&d->d_resident, d->h_resident, 0)) != hipSuccess) {
        jd_dec_destroy(d);
        return jd_fail(JD_EHIP, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    *d->h_resident = 0;
    d->stream_T.assign((size_t)max_streams, 0);
    d->stream_started.assign((size_t)max_streams, 0);
    d->lazy_in.assign((size_t)max_streams, 0);
    d->stream_dirty.assign((size_t)max_streams, 0);
    d->last_collect.assign((size_t)max_streams, -1);
    d->n_collect_host.assign((size_t)max_streams, 0);
    d->last_trace.assign((size_t)max_streams, -1);
    d->partial_label.resize((size_t)max_streams);
    d->partial_time.resize((size_t)max_streams);
    d->results.resize((size_t)max_streams);
#undef TRY
    *out = d;
    return JD_OK;
}

extern "C" int jd_dec_set_capacity(jd_dec *d, int64_t max_slots, int64_t max_paths, int64_t max_items)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_set_capacity: null");
    if (d->arenas_ready) return jd_fail(JD_ESTATE, "jd_dec_set_capacity: arenas already allocated");
    if (max_slots > 0) d->cap_slots = max_slots;
    if (max_paths > 0) d->cap_paths = max_paths;
    if (max_items > 0) d->cap_items = max_items;
    return JD_OK;
}

// setMaxAllocModels (WFSTDecoderLite.cpp:807-820): same argument convention - below 100 a percentage
// of the network's transitions, 100..7999 a memory limit in MB, from 8000 a number of instances.  In the
// reference it is a SOFT limit: it is compared with the NetInst objects the pools have handed out, and when there
// are more they are all dropped between two utterances (:164-169) - a cap on what stays cached, never a
// reason for a decode to fail.  Here instance memory is one arena of records per stream that every utterance
// reuses wholesale, so nothing is cached from one utterance to the next and there is nothing to drop: the value
// is validated and kept as a hint that can only RAISE the arena above its automatic size (a caller who asks
// for room for more instances gets it), never lower it.  A hard capacity is jd_dec_set_capacity's business.
extern "C" int jd_dec_set_max_alloc_models(jd_dec *d, int32_t max_alloc_models)
{
    if (!d || max_alloc_models <= 0) return jd_fail(JD_EINVAL, "setMaxAllocModels: maxAllocModels > 0");
    if (d->arenas_ready) return jd_fail(JD_ESTATE, "jd_dec_set_max_alloc_models: arenas already allocated");
    int64_t n;
    // (a lazily composed network's n_arcs is the capacity it may grow into: the most it can ever hold)
    if (max_alloc_models < 100) n = d->net->n_arcs * max_alloc_models / 100;                        // :809-811
    // :812-814, sizeof(NetInst) + sizeof(Token) * nStatePools
    else if (max_alloc_models < 8000) n = (int64_t)max_alloc_models * 1024 * 1024 / (40 + 24 * (int64_t)d->am->max_n);
    else n = max_alloc_models;                                                                       // :815-817
    d->slots_hint = std::min<int64_t>(n, d->net->n_arcs + 65536);      // (one instance per arc at most)
    return JD_OK;
}

// reset a stream's per-state records: no bids, no arrivals; the CSR row of the state (row_ptr == nullptr: a lazy
// graph, rows live elsewhere)
__global__ void jd_reset_srec_kernel(StateRec *srec, const int *row_ptr, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        (void)row_ptr;
        srec[i] = StateRec{0ULL, 0ULL, {0ULL, 0ULL}};
    }
}
static void reset_srec(StateRec *srec, const int *d_row_ptr, int64_t n_states)
{
    hipLaunchKernelGGL(jd_reset_srec_kernel, dim3((unsigned)((n_states + 255) / 256)), dim3(256), 0, 0, srec, d_row_ptr, (long long)n_states);
}

static int ensure_slab(jd_dec *d, size_t floats);
static int ensure_arenas_try(jd_dec *d, double mem_fraction)
{
    int rc = check_device(d->device);
    if (rc) return rc;
    const int B = d->max_streams, MN = d->am->max_n;
    const int64_t rec_bytes = (MN <= 5) ? RecLayout<3>::REC_BYTES : RecLayout<6>::REC_BYTES;
    // a stream's arenas: 256-byte aligned pieces of one block (two uses of the same list: sizes with blk = null, then pointers)
#define A(p, n) do { const size_t bytes_ = (std::max<size_t>((size_t)(n), 1) * sizeof(*(p)) + 255) & ~(size_t)255; \
                     if (blk) (p) = (typename std::remove_reference<decltype(p)>::type)(blk + off); off += bytes_; } while (0)
#define ARENAS() do { \
        A(S.rec, 2 * d->cap_slots * (rec_bytes / 4)); \
        A(S.live, d->net->n_arcs); \
        A(S.srec, d->net->n_states); \
        A(S.items, 4 * d->cap_items); \
        A(S.newl, d->cap_new); A(S.dirtyl, 2 * d->cap_new); \
        A(S.tot, TOT_N * MAXW); A(S.item_end, MAXW); \
        A(S.paths, d->cap_paths); A(S.paths2, d->cap_paths); A(S.gc_idx, d->cap_paths); A(S.gc_state, sizeof(GcState) / 4); \
        A(S.hist, 2 * HIST_MAX_BINS); } while (0)
    const bool auto_slots = d->cap_slots <= 0, auto_items = d->cap_items <= 0, auto_paths = d->cap_paths <= 0;
    // The slab a previous decoder left on this device (slab_take) is there without waiting for the driver to clear
    // memory - 5-6 s for the default share of a 288 GB device - but only if this decoder's arenas FIT it, and two
    // decoders never ask for quite the same: the free memory has moved by a table or a network since.  So when the
    // sizes come out above the cached slab by a few per cent (up to 6: a smaller Path arena means more collections - the
    // 14 M-arc bench graph lost 4 % of its throughput to arenas fitted to a slab a fifth short), they are scaled down
    // to it (second pass).
    double fit = 1.0;
    for (int pass = 0; pass < 3; ++pass) {
    if (pass) { if (auto_slots) d->cap_slots = 0; if (auto_items) d->cap_items = 0; if (auto_paths) d->cap_paths = 0; }
    {   // Capacities the caller did not set: sized for 288 GB of HBM, not for frugality.  70% of the
        // free memory is split over the streams; of a stream's share (after its per-arc / per-state
        // tables) 50% goes to instance records, 20% to frontier items, 30% to Path records - each
        // between a floor that suits narrow beams and the most the graph can ever need.  Record and
        // item arenas are addressed with 32-bit byte offsets (buffer descriptors): < 4 GiB each.
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        free_b += slab_cached_bytes(d->device);                        // (the cached slab is this decoder's to take)
        const double n_arcs = (double)d->net->n_arcs, n_states = (double)d->net->n_states;
        const double fixed = n_arcs * 1.0 + n_states * (double)sizeof(StateRec) + 2.0 * d->Fc * d->am->n_gmm * sizeof(float);
        const double budget = std::max(0.0, fit * mem_fraction * (double)free_b / B - fixed);
        const double rec_b = 2.0 * rec_bytes, item_b = 2.0 * (sizeof(Tok) + sizeof(int4)) + 32.0;
        const double path_b = 2.0 * sizeof(PathRec) + 4.0;
        auto pick = [](double share, int64_t lo, int64_t hi) {
            return std::max<int64_t>(std::min<int64_t>(hi, (int64_t)share), std::min(lo, hi));
        };
        const int64_t lim_rec = (0xe0000000LL / (2 * rec_bytes)) & ~63LL, lim_item = 0xe0000000LL / 64;
        if (d->cap_slots <= 0) {
            d->cap_slots = pick(0.5 * budget / rec_b, 1 << 19, std::min<int64_t>(d->net->n_arcs + 65536, lim_rec));
            if (d->slots_hint > d->cap_slots) d->cap_slots = std::min<int64_t>(d->slots_hint, lim_rec);   // setMaxAllocModels: more room, never less
        }
        if (d->cap_items <= 0)
            d->cap_items = pick(0.2 * budget / item_b, 1 << 21, std::min<int64_t>(std::max<int64_t>(2 * d->net->n_arcs + 65536, 1 << 21), lim_item));
        // (records and items stop at their addressing limits: what they leave of the budget goes to the Path
        // records - every collection of those is a stop of the stream's launch - up to 16 per arc of the
        // graph: small graphs do not write more, and tens of GB take seconds to allocate)
        if (d->cap_paths <= 0)
            d->cap_paths = pick(std::max(0.3 * budget, budget - (double)d->cap_slots * rec_b - (double)d->cap_items * item_b) / path_b,
                                1 << 21, std::min<int64_t>(0x40000000LL, std::max<int64_t>(1 << 24, 16 * d->net->n_arcs)));
        if (d->cap_slots > lim_rec || d->cap_items > lim_item || d->cap_paths > 0x7fffff00LL)
            return jd_fail(JD_EINVAL, "arena capacity too large (instance records and frontier items are addressed "
                           "with 32-bit byte offsets: at most %lld / %lld records)", (long long)lim_rec, (long long)lim_item);
        // every wave of a stream's cluster owns 1/NW of each arena; small arenas limit the cluster
        // size instead (launch_search), down to one workgroup per stream
        d->cap_slots = std::max<int64_t>(d->cap_slots, 64 * SW) & ~63LL;
        d->cap_items = std::max<int64_t>(d->cap_items, 64 * SW);
        d->cap_new = std::min<int64_t>(4 * d->cap_items, 0x7fffff00LL);
    }
    {
        StreamDev S;
        char *blk = nullptr;
        size_t off = 0;
        ARENAS();
        const double need = (double)off * B, cached = (double)slab_cached_bytes(d->device);
        if (cached <= 0.0 || need <= cached || need > 1.06 * cached || !(auto_slots || auto_items || auto_paths)) break;
        fit *= 0.998 * cached / need;
    }
    }
    d->C.cap_slots = (unsigned)d->cap_slots; d->C.cap_items = (unsigned)d->cap_items; d->C.cap_new = (unsigned)d->cap_new;
    d->C.cap_paths = (int)d->cap_paths;
    // a stream stops for a collection when its Path records pass this mark (checked between frames): half
    // of a small arena, all but the larger of an eighth / 4 M records of a large one (a frame writes at
    // most one record per frontier item; running out anyway is the JD_ENOMEM that names the arena)
    d->C.gc_threshold = (int)std::max<int64_t>(d->cap_paths / 2, d->cap_paths - std::max<int64_t>(d->cap_paths / 8, 4 << 20));
    d->h_streams.assign((size_t)B, StreamDev());
    rc = dmalloc(d, &d->d_res, (size_t)B * 5 * d->res_cap);
    if (rc) return rc;
    size_t stream_bytes = 0;                                           // (the same for every stream: known after the first sizing pass)
    for (int s = 0; s < B; ++s) {
        StreamDev &S = d->h_streams[(size_t)s];
        memset(&S, 0, sizeof S);
        // ONE slab for all streams (slab_take: the previous decoder's when there is one), a stream's share carved
        // up into 256-byte aligned pieces (two passes over the same list: sizes, then pointers).  What a decoder's
        // set-up costs is the BYTES the driver has to clear: 25-60 GB/s (tools/alloc_probe.py), i.e. seconds for
        // the default 70 % of a 288 GB device.
        char *blk = nullptr;
        size_t off = 0;
        ARENAS();
        if (s == 0) {
            stream_bytes = off;
            rc = slab_take(d->device, stream_bytes * (size_t)B, &d->slab);
            if (rc) return rc;
        }
        blk = (char *)d->slab.p + (size_t)s * stream_bytes;
        off = 0;
        ARENAS();
        {   // the five result arrays of all streams live in one arena [stream][array][res_cap]:
            // fetch_results brings a whole wave back with a single strided copy
            int *base = d->d_res + (size_t)s * 5 * d->res_cap;
            S.res_label = base; S.res_time = base + d->res_cap;
            S.res_score = (float *)(base + 2 * (size_t)d->res_cap); S.res_ac = (float *)(base + 3 * (size_t)d->res_cap);
            S.res_lm = (float *)(base + 4 * (size_t)d->res_cap);
        }
#undef ARENAS
#undef A
        S.res_cap = d->res_cap;
        HIPCHK(hipMemset(S.live, 0, (size_t)d->net->n_arcs));          // per arc: no instance
        reset_srec(S.srec, d->d_row_ptr, d->net->n_states);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemset(S.hist, 0, 2 * HIST_MAX_BINS * sizeof(int)));
        HIPCHK(hipMemset(S.tot, 0, TOT_N * MAXW * sizeof(int)));
        HIPCHK(hipMemset(S.item_end, 0, MAXW * sizeof(int)));
    }
    rc = dmalloc(d, &d->d_streams, (size_t)B);
    if (rc) return rc;
    HIPCHK(hipMemcpy(d->d_streams, d->h_streams.data(), (size_t)B * sizeof(StreamDev), hipMemcpyHostToDevice));
    rc = dmalloc(d, &d->d_T, (size_t)B);
    if (rc) return rc;
    rc = dmalloc(d, &d->d_ctl, (size_t)B);
    if (rc) return rc;
    rc = dmalloc(d, &d->d_status, 8);
    if (rc) return rc;
    HIPCHK(hipMemset(d->d_status, 0, 8 * sizeof(int)));
    if (rc) return rc;
    HIPCHK(hipHostMalloc((void **)&d->h_status, 8 * sizeof(int)));
    {
        std::vector<StreamCtl> hc((size_t)B);
        memset(hc.data(), 0, hc.size() * sizeof(StreamCtl));
        for (auto &c : hc) {
            c.needs_init = 1; c.best_emit = LZ;
            c.best_final.score = LZ; c.best_final.ac = LZ; c.best_final.lm = LZ; c.best_final.path = -1;
        }
        HIPCHK(hipMemcpy(d->d_ctl, hc.data(), hc.size() * sizeof(StreamCtl), hipMemcpyHostToDevice));
    }
    rc = ensure_slab(d, (size_t)d->Fc * d->am->n_gmm);                // streaming API: one stream, one chunk
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    d->arenas_ready = true;
    return JD_OK;
}

// The default capacities take 70% of the device's free memory.  When that cannot be had - another
// process on the same GPU sized itself at the same moment - the defaults are halved, up to three times.
static int ensure_arenas(jd_dec *d)
{
    if (d->arenas_ready) return JD_OK;
    const auto t_start = std::chrono::steady_clock::now();
    struct Report { jd_dec *d; std::chrono::steady_clock::time_point t0;
        ~Report() { if (getenv("JD_VERBOSE")) fprintf(stderr, "arenas: slots %lld, items %lld, Path records %lld per stream x %d streams in %.3f s\n",
                                                      (long long)d->cap_slots, (long long)d->cap_items, (long long)d->cap_paths, d->max_streams,
                                                      std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } } report{d, t_start};
    const int64_t u_slots = d->cap_slots, u_items = d->cap_items, u_paths = d->cap_paths;   // <= 0: not set by the caller
    double frac = 0.7;
    for (int attempt = 0;; ++attempt) {
        const size_t mark = d->allocs.size();
        const int rc = ensure_arenas_try(d, frac);
        if (rc == JD_OK) return JD_OK;
        for (size_t i = mark; i < d->allocs.size(); ++i) (void)hipFree(d->allocs[i]);
        d->allocs.resize(mark);
        slab_give(d->device, d->slab);
        d->slab = ArenaSlab();
        if (d->h_status) { (void)hipHostFree(d->h_status); d->h_status = nullptr; }
        if (d->d_ll_slab) { (void)hipFree(d->d_ll_slab); d->d_ll_slab = nullptr; d->ll_cap = 0; for (int i = 0; i < 3; ++i) d->d_ll[i] = nullptr; }
        d->d_res = nullptr; d->d_streams = nullptr; d->d_T = nullptr; d->d_ctl = nullptr; d->d_status = nullptr;
        (void)hipGetLastError();
        if (rc != JD_ENOMEM || attempt == 3 || (u_slots > 0 && u_items > 0 && u_paths > 0)) return rc;
        d->cap_slots = u_slots; d->cap_items = u_items; d->cap_paths = u_paths; d->cap_new = 0;
        frac *= 0.5;
    }
}

// After an error the lists a stream's next init would clean up from cannot be trusted (stale bids, arrival keys or
// instance flags would silently drop tokens of later utterances): everything per state and per arc is wiped.
static int wipe_stream(jd_dec *d, int s)
{
    const StreamDev &S = d->h_streams[(size_t)s];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(S.live, 0, (size_t)d->net->n_arcs));
    reset_srec(S.srec, d->d_row_ptr, d->net->n_states);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemset(S.tot, 0, TOT_N * MAXW * sizeof(int)));
    HIPCHK(hipMemset(S.item_end, 0, MAXW * sizeof(int)));
    HIPCHK(hipMemset(S.hist, 0, 2 * HIST_MAX_BINS * sizeof(int)));
    HIPCHK(hipDeviceSynchronize());
    d->stream_dirty[(size_t)s] = 0;
    return JD_OK;
}

// mark streams [s0, s0+n) for re-initialisation (IDecoder::init)
static int mark_init(jd_dec *d, int s0, int n, hipStream_t st)
{
    for (int s = s0; s < s0 + n; ++s)
        if (d->stream_dirty[(size_t)s]) { const int rc = wipe_stream(d, s); if (rc) return rc; }
    hipLaunchKernelGGL(jd_mark_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d->d_ctl, s0, n);
    HIPCHK(hipGetLastError());
    return JD_OK;
}

// results -> out[out_idx[i]] (out_idx == nullptr: out[out0 + i]): of streams [s0, s0+n), or - resn_v != null - of the virtual
// result slots [s0, s0+n) the batch pipeline exported them to (ctl_v / resn_v / res_v: their control blocks, word counts and
// result arrays; slot_of: the streams they ran on, for the messages and the dirty marks)
// A stream's device-side error word (StreamCtl::error) as the library's code + message
static int report_stream_error(jd_dec *d, int s_i, int error, int frame, int lst_nw)
{
    if (error == JD_EHIST) return jd_fail(JD_EHIST, "Histogram::addScore - score > maxScore (stream %d)", s_i);
    if (error == JDE_BARRIER)
        return jd_fail(JD_EHIP, "stream %d: a workgroup of the search cluster did not arrive at a barrier (frame %d)", s_i, frame);
    if (error == JDE_LAZY_INV)
        return jd_fail(JD_EHIP, "stream %d: internal error - a token reached a composed state that has not been expanded (frame %d)", s_i, frame);
    if (error == JDE_GEOM)
        return jd_fail(JD_ESTATE, "stream %d: the slot kernel was given a stream in the middle of an utterance (frame %d) whose lists were written by a "
                       "cluster of %d wave segments, not its own %d: the utterance is lost, the stream is reset by its next jd_stream_init", s_i, frame,
                       lst_nw, SW);
    if (error == JDE_LAZY) {
        d->lazy_failed = true;
        LazyDev L;
        int why = 0;
        if (hipMemcpy(&L, d->net->lazy_dev, sizeof L, hipMemcpyDeviceToHost) == hipSuccess)
            (void)hipMemcpy(&why, L.err, sizeof why, hipMemcpyDeviceToHost);
        return jd_fail(JD_ENOMEM, "stream %d: the lazily composed network ran out of %s at frame %d (capacity %d states, %lld arcs): "
                       "what was being decoded at once needs a network with larger max_states / max_arcs", s_i,
                       why == 1 ? "states" : why == 2 ? "arcs"
                                : "stack closing a state (epsilon / tee arcs more than a hundred deep, or a cycle of them)",
                       frame, d->net->n_states, (long long)d->net->n_arcs);
    }
    const char *what = error == JDE_SLOTS ? "instance slots" : error == JDE_ITEMS ? "frontier items"
                     : error == JDE_PATHS ? "Path records" : error == JDE_NEW ? "newly entered arcs" : "arena";
    const long long cap = error == JDE_SLOTS ? d->cap_slots : error == JDE_ITEMS ? d->cap_items : error == JDE_NEW ? d->cap_new : d->cap_paths;
    return jd_fail(JD_ENOMEM, "stream %d: device arena overflow at frame %d: %s (capacity %lld, split over %d wave segments); raise it with "
                   "jd_dec_set_capacity", s_i, frame, what, cap, error == JDE_PATHS ? 1 : std::max(lst_nw, 1));
}

static int fetch_results_from(jd_dec *d, const StreamCtl *ctl_v, const int *resn_v, const int *res_v, const int *slot_of, int s0, int n,
                              jd_hyp *out, int out0, const int *out_idx)
{
    std::vector<StreamDev> hs((size_t)n);
    std::vector<StreamCtl> hc((size_t)n);
    if (resn_v) {
        std::vector<int> rn((size_t)n);
        HIPCHK(hipMemcpy(rn.data(), resn_v + s0, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) hs[(size_t)i].res_n = rn[(size_t)i];
        HIPCHK(hipMemcpy(hc.data(), ctl_v + s0, (size_t)n * sizeof(StreamCtl), hipMemcpyDeviceToHost));
    } else {
        HIPCHK(hipMemcpy(hs.data(), d->d_streams + s0, (size_t)n * sizeof(StreamDev), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hc.data(), d->d_ctl + s0, (size_t)n * sizeof(StreamCtl), hipMemcpyDeviceToHost));
    }
    const int *res_base = resn_v ? res_v : d->d_res;
    int first_err = JD_OK;
    int kmax = 0;
    for (int i = 0; i < n; ++i) kmax = std::max(kmax, std::min(hs[(size_t)i].res_n, d->res_cap));
    std::vector<int> hres((size_t)n * 5 * std::max(kmax, 0));
    if (kmax > 0)                                                      // rows = (stream, array), first kmax words of each
        HIPCHK(hipMemcpy2D(hres.data(), (size_t)kmax * 4, res_base + (size_t)s0 * 5 * d->res_cap, (size_t)d->res_cap * 4,
                           (size_t)kmax * 4, (size_t)n * 5, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        const StreamDev &S = hs[(size_t)i];
        const StreamCtl &K = hc[(size_t)i];
        const int s_i = slot_of ? slot_of[i] : s0 + i;                  // the stream it ran on
        const int oi = out_idx ? out_idx[i] : out0 + i;
        HostResult &R = d->results[(size_t)oi];
        jd_hyp &H = out[oi];
        memset(&H, 0, sizeof H);
        if (K.error && first_err == JD_OK) first_err = report_stream_error(d, s_i, K.error, K.frame, K.lst_nw);
        if (K.error) d->stream_dirty[(size_t)(s_i)] = 1;            // arenas may be inconsistent after an abort: wiped before the next init
        H.stats.n_frames = K.frame;
        H.stats.tot_active_emit_hyps = K.st[ST_EMIT];
        H.stats.tot_active_end_hyps = K.st[ST_END];
        H.stats.tot_active_models = K.st[ST_MODELS];
        H.stats.tot_proc_emit_hyps = K.st[ST_PEMIT];
        H.stats.tot_proc_end_hyps = K.st[ST_PEND];
        H.stats.tot_arcs_visited = K.st[ST_ARCS];
        H.stats.tot_paths = K.st[ST_PATHS];
        H.stats.tot_insts_in = K.st[ST_INSTS];
        H.stats.tot_recs_read = K.st[ST_RECS]; H.stats.tot_new_attached = K.st[ST_NEWL]; H.stats.tot_recs_written = K.st[ST_SURV]; H.stats.tot_entry_items = K.st[ST_KEYS];
        H.stats.tot_items_expanded = K.st[ST_XITEMS]; H.stats.tot_arcs_walked = K.st[ST_WALK]; H.stats.tot_closure_items = K.st[ST_CLOS]; H.stats.tot_bids_placed = K.st[ST_BIDS];
        d->load_sum += (double)K.st[ST_INSTS] + (double)K.st[ST_ARCS]; d->load_frames += (double)K.frame;
        H.stats.ties = 0;
        int k = S.res_n;
        if (k > d->res_cap) {
            if (first_err == JD_OK) first_err = jd_fail(JD_ENOMEM, "stream %d: hypothesis has %d words (> %d)", s_i, k, d->res_cap);
            k = d->res_cap;
        }
        H.n = S.res_n < 0 ? -1 : k;
        const size_t kk = (size_t)std::max(k, 0);
        R.label.resize(kk); R.time.resize(kk); R.score.resize(kk); R.ac.resize(kk); R.lm.resize(kk);
        if (kk) {
            const int *row = hres.data() + (size_t)i * 5 * kmax;
            memcpy(R.label.data(), row, kk * 4); memcpy(R.time.data(), row + kmax, kk * 4);
            memcpy(R.score.data(), row + 2 * (size_t)kmax, kk * 4); memcpy(R.ac.data(), row + 3 * (size_t)kmax, kk * 4);
            memcpy(R.lm.data(), row + 4 * (size_t)kmax, kk * 4);
        }
        H.label = R.label.data(); H.time = R.time.data();
        H.score = R.score.data(); H.ac = R.ac.data(); H.lm = R.lm.data();
        if (kk) { H.tot_score = K.best_final.score; H.tot_ac = K.best_final.ac; H.tot_lm = K.best_final.lm; }
        else { H.tot_score = LZ; H.tot_ac = LZ; H.tot_lm = LZ; }      // DecHyp() defaults, DecHypHistPool.h
    }
    return first_err;
}
static int fetch_results(jd_dec *d, int s0, int n, jd_hyp *out, int out0, const int *out_idx = nullptr)
{
    return fetch_results_from(d, nullptr, nullptr, nullptr, nullptr, s0, n, out, out0, out_idx);
}

#include "jd_host_launch.h"

// Lay a wave of nb <= max_streams utterances out in likelihood tables.
// Frames per chunk.  A search launch lasts as long as its slowest stream and streams do not wait
// for each other inside a launch, so the fewer launches the better: one chunk covers the longest
// utterance when the likelihood table (frames of the batch x tied states floats, 288 GB of HBM to
// draw on) fits a quarter of the free memory; otherwise the batch is decoded in several chunks,
// scored alternately into two tables.  A chunk's table holds the frames the streams really have
// in it, packed stream after stream (only the last 128-row scoring tile of a chunk is partly empty).
static int plan_wave(jd_dec *d, int nb, const int64_t *ustart, const int64_t *ulen, WavePlan &P)
{
    const int G = d->am->n_gmm;
    P = WavePlan();
    P.nb = nb;
    P.T.assign((size_t)nb, 0);
    long long sumT = 0;
    for (int u = 0; u < nb; ++u) {
        const int64_t t = ulen[u];
        if (t < 0 || t > 0x3fffffff) return jd_fail(JD_EINVAL, "utterance %d: bad frame count", u);
        P.T[(size_t)u] = (int)t;
        P.maxT = std::max(P.maxT, (int)t);
        sumT += t;
    }
    int Fc = std::max(128, (P.maxT + 127) / 128 * 128);
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        // (three tables of one size are kept: ensure_slab)
        const double have = 0.25 * (double)free_b + 3.0 * (double)d->ll_cap * sizeof(float);
        const double per_frame = (double)nb * G * sizeof(float);       // (a chunk holds at most nb * Fc rows)
        if (3.0 * (double)(sumT + 128) * G * sizeof(float) > have && 3.0 * (double)Fc * per_frame > have)
            Fc = std::max(128, (int)(have / 3.0 / per_frame) / 128 * 128);
        if (d->Fw_env > 0) Fc = d->Fw_env;
    }
    P.Fc = Fc;
    const int n_chunks = P.n_chunks = std::max(1, (P.maxT + Fc - 1) / Fc);   // chunk 0 also carries recognitionStart
    // rows of a chunk: stream u's frames [c0, min(T_u, c1)) start at row_off[c][u]
    P.chunk_row0.assign((size_t)n_chunks + 1, 0);
    P.row_off.assign((size_t)n_chunks * nb, 0); P.chunk_rows.assign((size_t)n_chunks, 0);
    for (int c = 0; c < n_chunks; ++c) {
        long long r = 0;
        for (int u = 0; u < nb; ++u) {
            P.row_off[(size_t)c * nb + u] = (int)r;
            r += std::max(0, std::min(P.T[(size_t)u], (c + 1) * Fc) - c * Fc);
        }
        if (r > 0x7fffff00LL) return jd_fail(JD_EINVAL, "more than 2^31 frames in one chunk");
        P.chunk_rows[(size_t)c] = (int)((r + GMM_ROWS2 - 1) / GMM_ROWS2 * GMM_ROWS2);
        P.chunk_row0[(size_t)c + 1] = P.chunk_row0[(size_t)c] + (size_t)P.chunk_rows[(size_t)c];
        P.max_rows = std::max(P.max_rows, (size_t)P.chunk_rows[(size_t)c]);
    }
    P.max_rows = std::max<size_t>(P.max_rows, GMM_ROWS2);
    // row -> source frame table for all chunks
    P.n_rows_all = std::max<size_t>(P.chunk_row0[(size_t)n_chunks], 1);
    P.row_src.assign(P.n_rows_all, -1);
    for (int c = 0; c < n_chunks; ++c)
        for (int u = 0; u < nb; ++u) {
            const int n = std::max(0, std::min(P.T[(size_t)u], (c + 1) * Fc) - c * Fc);
            int *dst = P.row_src.data() + P.chunk_row0[(size_t)c] + (size_t)P.row_off[(size_t)c * nb + u];
            for (int dt = 0; dt < n; ++dt) {
                const int64_t src = ustart[u] + (int64_t)c * Fc + dt;
                if (src > 0x7fffffff) return jd_fail(JD_EINVAL, "more than 2^31 frames in one batch");
                dst[dt] = (int)src;
            }
        }
    return JD_OK;
}

// The likelihood tables: THREE of one size in one allocation (a search launch addresses the table of every stream it
// advances as a row of that slab: SearchArgs::ll + slot * G), each with its row -> source frame table.  Three, because
// three batches can be under way at once ("two batches in flight" below): the one the caller is waiting for, the one
// behind it whose utterances are already being searched, and the one behind that whose table is being scored.
// Growing the slab loses what is in it: only when nothing scored ahead is waiting (the callers see to that).
static int ensure_slab(jd_dec *d, size_t floats)
{
    if (floats <= d->ll_cap) return JD_OK;
    if (d->d_cells) { (void)hipFree(d->d_cells); d->d_cells = nullptr; d->cells_words = 0; }   // (the diagnostics bitmap is as long as the slab)
    if (d->d_ll_slab) (void)hipFree(d->d_ll_slab);
    d->d_ll_slab = nullptr; d->ll_cap = 0;
    for (int i = 0; i < 3; ++i) d->d_ll[i] = nullptr;
    const size_t G = (size_t)d->am->n_gmm;
    floats = (floats + G - 1) / G * G;                                 // whole rows: a table starts at a row of the slab
    hipError_t e = hipMalloc(&d->d_ll_slab, 3 * floats * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return jd_fail(e == hipErrorOutOfMemory ? JD_ENOMEM : JD_EHIP, "hipMalloc of %zu bytes (likelihood tables) failed: %s",
                       3 * floats * sizeof(float), hipGetErrorString(e));
    }
    d->ll_cap = floats;
    for (int i = 0; i < 3; ++i) d->d_ll[i] = d->d_ll_slab + (size_t)i * floats;
    return JD_OK;
}
static int ensure_rows(jd_dec *d, int i, size_t rows)
{
    if (rows > d->row_src_cap[i]) {
        if (d->d_row_src[i]) (void)hipFree(d->d_row_src[i]);
        d->d_row_src[i] = nullptr; d->row_src_cap[i] = 0;
        HIPCHK(hipMalloc(&d->d_row_src[i], rows * sizeof(int)));
        d->row_src_cap[i] = rows;
    }
    return JD_OK;
}
// table i holds `floats`, its row table `rows` entries (grown, never shrunk)
static int ensure_table(jd_dec *d, int i, size_t floats, size_t rows)
{
    int rc = ensure_slab(d, floats);
    if (rc) return rc;
    return ensure_rows(d, i, rows);
}
static long long table_row0(const jd_dec *d, int buf) { return (long long)buf * (long long)(d->ll_cap / (size_t)d->am->n_gmm); }

// Forget what was scored, searched or announced ahead (the scoring stream is drained first: a table being written is
// not re-used; utterances whose search had been started are simply started again when their batch is decoded)
static void pipe_drain(jd_dec *d);
static void pf_discard(jd_dec *d)
{
    pipe_drain(d);
    bool scoring = false;
    for (const Prefetch &F : d->pf_q) if (F.state == 2) scoring = true;
    if (scoring) (void)hipStreamSynchronize(d->s_gmm);
    for (Prefetch &F : d->pf_q) F.drop();
    d->pf_q.clear();
}
// ... what has been scored ahead only: the announcements stay (they are scored again, beside the next search)
static void pf_discard_scored(jd_dec *d)
{
    bool scoring = false;
    for (const Prefetch &F : d->pf_q) if (F.state == 2) scoring = true;
    if (scoring) (void)hipStreamSynchronize(d->s_gmm);
    for (Prefetch &F : d->pf_q) { F.drop_events(); F.state = 1; F.bank = -1; }
}
// a table that neither the running batch (fg_buf, -1: none) nor a batch scored ahead holds; -1: none
static int free_table(const jd_dec *d, int fg_buf)
{
    for (int t = 0; t < 3; ++t) {
        bool used = t == fg_buf;
        for (const Prefetch &F : d->pf_q) if (F.state == 2 && F.buf == t) used = true;
        if (!used) return t;
    }
    return -1;
}

// Start the scoring of the batches that have been announced and not scored yet, each into a table nobody holds.  Called
// by launch_search right behind the dispatch of a persistent search launch, once that launch is resident: the scoring
// kernel's blocks then only ever get the CUs that clusters of the search have left.
static int pf_launch(jd_dec *d)
{
    const size_t G = (size_t)d->am->n_gmm;
    bool any = false;
    for (Prefetch &F : d->pf_q) {
        if (F.state != 1) continue;
        // Scoring ahead is an optimisation: a batch that cannot be scored now (no table free, a table larger than the
        // slab - which cannot grow while tables are in use) is scored when it is decoded, as ever.
        const int buf = free_table(d, d->fg_buf);
        if (buf < 0 || F.plan.max_rows * G > d->ll_cap) break;
        if (ensure_rows(d, buf, F.plan.n_rows_all) != JD_OK) { (void)hipGetLastError(); break; }
        if (hipEventCreate(&F.ev0) != hipSuccess || hipEventCreate(&F.ev1) != hipSuccess) { (void)hipGetLastError(); F.drop_events(); break; }
        if (hipMemcpyAsync(d->d_row_src[buf], F.plan.row_src.data(), F.plan.n_rows_all * sizeof(int), hipMemcpyHostToDevice, d->s_gmm) != hipSuccess ||
            hipEventRecord(F.ev0, d->s_gmm) != hipSuccess ||
            launch_gmm(d->am, d->amb, F.feats, d->d_row_src[buf], F.plan.chunk_rows[0], d->d_ll[buf], d->s_gmm, 0, true) != JD_OK ||
            hipEventRecord(F.ev1, d->s_gmm) != hipSuccess) {
            (void)hipGetLastError(); F.drop_events(); break;
        }
        F.buf = buf; F.state = 2;
        any = true;
    }
    (void)any;
    HIPCHK(hipMemsetAsync(d->d_status + 4, 0, sizeof(int), d->s_gmm)); // (behind the scoring, if any: the search may be re-planned again)
    return JD_OK;
}
static bool pf_wants_scoring(const jd_dec *d)
{
    for (const Prefetch &F : d->pf_q) if (F.state == 1) return free_table(d, d->fg_buf) >= 0 && F.plan.max_rows * (size_t)d->am->n_gmm <= d->ll_cap;
    return false;
}
static bool pf_scoring_in_flight(const jd_dec *d)
{
    for (const Prefetch &F : d->pf_q)
        if (F.state == 2 && F.ev1 && hipEventQuery(F.ev1) != hipSuccess) { (void)hipGetLastError(); return true; }
    return false;
}
static double pf_scoring_rows(const jd_dec *d)
{
    double rows = 0.0;
    for (const Prefetch &F : d->pf_q) if (F.state == 1) rows += (double)F.plan.chunk_rows[0];
    return rows;
}

// Announce a wave: it joins the queue of batches ahead (at most three; one more is not taken) and is scored beside
// the next search launch that leaves a table free.  A wave whose table would be cut into chunks is not announced (it
// is scored when it is decoded, as ever).  front: in front of what the caller announced (the waves of one call).
static int pf_announce(jd_dec *d, int nb, const float *d_feats, const int64_t *ustart, const int64_t *ulen, bool front = false)
{
    if (nb <= 0 || nb > d->max_streams || d->pf_q.size() >= 3) return JD_OK;
    Prefetch F;
    F.ustart.assign(ustart, ustart + nb);
    F.ulen.assign(ulen, ulen + nb);
    const int rc = plan_wave(d, nb, F.ustart.data(), F.ulen.data(), F.plan);
    if (rc) return rc;
    if (F.plan.n_chunks != 1) return JD_OK;
    F.feats = d_feats; F.nb = nb; F.state = 1;
    if (front) d->pf_q.push_front(std::move(F)); else d->pf_q.push_back(std::move(F));
    return JD_OK;
}

// ---- two batches in flight.  A batch of 64 utterances lasts as long as its longest one - a chain of ~1150 dependent
// frames - while the clusters of the shorter ones are through after half of that: whatever the plan, workgroup-time is
// burnt waiting.  What a stream of batches allows: the utterances of the batch BEHIND the one the caller
// is waiting for are started beside it, one workgroup each (no cluster barrier at all), on stream slots of their own
// (the decoder's streams are two banks of max_streams / 2) - when that batch's turn comes its utterances are hundreds
// of frames in, and the step is that much shorter.  It takes a table scored one batch earlier, i.e. announcements
// that run two batches ahead of the decode (jd_dec_prefetch_scores twice before the first decode, once per decode
// afterwards).  Nothing changes for a caller who does not announce, announces one ahead, or fills more than half of
// the streams with one batch; results cannot depend on any of it (streams never interact).
// The batch behind the running one, if its table is there: its unfinished streams as work items for the launch that is
// being planned (started - initialised, frame counts set - the first time round).  heads: {frame, T, error, needs_init} of
// every stream, as of now (only read for a batch that has been started before).
static int pf_background(jd_dec *d, int fg_bank, const std::vector<int> *heads, hipStream_t st, std::vector<int2> *work, std::vector<double> *left)
{
    work->clear(); left->clear();
    if (d->pf_q.empty() || !d->pipeline || d->C.lazy) return JD_OK;
    // Searching ahead pays where a frame is a chain of dependent steps and workgroups wait - not where it is bytes: a
    // stream-frame of several hundred thousand instances keeps every workgroup it can get busy, and a second batch on the
    // same chip only splits them (measured: the 14 M-arc graph 119 k -> 55 k frames/s, configs[3] 5.3 k -> 0.8 k).  The
    // decoder knows its load from the batches it has decoded (load_scale: instances + arcs per stream-frame over
    // configs[1]'s 23.7 k); beyond JD_BG_MAX_LOAD times that the batch behind waits for its turn.
    if (d->load_scale > d->bg_max_load) return JD_OK;
    Prefetch &F = d->pf_q.front();
    const int B = d->max_streams / 2;
    if (F.state != 2 || F.nb > B || F.plan.n_chunks != 1) return JD_OK;
    if (F.bank < 0) {
        // its table was scored beside the search before this one and is about ready: worth a moment (JD_BG_WAIT_US)
        const auto tq0 = std::chrono::steady_clock::now();
        while (hipEventQuery(F.ev1) != hipSuccess) {
            (void)hipGetLastError();
            const double waited_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq0).count();
            if (waited_us > d->bg_wait_us) return JD_OK;              // still being scored
        }
        F.bank = fg_bank ^ 1;
        const int s0 = F.bank * B;
        int rc = mark_init(d, s0, F.nb, st);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(d->d_T + s0, F.plan.T.data(), (size_t)F.nb * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(jd_set_T_kernel, dim3((F.nb + 63) / 64), dim3(64), 0, st, d->d_ctl, s0, F.nb, d->d_T + s0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));                              // (plan.T may move with the queue)
        for (int u = 0; u < F.nb; ++u) {
            work->push_back(make_int2(s0 + u, (int)(table_row0(d, F.buf) + F.plan.row_off[(size_t)u])));
            left->push_back((double)F.plan.T[(size_t)u]);
        }
        return JD_OK;
    }
    if (!heads) return JD_OK;
    const int s0 = F.bank * B;
    for (int u = 0; u < F.nb; ++u) {
        const int *h = heads->data() + (size_t)(s0 + u) * 4;
        if (h[2] == 0 && h[0] < F.plan.T[(size_t)u]) {
            work->push_back(make_int2(s0 + u, (int)(table_row0(d, F.buf) + F.plan.row_off[(size_t)u])));
            left->push_back((double)(F.plan.T[(size_t)u] - h[0]));
        }
    }
    return JD_OK;
}

// Decode one wave of nb <= max_streams utterances held in device memory.
// Stream u decodes frames [ustart[u], ustart[u] + ulen[u]) of d_feats.  *s0_out: the first stream it ran on.
static int decode_wave(jd_dec *d, int nb, const float *d_feats, const int64_t *ustart, const int64_t *ulen,
                       hipStream_t user_stream, int *s0_out)
{
    const int G = d->am->n_gmm;
    const double gmm_before = d->timing.gmm_ms, search_before = d->timing.search_ms;
    // is this wave the one at the head of the batches ahead (jd_dec_prefetch_scores)?  An announcement speaks of a batch
    // BEHIND the decode that follows it (a stream of equal batches announces the very batch it is about to decode):
    // only a batch whose table has been scored ahead - beside an earlier decode - can be the one decoded now.
    Prefetch me;
    bool mine = false;
    if (!d->pf_q.empty()) {
        Prefetch &F = d->pf_q.front();
        mine = F.state == 2 && F.feats == d_feats && F.nb == nb && std::equal(ustart, ustart + nb, F.ustart.begin()) &&
               std::equal(ulen, ulen + nb, F.ulen.begin());
        if (mine) { me = std::move(F); d->pf_q.pop_front(); }
        else {
            // not the batch that was announced: what has been scored (or searched) ahead is for a batch that is not
            // coming now - dropped; announcements nothing has been done for yet stay (they are batches BEHIND this one)
            // (a wave that is decoded a second time - the retry of jd_decode_batch_device - finds the waves behind it there)
            bool worked = false;
            for (const Prefetch &Q : d->pf_q) if (Q.state == 2 || Q.bank >= 0) worked = true;
            if (worked && !d->pf_retry) pf_discard(d);
        }
    }
    struct MeGuard { Prefetch &p; ~MeGuard() { p.drop(); } } me_guard{me};
    const bool prefetched = mine;
    WavePlan plan_local;
    int rc = JD_OK;
    if (!mine) { rc = plan_wave(d, nb, ustart, ulen, plan_local); if (rc) return rc; }
    const WavePlan &P = mine ? me.plan : plan_local;
    const std::vector<int> &T = P.T;
    const int Fc = P.Fc, n_chunks = P.n_chunks, maxT = P.maxT;
    // stream bank: the utterances of a batch that fills at most half of the streams run on one of two banks, so that
    // the batch behind it can be started beside it (pf_background)
    const int B = d->max_streams / 2;
    const bool pipe = d->pipeline && B > 0 && nb <= B && n_chunks == 1 && !d->C.lazy;
    {
        bool scored = false, started = false;
        for (const Prefetch &Q : d->pf_q) { if (Q.state == 2) scored = true; if (Q.bank >= 0) started = true; }
        // a wave in chunks needs tables 0 and 1: nothing scored ahead survives it
        if (n_chunks > 1 && (scored || started)) pf_discard(d);
        // a wave outside the banks takes the streams a batch behind it has been started on: that batch starts again when
        // its turn comes (its table stays)
        else if (!pipe && started) for (Prefetch &Q : d->pf_q) Q.bank = -1;
    }
    const bool resumed = prefetched && pipe && me.bank >= 0;           // its utterances have been started beside the batch before
    int bank = 0;
    if (pipe) {
        bank = resumed ? me.bank : 0;
        if (!resumed) for (const Prefetch &Q : d->pf_q) if (Q.bank == bank) bank ^= 1;   // (at most one batch ahead holds a bank)
    }
    const int s0 = pipe ? bank * B : 0;
    if (s0_out) *s0_out = s0;
    // tables: a single-chunk wave sits in one of the three; chunks alternate between tables 0 and 1
    int b0 = 0;
    if (n_chunks == 1) {
        if (prefetched) b0 = me.buf;
        else {
            // the three tables are of one size: this wave's, or what a batch announced behind it needs if that is more
            // (with some room: a slab that has to grow loses whatever was scored ahead)
            size_t need = P.max_rows * (size_t)G;
            for (const Prefetch &Q : d->pf_q) need = std::max(need, Q.plan.max_rows * (size_t)G);
            if (need > d->ll_cap) { pf_discard_scored(d); need += need / 8; }
            rc = ensure_slab(d, need);
            if (rc) return rc;
            b0 = free_table(d, -1);
            if (b0 < 0) { pf_discard(d); b0 = 0; }
            rc = ensure_rows(d, b0, P.n_rows_all);
            if (rc) return rc;
        }
    } else {
        rc = ensure_slab(d, P.max_rows * (size_t)G);
        if (rc) return rc;
        for (int i = 0; i < 2; ++i) { rc = ensure_rows(d, i, i == 0 ? P.n_rows_all : 0); if (rc) return rc; }
    }
    d->fg_buf = n_chunks == 1 ? b0 : -2;                               // (-2: both of tables 0 / 1 - nothing is scored ahead meanwhile)
    struct FgGuard { jd_dec *d; ~FgGuard() { d->fg_buf = -1; } } fg_guard{d};
    // the caller's features may have been produced asynchronously on its stream (NULL = the
    // default stream): the decoder's own streams are non-blocking, so order against it explicitly
    HIPCHK(hipStreamSynchronize(user_stream));
    if (!prefetched)
        HIPCHK(hipMemcpyAsync(d->d_row_src[b0], P.row_src.data(), P.n_rows_all * sizeof(int), hipMemcpyHostToDevice, d->s_gmm));
    std::vector<int> frame0((size_t)nb, 0);                            // where the streams stand (a batch started ahead: hundreds of frames in)
    if (!resumed) {
        HIPCHK(hipMemcpyAsync(d->d_T + s0, T.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice, d->s_search));
        rc = mark_init(d, s0, nb, d->s_search);
        if (rc) return rc;
        hipLaunchKernelGGL(jd_set_T_kernel, dim3((nb + 63) / 64), dim3(64), 0, d->s_search, d->d_ctl, s0, nb, d->d_T + s0);
        HIPCHK(hipGetLastError());
    } else {
        std::vector<int> head((size_t)nb * 4);
        HIPCHK(hipMemcpy2D(head.data(), 16, d->d_ctl + s0, sizeof(StreamCtl), 16, (size_t)nb, hipMemcpyDeviceToHost));
        for (int u = 0; u < nb; ++u) { frame0[(size_t)u] = head[(size_t)u * 4]; d->timing.ahead_frames += head[(size_t)u * 4]; }
    }

    struct Events {                                                    // (destroyed on every way out, error paths included)
        std::vector<hipEvent_t> v;
        ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
    } ev_gs, ev_ge;
    ev_gs.v.assign((size_t)n_chunks, nullptr); ev_ge.v.assign((size_t)n_chunks, nullptr);
    std::vector<hipEvent_t> &gs = ev_gs.v, &ge = ev_ge.v;
    if (prefetched) {                                                  // the scoring's own events (Events destroys them)
        gs[0] = me.ev0; ge[0] = me.ev1;
        me.ev0 = me.ev1 = nullptr;
    } else
        for (int c = 0; c < n_chunks; ++c) { HIPCHK(hipEventCreate(&gs[(size_t)c])); HIPCHK(hipEventCreate(&ge[(size_t)c])); }
    auto w0 = std::chrono::steady_clock::now();
    // Scoring runs one chunk ahead of the search on its own stream.  launch_search returns when its
    // chunk is through (it synchronises to learn whether a stream stopped for garbage collection),
    // so by the time chunk c+1 is scored into buffer (c+1)&1 the search of chunk c-1 has left it.
    auto score_chunk = [&](int c) -> int {
        HIPCHK(hipEventRecord(gs[(size_t)c], d->s_gmm));
        // (a later chunk is launched while the previous one is searched: k_search holds every CU's
        // registers, so its workgroups start as the search's clusters finish - they fill the tail)
        if (P.chunk_rows[(size_t)c] == 0) { HIPCHK(hipEventRecord(ge[(size_t)c], d->s_gmm)); return JD_OK; }
        int r = launch_gmm(d->am, d->amb, d_feats, d->d_row_src[b0] + P.chunk_row0[(size_t)c], P.chunk_rows[(size_t)c], d->d_ll[b0 ^ (c & 1)], d->s_gmm,
                           0, true);
        if (r) return r;
        HIPCHK(hipEventRecord(ge[(size_t)c], d->s_gmm));
        return JD_OK;
    };
    if (!prefetched) { rc = score_chunk(0); if (rc) return rc; }
    double waited_ms = 0.0;
    std::vector<int2> work;
    std::vector<double> weight;
    long long frames_here = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const auto tw0 = std::chrono::steady_clock::now();
        HIPCHK(hipEventSynchronize(ge[(size_t)c]));                    // scores of this chunk are there
        waited_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        if (c + 1 < n_chunks) { rc = score_chunk(c + 1); if (rc) return rc; }
        // a launch lasts as long as its slowest stream: the clusters are sized by the frames ahead of each
        const int c0 = c * Fc, c1 = (c + 1) * Fc;
        const long long row0 = table_row0(d, b0 ^ (c & 1));            // (k_search reads row  slot + (f - c0)  of the slab)
        work.clear(); weight.clear();
        for (int u = 0; u < nb; ++u)
            if ((c == 0 && frame0[(size_t)u] == 0) || T[(size_t)u] > std::max(c0, frame0[(size_t)u])) {
                work.push_back(make_int2(s0 + u, (int)(row0 + P.row_off[(size_t)c * nb + u])));   // {stream, its first row in the chunk's table}
                weight.push_back((double)(std::min(T[(size_t)u], c1) - std::max(c0, frame0[(size_t)u])));
                frames_here += std::min(T[(size_t)u], c1) - std::max(c0, frame0[(size_t)u]);
            }
        const float *ll = d->d_ll_slab;
        if (d->load_scale == 1.0 && d->weighted && c == 0 && nb > 1 && std::min(maxT, c1) - c0 > 128 && !resumed) {
            // the decoder's very first batch: nothing is known about the load yet, and the cluster sizes depend
            // on it (a frame of 2 M instances is not a frame of 12 k) - a short launch finds out
            const int c_mid = c0 + 32;
            std::vector<double> wp(weight.size());
            for (size_t i = 0; i < wp.size(); ++i) wp[i] = std::min(weight[i], 32.0);
            rc = launch_search(d, work, ll, (long long)G, c0, c_mid, d->s_search, &wp);
            if (rc) { for (int u = 0; u < nb; ++u) d->stream_dirty[(size_t)(s0 + u)] = 1; return rc; }
            if (d->load_scale == 1.0) { rc = learn_load(d, work); if (rc) return rc; }
            std::vector<int2> w2; std::vector<double> wt2;
            for (size_t i = 0; i < work.size(); ++i) {
                const int u = work[i].x - s0;
                if (T[(size_t)u] > c_mid) { w2.push_back(work[i]); wt2.push_back((double)(std::min(T[(size_t)u], c1) - c_mid)); }
            }
            work.swap(w2); weight.swap(wt2);
        }
        d->pf_armed = n_chunks == 1;                                   // batches ahead are scored / started beside this launch
        d->fg_bank = pipe ? bank : -1;
        rc = launch_search(d, work, ll, (long long)G, c0, c1, d->s_search, &weight);
        d->pf_armed = false; d->fg_bank = -1;
        if (rc) { for (int u = 0; u < nb; ++u) d->stream_dirty[(size_t)(s0 + u)] = 1; return rc; }   // (nobody looks at the streams' error words)
    }
    hipLaunchKernelGGL(jd_finish_kernel, dim3((nb + 63) / 64), dim3(64), 0, d->s_search, d->d_ctl, d->d_streams, s0, nb);
    HIPCHK(hipGetLastError());
    {   // (a table being scored ahead is waited for by the wave that uses it)
        bool scoring = false;
        for (const Prefetch &Q : d->pf_q) if (Q.state == 2) scoring = true;
        if (!scoring) HIPCHK(hipStreamSynchronize(d->s_gmm));
    }
    HIPCHK(hipStreamSynchronize(d->s_search));
    auto w1 = std::chrono::steady_clock::now();
    d->timing.total_ms += std::chrono::duration<double, std::milli>(w1 - w0).count();
    d->last_wave_ms = std::chrono::duration<double, std::milli>(w1 - w0).count();
    for (int c = 0; c < n_chunks; ++c) {
        float gms = 0.0f;
        if (hipEventElapsedTime(&gms, gs[(size_t)c], ge[(size_t)c]) == hipSuccess) d->timing.gmm_ms += gms;
    }
    d->timing.gmm_wait_ms += waited_ms;
    d->timing.gmm_launches += n_chunks;
    d->timing.prefetched += prefetched ? 1 : 0;
    {   // what scoring (on its own) and search cost on this decoder: launch_search's choice when a table is scored ahead
        long long rows = 0;
        for (int c = 0; c < n_chunks; ++c) rows += P.chunk_rows[(size_t)c];
        if (!prefetched && rows > 0 && d->timing.gmm_ms > gmm_before) d->gmm_ms_per_row = (d->timing.gmm_ms - gmm_before) / (double)rows;
        // (a launch that also advanced the batch behind: its frames are not known here - the estimate stays)
        if (frames_here > 0 && d->timing.search_ms > search_before && !d->bg_ran)
            d->search_ms_per_frame = (d->timing.search_ms - search_before) / (double)frames_here;
        d->bg_ran = false;
    }
    for (int u = 0; u < nb; ++u) d->timing.search_frames += T[(size_t)u];
    d->timing.gmm_frames = d->timing.search_frames;
    d->timing.gmm_states = G;
    return JD_OK;
}

extern "C" int jd_decode_batch_device(jd_dec *d, int32_t n_utts, const float *d_feats, const int64_t *offs,
                                      void *hip_stream, jd_hyp *out)
{
    if (!d || !offs || !out || n_utts < 0) return jd_fail(JD_EINVAL, "jd_decode_batch_device: bad argument");
    int rc = check_device(d->device);
    if (rc) return rc;
    {   // the oldest batch of the pipeline (JD_PIPELINE=3)?
        int handled = 0;
        rc = pipe_decode(d, n_utts, d_feats, offs, out, &handled);
        if (rc || handled) return rc;
        // ... or more utterances than streams, nothing announced: through the pipeline's slots as well - a stream takes the next
        // utterance the moment its own is through, where the waves below end with their longest one
        if (d->pipe_mode && n_utts > (d->pipe_slots > 0 ? d->pipe_slots : d->max_streams) && !d->pipe_on) {
            HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));     // the features are there
            int taken = 0;
            rc = pipe_announce(d, n_utts, d_feats, offs, &taken);
            if (rc) return rc;
            if (taken) {
                rc = pipe_decode(d, n_utts, d_feats, offs, out, &handled);
                if (rc || handled) return rc;
            }
        }
    }
    rc = ensure_arenas(d);
    if (rc) return rc;
    if ((size_t)n_utts > d->results.size()) d->results.resize((size_t)n_utts);
    d->timing = jd_timing();
    int first_err = JD_OK;
    // More utterances than streams: successive waves.  A wave lasts as long as its longest
    // utterance, so the waves are formed from the utterances sorted by length (results do not
    // depend on which utterances share a wave).
    std::vector<int> order((size_t)n_utts);
    std::iota(order.begin(), order.end(), 0);
    if (n_utts > d->max_streams)
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return offs[a + 1] - offs[a] > offs[b + 1] - offs[b]; });
    std::vector<int64_t> ustart((size_t)d->max_streams), ulen((size_t)d->max_streams);
    std::vector<int64_t> nstart((size_t)d->max_streams), nlen((size_t)d->max_streams);
    // the batch takes every stream over: utterances of the streaming interface that were never finished leave the
    // (lazily composed) network now - before the first jd_lazy_enter, which starts a new arena generation only when
    // nobody is inside, or a full network would fail this call and, the release below never reached, every later one
    for (int s = 0; s < d->max_streams; ++s) {
        d->stream_started[(size_t)s] = 0; d->stream_T[(size_t)s] = 0;
        if (d->lazy_in[(size_t)s]) { d->lazy_in[(size_t)s] = 0; jd_lazy_leave(d->net, 1); }
    }
    // Scoring ahead inside the batch: every wave but the last announces the wave behind it, whose table is then scored
    // beside this wave's search (pf_launch); what the caller announced (the batches behind this one) rides on the last wave.
    const bool waves = n_utts > d->max_streams;
    std::deque<Prefetch> callers;
    if (waves) {
        // (a batch of several waves is never itself one of the announced ones: what has been worked ahead goes)
        bool worked = false;
        for (const Prefetch &Q : d->pf_q) if (Q.state == 2 || Q.bank >= 0) worked = true;
        if (worked) pf_discard(d);
        callers.swap(d->pf_q);
    }
    // (on an error way out nothing scored or announced ahead survives: the caller may free the features next)
    struct Restore {
        jd_dec *d; std::deque<Prefetch> &p; bool ok;
        ~Restore() { for (Prefetch &F : p) F.drop(); if (!ok) pf_discard(d); }
    } callers_guard{d, callers, false};
    for (int u0 = 0; u0 < n_utts; u0 += d->max_streams) {
        const int nb = std::min(d->max_streams, n_utts - u0);
        for (int i = 0; i < nb; ++i) {
            const int u = order[(size_t)(u0 + i)];
            ustart[(size_t)i] = offs[u]; ulen[(size_t)i] = offs[u + 1] - offs[u];
        }
        if (waves && u0 + nb < n_utts) {
            const int nn = std::min(d->max_streams, n_utts - (u0 + nb));
            for (int i = 0; i < nn; ++i) {
                const int u = order[(size_t)(u0 + nb + i)];
                nstart[(size_t)i] = offs[u]; nlen[(size_t)i] = offs[u + 1] - offs[u];
            }
            rc = pf_announce(d, nn, d_feats, nstart.data(), nlen.data());
            if (rc) return rc;
        } else if (waves) {
            for (Prefetch &F : callers) d->pf_q.push_back(std::move(F));
            callers.clear();
        }
        struct RetryGuard { jd_dec *d; ~RetryGuard() { d->pf_retry = false; } } retry_guard{d};
        for (int attempt = 0;; ++attempt) {
            // (lazily composed networks: the wave's utterances enter the network - which starts a new arena generation
            // when nobody is inside an utterance and it is nearly full, or has run out of room)
            bool net_failed = false;
            rc = jd_lazy_enter(d->net, nb, &net_failed);
            if (rc) return rc;
            if (net_failed) {
                jd_lazy_leave(d->net, nb);
                return jd_fail(JD_ENOMEM, "the lazily composed network has run out of room and utterances of other decoders are inside it: "
                               "it starts again when they are through (capacity %d states, %lld arcs)", d->net->n_states, (long long)d->net->n_arcs);
            }
            d->lazy_failed = false;
            const jd_timing t_before = d->timing;                      // (a wave that is decoded twice counts once)
            const double ls_before = d->load_sum, lf_before = d->load_frames;
            int s0 = 0;
            rc = decode_wave(d, nb, d_feats, ustart.data(), ulen.data(), (hipStream_t)hip_stream, &s0);
            const int rf = rc ? rc : fetch_results(d, s0, nb, out, 0, order.data() + u0);
            jd_lazy_leave(d->net, nb);
            if (rc) return rc;
            // out of graph room under way: once more - jd_lazy_enter gives the wave a fresh generation to itself
            if (d->lazy_failed && attempt == 0) {
                d->timing = t_before; d->load_sum = ls_before; d->load_frames = lf_before;
                // (what was scored ahead beside attempt 0 belongs to the waves BEHIND this one: the retry, which matches no
                // announced table, must not drop it)
                d->pf_retry = true;
                continue;
            }
            d->pf_retry = false;
            if (rf && first_err == JD_OK) first_err = rf;
            break;
        }
    }
    if (d->load_frames > 0.0) {                                        // the load this batch had -> the next batch's cluster sizes
        const double scale = std::min(1e5, std::max(0.25, d->load_sum / d->load_frames / 23700.0));
        d->load_scale = (d->load_scale == 1.0) ? scale : 0.5 * (d->load_scale + scale);
        d->load_sum = d->load_frames = 0.0;
    }
    callers_guard.ok = true;
    return first_err;
}

extern "C" int jd_dec_prefetch_scores(jd_dec *d, int32_t n_utts, const float *d_feats, const int64_t *offs, void *hip_stream)
{
    if (!d || n_utts < 0 || (n_utts > 0 && (!d_feats || !offs))) return jd_fail(JD_EINVAL, "jd_dec_prefetch_scores: bad argument");
    int rc = check_device(d->device);
    if (rc) return rc;
    if (n_utts == 0) { pf_discard(d); return JD_OK; }                  // nothing: what was scored or announced ahead is dropped
    // what cannot be scored ahead is scored when it is decoded, as ever: more utterances than streams (several waves,
    // formed by length: jd_decode_batch_device scores each of them beside the wave before it), a table cut into chunks
    if (d->pipe_mode) {   // JD_PIPELINE=3: the batch goes through the resident kernel, utterance by utterance (any number of them)
        HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));         // the features are there
        int taken = 0;
        rc = pipe_announce(d, n_utts, d_feats, offs, &taken);
        if (rc || taken) return rc;
    }
    if (n_utts > d->max_streams) return JD_OK;
    HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));             // the features are there
    std::vector<int64_t> ulen((size_t)n_utts);
    for (int u = 0; u < n_utts; ++u) ulen[(size_t)u] = offs[u + 1] - offs[u];
    return pf_announce(d, n_utts, d_feats, offs, ulen.data());
}

extern "C" int jd_decode_batch(jd_dec *d, int32_t n_utts, const float *const *feats, const int32_t *n_frames,
                               jd_hyp *out)
{
    if (!d || !feats || !n_frames || !out || n_utts < 0) return jd_fail(JD_EINVAL, "jd_decode_batch: bad argument");
    int rc = check_device(d->device);
    if (rc) return rc;
    const int D = d->am->D;
    std::vector<int64_t> offs((size_t)n_utts + 1, 0);
    for (int u = 0; u < n_utts; ++u) {
        if (n_frames[u] < 0) return jd_fail(JD_EINVAL, "jd_decode_batch: negative frame count");
        offs[(size_t)u + 1] = offs[(size_t)u] + n_frames[u];
    }
    float *d_feats = nullptr;
    HIPCHK(hipMalloc(&d_feats, std::max<size_t>((size_t)offs[(size_t)n_utts] * D, 1) * sizeof(float)));
    for (int u = 0; u < n_utts; ++u)
        if (n_frames[u] > 0)
            HIPCHK(hipMemcpy(d_feats + (size_t)offs[(size_t)u] * D, feats[u], (size_t)n_frames[u] * D * sizeof(float),
                             hipMemcpyHostToDevice));
    rc = jd_decode_batch_device(d, n_utts, d_feats, offs.data(), nullptr, out);
    (void)hipFree(d_feats);
    return rc;
}

// ---- streaming API: IDecoder::init / processFrame / finish for one stream

#include "jd_host_stream.h"

// Diagnostics: per-workgroup cycle accounting of k_search (100 MHz wall clock): for every
// workgroup of the grid {work lists A, phase A, workgroup wait, cluster barrier, work lists X, phase X,
// workgroup wait, cluster barriers, frames} (thread 0's timeline), summed over the launches since it
// was enabled.  enable >= 0 switches it on (and clears it), < 0 off;
// fetch (1024 x 16 int64, may be NULL) receives the current sums.
extern "C" int jd_dec_debug_trace(jd_dec *d, int32_t enable, int64_t *fetch)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_debug_trace: null");
    const size_t n = (size_t)1024 * 16;
    static_assert(sizeof(long long) == sizeof(int64_t), "");
    if (fetch && d->d_dbg) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(fetch, d->d_dbg, n * sizeof(long long), hipMemcpyDeviceToHost));
    } else if (fetch) memset(fetch, 0, n * sizeof(int64_t));
    if (enable >= 0 && !fetch) {
        if (!d->d_dbg_buf) {                                           // (one buffer per decoder, kept when tracing is switched off)
            long long *p = nullptr;
            HIPCHK(hipMalloc(&p, n * sizeof(long long)));
            d->allocs.push_back(p);
            d->d_dbg_buf = p;
        }
        d->d_dbg = d->d_dbg_buf;
        HIPCHK(hipMemset(d->d_dbg, 0, n * sizeof(long long)));
    } else if (enable < 0) d->d_dbg = nullptr;
    return JD_OK;
}

// Test hook for the bit-exactness claim of jd_expf: evaluates it for x[0..n) on HIP device
// `device`, or - device == -1 - its host twin compiled from the same source.
__global__ void jd_debug_expf_kernel(const float *x, long long n, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = jd_expf(x[i]);
}
extern "C" int jd_debug_expf(int32_t device, const float *x, int64_t n, float *out)
{
    if (!x || !out || n < 0) return jd_fail(JD_EINVAL, "jd_debug_expf: bad argument");
    if (device == -1) {
        for (int64_t i = 0; i < n; ++i) out[i] = jd_expf_impl(x[i], jd_exp2f_tab_host);
        return JD_OK;
    }
    int rc = check_device(device);
    if (rc) return rc;
    float *dx = nullptr, *dy = nullptr;
    HIPCHK(hipMalloc(&dx, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    HIPCHK(hipMalloc(&dy, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    HIPCHK(hipMemcpy(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(jd_debug_expf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, (long long)n, dy);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dy, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return JD_OK;
}

// Diagnostics: what part of a likelihood table does the search read?  (SURVEY.md 8d's Ug: the reference scores a tied
// state only when a token that passed the emit threshold asks for it, WFSTDecoderLite.cpp:409-411; here every state of
// every frame is scored.)  enable != 0: the slab's bitmap is cleared and every cell phase A adds to a token is marked from
// the next launch on; enable == 0: the marks are counted against the cells of the last decode's frames and marking stops.
__global__ void jd_popcount_kernel(const unsigned *w, size_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (unsigned)__popc(w[i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
extern "C" int jd_dec_debug_cells(jd_dec *d, int32_t enable, int64_t *cells_read, int64_t *cells_total)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_debug_cells: null");
    int rc = check_device(d->device);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    if (enable) {
        if (d->ll_cap == 0) return jd_fail(JD_ESTATE, "jd_dec_debug_cells: decode a batch first (the tables are sized by it)");
        const size_t words = (3 * d->ll_cap + 31) / 32 + 2;
        if (!d->d_cells || d->cells_words != words) {
            if (d->d_cells) (void)hipFree(d->d_cells);
            d->d_cells = nullptr;
            HIPCHK(hipMalloc(&d->d_cells, words * sizeof(unsigned)));
            d->cells_words = words;
        }
        HIPCHK(hipMemset(d->d_cells, 0, words * sizeof(unsigned)));
        return JD_OK;
    }
    unsigned long long n = 0;
    if (d->d_cells) {
        unsigned long long *dn = (unsigned long long *)d->d_cells;      // (the first two words: the counter)
        HIPCHK(hipMemset(dn, 0, sizeof(unsigned long long)));
        hipLaunchKernelGGL(jd_popcount_kernel, dim3(1024), dim3(256), 0, 0, d->d_cells + 2, d->cells_words - 2, dn);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(&n, dn, sizeof n, hipMemcpyDeviceToHost));
        (void)hipFree(d->d_cells);
        d->d_cells = nullptr; d->cells_words = 0;
    }
    if (cells_read) *cells_read = (int64_t)n;
    if (cells_total) *cells_total = (int64_t)d->timing.search_frames * d->am->n_gmm;
    return JD_OK;
}

extern "C" int jd_dec_info(const jd_dec *d, int32_t *max_streams, int32_t *vec_size)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_info: null");
    if (max_streams) *max_streams = d->max_streams;
    if (vec_size) *vec_size = d->am->D;
    return JD_OK;
}

extern "C" int jd_dec_last_timing(const jd_dec *d, jd_timing *out)
{
    if (!d || !out) return jd_fail(JD_EINVAL, "jd_dec_last_timing: null");
    *out = d->timing;
    return JD_OK;
}
