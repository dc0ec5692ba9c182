// jd_host_resident.h - host side of the search kernels that STAY on the device (included by jd_device.hip; not a translation unit of
// its own: it lives in jd_device.hip's TU because the kernels it launches are templates of that TU).
//
//   * Resident: the mailbox protocol of jd_resident.h / jd_slot.h (commands in host-mapped words, ready numbers counted up on the side
//     stream, reports, the host heartbeat) - jd_res_start / _stop / _post / _poll / _collect / _finish, driven by jd_broker.cpp;
//   * Pipe: batches through the resident slot kernel utterance by utterance (jd_dec_set_pipeline: JD_FLOW_RESIDENT) - pipe_announce,
//     pipe_pump, pipe_decode, jd_dec_quiesce, jd_dec_pipeline_stats.
// The launch-per-call paths (one batch, two batches in flight, re-planning, the streaming calls) stay in jd_device.hip.
#pragma once

// ------------------------------------------------------------------------------------------------------------------
// The resident search kernel (jd_resident.h) and its host side: what jd_broker.cpp drives instead of ticks.
// While it runs it owns the device's search lock (this process) and the GPU's file lock (other processes); nothing here
// allocates or frees device memory or synchronises the device - either would wait for the kernel.
#define RES_RING 256
#define RES_RING_W 2048
struct Resident {
    bool on = false;
    int n = 0, Cw = 0, rows = 0;                       // streams [0, n), workgroups per cluster, rows per likelihood buffer
    bool slot = false;                                 // one workgroup per stream: the slot kernel (jd_slot.h), SLOT_WG_PER_CU of them per CU
    ResMail *d_mail = nullptr;
    ResPost *h_post = nullptr;                         // host-mapped: the commands
    ResDone *h_done = nullptr;                         // host-mapped: the reports
    unsigned *h_beat = nullptr;                        // host-mapped: counted up whenever the host looks after the kernel (k_resident: beat)
    unsigned *d_ready = nullptr;                       // per stream: how far the side stream has come for it
    std::vector<unsigned> rid;                         // ... and the last number enqueued for it
    int *h_ring = nullptr, *d_ring = nullptr;          // row-tile lists of the scoring launches (RES_RING slots of RES_RING_W)
    int ring_turn = 0;
    float *d_feat = nullptr, *d_ll = nullptr;          // [n][2][rows] x D / x G
    int *d_src = nullptr;
    char *h_stage = nullptr;                           // pinned: the features of every buffer, [n][2][rows] x D
    std::vector<unsigned> seq;                         // last sequence number posted per stream
    std::vector<int> T_posted, T_done, err_done;
    std::vector<int> slot_posted;
    std::vector<char> busy;                            // a command is posted and its report not yet taken
    std::vector<char> init_pending;                    // ... and it begins an utterance (ResPost::init): a re-post must say so again
    long long run_ticks = 0;                           // (statistics) what the clusters spent on their commands, 100 MHz ticks
    long long n_collect = 0;                           // (statistics) Path collections between commands
    std::unique_lock<std::mutex> search_lock;
    GpuLockGuard *process_lock = nullptr;
    std::chrono::steady_clock::time_point t_start;     // when the kernel was last started (jd_dec_pipeline_stats: time on the device)
    hipStream_t st = nullptr;                          // the stream the kernel runs on (the decoder's search stream, or its CU-masked slot stream)
};

static void res_free(jd_dec *d);
static void res_free_fwd(jd_dec *d) { res_free(d); }
static void res_free(jd_dec *d)
{
    Resident *R = d->res;
    if (!R) return;
    if (R->d_mail) (void)hipFree(R->d_mail);
    if (R->h_post) (void)hipHostFree(R->h_post);
    if (R->h_done) (void)hipHostFree(R->h_done);
    if (R->h_beat) (void)hipHostFree(R->h_beat);
    if (R->d_ready) (void)hipFree(R->d_ready);
    if (R->h_ring) (void)hipHostFree(R->h_ring);
    if (R->d_ring) (void)hipFree(R->d_ring);
    if (R->d_feat) (void)hipFree(R->d_feat);
    if (R->d_ll) (void)hipFree(R->d_ll);
    if (R->d_src) (void)hipFree(R->d_src);
    if (R->h_stage) (void)hipHostFree(R->h_stage);
    delete R;
    d->res = nullptr;
}

// the report of stream s's command, if it is in
static bool res_harvest(jd_dec *d, int s)
{
    Resident *R = d->res;
    if (!R->busy[(size_t)s]) return true;
    if (__atomic_load_n(&R->h_done[s].seq, __ATOMIC_ACQUIRE) != R->seq[(size_t)s]) return false;
    if (d->pipe_on) {                                                  // (jd_dec_pipeline_stats: frames the slot has advanced)
        d->pipe_frames_searched += std::max(0, R->h_done[s].frame - R->T_done[(size_t)s]);
        d->pipe_busy_ticks += R->h_done[s].run_ticks;
    }
    R->T_done[(size_t)s] = R->h_done[s].frame; R->err_done[(size_t)s] = R->h_done[s].error;
    // (a stream that failed on the device - an arena overflow, a lost workgroup - may hold anything: wiped before its next init,
    // whether or not anybody fetches its result)
    if (R->err_done[(size_t)s] != 0) d->stream_dirty[(size_t)s] = 1;
    R->run_ticks += R->h_done[s].run_ticks;
    R->init_pending[(size_t)s] = 0;
    d->stream_T[(size_t)s] = R->T_done[(size_t)s];
    R->busy[(size_t)s] = 0;
    return true;
}

int jd_res_stop(jd_dec *d)
{
    if (!d) return JD_OK;
    std::lock_guard<std::recursive_mutex> guard(d->res_mu);            // (a finish that is being fetched goes first)
    if (!d->res || !d->res->on) return JD_OK;
    Resident *R = d->res;
    for (int s = 0; s < R->n; ++s) __atomic_store_n(&R->h_post[s].exit_req, 1, __ATOMIC_RELEASE);
    hipError_t e = hipStreamSynchronize(R->st ? R->st : d->s_search);  // (it also leaves by itself after RES_IDLE_TICKS)
    (void)hipStreamSynchronize(d->s_gmm);
    // (a cluster takes a command that is there before it looks at the exit request: whatever was posted is through)
    bool lost = false;
    for (int s = 0; s < R->n; ++s)
        if (!res_harvest(d, s)) {
            R->busy[(size_t)s] = 0;
            // a command the cluster never saw (it left by itself - idle for 5 s - just before the word was written) has not been
            // started: the stream stands where its last report says, short of what was posted, and whoever drives it posts the
            // rest again (the same way as behind a Path collection).  Anything else is a lost workgroup.
            if (!__atomic_load_n(&R->h_done[s].left, __ATOMIC_ACQUIRE)) { lost = true; d->stream_dirty[(size_t)s] = 1; }
        }
    R->on = false;
    if (d->pipe_on) d->pipe_on_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - R->t_start).count();
    delete R->process_lock; R->process_lock = nullptr;
    if (R->search_lock.owns_lock()) R->search_lock.unlock();
    if (e != hipSuccess) return jd_fail(JD_EHIP, "the resident search kernel did not end: %s", hipGetErrorString(e));
    if (lost) return jd_fail(JD_EHIP, "the resident search kernel ended with a command unanswered");
    return JD_OK;
}

// streams [0, n_streams) of the decoder, likelihood buffers of rows_per_buf rows (two per stream)
int jd_res_start(jd_dec *d, int n_streams, int rows_per_buf)
{
    if (!d || n_streams < 1 || n_streams > d->max_streams || rows_per_buf < 1) return jd_fail(JD_EINVAL, "jd_res_start: bad argument");
    std::lock_guard<std::recursive_mutex> guard(d->res_mu);
    if (d->net->lazy_dev || d->partial_interval > 0) return jd_fail(JD_ESTATE, "jd_res_start: not with a lazily composed network / partial traces");
    int rc = check_device(d->device);
    if (rc) return rc;
    rc = ensure_arenas(d);
    if (rc) return rc;
    if (!d->pipe_on) pf_discard(d);                      // (not when the batch pipeline's kernel comes back: jd_dec_quiesce)
    const int D = d->am->D, G = d->am->n_gmm;
    const int rows = (rows_per_buf + GMM_ROWS2 - 1) / GMM_ROWS2 * GMM_ROWS2;
    if (d->res && (d->res->n != n_streams || d->res->rows != rows)) { if (d->res->on) { rc = jd_res_stop(d); if (rc) return rc; } res_free(d); }
    if (!d->res) {
        Resident *R = new Resident();
        d->res = R;
        R->n = n_streams; R->rows = rows;
        const size_t tr = (size_t)n_streams * 2 * rows;
        if (n_streams > 1024 || 2 * n_streams * ((rows + GMM_ROWS2 - 1) / GMM_ROWS2) > RES_RING_W) {
            res_free(d);
            return jd_fail(JD_EINVAL, "jd_res_start: at most 1024 streams and %d row tiles per scoring launch", RES_RING_W);
        }
        if (hipMalloc(&R->d_mail, (size_t)n_streams * sizeof(ResMail)) != hipSuccess ||
            hipHostMalloc((void **)&R->h_post, (size_t)n_streams * sizeof(ResPost), hipHostMallocMapped) != hipSuccess ||
            hipHostMalloc((void **)&R->h_done, (size_t)n_streams * sizeof(ResDone), hipHostMallocMapped) != hipSuccess ||
            hipHostMalloc((void **)&R->h_beat, 64, hipHostMallocMapped) != hipSuccess ||
            hipMalloc(&R->d_ready, (size_t)n_streams * sizeof(unsigned)) != hipSuccess ||
            hipHostMalloc((void **)&R->h_ring, (size_t)RES_RING * RES_RING_W * sizeof(int)) != hipSuccess ||
            hipMalloc(&R->d_ring, (size_t)RES_RING * RES_RING_W * sizeof(int)) != hipSuccess ||
            hipMalloc(&R->d_feat, tr * D * sizeof(float)) != hipSuccess || hipMalloc(&R->d_ll, tr * G * sizeof(float)) != hipSuccess ||
            hipMalloc(&R->d_src, tr * sizeof(int)) != hipSuccess ||
            hipHostMalloc((void **)&R->h_stage, tr * D * sizeof(float)) != hipSuccess) {
            res_free(d);
            return jd_fail(JD_ENOMEM, "jd_res_start: no memory for %d streams x 2 x %d rows", n_streams, rows);
        }
        *R->h_beat = 0u;
        R->seq.assign((size_t)n_streams, 0u); R->T_posted.assign((size_t)n_streams, 0); R->T_done.assign((size_t)n_streams, 0);
        R->slot_posted.assign((size_t)n_streams, 0); R->err_done.assign((size_t)n_streams, 0); R->busy.assign((size_t)n_streams, 0); R->init_pending.assign((size_t)n_streams, 0);
        for (int t = 0; t < n_streams; ++t) R->T_done[(size_t)t] = R->T_posted[(size_t)t] = d->stream_T[(size_t)t];
        R->rid.assign((size_t)n_streams, 0u);
        std::vector<int> ident(tr);            // the row table of every scoring launch: row r of the table is row r of the features
        for (size_t r = 0; r < tr; ++r) ident[r] = (int)r;
        HIPCHK(hipMemcpy(R->d_src, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    Resident *R = d->res;
    if (R->on) return JD_OK;
    const bool ne3 = d->am->max_n <= 5;
    // clusters: what the arenas allow, and a sixth of the chip left to the scoring, collection and finish kernels
    const int cw_cap = (int)std::max<int64_t>(1, std::min<int64_t>(d->cap_slots / (64 * SW), d->cap_items / (512 * SW)));
    // (the scoring of what the streams search: about 1.6 CUs per stream at their pace, and a quarter of the chip at least -
    // sixteen C++ callers: 407 k frames/s with 24 CUs left, 433 k with 40, 469 k with 64, 462 k with 96)
    int free_cus = std::min(d->n_cus / 2, std::max(d->n_cus / 4, (n_streams * 8) / 5));
    if (const char *e = jd_dev_env("JD_RES_FREE_CUS")) { const int v = atoi(e); if (v >= 0 && v < d->n_cus) free_cus = v; }   // development
    R->Cw = std::max(1, std::min(std::min(d->max_cw, cw_cap), (d->n_cus * WG_PER_CU - free_cus) / n_streams));
    if (d->res_ll) R->Cw = 1;                                          // (the batch pipeline: every stream a slot of ONE workgroup, however few they are)
    // One workgroup per stream: the slot kernel (jd_slot.h) - compiled for four waves per SIMD, SLOT_WG_PER_CU workgroups per CU,
    // every per-frame word in LDS.  (JD_RES_SLOT=0, development: k_resident's one-workgroup clusters, one per CU.)
    R->slot = R->Cw == 1;
    if (const char *e = jd_dev_env("JD_RES_SLOT")) R->slot = R->slot && atoi(e) != 0;
    {
        int per_cu = 0;
        const void *kf = R->slot ? (ne3 ? (const void *)k_slot<3> : (const void *)k_slot<6>)
                                 : (ne3 ? (const void *)k_resident<3, false> : (const void *)k_resident<6, false>);
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kf, SNT, 0));
        const int need = R->slot ? SLOT_WG_PER_CU : WG_PER_CU;
        if (per_cu < need)
            return jd_fail(R->slot ? JD_EINVAL : JD_EHIP, "%s: %d workgroup(s) per CU fit, %d streams on %d CUs need %d", R->slot ? "k_slot" : "k_resident",
                           per_cu, n_streams, d->n_cus, need);
    }
    if (!R->slot && R->Cw * n_streams > d->n_cus * WG_PER_CU) return jd_fail(JD_EINVAL, "jd_res_start: %d streams do not fit the device", n_streams);
    // (the slot kernel's workgroups answer a mailbox: one that is never dispatched never answers - all of them resident, or none)
    if (R->slot && n_streams > d->n_cus * SLOT_WG_PER_CU)
        return jd_fail(JD_EINVAL, "jd_res_start: %d one-workgroup slots do not fit the device (%d CUs x %d workgroups of k_slot resident at once)",
                       n_streams, d->n_cus, SLOT_WG_PER_CU);
    memset(R->h_done, 0, (size_t)R->n * sizeof(ResDone));
    memset(R->h_post, 0, (size_t)R->n * sizeof(ResPost));
    std::fill(R->seq.begin(), R->seq.end(), 0u);
    std::fill(R->rid.begin(), R->rid.end(), 0u);
    {
        const size_t dev_i = (size_t)std::min(std::max(d->device, 0), JD_MAX_DEVICES - 1);
        if (d->res_yield_turn >= 0) {
            // this kernel has just made room for somebody who waits for the device (jd_res_yield): a mutex hands itself to
            // whoever asks first, which may well be the one who let go - so it asks only once the waiter has had its turn
            const auto t0 = std::chrono::steady_clock::now();
            while (g_search_turn[dev_i].load() == d->res_yield_turn && g_search_waiters[dev_i].load() > 0 &&
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 500.0)
                std::this_thread::sleep_for(std::chrono::microseconds(100));
            d->res_yield_turn = -1;
        }
        g_search_waiters[dev_i].fetch_add(1);
        R->search_lock = std::unique_lock<std::mutex>(g_search_mu[dev_i]);
        g_search_waiters[dev_i].fetch_sub(1);
        g_search_turn[dev_i].fetch_add(1);
    }
    R->process_lock = new GpuLockGuard(d->device);
    SearchArgs A;
    memset(&A, 0, sizeof A);
    A.C = d->C; A.ctl = d->d_ctl; A.streams = d->d_streams; A.work = nullptr; A.n_work = R->n; A.Cw = R->Cw; A.n_slots = 0;
    A.ll = d->res_ll ? d->res_ll : R->d_ll; A.ll_stride = (long long)G; A.f0 = 0; A.f_end = 0x7fffffff;
    A.status = d->d_status; A.dbg = d->d_dbg; A.cells = nullptr; A.resident = nullptr; A.rebalance_at = 0; A.n_prio = 0;   // (dbg: jd_dec_debug_trace)
    const dim3 rgrid((unsigned)(R->n * R->Cw));
    // (one workgroup per stream: the XCD-local flavour of the memory operations - a cluster of one sits on one XCD)
    bool xl = R->Cw == 1;
    const bool slot = R->slot;
    if (const char *e = jd_dev_env("JD_RES_XL")) xl = xl && atoi(e) != 0;   // development
    typedef void (*ResKernel)(SearchArgs, const ResPost *, ResMail *, const unsigned *, ResDone *, int, const unsigned *);
    const ResKernel rk = ne3 ? (xl ? k_resident<3, true> : k_resident<3, false>) : (xl ? k_resident<6, true> : k_resident<6, false>);
    // HIP maps streams onto a few hardware queues, and whatever is queued BEHIND a kernel that stays waits until it leaves:
    // the side stream's scoring, the null stream's copies back.  Which queue a stream gets is the runtime's business
    // (tools/resident_alias_probe.py: one fresh stream in fourteen lands behind the kernel), so the kernel is started, a
    // small kernel is sent down the side stream and the null stream, and if either has not come back in 150 ms the
    // resident kernel leaves again and comes back on a NEW search stream - a few times, then it is an error.
    // Where the slots go.  A grid of at most one workgroup per CU is dealt one per CU, and that is what the pipeline wants: a slot's eight
    // waves take two of a SIMD's four wave slots and half its registers, the scoring kernel's waves (127 VGPRs, no LDS) take the other half
    // of the SAME CU - a search that waits for memory beside arithmetic that does not: 256 slots on 256 CUs 18.1 ms per configs[1] batch,
    // against 21.5 with the slots two per CU on half the chip and the scoring on the other half (jd_slot.h: jd_park_kernel, which is how
    // such a split is made: JD_SLOT_KEEP_SE = CUs per shader engine the slots get, development), 19.1 with 272 and 21.1 with 304 slots
    // dealt over all CUs (the CUs that hold two slots have no room for the scoring).
    R->st = d->s_search;
    int park_cus = 0, park_fill = 0;
    if (slot) {
        // (whole CUs per shader engine: 32 engines of n_cus / 32 CUs each, every one keeps the same number for the slots)
        const int per_se = std::max(1, d->n_cus / 32);
        int keep_se = per_se;
        if (const char *e2 = jd_dev_env("JD_SLOT_KEEP_SE")) { const int v = atoi(e2); if (v >= 1 && v <= per_se && v * 32 * SLOT_WG_PER_CU >= R->n) keep_se = v; }
        park_cus = (per_se - keep_se) * 32;
        park_fill = std::min(R->n, keep_se * 32 * SLOT_WG_PER_CU);     // slots that find room while the others are parked
        if (park_cus > 0 && !d->h_park) {
            if (hipHostMalloc((void **)&d->h_park, 64, hipHostMallocMapped) != hipSuccess || hipMalloc(&d->d_park, 64 * sizeof(int)) != hipSuccess) {
                (void)hipGetLastError();
                park_cus = 0;
            }
        }
    }
    hipError_t e = hipSuccess;
    bool clear = false;
    ReadyList none; none.n = 0;
    struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } } ev_side, ev_null;
    HIPCHK(hipEventCreateWithFlags(&ev_side.e, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_null.e, hipEventDisableTiming));
    for (int attempt = 0; attempt < 8 && !clear; ++attempt) {
        hipLaunchKernelGGL(jd_res_reset_kernel, dim3((R->n + 63) / 64), dim3(64), 0, R->st, d->d_ctl, R->d_mail, R->d_ready, R->n);
        __atomic_fetch_add(R->h_beat, 1u, __ATOMIC_RELEASE);
        if (slot) {
            bool parked = false;
            if (park_cus > 0) {
                // the CUs of every XCD that the slots are NOT to get: parked until the slots are on theirs
                d->h_park[0] = d->h_park[1] = d->h_park[2] = d->h_park[3] = d->h_park[4] = 0u;
                if (hipMemsetAsync(d->d_park, 0, 64 * sizeof(int), d->s_gmm) == hipSuccess) {
                    hipLaunchKernelGGL(jd_park_kernel, dim3((unsigned)d->n_cus), dim3(64), 0, d->s_gmm, (unsigned *)d->d_park, park_cus / 32, d->h_park);
                    parked = hipGetLastError() == hipSuccess;
                }
                const auto tp = std::chrono::steady_clock::now();
                while (parked && __atomic_load_n(&d->h_park[0], __ATOMIC_ACQUIRE) + __atomic_load_n(&d->h_park[1], __ATOMIC_ACQUIRE) < (unsigned)d->n_cus &&
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count() < 50.0)
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            unsigned *started = parked ? d->h_park + 4 : nullptr;
            if (ne3) hipLaunchKernelGGL(k_slot<3>, rgrid, dim3(SNT), 0, R->st, A, R->h_post, R->d_ready, R->h_done, R->h_beat, started);
            else hipLaunchKernelGGL(k_slot<6>, rgrid, dim3(SNT), 0, R->st, A, R->h_post, R->d_ready, R->h_done, R->h_beat, started);
            if (parked) {                                              // every slot is on its CU (or 100 ms are over): the parked CUs are the scoring's
                const auto tp = std::chrono::steady_clock::now();
                while (__atomic_load_n(&d->h_park[4], __ATOMIC_ACQUIRE) < (unsigned)park_fill &&
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count() < 100.0)
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                if (getenv("JD_VERBOSE"))
                    fprintf(stderr, "k_slot: %u CUs parked (%u left free), %u of %d slots on their CUs after %.2f ms\n", d->h_park[0], d->h_park[1], d->h_park[4], R->n,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count());
                __atomic_store_n(&d->h_park[2], 1u, __ATOMIC_RELEASE);
            }
        } else
        hipLaunchKernelGGL(rk, rgrid, dim3(SNT), 0, R->st, A, R->h_post, R->d_mail, R->d_ready, R->h_done, R->Cw, R->h_beat);
        e = hipGetLastError();
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(jd_res_ready_kernel, dim3(1), dim3(64), 0, d->s_gmm, R->d_ready, none);
        (void)hipEventRecord(ev_side.e, d->s_gmm);
        hipLaunchKernelGGL(jd_res_ready_kernel, dim3(1), dim3(64), 0, (hipStream_t)0, R->d_ready, none);
        (void)hipEventRecord(ev_null.e, (hipStream_t)0);
        const auto t0 = std::chrono::steady_clock::now();
        while (!clear && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 150.0) {
            clear = hipEventQuery(ev_side.e) == hipSuccess && hipEventQuery(ev_null.e) == hipSuccess;
            if (!clear) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (clear) break;
        // behind the kernel: it leaves (the exit word), what waited for it runs, and the search stream is made anew
        for (int t = 0; t < R->n; ++t) __atomic_store_n(&R->h_post[t].exit_req, 1, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(R->st);
        (void)hipEventSynchronize(ev_side.e); (void)hipEventSynchronize(ev_null.e);
        memset(R->h_post, 0, (size_t)R->n * sizeof(ResPost));
        memset(R->h_done, 0, (size_t)R->n * sizeof(ResDone));
        hipStream_t fresh = nullptr;
        {
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            if (hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio_hi) != hipSuccess) { e = hipErrorUnknown; break; }
            d->res_old_streams.push_back(d->s_search);                 // (destroyed with the decoder: somebody may still hold it)
            d->s_search = fresh; R->st = fresh;
        }
        if (getenv("JD_VERBOSE")) fprintf(stderr, "k_resident: the side stream or the null stream was queued behind it - a new search stream (%d)\n", attempt + 1);
    }
    if (e != hipSuccess || !clear) {
        delete R->process_lock; R->process_lock = nullptr; R->search_lock.unlock();
        if (e != hipSuccess) return jd_fail(JD_EHIP, "k_resident: %s", hipGetErrorString(e));
        return jd_fail(JD_EHIP, "k_resident: no search stream whose hardware queue the side stream and the null stream do not share");
    }
    R->on = true;
    R->t_start = std::chrono::steady_clock::now();
    if (getenv("JD_VERBOSE")) fprintf(stderr, "%s: %d streams, clusters of %d workgroups, %d rows per buffer%s\n", R->slot ? "k_slot" : "k_resident", R->n, R->Cw, R->rows,
                                      park_cus > 0 ? " (the other CUs parked while its grid was dealt)" : "");
    return JD_OK;
}

int jd_res_cluster(const jd_dec *d) { return (d && d->res) ? d->res->Cw : 0; }
// the kernel leaves for somebody who waits for the device, and comes back behind them (jd_res_start)
int jd_res_yield(jd_dec *d)
{
    if (!d || !d->res || !d->res->on) return JD_OK;
    d->res_yield_turn = g_search_turn[(size_t)std::min(std::max(d->device, 0), JD_MAX_DEVICES - 1)].load();
    return jd_res_stop(d);
}
// somebody else of this process waits for the device's search lock (another decoder's launch, another broker's kernel)
int jd_res_should_yield(const jd_dec *d)
{
    return (d && d->res && d->res->on) ? g_search_waiters[(size_t)std::min(std::max(d->device, 0), JD_MAX_DEVICES - 1)].load() > 0 : 0;
}
long long jd_res_run_us(const jd_dec *d) { return (d && d->res) ? d->res->run_ticks / 100 : 0; }
long long jd_res_collections(const jd_dec *d) { return (d && d->res) ? d->res->n_collect : 0; }

// "the side stream has come this far" for these streams: a new ready number each, behind everything enqueued so far
static int res_bump(jd_dec *d, int n, const int *streams)
{
    Resident *R = d->res;
    for (int i0 = 0; i0 < n; i0 += 64) {
        ReadyList L;
        L.n = std::min(64, n - i0);
        for (int i = 0; i < L.n; ++i) { const int s = streams[i0 + i]; R->rid[(size_t)s] += 1; L.s[i] = s; L.id[i] = R->rid[(size_t)s]; }
        hipLaunchKernelGGL(jd_res_ready_kernel, dim3(1), dim3(64), 0, d->s_gmm, R->d_ready, L);
        HIPCHK(hipGetLastError());
    }
    return JD_OK;
}

// IDecoder::init of stream s (idle): recognitionStart runs with the stream's next command
int jd_res_init(jd_dec *d, int s)
{
    std::lock_guard<std::recursive_mutex> guard(d->res_mu);            // (the wipe of a failed stream stops and starts the kernel)
    Resident *R = d->res;
    if (!R || !R->on || s < 0 || s >= R->n) return jd_fail(JD_ESTATE, "jd_res_init: no resident kernel for stream %d", s);
    if (d->stream_dirty[(size_t)s]) {                                  // (after an error: the wipe synchronises the device)
        int rc = jd_res_stop(d);
        if (rc) return rc;
        rc = wipe_stream(d, s);
        if (rc) return rc;
        rc = jd_res_start(d, R->n, R->rows);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(jd_mark_init_kernel, dim3(1), dim3(64), 0, d->s_gmm, d->d_ctl, s, 1);
    HIPCHK(hipGetLastError());
    d->stream_T[(size_t)s] = 0; d->stream_started[(size_t)s] = 1;
    R->T_posted[(size_t)s] = 0; R->T_done[(size_t)s] = 0; R->err_done[(size_t)s] = 0;
    return res_bump(d, 1, &s);
}

// Frames of n streams into their likelihood buffers bufs[i] (0 / 1, free): an upload per stream from the buffer's own
// pinned staging region, ONE scoring launch over the row tiles concerned (asynchronous, on the side stream), and the
// streams' ready numbers behind it
int jd_res_stage_many(jd_dec *d, int n, const int *streams, const int *bufs, const float *const *frames, const int *n_frames)
{
    Resident *R = d->res;
    if (!R || !R->on) return jd_fail(JD_ESTATE, "jd_res_stage_many: no resident kernel");
    const int D = d->am->D;
    const size_t tr = (size_t)R->n * 2 * R->rows;
    int *list = R->h_ring + (size_t)R->ring_turn * RES_RING_W;
    int *d_list = R->d_ring + (size_t)R->ring_turn * RES_RING_W;
    int nt = 0, ns = 0;
    std::vector<int> who;
    for (int i = 0; i < n; ++i) {
        const int s = streams[i], buf = bufs[i], nf = n_frames[i];
        if (s < 0 || s >= R->n || (buf != 0 && buf != 1) || nf < 0 || nf > R->rows || (nf > 0 && !frames[i]))
            return jd_fail(JD_EINVAL, "jd_res_stage_many: bad argument");
        if (nf == 0) continue;
        const size_t r0 = ((size_t)s * 2 + (size_t)buf) * (size_t)R->rows;
        float *hf = (float *)R->h_stage + r0 * D;
        memcpy(hf, frames[i], (size_t)nf * D * sizeof(float));
        HIPCHK(hipMemcpyAsync(R->d_feat + r0 * D, hf, (size_t)nf * D * sizeof(float), hipMemcpyHostToDevice, d->s_gmm));
        for (int t = 0; t < (nf + GMM_ROWS2 - 1) / GMM_ROWS2; ++t) list[nt++] = (int)r0 + t * GMM_ROWS2;
        who.push_back(s);
        ++ns;
    }
    if (ns == 0) return JD_OK;
    R->ring_turn = (R->ring_turn + 1) % RES_RING;
    HIPCHK(hipMemcpyAsync(d_list, list, (size_t)nt * sizeof(int), hipMemcpyHostToDevice, d->s_gmm));
    // (a tile's rows behind the chunk's last frame are scored too - whatever the buffer holds there - and read by nobody)
    const int rc = launch_gmm(d->am, d->amb, R->d_feat, R->d_src, (int)tr, R->d_ll, d->s_gmm, 0, 0, nt, d_list, nt);
    if (rc) return rc;
    return res_bump(d, ns, who.data());
}

// the command itself: a word in host-mapped memory (the cluster's first workgroup polls it)
static void res_write_post(Resident *R, int s, int T, int slot, int init = 0)
{
    __atomic_fetch_add(R->h_beat, 1u, __ATOMIC_RELAXED);               // (a sign of life: k_resident's `beat`)
    ResPost &P = R->h_post[s];
    P.T = T;
    P.init = init;
    if (init) R->init_pending[(size_t)s] = 1;
    P.ready_id = R->rid[(size_t)s];
    __atomic_store_n(&P.word, ((unsigned long long)R->seq[(size_t)s] << 32) | (unsigned)slot, __ATOMIC_RELEASE);
}

// The command "frames up to T + n_frames are scored in buffer buf" for stream s (idle), behind what has been staged
int jd_res_post(jd_dec *d, int s, int buf, int n_frames)
{
    Resident *R = d->res;
    if (!R || !R->on || s < 0 || s >= R->n) return jd_fail(JD_ESTATE, "jd_res_post: no resident kernel for stream %d", s);
    const int T0 = R->T_done[(size_t)s], T1 = T0 + n_frames;
    const long long slot = ((long long)s * 2 + buf) * R->rows - T0;    // (k_search reads row  slot + f: see jd_streams_push)
    R->seq[(size_t)s] += 1;
    R->busy[(size_t)s] = 1;
    R->T_posted[(size_t)s] = T1; R->slot_posted[(size_t)s] = (int)slot;
    res_write_post(R, s, T1, (int)slot);
    return JD_OK;
}

// Where stream s stands: *idle = its last command is through (then *frame = frames processed, *error = StreamCtl::error,
// *stopped = it stopped short of what was posted - a Path collection is due: jd_res_collect)
int jd_res_poll(jd_dec *d, int s, int *idle, int *frame, int *error, int *stopped)
{
    Resident *R = d->res;
    if (!R || !R->on || s < 0 || s >= R->n) return jd_fail(JD_ESTATE, "jd_res_poll: no resident kernel for stream %d", s);
    __atomic_fetch_add(R->h_beat, 1u, __ATOMIC_RELAXED);               // (a sign of life: k_resident's `beat`)
    const bool through = res_harvest(d, s);
    *idle = through ? 1 : 0;
    if (frame) *frame = R->T_done[(size_t)s];
    if (error) *error = R->err_done[(size_t)s];
    if (stopped) *stopped = (through && R->err_done[(size_t)s] == 0 && R->T_done[(size_t)s] < R->T_posted[(size_t)s]) ? 1 : 0;
    if (through) return JD_OK;
    if (__atomic_load_n(&R->h_done[s].left, __ATOMIC_ACQUIRE))
        return jd_fail(JD_ESTATE, "the resident search kernel has ended (no command for 5 s, or a lost workgroup)");
    return JD_OK;
}

// the device-side error a poll reported for stream s, as the library's code (jd_last_error() has the text)
int jd_res_stream_error(jd_dec *d, int s, int dev_error, int frame)
{
    return report_stream_error(d, s, dev_error, frame, 1);
}

// collectPaths for stream s (idle, stopped), then the rest of its command again
int jd_res_collect(jd_dec *d, int s)
{
    Resident *R = d->res;
    if (!R || !R->on || s < 0 || s >= R->n) return jd_fail(JD_ESTATE, "jd_res_collect: no resident kernel for stream %d", s);
    launch_gc(d->C, d->d_ctl, d->d_streams, nullptr, 1, s, d->am->max_n <= 5, std::max(8, d->n_cus / 6), d->s_gmm);
    HIPCHK(hipGetLastError());
    R->seq[(size_t)s] += 1;
    R->busy[(size_t)s] = 1;
    R->n_collect += 1;
    if (d->pipe_on) d->pipe_collections += 1;
    const int rc = res_bump(d, 1, &s);                                 // (the command waits for the collection)
    if (rc) return rc;
    res_write_post(R, s, R->T_posted[(size_t)s], R->slot_posted[(size_t)s], R->init_pending[(size_t)s]);
    return JD_OK;
}

// IDecoder::finish of stream s (idle, every frame it was given processed)
int jd_res_finish(jd_dec *d, int s, jd_hyp *out)
{
    // (the broker's finisher thread, beside its worker: the kernel is neither stopped nor started while a result is fetched -
    // a stop wipes device state and synchronises the device - and this thread's launches go to the decoder's device)
    std::lock_guard<std::recursive_mutex> guard(d->res_mu);
    int rc0 = check_device(d->device);
    if (rc0) return rc0;
    Resident *R = d->res;
    if (!R || !R->on || s < 0 || s >= R->n || !out) return jd_fail(JD_ESTATE, "jd_res_finish: no resident kernel for stream %d", s);
    hipLaunchKernelGGL(jd_finish_kernel, dim3(1), dim3(64), 0, d->s_gmm, d->d_ctl, d->d_streams, s, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->s_gmm));
    std::vector<jd_hyp> tmp((size_t)d->max_streams);
    const int rc = fetch_results(d, s, 1, tmp.data(), s);
    *out = tmp[(size_t)s];
    return rc;
}


// ------------------------------------------------------------------------------------------------------------------
// Batches through the resident kernel, UTTERANCE BY UTTERANCE (JD_PIPELINE=3).  A batch lasts as long as its longest
// utterance; with two batches in flight (above) the streams of a bank still wait for their bank to be handed back.  Here
// every stream is a slot of the resident kernel with ONE workgroup: announced batches (jd_dec_prefetch_scores, up to
// pipe_depth of them) are scored whole into a table of their own, their utterances queue up, and a slot whose utterance is
// through takes the next one at once - its result exported to a virtual result slot first (jd_finish_export_kernel) - so
// that no workgroup waits for anybody.  jd_decode_batch_device of the OLDEST announced batch waits until its utterances are
// through and hands them back; it is also what keeps the slots fed (the pump runs inside the calls - no thread).  Results
// are those of any other path; what changes is that a batch takes as long as its longest utterance on one workgroup.
struct PipeUtt { int state = 0, slot = -1, T = 0; long long row0 = 0; };      // state: 0 queued, 1 running, 2 through
#define PIPE_CHUNK 128                  // frames per command: what a slot runs before it looks at its mailbox again (jd_dec_quiesce waits that long)
struct PipeBatch {
    const float *feats = nullptr; int n = 0; int table = 0; int next = 0, n_done = 0;
    size_t rows = 0, rows_scored = 0;                  // rows of its table, and how many of them have a scoring launch enqueued
    std::vector<int64_t> offs;
    std::vector<PipeUtt> u;
    std::vector<int> order;                            // its utterances by length, longest first: the order in which slots take them
};
struct Pipe {
    bool on = false;
    int K = 0, max_batch = 0, n_slots = 0;
    size_t table_rows = 0;
    float *d_ll = nullptr;                             // K tables
    int *d_ident = nullptr;                            // row r is frame r of the batch's features
    StreamCtl *d_vctl = nullptr; int *d_vresn = nullptr, *d_vres = nullptr;   // K x max_batch virtual result slots
    std::deque<PipeBatch> q;
    std::vector<char> table_used;
    std::vector<int> slot_batch_id, slot_utt;          // per slot: the batch (its serial number) and utterance it runs, -1: free
    std::vector<char> slot_dirty;
    long long serial0 = 0;                             // serial number of q.front()
    int chunk = PIPE_CHUNK;
    std::chrono::steady_clock::time_point t_on;        // (statistics)
    long long frames_done = 0;
    hipEvent_t ev_piece = nullptr;                     // behind the last scoring launch enqueued
    bool piece_out = false;
    size_t piece_rows = 6144;                          // rows per scoring launch (JD_PIPE_PIECE)
};

static void pipe_free(jd_dec *d);
static void pipe_free_fwd(jd_dec *d) { pipe_free(d); }
static void pipe_free(jd_dec *d)
{
    Pipe *P = d->pipe;
    if (!P) return;
    if (P->d_ll) (void)hipFree(P->d_ll);
    if (P->d_ident) (void)hipFree(P->d_ident);
    if (P->d_vctl) (void)hipFree(P->d_vctl);
    if (P->d_vresn) (void)hipFree(P->d_vresn);
    if (P->d_vres) (void)hipFree(P->d_vres);
    if (P->ev_piece) (void)hipEventDestroy(P->ev_piece);
    delete P;
    d->pipe = nullptr;
}

// everything in flight is dropped (the batches concerned are decoded from scratch when their turn comes), the kernel leaves
static void pipe_drain(jd_dec *d)
{
    Pipe *P = d->pipe;
    if (!P || !P->on) return;
    (void)jd_res_stop(d);                                              // (running utterances run out first)
    if (getenv("JD_VERBOSE") && d->res) {                              // development: how busy the slots were
        const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - P->t_on).count();
        fprintf(stderr, "pipeline: %d slots for %.1f ms, busy %.1f %% of it on %lld frames (%.1f us per frame on the slot's clock)\n", P->n_slots,
                wall_us / 1e3, 100.0 * (double)(d->res->run_ticks / 100) / (wall_us * P->n_slots), P->frames_done,
                P->frames_done ? (double)(d->res->run_ticks / 100) / (double)P->frames_done : 0.0);
    }
    P->on = false; d->pipe_on = false; P->piece_out = false;
    P->q.clear();
    std::fill(P->table_used.begin(), P->table_used.end(), 0);
    std::fill(P->slot_batch_id.begin(), P->slot_batch_id.end(), -1);
    d->res_ll = nullptr;
    for (int s = 0; s < P->n_slots; ++s) {                             // (streams left in the middle of an utterance, or failed)
        if (P->slot_dirty[(size_t)s]) d->stream_dirty[(size_t)s] = 1;
        P->slot_dirty[(size_t)s] = 0;
    }
}

// slots whose utterance is through -> their results exported, the slots free; free slots -> the next queued utterances
static int pipe_pump(jd_dec *d)
{
    Pipe *P = d->pipe;
    Resident *R = d->res;
    ExportList EL; EL.n = 0;
    auto flush_exports = [&]() -> int {
        if (EL.n == 0) return JD_OK;
        hipLaunchKernelGGL(jd_finish_export_kernel, dim3((unsigned)EL.n), dim3(64), 0, d->s_gmm, d->d_ctl, d->d_streams, EL, P->d_vctl, P->d_vresn,
                           P->d_vres, d->res_cap);
        HIPCHK(hipGetLastError());
        EL.n = 0;
        return JD_OK;
    };
    if (R->h_beat) __atomic_fetch_add(R->h_beat, 1u, __ATOMIC_RELAXED);   // (a sign of life: k_resident's `beat`)
    if (R->on) {
        // the kernel has gone by itself: nobody gave it a command for 5 s (a caller that was away between two calls) - seen
        // BEFORE anything is posted to it: the reports are all in, and it comes back like behind jd_dec_quiesce
        bool left = false;
        for (int s = 0; s < P->n_slots && !left; ++s) left = __atomic_load_n(&R->h_done[s].left, __ATOMIC_ACQUIRE) != 0;
        if (left) { const int rc = jd_res_stop(d); if (rc) return rc; }
    }
    if (!R->on) {                                                      // (after jd_dec_quiesce: the kernel comes back, the slots go on where they were)
        const int rc = jd_res_start(d, P->n_slots, GMM_ROWS2);
        if (rc) return rc;
    }
    for (int s = 0; s < P->n_slots; ++s) {
        if (P->slot_batch_id[(size_t)s] < 0) continue;
        if (R->busy[(size_t)s] && !res_harvest(d, s)) continue;
        const int er = R->err_done[(size_t)s], fr = R->T_done[(size_t)s];
        PipeBatch &B = P->q[(size_t)(P->slot_batch_id[(size_t)s] - P->serial0)];
        const int ui = P->slot_utt[(size_t)s];
        const PipeUtt &U = B.u[(size_t)ui];
        if (er == 0 && fr < R->T_posted[(size_t)s]) {                  // stopped for a Path collection: collect, go on
            const int rc = jd_res_collect(d, s);
            if (rc) return rc;
            continue;
        }
        if (er == 0 && fr < U.T) {                                     // its next frames
            R->seq[(size_t)s] += 1; R->busy[(size_t)s] = 1;
            R->T_posted[(size_t)s] = std::min(U.T, fr + P->chunk);
            res_write_post(R, s, R->T_posted[(size_t)s], (int)U.row0, 0);
            continue;
        }
        EL.slot[EL.n] = s; EL.vslot[EL.n] = B.table * P->max_batch + ui; EL.n += 1;
        if (EL.n == 64) { const int rc = flush_exports(); if (rc) return rc; }
        B.u[(size_t)ui].state = 2; B.n_done += 1; P->frames_done += fr; d->pipe_utts_through += 1;
        if (er) P->slot_dirty[(size_t)s] = 1;                          // (its arenas may be inconsistent: out of the game until the pipeline stops)
        P->slot_batch_id[(size_t)s] = -1;
    }
    int rc = flush_exports();
    if (rc) return rc;
    // scoring, a piece at a time: a batch's table in ONE launch holds the side stream for ~20 ms, and the exports and ready
    // numbers of every slot that finishes meanwhile queue up behind it (measured: slots 12 % idle); the next piece goes out when
    // the one before it is through, so that those wait for a piece at most
    if (P->piece_out && hipEventQuery(P->ev_piece) == hipSuccess) P->piece_out = false;
    if (!P->piece_out)
        for (PipeBatch &B : P->q) {
            if (B.rows_scored >= B.rows) continue;
            const size_t n = std::min(P->piece_rows, B.rows - B.rows_scored);
            const size_t base = (size_t)B.table * P->table_rows + B.rows_scored;
            rc = launch_gmm(d->am, d->amb, B.feats + ((size_t)B.offs[0] + B.rows_scored) * (size_t)d->am->D, P->d_ident, (int)n,
                            P->d_ll + base * (size_t)d->am->n_gmm, d->s_gmm);
            if (rc) return rc;
            HIPCHK(hipEventRecord(P->ev_piece, d->s_gmm));
            B.rows_scored += n; P->piece_out = true; d->pipe_rows_scored += (long long)n;
            break;
        }
    // refill (from batches whose scoring is enqueued to the last row: a slot's ready number is counted up behind it)
    std::vector<int> who;
    std::vector<std::pair<int, int>> what;                             // (batch index in q, utterance)
    size_t bi = 0;
    for (int s = 0; s < P->n_slots; ++s) {
        if (P->slot_batch_id[(size_t)s] >= 0 || P->slot_dirty[(size_t)s]) continue;
        while (bi < P->q.size() && P->q[bi].next >= P->q[bi].n) ++bi;
        if (bi >= P->q.size() || P->q[bi].rows_scored < P->q[bi].rows) break;
        PipeBatch &B = P->q[bi];
        const int ui = B.order[(size_t)B.next++];                      // (longest first: a batch is handed back when its LAST utterance is through)
        B.u[(size_t)ui].state = 1; B.u[(size_t)ui].slot = s;
        P->slot_batch_id[(size_t)s] = (int)(P->serial0 + (long long)bi); P->slot_utt[(size_t)s] = ui;
        who.push_back(s); what.push_back(std::make_pair((int)bi, ui));
    }
    if (who.empty()) return JD_OK;
    rc = res_bump(d, (int)who.size(), who.data());                    // (behind the exports and every scoring launch enqueued so far)
    if (rc) return rc;
    for (size_t k = 0; k < who.size(); ++k) {
        const int s = who[k];
        const PipeUtt &U = P->q[(size_t)what[k].first].u[(size_t)what[k].second];
        R->seq[(size_t)s] += 1; R->busy[(size_t)s] = 1;
        R->T_done[(size_t)s] = 0; R->err_done[(size_t)s] = 0;
        R->T_posted[(size_t)s] = std::min(U.T, P->chunk); R->slot_posted[(size_t)s] = (int)U.row0;
        res_write_post(R, s, R->T_posted[(size_t)s], (int)U.row0, 1);
    }
    return JD_OK;
}

// The decoder's work on the device comes to rest: a search kernel of its own that stays on the device (the batch pipeline)
// lets the commands that are running run out (PIPE_CHUNK frames at most) and leaves; nothing that is announced or under
// way is lost - the kernel comes back with the next call and the slots go on where they were.  What a caller needs before
// a device-wide synchronisation (hipDeviceSynchronize, torch.cuda.synchronize) while batches are announced.
extern "C" int jd_dec_quiesce(jd_dec *d)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_quiesce: null");
    if (d->pipe && d->pipe->on && d->res && d->res->on) {
        int rc = check_device(d->device);
        if (rc) return rc;
        rc = jd_res_stop(d);
        if (rc) return rc;
    }
    return JD_OK;
}

// How batches that follow each other share the chip (include/juicer_amd.h).  Whatever is announced or under way under the old
// setting is dropped: results never depend on announcements, the batches concerned are decoded from scratch when their turn comes.
extern "C" int jd_dec_set_pipeline(jd_dec *d, int32_t mode, int32_t depth, int32_t slots)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_set_pipeline: null");
    if (mode != JD_FLOW_SERIAL && mode != JD_FLOW_TWO_IN_FLIGHT && mode != JD_FLOW_RESIDENT)
        return jd_fail(JD_EINVAL, "jd_dec_set_pipeline: mode %d (JD_FLOW_SERIAL, JD_FLOW_TWO_IN_FLIGHT or JD_FLOW_RESIDENT)", mode);
    if (mode == JD_FLOW_RESIDENT) {
        if (slots == 0) slots = d->max_streams;
        // (batches in the slots + the one being handed back + what is being scored: 256 slots and batches of 64 need ten - a batch
        // is handed back when its longest utterance is through, ~150 ms after it was taken up at 130 us per frame)
        if (depth == 0) depth = std::min(32, std::max(8, slots / 32 + 2));
        if (depth < 2 || depth > 32) return jd_fail(JD_EINVAL, "jd_dec_set_pipeline: depth %d (2..32 batches announced and not handed back)", depth);
        if (slots < 1 || slots > d->max_streams) return jd_fail(JD_EINVAL, "jd_dec_set_pipeline: %d slots, the decoder has %d streams", slots, d->max_streams);
        // the slots are workgroups of a mailbox kernel: ALL of them have to be on the device at once (SLOT_WG_PER_CU per CU at most) -
        // a workgroup that never gets a CU never answers, and the batch whose utterance was posted to it never comes back
        if (slots > d->n_cus * SLOT_WG_PER_CU)
            return jd_fail(JD_EINVAL, "jd_dec_set_pipeline: %d slots, but the device holds %d at once (%d CUs x %d workgroups of the slot kernel); "
                           "one per CU (%d) is what leaves the scoring room beside them", slots, d->n_cus * SLOT_WG_PER_CU, d->n_cus, SLOT_WG_PER_CU, d->n_cus);
        if (d->net->lazy_dev || d->am->hybrid)
            return jd_fail(JD_ESTATE, "jd_dec_set_pipeline: JD_FLOW_RESIDENT not with a lazily composed network / hybrid scoring");
    }
    if (d->res && d->res->on && !d->pipe_on) return jd_fail(JD_ESTATE, "jd_dec_set_pipeline: a broker drives this decoder's resident kernel");
    int rc = check_device(d->device);
    if (rc) return rc;
    pipe_drain(d);
    pipe_free(d);
    pf_discard(d);
    d->pipeline = mode != JD_FLOW_SERIAL;
    d->pipe_mode = mode == JD_FLOW_RESIDENT;
    if (d->pipe_mode) { d->pipe_depth = depth; d->pipe_slots = slots; }
    return JD_OK;
}

// How the likelihood tables are scored (include/juicer_amd.h).  Whatever was scored or announced under the other setting is dropped.
extern "C" int jd_dec_set_scoring(jd_dec *d, int32_t mode)
{
    if (!d) return jd_fail(JD_EINVAL, "jd_dec_set_scoring: null");
    if (mode != JD_SCORE_EXACT && mode != JD_SCORE_FAST) return jd_fail(JD_EINVAL, "jd_dec_set_scoring: mode %d (JD_SCORE_EXACT or JD_SCORE_FAST)", mode);
    if (mode == JD_SCORE_FAST && (d->am->D != 39 || d->am->hybrid))
        return jd_fail(JD_EINVAL, "jd_dec_set_scoring: JD_SCORE_FAST serves 39-dimensional GMM models (this decoder's: D = %d%s)", d->am->D,
                       d->am->hybrid ? ", hybrid" : "");
    if (d->res && d->res->on && !d->pipe_on) return jd_fail(JD_ESTATE, "jd_dec_set_scoring: a broker drives this decoder's resident kernel");
    int rc = check_device(d->device);
    if (rc) return rc;
    if ((d->amb.fast != 0) == (mode == JD_SCORE_FAST)) return JD_OK;
    pipe_drain(d);
    pf_discard(d);
    if (mode == JD_SCORE_FAST) { rc = upload_am_fast(d->am, d->amb); if (rc) return rc; }
    d->amb.fast = mode == JD_SCORE_FAST ? 1 : 0;
    return JD_OK;
}

extern "C" int jd_dec_pipeline_stats(const jd_dec *d, jd_pipe_stats *out)
{
    if (!d || !out) return jd_fail(JD_EINVAL, "jd_dec_pipeline_stats: null");
    memset(out, 0, sizeof *out);
    out->mode = d->pipe_mode ? JD_FLOW_RESIDENT : (d->pipeline ? JD_FLOW_TWO_IN_FLIGHT : JD_FLOW_SERIAL);
    out->depth = d->pipe_mode ? d->pipe_depth : 0;
    out->slots = d->pipe_mode ? (d->pipe ? d->pipe->n_slots : (d->pipe_slots > 0 ? d->pipe_slots : d->max_streams)) : 0;
    out->resident = (d->pipe_on && d->res && d->res->on) ? 1 : 0;
    out->batches_announced = d->pipe ? (int32_t)d->pipe->q.size() : 0;
    out->frames_searched = d->pipe_frames_searched; out->utts_through = d->pipe_utts_through; out->rows_scored = d->pipe_rows_scored;
    out->batches_back = d->pipe_batches_back; out->collections = d->pipe_collections;
    out->slot_busy_us = (double)d->pipe_busy_ticks / 100.0;
    out->on_us = d->pipe_on_us;
    if (out->resident) out->on_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - d->res->t_start).count();
    return JD_OK;
}

// jd_dec_prefetch_scores in pipe mode: 1 = taken, 0 = not this way (the caller goes on with the usual announcement)
static int pipe_announce(jd_dec *d, int n_utts, const float *d_feats, const int64_t *offs, int *taken)
{
    *taken = 0;
    if (!d->pipe_mode || d->net->lazy_dev || d->partial_interval > 0 || d->am->hybrid || n_utts < 1) return JD_OK;
    const int G = d->am->n_gmm;
    const size_t rows = (size_t)(offs[n_utts] - offs[0]);
    Pipe *P = d->pipe;
    if (P && (n_utts > P->max_batch || rows > P->table_rows)) {        // a larger batch than the tables were made for: not this way
        pipe_drain(d);
        pipe_free(d);
        P = nullptr;
    }
    int rc = check_device(d->device);
    if (rc) return rc;
    if (!P) {
        rc = ensure_arenas(d);
        if (rc) return rc;
        if (d->res && d->res->on) return JD_OK;                        // (a broker owns the resident kernel)
        P = new Pipe();
        d->pipe = P;
        // (tables and result slots for batches up to twice this one: a larger one later starts the pipeline again, with larger ones)
        P->K = d->pipe_depth; P->max_batch = 2 * n_utts;
        P->n_slots = (d->pipe_slots > 0 && d->pipe_slots <= d->max_streams) ? d->pipe_slots : d->max_streams;
        if (const char *e = jd_dev_env("JD_PIPE_CHUNK")) { const int v = atoi(e); if (v >= 16) P->chunk = v; }   // development
        if (const char *e = jd_dev_env("JD_PIPE_PIECE")) { const int v = atoi(e); if (v >= 128) P->piece_rows = (size_t)v / GMM_ROWS2 * GMM_ROWS2; }
        if (hipEventCreateWithFlags(&P->ev_piece, hipEventDisableTiming) != hipSuccess) { pipe_free(d); return jd_fail(JD_EHIP, "hipEventCreate failed"); }
        P->table_rows = ((2 * rows + 1024) + GMM_ROWS2 - 1) / GMM_ROWS2 * GMM_ROWS2;
        const size_t V = (size_t)P->K * P->max_batch;
        if (hipMalloc(&P->d_ll, (size_t)P->K * P->table_rows * G * sizeof(float)) != hipSuccess ||
            hipMalloc(&P->d_ident, P->table_rows * sizeof(int)) != hipSuccess ||
            hipMalloc(&P->d_vctl, V * sizeof(StreamCtl)) != hipSuccess || hipMalloc(&P->d_vresn, V * sizeof(int)) != hipSuccess ||
            hipMalloc(&P->d_vres, V * 5 * (size_t)d->res_cap * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            pipe_free(d);
            return jd_fail(JD_ENOMEM, "jd_dec_prefetch_scores: no memory for %d likelihood tables of %zu rows", d->pipe_depth, rows);
        }
        std::vector<int> ident(P->table_rows);
        for (size_t r = 0; r < P->table_rows; ++r) ident[r] = (int)r;
        HIPCHK(hipMemcpy(P->d_ident, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
        P->table_used.assign((size_t)P->K, 0);
        P->slot_batch_id.assign((size_t)P->n_slots, -1); P->slot_utt.assign((size_t)P->n_slots, -1); P->slot_dirty.assign((size_t)P->n_slots, 0);
    }
    if ((int)P->q.size() >= P->K)
        return jd_fail(JD_ESTATE, "jd_dec_prefetch_scores: %d batches are announced and not decoded - the pipeline is %d deep (JD_PIPE_DEPTH)",
                       (int)P->q.size(), P->K);
    if (!P->on) {
        pf_discard(d);                                                 // (what the other way of working ahead holds)
        for (int s = 0; s < P->n_slots; ++s)
            if (d->stream_dirty[(size_t)s]) { rc = wipe_stream(d, s); if (rc) return rc; }
        d->res_ll = P->d_ll;
        rc = jd_res_start(d, P->n_slots, GMM_ROWS2);
        if (rc) { d->res_ll = nullptr; return rc; }
        P->on = true; d->pipe_on = true;
        P->serial0 = 0; P->t_on = std::chrono::steady_clock::now(); P->frames_done = 0; d->res->run_ticks = 0;
    }
    PipeBatch B;
    B.feats = d_feats; B.n = n_utts; B.offs.assign(offs, offs + n_utts + 1);
    int t = 0;
    while (t < P->K && P->table_used[(size_t)t]) ++t;
    B.table = t; P->table_used[(size_t)t] = 1;
    B.u.resize((size_t)n_utts);
    const long long base = (long long)t * (long long)P->table_rows;
    for (int u = 0; u < n_utts; ++u) { B.u[(size_t)u].T = (int)(offs[u + 1] - offs[u]); B.u[(size_t)u].row0 = base + (offs[u] - offs[0]); }
    B.order.resize((size_t)n_utts);
    std::iota(B.order.begin(), B.order.end(), 0);
    std::stable_sort(B.order.begin(), B.order.end(), [&](int a, int b) { return B.u[(size_t)a].T > B.u[(size_t)b].T; });
    B.rows = rows; B.rows_scored = 0;                                  // (scored by the pump, a piece at a time, on the CUs the slots leave)
    P->q.push_back(std::move(B));
    *taken = 1;
    return pipe_pump(d);
}

// jd_decode_batch_device in pipe mode: 1 = handled (the oldest announced batch, handed back), 0 = not this way
static int pipe_decode(jd_dec *d, int n_utts, const float *d_feats, const int64_t *offs, jd_hyp *out, int *handled)
{
    *handled = 0;
    Pipe *P = d->pipe;
    if (!P || !P->on || P->q.empty()) return JD_OK;
    {
        const PipeBatch &F = P->q.front();
        bool same = F.feats == d_feats && F.n == n_utts;
        for (int u = 0; same && u <= n_utts; ++u) same = F.offs[(size_t)u] == offs[u];
        if (!same) { pipe_drain(d); return JD_OK; }                    // not the announced one: as if nothing had been announced
    }
    const auto w0 = std::chrono::steady_clock::now();
    int restarts = 0;
    long long seen_frames = -1;
    auto t_progress = w0;
    for (;;) {
        const int rc = pipe_pump(d);
        if (rc) { pipe_drain(d); return rc; }
        if (P->q.front().n_done == P->q.front().n) break;
        {   // (no utterance through for 30 s: something is stuck - better an error, and the other paths, than a caller that waits for ever)
            const auto now = std::chrono::steady_clock::now();
            if (P->frames_done != seen_frames) { seen_frames = P->frames_done; t_progress = now; }
            else if (std::chrono::duration<double>(now - t_progress).count() > 30.0) {
                pipe_drain(d);
                return jd_fail(JD_EHIP, "the batch pipeline has not finished an utterance for 30 s");
            }
        }
        bool left = false;
        for (int s = 0; s < P->n_slots && d->res->on && !left; ++s) left = __atomic_load_n(&d->res->h_done[s].left, __ATOMIC_ACQUIRE) != 0;
        if (left) {
            // the kernel has gone by itself: nobody gave it a command for 5 s (a caller that was away between two calls) - the
            // reports are taken and it comes back like behind jd_dec_quiesce; a command that was never answered is a lost workgroup
            const int rs = jd_res_stop(d);
            if (rs || ++restarts > 3) {
                pipe_drain(d);
                return rs ? rs : jd_fail(JD_EHIP, "the resident search kernel keeps ending under a batch");
            }
            continue;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    HIPCHK(hipStreamSynchronize(d->s_gmm));                            // (the exports)
    PipeBatch &F = P->q.front();
    std::vector<int> slot_of((size_t)n_utts);
    for (int u = 0; u < n_utts; ++u) slot_of[(size_t)u] = F.u[(size_t)u].slot;
    if ((size_t)n_utts > d->results.size()) d->results.resize((size_t)n_utts);
    d->timing = jd_timing();
    const int rc = fetch_results_from(d, P->d_vctl, P->d_vresn, P->d_vres, slot_of.data(), F.table * P->max_batch, n_utts, out, 0, nullptr);
    for (int u = 0; u < n_utts; ++u) d->timing.search_frames += F.u[(size_t)u].T;
    d->timing.gmm_frames = d->timing.search_frames; d->timing.gmm_states = d->am->n_gmm;
    d->timing.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    d->timing.search_ms = d->timing.total_ms; d->timing.search_launches = 0; d->timing.cluster_wgs = 1; d->timing.prefetched = 1;
    d->load_sum = d->load_frames = 0.0;
    P->table_used[(size_t)F.table] = 0;
    P->q.pop_front();
    d->pipe_batches_back += 1;
    P->serial0 += 1;
    if (P->q.empty()) pipe_drain(d);                                   // nothing announced behind it: the kernel leaves the device
    *handled = 1;
    return rc;
}

