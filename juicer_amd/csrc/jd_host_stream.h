// jd_host_stream.h - the IDecoder seam of the C ABI (included by jd_device.hip): jd_stream_init / _push / _finish (IDecoder::init /
// processFrame / finish, src/Decoder.h:18-30, src/WFSTDecoderLite.cpp:139-372), jd_streams_push (several callers' frames in one launch),
// PARTIAL_DECODING (jd_dec_set_partial_interval, jd_stream_partial: src/WFSTDecoderLite.cpp:822-896) and the collection bookkeeping.
#pragma once

extern "C" int jd_stream_init(jd_dec *d, int32_t s)
{
    if (!d || s < 0 || s >= d->max_streams) return jd_fail(JD_EINVAL, "jd_stream_init: bad stream");
    int rc = check_device(d->device);
    if (rc) return rc;
    rc = ensure_arenas(d);
    if (rc) return rc;
    // (the streaming interface names its streams itself: whatever a stream of batches had been started ahead on them is
    // dropped - the batch concerned starts again when it is decoded)
    pf_discard(d);
    if (d->lazy_in[(size_t)s]) { d->lazy_in[(size_t)s] = 0; jd_lazy_leave(d->net, 1); }   // (an utterance that was never finished)
    bool net_failed = false;
    rc = jd_lazy_enter(d->net, 1, &net_failed);                       // (may start a new arena generation)
    if (rc) return rc;
    if (net_failed) {                                                  // as in jd_decode_batch_device
        jd_lazy_leave(d->net, 1);
        return jd_fail(JD_ENOMEM, "the lazily composed network has run out of room and other utterances are inside it: "
                       "it starts again when they are through (capacity %d states, %lld arcs)", d->net->n_states, (long long)d->net->n_arcs);
    }
    d->lazy_in[(size_t)s] = d->net->lazy_dev != nullptr;
    rc = mark_init(d, s, 1, d->s_search);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(d->s_search));
    d->stream_T[(size_t)s] = 0;
    d->stream_started[(size_t)s] = 1;
    d->last_collect[(size_t)s] = -1; d->last_trace[(size_t)s] = -1;    // WFSTDecoderLite.cpp:179-181, 202-206
    d->n_collect_host[(size_t)s] = 0;
    d->partial_label[(size_t)s].clear(); d->partial_time[(size_t)s].clear();
    return JD_OK;
}

// tracePartialPath (:824-868) on stream s at the frame it has reached; extends the stream's
// partialPaths when a converged record is found
static int trace_partial(jd_dec *d, int s, int *found)
{
    if (!d->d_partial_out) { int rc = dmalloc(d, &d->d_partial_out, 2); if (rc) return rc; }
    std::vector<int32_t> &L = d->partial_label[(size_t)s], &Tm = d->partial_time[(size_t)s];
    const int last_frame = Tm.empty() ? -1 : Tm.back();
    hipStream_t st = d->s_search;
    if (d->am->max_n <= 5)
        hipLaunchKernelGGL(k_partial<3>, dim3(1), dim3(1024), 0, st, d->C, d->d_ctl, d->d_streams, s, last_frame, d->d_partial_out);
    else hipLaunchKernelGGL(k_partial<6>, dim3(1), dim3(1024), 0, st, d->C, d->d_ctl, d->d_streams, s, last_frame, d->d_partial_out);
    HIPCHK(hipGetLastError());
    int ho[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(ho, d->d_partial_out, sizeof ho, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    d->last_trace[(size_t)s] = d->stream_T[(size_t)s] - 1;             // :867
    if (found) *found = ho[0];
    if (!ho[0]) return JD_OK;
    if (ho[1] > d->res_cap) return jd_fail(JD_ENOMEM, "stream %d: partial path has %d records (> %d)", s, ho[1], d->res_cap);
    // the chain from the root to the found record; the records traced before are its prefix
    L.resize((size_t)ho[1]); Tm.resize((size_t)ho[1]);
    const int *base = d->d_res + (size_t)s * 5 * d->res_cap;           // res_label, res_time: arrays 0 and 1 of the stream
    HIPCHK(hipMemcpy(L.data(), base, (size_t)ho[1] * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(Tm.data(), base + d->res_cap, (size_t)ho[1] * 4, hipMemcpyDeviceToHost));
    return JD_OK;
}

extern "C" int jd_stream_push(jd_dec *d, int32_t s, const float *frames, int32_t n_frames)
{
    if (!d || s < 0 || s >= d->max_streams || n_frames < 0 || (n_frames > 0 && !frames))
        return jd_fail(JD_EINVAL, "jd_stream_push: bad argument");
    if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_stream_push before jd_stream_init");
    int rc = check_device(d->device);
    if (rc) return rc;
    const int D = d->am->D, G = d->am->n_gmm, Fc = d->Fc;
    hipStream_t st = d->s_search;
    pf_discard(d);                                                     // (the streaming path scores into table 0)
    for (int done = 0, n = 0; done < n_frames; done += n) {
        n = std::min(Fc, n_frames - done);
        // PARTIAL_DECODING rides on the path collection (:362-368): a chunk ends at the frame rule's frame
        // (the first frame f with f - lastPathCollectFrame > 100)
        const int f_collect = d->last_collect[(size_t)s] + 101;
        if (d->partial_interval > 0) n = std::min(n, std::max(1, f_collect + 1 - d->stream_T[(size_t)s]));
        if ((size_t)n * D > d->push_cap) {
            if (d->d_push) (void)hipFree(d->d_push);
            HIPCHK(hipMalloc(&d->d_push, (size_t)Fc * D * sizeof(float)));
            d->push_cap = (size_t)Fc * D;
        }
        HIPCHK(hipMemcpyAsync(d->d_push, frames + (size_t)done * D, (size_t)n * D * sizeof(float),
                              hipMemcpyHostToDevice, st));
        std::vector<int> src((size_t)Fc, -1);
        for (int i = 0; i < n; ++i) src[(size_t)i] = i;
        if ((size_t)Fc > d->row_src_cap[0]) {
            if (d->d_row_src[0]) (void)hipFree(d->d_row_src[0]);
            d->d_row_src[0] = nullptr; d->row_src_cap[0] = 0;
            HIPCHK(hipMalloc(&d->d_row_src[0], (size_t)Fc * sizeof(int)));
            d->row_src_cap[0] = (size_t)Fc;
        }
        HIPCHK(hipMemcpyAsync(d->d_row_src[0], src.data(), (size_t)Fc * sizeof(int), hipMemcpyHostToDevice, st));
        const int f0 = d->stream_T[(size_t)s];
        const int Tnew = f0 + n;
        HIPCHK(hipMemcpyAsync(d->d_T + s, &Tnew, sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(jd_set_T_kernel, dim3(1), dim3(64), 0, st, d->d_ctl, s, 1, d->d_T + s);
        rc = launch_gmm(d->am, d->amb, d->d_push, d->d_row_src[0], n, d->d_ll[0], st);
        if (rc) return rc;
        if (d->partial_interval <= 0) {
            rc = launch_search(d, std::vector<int2>(1, make_int2(s, 0)), d->d_ll[0], (long long)Fc * G, f0, Tnew, st);
            if (rc) { d->stream_dirty[(size_t)s] = 1; return rc; }
            d->stream_T[(size_t)s] = Tnew;
            continue;
        }
        // PARTIAL_DECODING: collectPaths runs after a frame when the count rule fires (path_rule_fires, evaluated by
        // the kernel after every frame: the launch then stops, the collection runs and launch_search comes back) or
        // the frame rule does (the chunk ends there), and the trace rides on it (:362-368)
        for (;;) {
            d->return_on_collect = true; d->collected_now = false;
            rc = launch_search(d, std::vector<int2>(1, make_int2(s, 0)), d->d_ll[0], (long long)Fc * G, f0, Tnew, st);
            d->return_on_collect = false;
            if (rc) { d->stream_dirty[(size_t)s] = 1; return rc; }
            StreamCtl hc;
            HIPCHK(hipMemcpy(&hc, d->d_ctl + s, sizeof hc, hipMemcpyDeviceToHost));
            if (hc.error != 0) { d->stream_T[(size_t)s] = Tnew; d->stream_dirty[(size_t)s] = 1; break; }   // (reported by jd_stream_finish)
            const int at = hc.frame - 1;                               // the last frame processed
            // (with the reference's counts a collection that only the arena asked for is none of the reference's: it neither
            // counts nor carries a trace, and the launch goes on behind it)
            const bool ref = d->C.pcount != nullptr;
            bool collected = ref ? hc.n_collect > d->n_collect_host[(size_t)s] : d->collected_now;
            if (!collected && hc.frame >= Tnew && (at - d->last_collect[(size_t)s] > 100 ||
                                                   path_rule_fires(ref ? hc.n_paths_ref : hc.n_paths, ref ? hc.path_new_ref : hc.path_new))) {
                // the rule fires behind the chunk's last frame (the kernel looks before a frame, not after the last one)
                DecConst Cg = d->C;
                Cg.gc_threshold = -1;                                  // (every started stream collects)
                launch_gc(Cg, d->d_ctl, d->d_streams, nullptr, 1, s, d->am->max_n <= 5, d->n_cus, st);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(st));
                collected = true;
            }
            d->stream_T[(size_t)s] = hc.frame;
            if (collected) {
                d->last_collect[(size_t)s] = at;                       // :746
                d->n_collect_host[(size_t)s] += 1;
                if (at - d->last_trace[(size_t)s] > d->partial_interval) {
                    rc = trace_partial(d, s, nullptr);
                    if (rc) return rc;
                }
            }
            if (hc.frame >= Tnew) break;
        }
    }
    return JD_OK;
}

// jd_stream_push for SEVERAL streams at once: the frames of all of them are scored by ONE launch of the scoring
// kernel and searched by ONE persistent launch, every stream a cluster of its own - what a broker that serves
// many IDecoder instances (jd_broker_*, jd_broker.cpp) makes of the pushes that have arrived since its last tick.
// A call takes any number of frames per stream (the tables are sized for the call).  The streams may sit at
// different frames: row r of the common table is frame stream_T[s] + (r - first row of s) of stream s.
// PARTIAL_DECODING rides on single-stream pushes (its traces are taken between launches): not here.
extern "C" int jd_streams_push(jd_dec *d, int32_t n, const int32_t *streams, const float *const *frames, const int32_t *n_frames)
{
    if (!d || n < 0 || (n > 0 && (!streams || !frames || !n_frames))) return jd_fail(JD_EINVAL, "jd_streams_push: bad argument");
    if (d->partial_interval > 0) return jd_fail(JD_ESTATE, "jd_streams_push: partial traces ride on jd_stream_push");
    int rc = check_device(d->device);
    if (rc) return rc;
    const int D = d->am->D, G = d->am->n_gmm;
    std::vector<char> seen((size_t)d->max_streams, 0);
    long long rows = 0;
    for (int i = 0; i < n; ++i) {
        const int s = streams[i];
        if (s < 0 || s >= d->max_streams || n_frames[i] < 0 || (n_frames[i] > 0 && !frames[i]) || seen[(size_t)s])
            return jd_fail(JD_EINVAL, "jd_streams_push: bad stream %d (each stream once)", s);
        if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_streams_push before jd_stream_init (stream %d)", s);
        seen[(size_t)s] = 1;
        rows += n_frames[i];
    }
    if (rows == 0) return JD_OK;
    if (rows > 0x7fffff00LL) return jd_fail(JD_EINVAL, "jd_streams_push: more than 2^31 frames in one call");
    hipStream_t st = d->s_search;
    pf_discard(d);                                                     // (the streaming path scores into table 0)
    const size_t tile_rows = ((size_t)rows + GMM_ROWS2 - 1) / GMM_ROWS2 * GMM_ROWS2;
    rc = ensure_table(d, 0, tile_rows * (size_t)G, tile_rows);
    if (rc) return rc;
    if ((size_t)rows * D > d->push_cap) {
        if (d->d_push) (void)hipFree(d->d_push);
        d->d_push = nullptr; d->push_cap = 0;
        HIPCHK(hipMalloc(&d->d_push, tile_rows * D * sizeof(float)));
        d->push_cap = tile_rows * D;
    }
    // rows: stream after stream, packed; row_src is the identity (the frames are packed the same way).  Everything the
    // launch needs from the host - frames, row table, frames available per stream - goes through ONE pinned staging
    // buffer: a tick of the broker is sixteen callers' frames, and sixteen copies from pageable memory were a tenth of it.
    const size_t need = (size_t)rows * D * sizeof(float) + tile_rows * sizeof(int) + (size_t)d->max_streams * sizeof(int);
    if (need > d->stage_cap) {
        if (d->h_stage) (void)hipHostFree(d->h_stage);
        d->h_stage = nullptr; d->stage_cap = 0;
        HIPCHK(hipHostMalloc((void **)&d->h_stage, need + need / 2));
        d->stage_cap = need + need / 2;
    }
    float *h_frames = (float *)d->h_stage;
    int *h_src = (int *)(d->h_stage + (size_t)rows * D * sizeof(float));
    int *h_T = h_src + tile_rows;
    std::vector<int> Tnew((size_t)n);
    std::vector<int2> work;
    std::vector<double> weight;
    size_t r0 = 0;
    int f_end = 0;
    for (int s = 0; s < d->max_streams; ++s) h_T[s] = d->stream_T[(size_t)s];     // (the other streams keep theirs)
    for (int i = 0; i < n; ++i) {
        if (n_frames[i] == 0) continue;
        const int s = streams[i];
        memcpy(h_frames + r0 * D, frames[i], (size_t)n_frames[i] * D * sizeof(float));
        Tnew[(size_t)i] = d->stream_T[(size_t)s] + n_frames[i];
        h_T[s] = Tnew[(size_t)i];
        // (k_search reads row  slot + (f - f0)  with f0 = 0: the slot is the stream's first row minus its first frame)
        work.push_back(make_int2(s, (int)((long long)r0 - d->stream_T[(size_t)s])));
        weight.push_back((double)n_frames[i]);
        f_end = std::max(f_end, Tnew[(size_t)i]);
        r0 += (size_t)n_frames[i];
    }
    for (size_t r = 0; r < tile_rows; ++r) h_src[r] = r < (size_t)rows ? (int)r : -1;
    HIPCHK(hipMemcpyAsync(d->d_push, h_frames, (size_t)rows * D * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d->d_row_src[0], h_src, tile_rows * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d->d_T, h_T, (size_t)d->max_streams * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(jd_set_T_kernel, dim3((d->max_streams + 63) / 64), dim3(64), 0, st, d->d_ctl, 0, d->max_streams, d->d_T);
    HIPCHK(hipGetLastError());
    rc = launch_gmm(d->am, d->amb, d->d_push, d->d_row_src[0], (int)rows, d->d_ll[0], st);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }             // (the staging buffer is the next call's too)
    rc = launch_search(d, work, d->d_ll[0], (long long)G, 0, f_end, st, &weight);
    if (rc) { (void)hipStreamSynchronize(st); for (const int2 &w : work) d->stream_dirty[(size_t)w.x] = 1; return rc; }
    for (int i = 0; i < n; ++i) if (n_frames[i] > 0) d->stream_T[(size_t)streams[i]] = Tnew[(size_t)i];
    return JD_OK;
}


#include "jd_host_resident.h"

// collectPaths runs (WFSTDecoderLite.cpp:362) of stream s since its init, and the frame after which the last one ran
// (lastPathCollectFrame, :746; -1: none yet).  Counted while PARTIAL_DECODING is on (jd_dec_set_partial_interval > 0).
extern "C" int jd_stream_collect_info(jd_dec *d, int32_t s, int32_t *n_collections, int32_t *last_collect_frame)
{
    if (!d || s < 0 || s >= d->max_streams) return jd_fail(JD_EINVAL, "jd_stream_collect_info: bad argument");
    if (n_collections) *n_collections = d->n_collect_host[(size_t)s];
    if (last_collect_frame) *last_collect_frame = d->last_collect[(size_t)s];
    return JD_OK;
}

// setPartialDecodeOptions (WFSTDecoderLite.cpp:892-896; the reference reads PartialTraceInterval
// from the environment, :116-119)
// The Path objects WFSTDecoderLite::propagateToken creates behind ONE token that arrives at state q (:497-509 inside
// the recursion of :533-541 and :583-599): one per labelled epsilon arc and per labelled arc of a tee model that leaves
// q, plus what arrives behind each of those arcs - with multiplicity, the recursion does not recombine.  Static as
// long as nothing prunes inside the closure, i.e. with the end and word beams off (the thresholds of :538, :591-596 are
// LOG_ZERO then).  Saturates at 2^20 (the rule's own mark is 10000).  false: the label-less part of the graph has a
// cycle (the reference would not come back from it).
static bool closure_path_counts(const jd_net *net, const jd_am *am, std::vector<int> &P)
{
    const int nS = net->n_states;
    P.assign((size_t)nS, -1);
    std::vector<char> open((size_t)nS, 0);
    std::vector<std::pair<int, int>> stack;                           // (state, next arc)
    auto passes = [&](const JdArc &a) { return a.in == 0 || am->hmm_tee[(size_t)a.in - 1] > LZ; };
    for (int q0 = 0; q0 < nS; ++q0) {
        if (P[(size_t)q0] >= 0) continue;
        stack.assign(1, std::make_pair(q0, net->row_ptr[(size_t)q0]));
        open[(size_t)q0] = 1;
        while (!stack.empty()) {
            const int q = stack.back().first;
            int &a = stack.back().second;
            bool descended = false;
            for (; a < net->row_ptr[(size_t)q + 1]; ++a) {
                const JdArc &arc = net->arcs[(size_t)a];
                if (!passes(arc) || P[(size_t)arc.to] >= 0) continue;
                if (open[(size_t)arc.to]) return false;
                open[(size_t)arc.to] = 1;
                stack.push_back(std::make_pair(arc.to, net->row_ptr[(size_t)arc.to]));
                descended = true;
                break;
            }
            if (descended) continue;
            long long sum = 0;
            for (int b = net->row_ptr[(size_t)q]; b < net->row_ptr[(size_t)q + 1]; ++b) {
                const JdArc &arc = net->arcs[(size_t)b];
                if (passes(arc)) sum += (arc.out != 0 ? 1 : 0) + P[(size_t)arc.to];
            }
            P[(size_t)q] = (int)std::min<long long>(sum, 1 << 20);
            open[(size_t)q] = 0;
            stack.pop_back();
        }
    }
    return true;
}

// Diagnostics / tests (host only, no device): the per-state counts of closure_path_counts - what jd_dec_set_partial_interval puts
// on the device for collectPaths' count trigger - into out[n_states]; *acyclic = 0 when the label-less part of the graph has
// a cycle (the counts are then not used)
extern "C" int jd_debug_closure_path_counts(const jd_net *net, const jd_am *am, int32_t *out, int32_t *acyclic)
{
    if (!net || !am || !out || !acyclic) return jd_fail(JD_EINVAL, "jd_debug_closure_path_counts: null argument");
    if (net->lazy_dev) return jd_fail(JD_EINVAL, "jd_debug_closure_path_counts: not for a lazily composed network");
    std::vector<int> P;
    *acyclic = closure_path_counts(net, am, P) ? 1 : 0;
    for (int q = 0; q < net->n_states; ++q) out[q] = P[(size_t)q];
    return JD_OK;
}

extern "C" int jd_dec_set_partial_interval(jd_dec *d, int32_t interval)
{
    if (!d || interval < 0) return jd_fail(JD_EINVAL, "jd_dec_set_partial_interval: traceInterval >= 0");
    d->partial_interval = interval;
    d->C.path_rule = interval > 0 ? 1 : 0;             // (the kernel then watches collectPaths' count rule as well)
    d->C.pcount = nullptr;
    if (interval > 0 && !d->net->lazy_dev && d->C.end_win <= 0.0f && d->C.word_win <= 0.0f) {
        // ... on the reference's own counts where they are a static property of the graph
        if (!d->d_pcount) {
            int rc = check_device(d->device);
            if (rc) return rc;
            std::vector<int> P;
            if (closure_path_counts(d->net, d->am, P)) {
                if (!d->state_new.empty()) {                           // (the kernels index it by the decoder's own state numbers)
                    std::vector<int> Q(P.size());
                    for (size_t q = 0; q < P.size(); ++q) Q[(size_t)d->state_new[q]] = P[q];
                    P.swap(Q);
                }
                rc = dupload(d, &d->d_pcount, P.data(), P.size());
                if (rc) return rc;
            }
        }
        d->C.pcount = d->d_pcount;
    }
    return JD_OK;
}

// nPath and nPathNew of stream s as collectPaths' trigger reads them (WFSTDecoderLite.cpp:360): the reference's counts
// where this decoder keeps them (*exact = 1), else the records in this build's arena and what its last collection kept
extern "C" int jd_stream_path_counts(jd_dec *d, int32_t s, int32_t *n_path, int32_t *n_path_new, int32_t *exact)
{
    if (!d || s < 0 || s >= d->max_streams) return jd_fail(JD_EINVAL, "jd_stream_path_counts: bad argument");
    if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_stream_path_counts before jd_stream_init");
    int rc = check_device(d->device);
    if (rc) return rc;
    StreamCtl hc;
    HIPCHK(hipMemcpy(&hc, d->d_ctl + s, sizeof hc, hipMemcpyDeviceToHost));
    const bool ref = d->C.pcount != nullptr;
    if (n_path) *n_path = ref ? hc.n_paths_ref : hc.n_paths;
    if (n_path_new) *n_path_new = ref ? hc.path_new_ref : hc.path_new;
    if (exact) *exact = ref ? 1 : 0;
    return JD_OK;
}

extern "C" int jd_stream_partial(jd_dec *d, int32_t s, int32_t trace_now, int32_t cap, int32_t *n, int32_t *labels,
                                 int32_t *times, int32_t *found)
{
    if (!d || s < 0 || s >= d->max_streams || cap < 0 || !n) return jd_fail(JD_EINVAL, "jd_stream_partial: bad argument");
    if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_stream_partial before jd_stream_init");
    int rc = check_device(d->device);
    if (rc) return rc;
    int fnd = 0;
    if (trace_now && d->stream_T[(size_t)s] > 0) {
        rc = trace_partial(d, s, &fnd);
        if (rc) return rc;
    }
    const std::vector<int32_t> &L = d->partial_label[(size_t)s], &Tm = d->partial_time[(size_t)s];
    *n = (int32_t)L.size();
    for (int k = 0; k < std::min<int>(cap, *n); ++k) {
        if (labels) labels[k] = L[(size_t)k];
        if (times) times[k] = Tm[(size_t)k];
    }
    if (found) *found = fnd;
    return JD_OK;
}

extern "C" int jd_stream_finish(jd_dec *d, int32_t s, jd_hyp *out)
{
    if (!d || s < 0 || s >= d->max_streams || !out) return jd_fail(JD_EINVAL, "jd_stream_finish: bad argument");
    if (!d->stream_started[(size_t)s]) return jd_fail(JD_ESTATE, "jd_stream_finish before jd_stream_init");
    int rc = check_device(d->device);
    if (rc) return rc;
    if (d->stream_T[(size_t)s] == 0) {                                 // init() directly followed by finish(): recognitionStart only
        rc = launch_search(d, std::vector<int2>(1, make_int2(s, 0)), d->d_ll[0], 0, 0, 0, d->s_search);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(jd_finish_kernel, dim3(1), dim3(64), 0, d->s_search, d->d_ctl, d->d_streams, s, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->s_search));
    // results of stream s are stored at result slot s
    std::vector<jd_hyp> tmp((size_t)d->max_streams);
    rc = fetch_results(d, s, 1, tmp.data(), s);
    *out = tmp[(size_t)s];
    if (d->lazy_in[(size_t)s]) { d->lazy_in[(size_t)s] = 0; jd_lazy_leave(d->net, 1); }   // the utterance has left the network
    if (d->partial_interval > 0 && out->n >= 0) {                      // :245-251 one more trace, from the best token
        std::vector<int32_t> &L = d->partial_label[(size_t)s], &Tm = d->partial_time[(size_t)s];
        L.resize((size_t)out->n); Tm.resize((size_t)out->n);
        for (int k = 0; k < out->n; ++k) { L[(size_t)k] = out->label[out->n - 1 - k]; Tm[(size_t)k] = out->time[out->n - 1 - k]; }
    }
    return rc;
}
