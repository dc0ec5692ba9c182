// jd_host_launch.h - launch_search (included by jd_device.hip): how the streams of a work list get their workgroups for ONE persistent
// launch of the search - uniform clusters, the weighted plan (clusters sized so that the streams finish together), its XCD-local
// packing, the batch behind the running one beside it, the slot kernel as a plain launch for more streams than CUs - and the loop
// around the launches (Path collections, re-planning cuts).  Reference: the per-file loop of DecoderBatchTest::run,
// src/DecoderBatchTest.cpp:738-771, as one launch for many utterances.
#pragma once

// The load (instances + arcs per stream-frame) of the streams of a work list so far, from their statistics:
// scales the cost model's b (jd_dec::load_scale).  Called when the decoder has not seen a batch yet.
static int learn_load(jd_dec *d, const std::vector<int2> &work_in)
{
    std::vector<long long> st((size_t)d->max_streams * ST_N);
    std::vector<int> head((size_t)d->max_streams * 4);
    HIPCHK(hipMemcpy2D(st.data(), ST_N * sizeof(long long), (const char *)d->d_ctl + offsetof(StreamCtl, st), sizeof(StreamCtl),
                       ST_N * sizeof(long long), (size_t)d->max_streams, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy2D(head.data(), 16, d->d_ctl, sizeof(StreamCtl), 16, (size_t)d->max_streams, hipMemcpyDeviceToHost));
    double work = 0.0, frames = 0.0;
    for (const int2 &w : work_in) {
        work += (double)st[(size_t)w.x * ST_N + ST_INSTS] + (double)st[(size_t)w.x * ST_N + ST_ARCS];
        frames += (double)head[(size_t)w.x * 4];
    }
    if (frames > 0.0) d->load_scale = std::min(1e5, std::max(0.25, work / frames / 23700.0));
    return JD_OK;
}

// the flavours of k_search (jd_search.h): HMMs of up to 5 / 8 states, agent-scope / XCD-local memory model, static / lazily
// composed graph
typedef void (*SearchKernel)(SearchArgs);
static SearchKernel search_kernel(bool ne3, bool xl, bool lazy)
{
    static const SearchKernel tab[8] = {
        k_search<6, false, false>, k_search<6, false, true>, k_search<6, true, false>, k_search<6, true, true>,
        k_search<3, false, false>, k_search<3, false, true>, k_search<3, true, false>, k_search<3, true, true>,
    };
    return tab[(ne3 ? 4 : 0) + (xl ? 2 : 0) + (lazy ? 1 : 0)];
}

// Advance the streams of `work` ({stream, likelihood slot}) through frames [.., f_end) with ONE
// persistent launch (k_search): every stream gets a cluster of workgroups, one 512-thread workgroup
// per CU in total, all resident at once (the clusters synchronise with barriers of their own).  A
// launch stops a stream early when its Path arena needs collecting; the collection (k_gc_*) runs
// after such a launch and the launch is repeated until every stream is through.
static int pf_launch(jd_dec *d);
static bool pf_wants_scoring(const jd_dec *d);
static double pf_scoring_rows(const jd_dec *d);
static bool pf_scoring_in_flight(const jd_dec *d);
static int pf_background(jd_dec *d, int fg_bank, const std::vector<int> *heads, hipStream_t st, std::vector<int2> *work, std::vector<double> *left);
static int mark_init(jd_dec *d, int s0, int n, hipStream_t st);
static int launch_search(jd_dec *d, const std::vector<int2> &work_first, const float *ll, long long ll_stride, int f0, int f_end,
                         hipStream_t st, const std::vector<double> *weight_first = nullptr)
{
    if (work_first.empty()) return JD_OK;
    if ((int)work_first.size() > d->work_cap) {
        if (d->d_work) (void)hipFree(d->d_work);
        d->d_work = nullptr; d->work_cap = 0;
        const size_t cap = std::max<size_t>(work_first.size(), (size_t)d->max_streams);
        HIPCHK(hipMalloc(&d->d_work, cap * sizeof(int4)));
        d->work_cap = (int)cap;
    }
    std::vector<int2> work_in = work_first;
    std::vector<int2> bg;                                              // streams of the batch behind, advanced beside these (pf_background)
    std::vector<double> bg_left;                                       // ... and the frames each has ahead
    std::vector<int> heads;
    std::vector<double> weight_now;
    std::vector<int> frame_before;                                     // per stream: where the previous launch found it
    const std::vector<double> *weight = weight_first;
    const bool ne3 = d->am->max_n <= 5;
    const int max_rounds = (f_end - f0) + 64;                          // every launch makes at least one frame of progress
    for (int it = 0;; ++it) {
    const int n_work = (int)work_in.size();
    int nwg_all = std::max(1, d->n_cus * WG_PER_CU);
    // Two batches in flight: the plan below makes the clusters of this batch finish together and the utterances of the
    // batch behind fill what is left - nothing idles, and the scoring of the table after that, which lives on idle CUs,
    // would finish long after the launch (and the batch behind the next one start late).  So the scoring gets CUs of its
    // own: as many as carry its cost over a launch as long as the last wave was (a multiple of eight: the XCD-local
    // numbering), a third of the chip at most; the search is planned on the rest.
    int reserve = 0;
    if (d->fg_bank >= 0 && d->pf_armed && weight && d->weighted && (pf_wants_scoring(d) || pf_scoring_in_flight(d))) {
        if (it == 0) {
            d->reserve_now = d->score_reserve;
            if (d->reserve_now < 0) d->reserve_now = (d->gmm_ms_per_row > 0.0 && d->last_wave_ms > 0.0)
                                                   ? (int)std::ceil(d->gmm_ms_per_row * pf_scoring_rows(d) * nwg_all / d->last_wave_ms) : 0;
            d->reserve_now = std::min((d->reserve_now + 7) & ~7, (nwg_all / 3) & ~7);
        }
        reserve = d->reserve_now;                                      // (the legs of a re-planned launch leave the same CUs alone)
    }
    // two batches in flight: the utterances of the batch behind this one, one workgroup each at least, on a quarter of the grid at most
    bg.clear();
    if (d->fg_bank >= 0 && weight && d->weighted) {
        const bool started = !d->pf_q.empty() && d->pf_q.front().bank >= 0;
        if (started) {
            heads.assign((size_t)d->max_streams * 4, 0);
            HIPCHK(hipMemcpy2D(heads.data(), 16, d->d_ctl, sizeof(StreamCtl), 16, (size_t)d->max_streams, hipMemcpyDeviceToHost));
        }
        const int br = pf_background(d, d->fg_bank, started ? &heads : nullptr, st, &bg, &bg_left);
        if (br) return br;
        if ((int)bg.size() > nwg_all / 4 || nwg_all - (int)bg.size() < 2 * n_work) { bg.clear(); bg_left.clear(); }
    }
    const int n_bg = (int)bg.size();
    while (reserve > 0 && nwg_all - reserve - n_bg < 2 * n_work) reserve -= 8;
    nwg_all -= std::max(reserve, 0);
    if ((int)(n_work + n_bg) > d->work_cap) {
        if (d->d_work) (void)hipFree(d->d_work);
        d->d_work = nullptr; d->work_cap = 0;
        HIPCHK(hipMalloc(&d->d_work, (size_t)(n_work + n_bg) * sizeof(int4)));
        d->work_cap = n_work + n_bg;
    }
    SearchArgs A;
    A.C = d->C; A.ctl = d->d_ctl; A.streams = d->d_streams; A.work = d->d_work; A.n_work = n_work; A.n_prio = 0;
    A.cells = (d->d_cells && ll == d->d_ll_slab) ? d->d_cells + 2 : nullptr;   // (the first two words: the counter of jd_dec_debug_cells)
    const int nwg = nwg_all - n_bg;                                    // what the plan of THESE streams may use
    // a wave segment holds at least one 64-record chunk of instances and 512 frontier items (one wave
    // writes the whole epsilon closure of the items it expands)
    const int cw_cap = (int)std::max<int64_t>(1, std::min<int64_t>(d->cap_slots / (64 * SW), d->cap_items / (512 * SW)));
    const int max_cw = std::min(d->max_cw, cw_cap);
    A.Cw = std::max(1, std::min(max_cw, nwg / n_work));
    A.n_slots = std::min(n_work, std::max(1, nwg / A.Cw));
    std::vector<int4> work((size_t)n_work);
    for (int k = 0; k < n_work; ++k) work[(size_t)k] = make_int4(work_in[(size_t)k].x, work_in[(size_t)k].y, k * A.Cw, A.Cw);
    int grid = A.n_slots * A.Cw;
    bool xl = false;
    int rebalance_at = 0;
    if (weight && d->weighted && (n_work > 1 || n_bg > 0) && max_cw > 1 && nwg >= 2 * n_work) {
        // Weighted mode.  weight[k] = frames stream k has in this launch.  A stream's frame costs about
        // a + b / workgroups  (a: the barriers and list set-up of a frame; b: the part that divides over
        // the cluster), so stream k finishes after  frames_k * (a + b / C_k).  The launch ends with its
        // last stream: the C_k that make all streams finish together solve  C_k = b / (tau / frames_k - a)
        // for the smallest common tau the device's workgroups allow (bisection).  a and b were fitted on
        // configs[1] (DESIGN.md "cluster sizes"); sizing by a stream's measured work per frame (a pilot
        // launch, or the previous chunk's counters) was tried and is slower - the work of the frames
        // ahead is not the work of the frames behind.
        // plan_mode 1 (JD_PLAN=1; measured, not the default): the MEASURED curve - configs[1]'s longest stream with every
        // cluster capped at C = 1 .. 8 workgroups takes 86, 56, 46, 41, 38, 35.5, -, 32 us per frame: a + b / C with a = 24.6,
        // b = 61.4 to within 2 % - and whole workgroups dealt GREEDILY: every stream starts with plan_min_cw, the next one
        // goes to the stream that would finish last, until the grid is used up; beside a launch the next batch's table is
        // scored, so workgroups that would shorten the launch below what the chip needs for BOTH (a cluster's barrier
        // share a * C burns CU-time) are not dealt.  The launch itself gets much shorter (configs[1]: 38.5 -> 35.3 ms
        // un-cut, 29.6 with every workgroup dealt) but the step does not: scored ahead it is bound by CU-time either
        // way (38.2-39.2 against 38.3-39.6 ms), in the serial order it gains 5-7 % with b = 61.4 and nothing with a
        // b that is safe for streams heavier than the longest one (one b serves all streams, and a stream that gets
        // ONE workgroup on a b that is too small is the straggler), and the configs[4] graph loses 1-2 %.  plan_mode 0:
        // the constants fitted in round 2 (a = 10, b = 360 - right at 58 ms per step, wrong now, but erring towards
        // larger short clusters, which is what re-planning and scoring ahead forgive) and the floor of the continuous
        // solution.
        // (with the batch behind beside it: past eight workgroups a cluster gains little - 32 us per frame against 28 at
        // sixteen - and the workgroups do more for the streams of the batch behind, JD_FG_CW)
        const int mcw = n_bg > 0 ? std::min(max_cw, d->fg_cw_cap) : max_cw;
        const bool greedy = d->plan_mode == 1;
        const double a_us = greedy ? d->model2_a_us : d->model_a_us, b_us = (greedy ? d->model2_b_us : d->model_b_us) * d->load_scale;
        std::vector<int> cw((size_t)n_work, 1);
        int used = 0;
        if (greedy) {
            std::priority_queue<std::pair<double, int>> pq;
            double cu_us = 0.0;                                            // CU-time of the plan so far
            const int c0 = std::max(1, std::min(std::min(d->plan_min_cw, mcw), nwg / n_work));   // every stream starts with this many
            for (int k = 0; k < n_work; ++k) {
                const double fr = std::max((*weight)[(size_t)k], 1.0);
                cw[(size_t)k] = c0;
                pq.push({fr * (a_us + b_us / c0), k});
                cu_us += fr * (a_us * c0 + b_us);
            }
            used = n_work * c0;
            const double gmm_cu_us = (d->pf_armed && pf_wants_scoring(d))
                                   ? d->pf_gmm_weight * 1e3 * d->gmm_ms_per_row * pf_scoring_rows(d) * nwg : 0.0;
            while (used < nwg && !pq.empty()) {
                const std::pair<double, int> top = pq.top();
                const int k = top.second;
                if (cw[(size_t)k] >= mcw) break;                        // the launch cannot end sooner than this stream
                if (gmm_cu_us > 0.0 && top.first <= (cu_us + gmm_cu_us) / nwg) break;
                pq.pop();
                const double fr = std::max((*weight)[(size_t)k], 1.0);
                ++cw[(size_t)k]; ++used;
                cu_us += fr * a_us;
                pq.push({fr * (a_us + b_us / cw[(size_t)k]), k});
            }
        }
        // One plan for both batches: a stream of the batch behind counts with a part of the frames it has ahead
        // (bg_weight: its turn as the batch the caller waits for is still to come - it has two launches to get through)
        // and so gets workgroups by its length like everybody else: the long utterances, which are what the NEXT launch
        // will last as long as, are the ones that get ahead.
        const int n_plan = greedy ? n_work : n_work + n_bg;
        const int nwg_plan = greedy ? nwg : nwg + n_bg;
        const int mcw_bg = std::max(1, std::min(d->bg_cw_cap, max_cw));
        std::vector<double> wt_plan(weight->begin(), weight->begin() + n_work);
        if (!greedy) for (int i = 0; i < n_bg; ++i) wt_plan.push_back(d->bg_weight * bg_left[(size_t)i]);
        if (!greedy) cw.assign((size_t)n_plan, 1);
        auto cap_of = [&](int k) { return k < n_work ? mcw : mcw_bg; };
        auto need = [&](double tau, std::vector<double> *out) {
            double tot = 0.0;
            for (int k = 0; k < n_plan; ++k) {
                const double fr = std::max(wt_plan[(size_t)k], 1.0);
                const double slack = tau / fr - a_us;
                double c = slack > 1e-9 ? b_us / slack : 1e9;
                c = std::min(std::max(c, 1.0), (double)cap_of(k));
                if (out) (*out)[(size_t)k] = c;
                tot += c;
            }
            return tot;
        };
        if (!greedy) {
        double lo_t = 0.0, hi_t = 1.0;
        while (need(hi_t, nullptr) > nwg_plan && hi_t < 1e15) hi_t *= 2.0;
        for (int it = 0; it < 60; ++it) { const double mid = 0.5 * (lo_t + hi_t); if (need(mid, nullptr) > nwg_plan) lo_t = mid; else hi_t = mid; }
        std::vector<double> want((size_t)n_plan);
        need(hi_t, &want);
        std::vector<std::pair<double, int>> frac;
        for (int k = 0; k < n_plan; ++k) {
            cw[(size_t)k] = std::max(1, std::min(cap_of(k), (int)want[(size_t)k]));
            used += cw[(size_t)k];
            frac.push_back({want[(size_t)k] - (int)want[(size_t)k], k});
        }
        std::sort(frac.begin(), frac.end(), [](const std::pair<double, int> &x, const std::pair<double, int> &y) { return x.first > y.first; });
        for (int pass = 0; pass < 4 && used < nwg_plan; ++pass)            // left-over workgroups: largest remainders first
            for (size_t i = 0; i < frac.size() && used < nwg_plan; ++i)
                if (cw[(size_t)frac[i].second] < cap_of(frac[i].second)) { ++cw[(size_t)frac[i].second]; ++used; }
        while (used > nwg_plan) {                                          // (rounding can only overshoot by the floor of ones)
            int big = 0;
            for (int k = 1; k < n_plan; ++k) if (cw[(size_t)k] > cw[(size_t)big]) big = k;
            if (cw[(size_t)big] <= 1) break;
            --cw[(size_t)big]; --used;
        }
        }
        // The streams of the batch behind join the plan: one workgroup each, plus what the plan of this batch leaves,
        // dealt evenly (up to JD_BG_CW) - they are ordinary clusters from here on, only not what the launch waits for.
        const int n_tot = n_work + n_bg;
        std::vector<int> cw_all(cw);
        std::vector<double> wt_all(weight->begin(), weight->begin() + n_work);
        if (n_bg > 0) {
            if (greedy) {                                              // (the measured-curve plan deals this batch only: the rest, evenly)
                const int spare = std::max(0, nwg - used);
                cw_all.insert(cw_all.end(), (size_t)n_bg, std::max(1, std::min(mcw_bg, 1 + spare / n_bg)));
            }
            wt_all.insert(wt_all.end(), (size_t)n_bg, 0.0);            // (not on the launch's critical path)
            for (int i = 0; i < n_bg; ++i) work.push_back(make_int4(bg[(size_t)i].x, bg[(size_t)i].y, 0, 1));
        }
        const int prio_flag = n_bg > 0 ? 0x40000000 : 0;               // (SearchArgs::n_prio: which work items the launch is there for)
        // XCD-local launch (jd_search.h): every cluster inside one eighth of the grid - the clusters go, largest
        // first, into the eighth with the most room; one that fits nowhere shrinks to the room there is, and what
        // an eighth has left over in the end goes to its cluster with the latest predicted finish.  The packed plan
        // is taken if the model says it ends no more than 4 % after the unpacked one (what plain stores and L2
        // atomics are measured to be worth): a cluster squeezed into a corner is a long tail.
        const int bin = nwg_all / 8;
        if (d->xl_ok && (nwg_all & 7) == 0) {
            auto t_of = [&](int k, int c) {
                return wt_all[(size_t)k] <= 0.0 ? 0.0 : std::max(wt_all[(size_t)k], 1.0) * (a_us + b_us / std::max(c, 1));
            };
            std::vector<int> order((size_t)n_tot), pos((size_t)n_tot, 0), room(8, bin), cwx = cw_all;
            std::vector<std::vector<int>> member(8);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cwx[(size_t)x] > cwx[(size_t)y]; });
            bool fits = true;
            for (int k : order) {
                int b = 0;
                for (int q = 1; q < 8; ++q) if (room[(size_t)q] > room[(size_t)b]) b = q;
                if (room[(size_t)b] <= 0) { fits = false; break; }
                cwx[(size_t)k] = std::min(cwx[(size_t)k], room[(size_t)b]);
                member[(size_t)b].push_back(k);
                room[(size_t)b] -= cwx[(size_t)k];
            }
            double tau_plain = 0.0, tau_xl = 0.0;
            if (fits) {
                for (int b = 0; b < 8; ++b) {
                    while (room[(size_t)b] > 0 && !member[(size_t)b].empty()) {
                        int late = -1;
                        for (int k : member[(size_t)b])
                            if (cwx[(size_t)k] < (k < n_work ? mcw : std::min(d->bg_cw_cap, max_cw)) &&
                                (late < 0 || t_of(k, cwx[(size_t)k]) > t_of(late, cwx[(size_t)late]))) late = k;
                        if (late < 0) break;
                        ++cwx[(size_t)late]; --room[(size_t)b];
                    }
                    int at = b * bin;
                    for (int k : member[(size_t)b]) { pos[(size_t)k] = at; at += cwx[(size_t)k]; }
                }
                for (int k = 0; k < n_work; ++k) {
                    tau_plain = std::max(tau_plain, t_of(k, cw_all[(size_t)k]));
                    tau_xl = std::max(tau_xl, t_of(k, cwx[(size_t)k]));
                }
                if (tau_xl > d->xl_slack * tau_plain) fits = false;
            }
            if (fits) {
                for (int k = 0; k < n_tot; ++k) {
                    work[(size_t)k].z = pos[(size_t)k];
                    work[(size_t)k].w = cwx[(size_t)k] | (k < n_work ? prio_flag : 0);
                }
                // (the kernel searches by first workgroup)
                std::sort(work.begin(), work.end(), [](const int4 &x, const int4 &y) { return x.z < y.z; });
                grid = nwg_all;
                xl = true;
            }
        }
        if (!xl) {
            int first = 0;
            for (int k = 0; k < n_tot; ++k) {
                work[(size_t)k].z = first;
                work[(size_t)k].w = cw_all[(size_t)k] | (k < n_work ? prio_flag : 0);
                first += cw_all[(size_t)k];
            }
            grid = first;
        }
        if (n_bg > 0) { A.n_prio = n_work; A.n_work = n_tot; d->bg_ran = true; }
        // (two batches in flight: clusters of one or two workgroups - one chunk of items per wave, i.e. full 64-item passes,
        // does better there than two: 31.2 against 31.9 ms per step; the heavy workloads lose 3-5 % with one)
        if (n_bg > 0 && !d->xch_forced) A.C.x_chunks = 1;
        A.n_slots = 0;
        // Re-planning under way (SearchArgs::rebalance_at): the plan above makes the streams finish together only as
        // far as frames predict work; when a fifth of the grid has run out of work the launch is cut short and the
        // rest planned anew - worth it while the rest is long against the ~0.2 ms a relaunch costs.
        // (not with the batch behind beside it: a cut stops ITS streams too, every leg pays the launch's set-up again and
        // the workgroups a finished cluster leaves are few against what the batch behind keeps busy anyway - measured at
        // configs[1]: 35.4 ms per step with cuts, 30.2 without; JD_BG_REBALANCE=1 brings them back)
        rebalance_at = 0;
        if (d->rebalance && n_work >= 4 && (n_bg == 0 || d->bg_rebalance)) {
            double tau = 0.0;
            for (int k = 0; k < n_work; ++k)
                tau = std::max(tau, std::max((*weight)[(size_t)k], 1.0) * (a_us + b_us / std::max(work[(size_t)k].w & 0xffff, 1)));
            if (tau > d->rebalance_min_us) rebalance_at = std::max(1, (int)(d->rebalance_frac * grid));
        }

    } else if (d->xl_ok && A.Cw > 1 && (grid & 7) == 0 && ((grid >> 3) % A.Cw) == 0) xl = true;   // uniform clusters that tile the eighths
    HIPCHK(hipMemcpyAsync(d->d_work, work.data(), work.size() * sizeof(int4), hipMemcpyHostToDevice, st));
    A.ll = ll; A.ll_stride = ll_stride; A.f0 = f0; A.f_end = f_end;
    A.status = d->d_status; A.dbg = d->d_dbg; A.rebalance_at = rebalance_at;
    A.xl_selftest = jd_dev_env("JD_XL_SELFTEST") ? 1 : 0;                 // (test knob, see SearchArgs)
    A.resident = d->d_resident; A.launch_seq = ++d->launch_seq;
        struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } } ev0, ev1;
        HIPCHK(hipEventCreate(&ev0.e)); HIPCHK(hipEventCreate(&ev1.e));
        const hipEvent_t e0 = ev0.e, e1 = ev1.e;
        // k_search is persistent and its clusters spin at barriers of their own: ALL its workgroups have to be resident
        // at once (one per CU).  Two such launches dispatched side by side - two decoders of this process on one device,
        // driven from two host threads - could each hold part of the CUs and wait for the rest until the barriers time
        // out: launches on one device are serialised here, from dispatch to completion.  (Other PROCESSES on the device
        // are outside this lock: the dispatcher starts the workgroups of a kernel in order, and a kernel that cannot
        // become fully resident ends in JDE_BARRIER after 30 s instead of hanging.)
        if (!d->occupancy_ok) {
            // ... and the kernel must fit a CU the way the grid assumes: asked of the runtime once per decoder, for the
            // flavours it may launch (a build with other SW / WG_PER_CU / LDS sizes, or a device with smaller CUs,
            // fails here with a message instead of after a 30 s barrier time-out)
            const bool lz = d->C.lazy != nullptr;
            for (int v = 0; v < 2; ++v) {
                int per_cu = 0;
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)search_kernel(ne3, v != 0, lz), SNT, 0));
                if (per_cu < WG_PER_CU)
                    return jd_fail(JD_EHIP, "k_search needs %d workgroup(s) of %d threads resident per CU, the device takes %d: "
                                   "its clusters could not all be resident at once", WG_PER_CU, SNT, per_cu);
            }
            d->occupancy_ok = true;
        }
        const size_t dev_i = (size_t)std::min(std::max(d->device, 0), JD_MAX_DEVICES - 1);
        g_search_waiters[dev_i].fetch_add(1);
        std::unique_lock<std::mutex> search_lock(g_search_mu[dev_i]);
        g_search_waiters[dev_i].fetch_sub(1);
        g_search_turn[dev_i].fetch_add(1);
        GpuLockGuard process_lock(d->device);                              // (other processes on this GPU: see GpuFileLock)
        // (a launch beside which the next batch's table is scored is not cut short for a re-plan while that scoring runs -
        // status[4]: its blocks sit on the CUs that finished clusters left, and a relaunch would wait for them to drain)
        // Whether that pays depends on what the scoring is against the search: where it is a fifth of it (configs[1]) the
        // whole scoring fits the tail the clusters leave and the step is the one uncut launch (47.3 -> 40.9 ms; re-planned
        // at will: 46.3); where it is a few per cent (the 14 M-arc graph: 13 of 340 ms) the tail begins late, the scoring
        // would hold the re-planning up for the whole launch (348 against 343 ms serial) and is better slotted in at the
        // cuts (340).  Decided by the measured costs of this decoder's last waves.
        bool hold_replan = false;
        if (d->pf_armed && pf_wants_scoring(d)) {
            double frames_now = 0.0;
            if (weight) for (double w : *weight) frames_now += w;
            const double est_gmm = d->gmm_ms_per_row * pf_scoring_rows(d);
            const double est_search = d->search_ms_per_frame * frames_now;
            hold_replan = d->pf_rebalance == 0 || (d->pf_rebalance < 0 && est_gmm > 0.0 && est_search > 0.0 && est_gmm >= 0.1 * est_search);
        }
        // the kernel flavour: HMM size class x XCD-local x lazily composed graph
        // More streams than the chip has CUs, one workgroup each: the slot kernel as a plain launch (jd_slot.h: k_slot_batch) - a
        // workgroup per stream, two per CU, the dispatcher deals the next one when one leaves - instead of k_search's
        // one-per-CU workgroups that take their streams one after the other.  (JD_SLOT_BATCH=1 / 0, development: always / never.)
        bool slot_batch = A.n_slots > 0 && A.Cw == 1 && n_bg == 0 && n_work > nwg_all && !d->C.lazy && !A.cells;
        if (const char *e = jd_dev_env("JD_SLOT_BATCH")) slot_batch = atoi(e) != 0 && A.Cw == 1 && n_bg == 0 && !d->C.lazy && !A.cells && (A.n_slots > 0 || n_work == 1);
        if (slot_batch) {
            // The slot kernel reads lists of ITS geometry only (eight wave segments, slot_run: JDE_GEOM), k_search those of any.  A stream
            // in the middle of an utterance whose last frames were written by a cluster of several workgroups - jd_streams_push or the
            // broker's ticks served 100 streams with clusters of two, then more streams joined - stays with k_search for this launch:
            // the shape of the launch alone does not decide.  (The streams' heads as they stand behind everything queued on `st`.)
            std::vector<int> hd((size_t)d->max_streams * 10);
            HIPCHK(hipMemcpy2DAsync(hd.data(), 40, d->d_ctl, sizeof(StreamCtl), 40, (size_t)d->max_streams, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (const int2 &w : work_in) {
                const int *h = hd.data() + (size_t)w.x * 10;           // {frame, T, error, needs_init, started, lst_nw, n_rec_hint, best_emit, dirty_nw[2]}
                if (h[4] && !h[3] && h[2] == 0 && (h[5] != SW || h[8] != SW || h[9] != SW)) { slot_batch = false; break; }
            }
            if (!slot_batch && getenv("JD_VERBOSE"))
                fprintf(stderr, "k_slot_batch: a stream's lists were written by a cluster of several workgroups - this launch stays with k_search\n");
        }
        hipLaunchKernelGGL(jd_zero_bar_kernel, dim3((A.n_work + 255) / 256), dim3(256), 0, st, d->d_ctl, d->d_work, A.n_work, d->d_status,
                           hold_replan ? 1 : 0);
        HIPCHK(hipEventRecord(e0, st));
        if (slot_batch) {
            if (ne3) hipLaunchKernelGGL(k_slot_batch<3>, dim3((unsigned)n_work), dim3(SNT), 0, st, A);
            else hipLaunchKernelGGL(k_slot_batch<6>, dim3((unsigned)n_work), dim3(SNT), 0, st, A);
        } else
        hipLaunchKernelGGL(search_kernel(ne3, xl, d->C.lazy != nullptr), dim3(grid), dim3(SNT), 0, st, A);
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipGetLastError());
        if (d->pf_armed && pf_wants_scoring(d)) {
            // the next batch's table is scored beside this launch (jd_dec_prefetch_scores): its kernel is enqueued once
            // the search is resident - the last workgroup of the grid says so in a host-mapped word - so that scoring
            // blocks never sit on a CU a search workgroup is waiting for (2 ms: it goes ahead anyway)
            const auto tr0 = std::chrono::steady_clock::now();
            // (k_slot_batch: nothing has to be resident at once - its workgroups come and go - and nobody writes the word)
            while (!slot_batch && __atomic_load_n(d->h_resident, __ATOMIC_ACQUIRE) != A.launch_seq &&
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count() < 2.0) { }
            const int pr = pf_launch(d);
            if (pr) { (void)hipStreamSynchronize(st); return pr; }       // (k_search is in flight: not left behind with the launch lock released)
        }
        HIPCHK(hipMemcpyAsync(d->h_status, d->d_status, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) d->timing.search_ms += ms;
        if (getenv("JD_VERBOSE")) {                                        // development
            int cmin = 1 << 30, cmax = 0;
            for (const int4 &w : work) { cmin = std::min(cmin, w.w & 0xffff); cmax = std::max(cmax, w.w & 0xffff); }
            fprintf(stderr, "%s: %d streams, grid %d (clusters %d..%d workgroups, %s%s), frames [%d, %d): %.3f ms\n", slot_batch ? "k_slot_batch" : "k_search", n_work,
                    slot_batch ? n_work : grid, cmin, cmax, A.n_slots ? "uniform" : "weighted", xl ? ", XCD-local" : "", f0, f_end, ms);
            if (d->h_status[3]) fprintf(stderr, "          cut short for a re-plan: %d streams go on\n", d->h_status[0]);
        }
        if (d->h_status[1] != 0) {
            // a cluster of an XCD-local launch found itself on several XCDs and left its stream untouched: from
            // now on this decoder launches the agent-scope kernel (the loop below repeats the launch)
            d->xl_ok = false;
            if (getenv("JD_VERBOSE")) fprintf(stderr, "k_search: %d cluster(s) not on one XCD - agent-scope launches from here on\n", d->h_status[1]);
        }
        d->timing.search_launches += 1;
        if (slot_batch) d->timing.slot_launches += 1;
        d->timing.cluster_wgs = A.Cw;
        if (d->h_status[0] == 0 && d->h_status[1] == 0) break;
        d->timing.relaunches += 1;
        if (it >= max_rounds) return jd_fail(JD_ENOMEM, "Path arena too small: no progress after %d garbage collections", it);
        // Some streams stopped for a collection of their Path records (k_gc_*: no-ops for the streams below
        // their mark): the launch is repeated for the streams that are not through, with clusters sized for
        // what each of them still has ahead.
        // (not when every stop was for a re-plan, or a stream ahead of its turn stopping with the launch)
        if (d->h_status[0] > d->h_status[3] + d->h_status[6]) {
            launch_gc(d->C, d->d_ctl, d->d_streams, d->d_work, A.n_work, 0, ne3, d->n_cus, st);
            HIPCHK(hipGetLastError());
            // PARTIAL_DECODING: the trace rides on the collection (:362-368) - the caller has to see the stream as it
            // stands right after one (jd_stream_push; it goes on from there)
            if (d->return_on_collect) { HIPCHK(hipStreamSynchronize(st)); d->collected_now = true; return JD_OK; }
        }
        // A stream that stopped for a collection after n frames will, by and large, stop again after as many
        // (its records per frame change slowly): what it has ahead IN THE NEXT LAUNCH is the smaller of that
        // and the frames it has left - sized by that, the streams stop together instead of idling.
        std::vector<int> head((size_t)d->max_streams * 4);
        HIPCHK(hipMemcpy2D(head.data(), 16, d->d_ctl, sizeof(StreamCtl), 16, (size_t)d->max_streams, hipMemcpyDeviceToHost));
        std::vector<int2> rest;
        weight_now.clear();
        if (d->load_scale == 1.0) { const int lr = learn_load(d, work_in); if (lr) return lr; }   // first batch of this decoder
        if (frame_before.empty()) frame_before.assign((size_t)d->max_streams, f0);
        for (const int2 &w : work_in) {
            const int *h = head.data() + (size_t)w.x * 4;              // {frame, T, error, needs_init}
            const int end = std::min(h[1], f_end), left = end - h[0];
            const int done = h[0] - frame_before[(size_t)w.x];
            frame_before[(size_t)w.x] = h[0];
            // (a launch cut short for a re-plan says nothing about when a stream's arena fills up: then the frames left count)
            if (left > 0 && h[2] == 0) {
                rest.push_back(w);
                weight_now.push_back((double)((done > 0 && d->h_status[3] == 0) ? std::min(left, done) : left));
            }
        }
        if (rest.empty()) break;
        work_in.swap(rest);
        weight = weight_first ? &weight_now : nullptr;
    }
    return JD_OK;
}
