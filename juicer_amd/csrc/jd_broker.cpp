// jd_broker.cpp - many IDecoder instances on the streams of ONE decoder.
//
// The reference's harness is serial: DecoderBatchTest::run decodes one list entry after the other through one
// IDecoder (src/DecoderBatchTest.cpp:738-771 -> DecoderSingleTest::decodeUtterance, src/DecoderSingleTest.cpp:259-324),
// and its users scale out by running several such processes over split file lists (doc/userman/juicer_userman.tex:584).
// A GPU decoder serves one utterance at a few hundred times real time but a batch of them at ten thousand: the
// search is a chain of dependent steps per frame, and the chip is filled by running many utterances side by side.
// The broker is what lets N serial harness threads do that without knowing of each other: every thread drives its own
// client with the IDecoder protocol - init, push (processFrame x n), finish - and a worker thread turns whatever has
// been pushed since its last tick into ONE scoring launch and ONE persistent search launch over all the streams
// concerned (jd_streams_push).  Clients never touch the decoder: only the worker thread does, so the decoder's
// single-caller rule holds.  Host code over the C ABI only - nothing here knows about HIP.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "jd_internal.h"

namespace {
struct Client {
    bool open = false;
    bool want_init = false, want_finish = false;   // requests the worker has not served yet
    bool inited = false;                           // between init and finish
    std::vector<float> pending;                    // frames pushed and not yet handed to the decoder
    int err = JD_OK;                               // first error of a tick that concerned this client (reported by its next call)
    std::string errmsg;
    bool failed = false;                           // its init or its stream failed: every call up to the next init() reports err
    jd_hyp result;                                 // of the last finish (arrays owned by the decoder, valid until the stream's next init)
    // the resident kernel's worker (broker_loop_resident): chunks of this client's frames that are scored (or being scored)
    // and not yet posted - at most two, one per likelihood buffer - and the one its cluster is running
    int staged_buf[2] = {0, 0}, staged_n[2] = {0, 0}, n_staged = 0;
    bool running = false; int run_buf = 0;
    bool fresh = false;                            // initialised and nothing posted yet (recognitionStart is still to run)
    bool finishing = false;                        // its result is being fetched (the finisher thread)
    unsigned init_req = 0;                         // counts the client's init() calls: one that arrives while the worker serves
                                                   // another (outside the lock) is still to be served behind it
    bool in_flight = false;                        // the worker holds something of this client outside the lock (frames being
                                                   // staged, a tick it is part of): close waits for that - the slot is somebody else's after it
    std::chrono::steady_clock::time_point t_post, t_idle;   // (statistics) its last command posted / found through
    bool was_idle = false;
    std::vector<float> taken;                      // frames on their way into a likelihood buffer
};
}  // namespace

struct jd_broker {
    jd_dec *dec = nullptr;
    int D = 0, n_clients = 0;
    int max_tick_frames = 192;                     // frames of one client a tick takes at most
    int max_pending_frames = 1024;                 // a push waits while its client holds more than this
    int coalesce_us = 300;                         // a tick waits this long for the other open clients' frames
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::thread worker;
    bool stop = false;
    std::vector<Client> clients;
    jd_broker_stats stats{};
    bool resident = false;                         // the worker drives the resident search kernel (jd_res_*) instead of ticks
    int fail_init_of = -1;                         // test hook (JD_DEV=1 JD_BROKER_FAIL_INIT=<client>): that client's inits fail
    // results are fetched by a thread of their own: recognitionFinish's kernel, a synchronisation of the side stream and
    // three copies back are a third of a millisecond in which the worker would post nothing to anybody
    std::thread finisher;
    std::deque<int> fin_q;
    std::condition_variable cv_fin;
};

static int client_error(jd_broker *b, Client &c)
{
    if (c.err == JD_OK) return JD_OK;
    const int e = c.err;
    if (!c.failed) c.err = JD_OK;                  // (a failed init / a failed stream: reported until the next init())
    (void)b;
    return jd_fail(e, "%s", c.errmsg.c_str());
}
// An init that failed, or a stream that failed on the device: nothing more can be done for the utterance - what the client has
// pushed is dropped and whoever waits (a push for room, a finish for its result) is woken with the error.  (lk held)
static void client_failed(jd_broker *b, Client &c, int rc, const std::string &msg)
{
    if (c.err == JD_OK) { c.err = rc; c.errmsg = msg; }
    c.failed = true;
    c.pending.clear(); c.n_staged = 0;
    c.want_finish = false;
    b->cv_done.notify_all();
}

static void broker_loop(jd_broker *b)
{
    std::unique_lock<std::mutex> lk(b->mu);
    std::vector<int> inits, pushers, finishers;
    std::vector<std::vector<float>> taken((size_t)b->n_clients);
    std::vector<unsigned> init_seen((size_t)b->n_clients, 0u);
    double search_ms_seen = 0.0;                   // (jd_timing accumulates over the streaming calls)
    for (;;) {
        auto has_work = [&]() {
            if (b->stop) return true;
            for (const Client &c : b->clients)
                if (c.open && (c.want_init || c.want_finish || (c.inited && !c.pending.empty()))) return true;
            return false;
        };
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto us_since = [&](std::chrono::steady_clock::time_point t) {
            return (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(now() - t).count();
        };
        auto t_mark = now();
        b->cv_work.wait(lk, has_work);
        if (b->stop) return;
        b->stats.us_idle += us_since(t_mark); t_mark = now();
        // a tick is the fuller the more clients have frames waiting: give the open ones that have nothing pending yet
        // a moment to deliver - they are all being fed at about the same rate, and one that has just been handed its
        // result is about to start its next utterance (a caller that is through closes its client)
        if (b->coalesce_us > 0) {
            auto all_ready = [&]() {
                if (b->stop) return true;
                for (const Client &c : b->clients)
                    if (c.open && !c.want_finish && c.pending.empty()) return false;
                return true;
            };
            b->cv_work.wait_for(lk, std::chrono::microseconds(b->coalesce_us), all_ready);
            if (b->stop) return;
            b->stats.us_coalesce += us_since(t_mark);
        }
        inits.clear(); pushers.clear(); finishers.clear();
        for (int i = 0; i < b->n_clients; ++i) {
            Client &c = b->clients[(size_t)i];
            if (!c.open) continue;
            if (c.want_init) { inits.push_back(i); init_seen[(size_t)i] = c.init_req; }
            // (frames pushed behind an init that has not been served yet go with this tick too: the init comes first)
            if ((c.inited || c.want_init) && !c.pending.empty()) {
                const size_t have = c.pending.size() / (size_t)b->D;
                const size_t take = std::min(have, (size_t)b->max_tick_frames);
                taken[(size_t)i].assign(c.pending.begin(), c.pending.begin() + (ptrdiff_t)(take * b->D));
                c.pending.erase(c.pending.begin(), c.pending.begin() + (ptrdiff_t)(take * b->D));
                pushers.push_back(i);
            }
        }
        for (int i = 0; i < b->n_clients; ++i) {
            Client &c = b->clients[(size_t)i];
            if (c.open && c.want_finish && c.pending.empty()) finishers.push_back(i);   // (its last frames - and its init - go with this tick)
        }
        for (int i : inits) b->clients[(size_t)i].in_flight = true;
        for (int i : pushers) b->clients[(size_t)i].in_flight = true;
        for (int i : finishers) b->clients[(size_t)i].in_flight = true;
        lk.unlock();
        b->cv_done.notify_all();                                       // (pushes that waited for room)
        // ---- the decoder is touched from here only
        long long tick_frames = 0, tick_streams = 0;
        std::vector<int> rc_of((size_t)b->n_clients, JD_OK);
        std::vector<std::string> msg_of((size_t)b->n_clients);
        int64_t us_init = 0, us_push = 0, us_finish = 0, us_search = 0;
        t_mark = now();
        for (int i : inits) {
            rc_of[(size_t)i] = i == b->fail_init_of ? jd_fail(JD_EHIP, "jd_broker: init failure injected for client %d", i) : jd_stream_init(b->dec, i);
            if (rc_of[(size_t)i]) msg_of[(size_t)i] = jd_last_error();
        }
        us_init = us_since(t_mark); t_mark = now();
        if (!pushers.empty()) {
            std::vector<int32_t> ss, nn;
            std::vector<const float *> ff;
            long long fr = 0;
            for (int i : pushers) {
                if (rc_of[(size_t)i] != JD_OK) continue;               // (its init failed)
                ss.push_back(i); ff.push_back(taken[(size_t)i].data()); nn.push_back((int32_t)(taken[(size_t)i].size() / (size_t)b->D));
                fr += nn.back();
            }
            const int rc = ss.empty() ? JD_OK : jd_streams_push(b->dec, (int32_t)ss.size(), ss.data(), ff.data(), nn.data());
            if (rc) {
                const std::string m = jd_last_error();
                for (int i : pushers) if (rc_of[(size_t)i] == JD_OK) { rc_of[(size_t)i] = rc; msg_of[(size_t)i] = m; }
            }
            tick_frames = fr; tick_streams = (long long)ss.size();
            jd_timing tm;
            if (!ss.empty() && jd_dec_last_timing(b->dec, &tm) == JD_OK) {
                us_search = (int64_t)((tm.search_ms - search_ms_seen) * 1e3);
                search_ms_seen = tm.search_ms;
            }
        }
        us_push = us_since(t_mark); t_mark = now();
        std::vector<jd_hyp> res((size_t)b->n_clients);
        auto serve_finishes = [&](const std::vector<int> &who) {
            for (int i : who) {
                memset(&res[(size_t)i], 0, sizeof(jd_hyp));
                const int rc = jd_stream_finish(b->dec, i, &res[(size_t)i]);
                if (rc && rc_of[(size_t)i] == JD_OK) { rc_of[(size_t)i] = rc; msg_of[(size_t)i] = jd_last_error(); }
            }
        };
        serve_finishes(finishers);
        {   // a caller whose last frames went with this tick and who asked for its result while the search ran is served
            // now - not at the end of the next tick, which it would sit out
            std::vector<int> late;
            lk.lock();
            for (int i = 0; i < b->n_clients; ++i) {
                const Client &c = b->clients[(size_t)i];
                if (c.open && c.want_finish && c.pending.empty() && !c.want_init && c.inited && rc_of[(size_t)i] == JD_OK &&
                    std::find(finishers.begin(), finishers.end(), i) == finishers.end())
                    late.push_back(i);
            }
            lk.unlock();
            serve_finishes(late);
            finishers.insert(finishers.end(), late.begin(), late.end());
        }
        us_finish = us_since(t_mark);
        lk.lock();
        b->stats.us_init += us_init; b->stats.us_push += us_push; b->stats.us_finish += us_finish;
        b->stats.us_search += us_search;
        if (tick_streams) { b->stats.ticks += 1; b->stats.frames += tick_frames; b->stats.stream_ticks += tick_streams; }
        for (int i = 0; i < b->n_clients; ++i) {
            Client &c = b->clients[(size_t)i];
            if (rc_of[(size_t)i] != JD_OK && c.err == JD_OK) { c.err = rc_of[(size_t)i]; c.errmsg = msg_of[(size_t)i]; }
        }
        for (int i : inits) {
            Client &c = b->clients[(size_t)i];
            if (c.init_req == init_seen[(size_t)i]) c.want_init = false;   // (else: init() again meanwhile - the next tick's)
            c.inited = rc_of[(size_t)i] == JD_OK;
            // (a failed init: the frames behind it and a finish that waits for them would wait for ever - pushers need inited, finishers
            // an empty queue)
            if (!c.inited && !c.want_init) client_failed(b, c, rc_of[(size_t)i], msg_of[(size_t)i]);
        }
        for (int i : finishers) { Client &c = b->clients[(size_t)i]; c.want_finish = false; c.inited = false; c.result = res[(size_t)i]; }
        for (Client &c : b->clients) c.in_flight = false;
        b->cv_done.notify_all();
    }
}


// The worker over the RESIDENT search kernel (jd_resident.h): no ticks.  Every client's stream has a cluster of its own in a
// kernel that stays; its frames are scored into one of the stream's two likelihood buffers as they come (on the side
// stream, beside the search) and posted to the cluster as soon as it is through with the chunk before - every stream at
// its own pace.  init, finish and the Path collections are small kernels on the side stream between two commands.
static void broker_loop_resident(jd_broker *b)
{
    std::unique_lock<std::mutex> lk(b->mu);
    auto now = []() { return std::chrono::steady_clock::now(); };
    std::vector<int> st_s, st_b, st_n;                                // this round's staging: streams, buffers, frames
    std::vector<const float *> st_f;
    bool on = false;
    bool draining = false;                                            // somebody waits for the device: no new commands, then the kernel makes room
    auto idle_since = now(), on_since = now();
    auto fail_all = [&](int rc, const std::string &msg) {              // (lk held)
        for (Client &c : b->clients)
            if (c.open && c.err == JD_OK) { c.err = rc; c.errmsg = msg; }
        for (Client &c : b->clients) { c.want_init = false; c.want_finish = false; c.pending.clear(); c.n_staged = 0; c.running = false; }
        b->cv_done.notify_all();
    };
    for (;;) {
        bool active = false;
        for (const Client &c : b->clients)
            if (c.open && (c.want_init || c.want_finish || c.running || c.finishing || c.n_staged > 0 || (c.inited && !c.pending.empty()))) {
                active = true;
                break;
            }
        if (b->stop) { if (on) { lk.unlock(); (void)jd_res_stop(b->dec); lk.lock(); } return; }
        if (!active) {
            // nothing to do: the kernel leaves the chip after a few milliseconds (other decoders, other processes, a
            // caller's device-wide synchronisation all wait for it) and comes back with the next request
            if (on && std::chrono::duration<double, std::milli>(now() - idle_since).count() > 3.0) {
                lk.unlock(); (void)jd_res_stop(b->dec); lk.lock();
                on = false;
                continue;
            }
            if (on) b->cv_work.wait_for(lk, std::chrono::microseconds(200));
            else b->cv_work.wait(lk);
            continue;
        }
        if (!on) {
            lk.unlock();
            const int rc = jd_res_start(b->dec, b->n_clients, b->max_tick_frames);
            const std::string m = rc ? jd_last_error() : "";
            lk.lock();
            if (rc) { fail_all(rc, m); continue; }
            on = true; draining = false; on_since = now();
        }
        // another decoder's launch (or another broker's kernel) of this process waits for the device: once this kernel has had
        // 20 ms, the chunks that are running run out, the kernel leaves, and comes back behind the one that waited
        if (!draining && jd_res_should_yield(b->dec) && std::chrono::duration<double, std::milli>(now() - on_since).count() > 20.0) draining = true;
        if (draining) {
            bool any_running = false;
            for (const Client &c : b->clients) any_running = any_running || c.running || c.finishing;
            if (!any_running) {
                lk.unlock(); (void)jd_res_yield(b->dec); lk.lock();
                on = false; draining = false;
                continue;
            }
        }
        bool progress = false;
        auto try_post = [&](int i) {                                   // (lk held)
            Client &c = b->clients[(size_t)i];
            int rc = JD_OK;
            // 4. the next chunk to the cluster (a finish without frames still runs recognitionStart: a chunk of none)
            if (!draining && !c.running && c.inited && (c.n_staged > 0 || (c.fresh && c.want_finish && c.pending.empty()))) {
                const int buf = c.n_staged > 0 ? c.staged_buf[0] : 0, nf = c.n_staged > 0 ? c.staged_n[0] : 0;
                lk.unlock();
                rc = jd_res_post(b->dec, i, buf, nf);
                const std::string m = rc ? jd_last_error() : "";
                lk.lock();
                if (rc && c.err == JD_OK) { c.err = rc; c.errmsg = m; }
                if (c.n_staged > 0) { c.staged_buf[0] = c.staged_buf[1]; c.staged_n[0] = c.staged_n[1]; c.n_staged -= 1; }
                if (!rc) {
                    c.running = true; c.run_buf = buf; c.fresh = false; b->stats.ticks += 1; b->stats.frames += nf; b->stats.stream_ticks += 1;
                    c.t_post = now();
                    // (statistics: what a cluster waited between two chunks of ONE utterance)
                    if (c.was_idle) b->stats.us_idle += (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(c.t_post - c.t_idle).count();
                }
                progress = true;
            }
        };
        auto try_finish = [&](int i) {                                 // (lk held)
            Client &c = b->clients[(size_t)i];
            // 5. IDecoder::finish (handed to the finisher thread)
            if (!c.running && c.n_staged == 0 && c.inited && !c.fresh && c.want_finish && c.pending.empty()) {
                c.finishing = true;
                b->fin_q.push_back(i);
                b->cv_fin.notify_one();
                progress = true;
            }
        };
        for (int i = 0; i < b->n_clients; ++i) {
            Client &c = b->clients[(size_t)i];
            if (!c.open || c.finishing) continue;
            int rc = JD_OK;
            // 1. where its cluster stands
            if (c.running) {
                int idle = 0, frame = 0, err = 0, stopped = 0;
                lk.unlock();
                rc = jd_res_poll(b->dec, i, &idle, &frame, &err, &stopped);
                if (rc == JD_OK && idle && stopped) { rc = jd_res_collect(b->dec, i); idle = 0; progress = true; }
                const std::string m = rc ? jd_last_error() : "";
                lk.lock();
                if (rc) { fail_all(rc, m); on = false; lk.unlock(); (void)jd_res_stop(b->dec); lk.lock(); break; }
                if (idle && err != 0 && !c.failed) {
                    // the stream failed on the device (an arena overflow, Histogram's ceiling, a lost workgroup): no more chunks are
                    // scored or posted for it, the client's next call says why (its finish still fetches - and reports - the same)
                    lk.unlock();
                    const int er = jd_res_stream_error(b->dec, i, err, frame);
                    const std::string em = jd_last_error();
                    lk.lock();
                    const bool wf = c.want_finish;
                    client_failed(b, c, er, em);
                    c.want_finish = wf;                                  // (a finish under way goes through try_finish as usual)
                }
                if (idle) {
                    c.running = false; progress = true; b->cv_done.notify_all();
                    c.t_idle = now(); c.was_idle = true;
                    b->stats.us_search += (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(c.t_idle - c.t_post).count();
                }
            }
            // 2. IDecoder::init
            if (!c.running && c.n_staged == 0 && c.want_init) {
                const unsigned req = c.init_req;
                lk.unlock();
                rc = i == b->fail_init_of ? jd_fail(JD_EHIP, "jd_broker: init failure injected for client %d", i) : jd_res_init(b->dec, i);
                const std::string m = rc ? jd_last_error() : "";
                lk.lock();
                if (c.init_req == req) c.want_init = false;           // (else: init() again meanwhile - served in the next round)
                c.inited = rc == JD_OK; c.fresh = rc == JD_OK; c.was_idle = false;
                if (rc && !c.want_init) client_failed(b, c, rc, m);    // (a push that waits for room, a finish: woken with the error)
                else if (rc && c.err == JD_OK) { c.err = rc; c.errmsg = m; }
                b->cv_done.notify_all();
                progress = true;
            }
            try_post(i);                                                // (what is scored already goes first: a word in host memory)
            // 3. frames for a free likelihood buffer: taken here, scored below - one launch for all the streams of this round
            if (c.inited && !c.failed && !c.want_init && !c.pending.empty() && c.n_staged + (c.running ? 1 : 0) < 2) {
                int buf = 0;
                if (c.running && c.run_buf == 0) buf = 1;
                if (c.n_staged == 1 && c.staged_buf[0] == buf) buf ^= 1;
                if (!(c.running && c.run_buf == buf)) {
                    const size_t have = c.pending.size() / (size_t)b->D;
                    const size_t take = std::min(have, (size_t)b->max_tick_frames);
                    c.taken.assign(c.pending.begin(), c.pending.begin() + (ptrdiff_t)(take * b->D));
                    c.pending.erase(c.pending.begin(), c.pending.begin() + (ptrdiff_t)(take * b->D));
                    st_s.push_back(i); st_b.push_back(buf); st_f.push_back(c.taken.data()); st_n.push_back((int)take);
                    c.in_flight = true;
                }
            }
        }
        if (!st_s.empty()) {
            lk.unlock();
            b->cv_done.notify_all();                                   // (pushes that waited for room)
            const auto t_st = now();
            const int rc = jd_res_stage_many(b->dec, (int)st_s.size(), st_s.data(), st_b.data(), st_f.data(), st_n.data());
            const std::string m = rc ? jd_last_error() : "";
            lk.lock();
            b->stats.us_push += (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(now() - t_st).count();
            for (size_t k = 0; k < st_s.size(); ++k) {
                Client &c = b->clients[(size_t)st_s[k]];
                if (rc) { if (c.err == JD_OK) { c.err = rc; c.errmsg = m; } }
                else { c.staged_buf[c.n_staged] = st_b[k]; c.staged_n[c.n_staged] = st_n[k]; c.n_staged += 1; }
                c.in_flight = false;
            }
            b->cv_done.notify_all();
            st_s.clear(); st_b.clear(); st_f.clear(); st_n.clear();
            progress = true;
        }
        for (int i = 0; i < b->n_clients; ++i) {
            const Client &c = b->clients[(size_t)i];
            if (!c.open || c.finishing) continue;
            try_post(i);
            try_finish(i);
        }
        if (progress) idle_since = now();
        else {
            // (the clusters report through host-mapped words: looking again costs nothing on the device)
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            lk.lock();
            idle_since = now();
        }
    }
}

static void broker_finisher(jd_broker *b)
{
    std::unique_lock<std::mutex> lk(b->mu);
    for (;;) {
        b->cv_fin.wait(lk, [&]() { return b->stop || !b->fin_q.empty(); });
        if (b->fin_q.empty()) return;                                  // (stop, and nothing left to hand out)
        const int i = b->fin_q.front();
        b->fin_q.pop_front();
        jd_hyp res;
        memset(&res, 0, sizeof res);
        lk.unlock();
        const auto t_f = std::chrono::steady_clock::now();
        const int rc = jd_res_finish(b->dec, i, &res);
        const std::string m = rc ? jd_last_error() : "";
        lk.lock();
        b->stats.us_finish += (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_f).count();
        Client &c = b->clients[(size_t)i];
        if (rc && c.err == JD_OK) { c.err = rc; c.errmsg = m; }
        c.want_finish = false; c.inited = false; c.finishing = false; c.result = res;
        b->cv_done.notify_all();
        b->cv_work.notify_all();
    }
}

extern "C" int jd_broker_create(jd_broker **out, jd_dec *dec, int32_t n_clients)
{
    if (!out || !dec || n_clients < 1) return jd_fail(JD_EINVAL, "jd_broker_create: bad argument");
    int32_t ms = 0, D = 0;
    int rc = jd_dec_info(dec, &ms, &D);
    if (rc) return rc;
    if (n_clients > ms) return jd_fail(JD_EINVAL, "jd_broker_create: %d clients on a decoder of %d streams", n_clients, ms);
    jd_broker *b = new jd_broker();
    b->dec = dec; b->D = D; b->n_clients = n_clients;
    b->clients.resize((size_t)n_clients);
    if (const char *e = jd_dev_env("JD_BROKER_TICK_FRAMES")) { const int v = atoi(e); if (v >= 1 && v <= 65536) b->max_tick_frames = v; }
    if (const char *e = jd_dev_env("JD_BROKER_COALESCE_US")) { const int v = atoi(e); if (v >= 0 && v <= 1000000) b->coalesce_us = v; }
    b->max_pending_frames = 4 * b->max_tick_frames;
    // the resident search kernel instead of ticks (JD_BROKER_RESIDENT=0: ticks): not with a lazily composed network or
    // partial traces - jd_res_start says so and the clients' first calls would fail, so those decoders keep the ticks
    // (up to 64 clients: the ready list of a scoring launch and the chip's room for clusters and their scoring side by side)
    b->resident = n_clients <= 64;
    if (const char *e = jd_dev_env("JD_BROKER_RESIDENT")) b->resident = atoi(e) != 0;
    if (const char *e = jd_dev_env("JD_BROKER_FAIL_INIT")) b->fail_init_of = atoi(e);
    if (b->resident && !jd_dev_env("JD_BROKER_TICK_FRAMES")) {            // (whole scoring tiles)
        b->max_tick_frames = 256;
        b->max_pending_frames = 4 * b->max_tick_frames;
    }
    if (b->resident && jd_res_start(dec, n_clients, b->max_tick_frames) != JD_OK) b->resident = false;
    else if (b->resident) (void)jd_res_stop(dec);                      // (it comes back with the first request)
    b->worker = b->resident ? std::thread(broker_loop_resident, b) : std::thread(broker_loop, b);
    if (b->resident) b->finisher = std::thread(broker_finisher, b);
    *out = b;
    return JD_OK;
}

extern "C" void jd_broker_destroy(jd_broker *b)
{
    if (!b) return;
    { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; }
    b->cv_work.notify_all(); b->cv_done.notify_all(); b->cv_fin.notify_all();
    if (b->finisher.joinable()) b->finisher.join();
    if (b->worker.joinable()) b->worker.join();
    delete b;
}

extern "C" int jd_broker_open(jd_broker *b, int32_t *client)
{
    if (!b || !client) return jd_fail(JD_EINVAL, "jd_broker_open: bad argument");
    std::lock_guard<std::mutex> lk(b->mu);
    for (int i = 0; i < b->n_clients; ++i)
        if (!b->clients[(size_t)i].open) { b->clients[(size_t)i] = Client(); b->clients[(size_t)i].open = true; *client = i; return JD_OK; }
    return jd_fail(JD_ESTATE, "jd_broker_open: all %d clients are taken", b->n_clients);
}

extern "C" int jd_broker_close(jd_broker *b, int32_t client)
{
    if (!b || client < 0 || client >= b->n_clients) return jd_fail(JD_EINVAL, "jd_broker_close: bad client");
    std::unique_lock<std::mutex> lk(b->mu);
    Client &c = b->clients[(size_t)client];
    if (!c.open) return jd_fail(JD_ESTATE, "jd_broker_close: client %d is not open", client);
    // (an utterance that was never finished: its frames are dropped; the stream's next init clears what it left)
    c.pending.clear(); c.want_finish = false;
    // (... and what its cluster still has of them runs out first: the stream is somebody else's after this)
    b->cv_work.notify_all();
    b->cv_done.wait(lk, [&]() { return b->stop || (!c.want_init && !c.running && c.n_staged == 0 && !c.in_flight && !c.finishing); });
    c.pending.clear(); c.want_finish = false; c.inited = false; c.open = false;
    return JD_OK;
}

extern "C" int jd_broker_init(jd_broker *b, int32_t client)
{
    if (!b || client < 0 || client >= b->n_clients) return jd_fail(JD_EINVAL, "jd_broker_init: bad client");
    std::unique_lock<std::mutex> lk(b->mu);
    Client &c = b->clients[(size_t)client];
    if (!c.open) return jd_fail(JD_ESTATE, "jd_broker_init: client %d is not open", client);
    // (an init that is still waiting to be served - init() twice - is this one)
    c.pending.clear(); c.want_finish = false; c.err = JD_OK; c.failed = false;   // (init() in the middle of an utterance drops it, as the reference does)
    c.want_init = true;
    c.init_req += 1;
    // nobody waits for the worker here: the stream is initialised at the head of the next tick, in front of whatever
    // frames this client has pushed by then (an error of it comes back with the next call)
    b->cv_work.notify_all();
    return JD_OK;
}

extern "C" int jd_broker_push(jd_broker *b, int32_t client, const float *frames, int32_t n_frames)
{
    if (!b || client < 0 || client >= b->n_clients || n_frames < 0 || (n_frames > 0 && !frames))
        return jd_fail(JD_EINVAL, "jd_broker_push: bad argument");
    std::unique_lock<std::mutex> lk(b->mu);
    Client &c = b->clients[(size_t)client];
    if (!c.open) return jd_fail(JD_ESTATE, "jd_broker_push: client %d is not open", client);
    int rc = client_error(b, c);                                       // (a failed init comes back here, not as "not between init and finish")
    if (rc) return rc;
    if (!(c.inited || c.want_init) || c.want_finish)
        return jd_fail(JD_ESTATE, "jd_broker_push: client %d is not between init and finish", client);
    b->cv_done.wait(lk, [&]() { return b->stop || c.failed || (int)(c.pending.size() / (size_t)b->D) <= b->max_pending_frames; });
    rc = client_error(b, c);                                           // (... or here, when it failed while this push waited for room)
    if (rc) return rc;
    c.pending.insert(c.pending.end(), frames, frames + (size_t)n_frames * (size_t)b->D);
    b->cv_work.notify_all();
    return JD_OK;
}

extern "C" int jd_broker_finish(jd_broker *b, int32_t client, jd_hyp *out)
{
    if (!b || client < 0 || client >= b->n_clients || !out) return jd_fail(JD_EINVAL, "jd_broker_finish: bad argument");
    std::unique_lock<std::mutex> lk(b->mu);
    Client &c = b->clients[(size_t)client];
    if (!c.open) return jd_fail(JD_ESTATE, "jd_broker_finish: client %d is not open", client);
    if (c.failed && !c.inited) return client_error(b, c);              // (its init failed: there is no utterance to finish)
    if (!(c.inited || c.want_init)) return jd_fail(JD_ESTATE, "jd_broker_finish: client %d is not between init and finish", client);
    c.want_finish = true;
    b->cv_work.notify_all();
    b->cv_done.wait(lk, [&]() { return b->stop || !c.want_finish; });
    const int rc = client_error(b, c);
    if (rc) return rc;
    *out = c.result;
    return JD_OK;
}

extern "C" int jd_broker_get_stats(jd_broker *b, jd_broker_stats *out)
{
    if (!b || !out) return jd_fail(JD_EINVAL, "jd_broker_get_stats: bad argument");
    std::lock_guard<std::mutex> lk(b->mu);
    *out = b->stats;
    out->resident = b->resident ? jd_res_cluster(b->dec) : 0;
    if (b->resident) {                                                 // (the clusters' own time on their chunks; Path collections)
        out->us_coalesce = jd_res_run_us(b->dec);
        out->us_init = jd_res_collections(b->dec);
    }
    return JD_OK;
}
