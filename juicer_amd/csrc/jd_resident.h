// jd_resident.h - the search kernel that STAYS (included by jd_device.hip behind jd_search.h; gfx950 only).
//
// k_search is launched per call and lives as long as the longest stream of the call has frames.  A broker that serves
// many serial IDecoder callers (jd_broker.cpp) pays for that with ticks: every launch waits for its longest stream,
// streams whose caller is between two utterances sit a launch out, and the host's share of a tick is time the chip
// idles.  k_resident is the same per-stream machinery (run_stream: one cluster of workgroups per stream, phases and
// cluster barriers as in k_search) under a loop that takes its work from a MAILBOX per stream: the host posts "frames
// up to T are scored, their rows start at this slot", the stream's cluster runs them and reports where it stands in a
// host-mapped word - every stream at its own
// pace, no common launch to wait for.  (Round 4, second cut: the command is a word the host writes into host-mapped memory
// - a post kernel on the side stream waited behind other streams' scoring launches, 0.9 ms per chunk with sixteen streams -
// and what has to have happened on the side stream before it may start is a number the side stream counts up, ready[s].)
// Between two commands other kernels touch the stream's state (recognitionStart's
// mark, the Path collection, recognitionFinish's walk, the scoring of the next rows): a command begins with an acquire
// at agent scope (vector L1 and stale L2 lines dropped, the scalar cache too) and ends with a release before anybody is told.
#pragma once

struct __align__(64) ResPost {  // host-mapped, one per stream: the HOST writes a command here (no kernel launch in a post)
    unsigned long long word;   // (sequence number << 32) | (likelihood slot as an unsigned 32-bit word): written last
    int T;                     // frames available with this command (StreamCtl::T)
    unsigned ready_id;         // the command may start once the side stream has come this far for the stream (ready[s]): the
                               // scoring of its rows, recognitionStart's mark, a Path collection
    int exit_req;              // != 0: leave the kernel
    int init;                  // != 0: a new utterance begins with this command (IDecoder::init: what jd_mark_init_kernel does,
                               // done by the cluster's first workgroup - the batch pipeline, jd_pipe_*)
    int pad[10];
};
struct __align__(128) ResMail { // device memory, one per stream: workgroup 0 of the cluster passes the command on to the others
    unsigned long long word;
    int exit_req;
    int pad[29];
};
struct ResDone {               // host-mapped, one per stream: written by workgroup 0 of the stream's cluster behind every command
    unsigned seq;              // the command that is through (written last)
    int frame;                 // StreamCtl::frame behind it: < T when the stream stopped for a Path collection
    int error;                 // StreamCtl::error
    int left;                  // 1: the cluster has left the kernel (exit request, or nobody posted anything for RES_IDLE_TICKS)
    long long run_ticks;       // (statistics) 100 MHz ticks the cluster spent on the command
};
#define RES_IDLE_TICKS 500000000LL      // 5 s at 100 MHz without a command AND without a sign of life from the host: the kernel ends by itself

// "the side stream has come this far": enqueued behind a scoring launch / a mark / a collection, for the streams concerned
struct ReadyList { int n; int s[64]; unsigned id[64]; };
__global__ void jd_res_ready_kernel(unsigned *ready, ReadyList L)
{
    const int i = threadIdx.x;
    if (i < L.n) __hip_atomic_store(&ready[L.s[i]], L.id[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void jd_res_reset_kernel(StreamCtl *ctl, ResMail *mail, unsigned *ready, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) {
        mail[s].word = 0ULL; mail[s].exit_req = 0; ready[s] = 0u;
        ctl[s].bar = 0u; ctl[s].xbar = 0u; ctl[s].xmask = 0u; ctl[s].stop_req = 0;
    }
}

// grid = n_streams x Cw workgroups: workgroup b serves stream b / Cw as member b % Cw of its cluster (agent-scope flavour:
// nothing is assumed about where the workgroups run).  All of them resident at once, like k_search's.
// XL: the workgroup-scope flavour of the search's memory operations (plain stores, atomics performed in the XCD's L2 - see
// jd_search.h) - for clusters of ONE workgroup, which sit on one XCD by definition; a command's closing release writes
// the L2 back for the kernels beside it, its opening acquire drops what they have made stale.
template <int NE, bool XL>
// beat: a host-mapped word the host counts up whenever it looks after the kernel (jd_res_poll / jd_res_post / the batch pipeline's pump).  A cluster
// without a command does not leave while that word moves: one caller of a broker may pause for as long as it likes while the others keep the
// kernel busy (round 4 left after 5 s without a command for ITS stream, and the paused caller's next command was never served).  It leaves when
// the host has given no sign of life for RES_IDLE_TICKS either - a caller that went away between two calls, a process that died - because the
// kernel holds the device's search lock and the GPU's file lock.
__global__ JD_KBOUNDS void k_resident(SearchArgs A, const ResPost *post, ResMail *mail, const unsigned *ready, ResDone *done, int Cw, const unsigned *beat)
{
    __shared__ SearchShared sh;
    __shared__ unsigned long long sh_word;
    __shared__ int sh_exit;
    const int s = (int)(blockIdx.x / (unsigned)Cw), jw = (int)(blockIdx.x % (unsigned)Cw);
    const int tid = threadIdx.x;
    StreamCtl &c = A.ctl[s];
    unsigned seen = 0u, nbar = 0u;
    if (tid == 0) sh.abort = 0;
    __syncthreads();
    for (;;) {
        if (tid == 0) {
            long long t_idle = wall_clock64() + RES_IDLE_TICKS;
            unsigned last_beat = __hip_atomic_load(beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // the time is up: has the host looked after the kernel since this wait began (or since the last time this was asked)?
            auto host_gone = [&](long long slack) __attribute__((always_inline)) {
                if (wall_clock64() <= t_idle + slack) return false;
                const unsigned bt = __hip_atomic_load(beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (bt == last_beat) return true;
                last_beat = bt; t_idle = wall_clock64() + RES_IDLE_TICKS;
                return false;
            };
            unsigned long long w = 0ULL;
            int ex = 0;
            unsigned spins = 0;
            if (jw == 0) {
                // the cluster's first workgroup reads the host's word (a read across the bus every few microseconds), waits
                // for the side stream to have come as far as the command says, and passes it on in device memory
                for (;;) {
                    w = __hip_atomic_load(&post[s].word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)(w >> 32) != seen) break;
                    ex = __hip_atomic_load(&post[s].exit_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (ex) break;
                    __builtin_amdgcn_s_sleep(48);
                    if ((++spins & 255u) == 0 && host_gone(0LL)) { ex = 1; break; }
                }
                if (!ex) {
                    const int T = __hip_atomic_load(&post[s].T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned rid = __hip_atomic_load(&post[s].ready_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    while ((int)(__hip_atomic_load(&ready[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - rid) < 0) {
                        __builtin_amdgcn_s_sleep(16);
                        if ((++spins & 255u) == 0 && host_gone(0LL)) { ex = 1; break; }
                    }
                    if (!ex) {
                        if (__hip_atomic_load(&post[s].init, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                            __hip_atomic_store(&c.needs_init, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&c.started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&c.error, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __hip_atomic_store(&c.T, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&mail[s].word, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (ex) __hip_atomic_store(&mail[s].exit_req, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (;;) {
                    w = __hip_atomic_load(&mail[s].word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(w >> 32) != seen) break;
                    ex = __hip_atomic_load(&mail[s].exit_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ex) break;
                    __builtin_amdgcn_s_sleep(16);
                    if ((++spins & 255u) == 0 && host_gone(100000000LL)) { ex = 1; break; }   // (a second behind workgroup 0)
                }
            }
            sh_word = w; sh_exit = ex;
        }
        __syncthreads();
        const unsigned long long w = sh_word;
        const int ex = sh_exit;
        __syncthreads();
        if (ex) {
            if (jw == 0 && tid == 0) __hip_atomic_store(&done[s].left, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const unsigned seq = (unsigned)(w >> 32);
        const int ll_slot = (int)(unsigned)(w & 0xffffffffULL);
        // what other kernels wrote since the last command - frames available, the likelihood rows, arenas swapped by a
        // collection, recognitionStart's mark - is read from memory, not from what this CU or this XCD's L2 still holds
        const long long t_cmd = wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_dcache_inv();
        run_stream<NE, XL, false>(A, sh, s, ll_slot, jw, Cw, true, &nbar);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        // every workgroup of the cluster is through with the command (its end-of-launch words are written) before the host
        // hears of it: the collection and the finish kernels read them
        if (Cw > 1 && sh.abort == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            ++nbar;
            if (tid == 0) {
                constexpr bool XL_ = false;
                GADD(&c.bar, 1u);
                const unsigned target = nbar * (unsigned)Cw;
                const long long t_lim = wall_clock64() + 200000000LL;          // 2 s
                unsigned spins = 0;
                while (CL(&c.bar) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0 && wall_clock64() > t_lim) break;
                }
            }
            __syncthreads();
        }
        if (jw == 0 && tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int fr = __hip_atomic_load(&c.frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int er = __hip_atomic_load(&c.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            done[s].frame = fr; done[s].error = er; done[s].run_ticks = wall_clock64() - t_cmd;
            __hip_atomic_store(&done[s].seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        seen = seq;
    }
}
