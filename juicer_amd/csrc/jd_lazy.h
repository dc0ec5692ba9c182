// jd_lazy.h - search-driven composition of C.L and G: the graph the
// search runs on grows while it runs.  Included by jd_search.h; the host side is in jd_compose.hip.
//
// The expansion step is jd_compose.hip's (binary-search match of the C.L arc's output label among the
// G state's input-sorted arcs, back-off epsilons after a word only, interval look-ahead) applied to
// ONE composed state at a time, by the wave of the search that needs it:
//   invariant   every arc a live instance sits on leads to a CLOSED state: one that has its arcs and whose
//               epsilon / tee arcs lead to closed states (what a token can reach within a frame) - phase X
//               therefore never meets a state without arcs;
//   who         after the last round of phase X every wave walks the arcs IT entered in this frame (its
//               segment of the new list) and expands the destinations that are not ready, closure included;
//   sharing     the graph belongs to the decoder, not to a stream: a state is claimed with an agent-scope
//               compare-and-swap (unknown -> expanding), its arcs are appended to one arena (bump pointer;
//               written with write-through stores and read with `sc1` loads like every mutable word of the
//               search, so blocks need no alignment) and its row
//               {first, count, status, final weight} is published after the arcs have drained; a wave that
//               loses a claim remembers the state and looks again.  Expansions never wait for each other;
//   closed      "has arcs" (EXPANDED) is not yet "its closure has arcs".  A wave closes a state depth first
//               (lz_close: an explicit stack of {state, next arc} in LDS) and marks it CLOSED when every
//               epsilon / tee arc of it leads to a closed state.  The invariant is about CLOSED states; a wave
//               that finds somebody else's EXPANDED state does not wait for whoever is closing it (two could
//               wait for each other) but walks it itself.
#ifndef JD_LAZY_H
#define JD_LAZY_H

#ifndef TEE_FLAG
#define TEE_FLAG 0x40000000          // (jd_search.h)
#endif
#ifndef LZ
#define LZ (-3.402823466e+38f)       // LOG_ZERO (jd_device.hip)
#endif

enum { LZ_UNKNOWN = 0, LZ_EXPANDING = 1, LZ_EXPANDED = 2, LZ_CLOSED = 3 };
#define LZ_FLAG 0x80000000u          // the composition filter's flag, in the top bit of the stored C.L state
#define LZQ 128                      // depth of a wave's stack when it closes a state (lz_close; at most 128: two ballots look through it)

// Look-ahead entry of a C.L state (jd_compose.hip, cl_lookahead): {lo, hi | MAYFIN}.  [lo, hi] bounds the first word
// labels reachable through label-less arcs; MAYFIN (the sign bit of hi) says a FINAL C.L state is reachable that
// way - the tail of the last word, `... -sil:eps-> final`: such arcs are followed whenever the G state is final,
// whatever its arcs (the reference always follows the transitions before the C.L final states: they carry
// NONPUSHING_OUTLABEL, WFSTOnTheFlyDecoder.cpp:2665-2697).
#define LA_MAYFIN 0x80000000u
__device__ __host__ __forceinline__ int la_hi(int2 la) { return (int)((unsigned)la.y & ~LA_MAYFIN); }
__device__ __host__ __forceinline__ bool la_mayfin(int2 la) { return ((unsigned)la.y & LA_MAYFIN) != 0; }

struct LazyDev {
    const int *cl_row; const JdArc *cl_arcs; const float *cl_fin; const int2 *cl_la;
    const int *g_row; const JdArc *g_arcs; const float *g_fin;
    unsigned long long *keys; int *vals; unsigned long long mask;
    int *st_c, *st_g;                 // composed state -> (C.L state | flag, G state)
    int4 *rows;                       // composed state -> {first arc, arcs, status, final weight bits}
    JdArc *arcs;                      // the arena
    int *n_states; unsigned long long *n_arcs;
    int max_states; long long max_arcs;
    int push;                         // weights pushed along the lexicon tree (jd_compose.hip "pushing")
    int *err;                         // 1: states exhausted, 2: arcs exhausted, 3: a wave's queue overflowed
};

// 8-byte write-through (agent-scope) stores: the graph is read by other workgroups, on any XCD, with `sc1` loads
__device__ __forceinline__ void lz_st8(void *p, int lo, int hi)
{
    __hip_atomic_store((unsigned long long *)p, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lz_store_arc(JdArc *p, int to, float w, int in, int out)
{
    lz_st8(p, to, __float_as_int(w));
    lz_st8(&p->in, in, out);
}

__device__ __forceinline__ unsigned long long lz_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

// weight look-ahead (pushing): the best weight among the arcs of G state g with an input label in [lo, hi] (0 if none)
__device__ __forceinline__ float lz_potential(const LazyDev &L, int g, int lo_l, int hi_l)
{
    if (lo_l > hi_l) return 0.0f;
    int lo = L.g_row[g], hi = L.g_row[g + 1];
    const int end = hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.g_arcs[mid].in < lo_l) lo = mid + 1; else hi = mid;
    }
    float best = 0.0f;
    bool have = false;
    for (; lo < end && L.g_arcs[lo].in <= hi_l; ++lo) {
        const float w = L.g_arcs[lo].w;
        if (!have || w > best) best = w;
        have = true;
    }
    return best;
}
// what the arcs into (c, g, flag) have already paid of the word that is under way
__device__ __forceinline__ float lz_paid(const LazyDev &L, unsigned cf, int g)
{
    if (!L.push || (cf & LZ_FLAG)) return 0.0f;
    const int2 la = L.cl_la[cf];
    return lz_potential(L, g, la.x, la_hi(la));
}

// id of the composed state (cf, g); the lane that creates it writes its row {0, 0, UNKNOWN, final weight}
// before the id becomes visible.  (Never blocks inside a branch, see jd_compose.hip.)
__device__ int lz_state_id(const LazyDev &L, unsigned cf, int g)
{
    const unsigned long long key = (((unsigned long long)cf << 32) | (unsigned)g) + 1ULL;
    unsigned long long slot = lz_hash(key) & L.mask;
    bool mine = false;
    int id = -1;
    unsigned long long probes = 0;
    if (__hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 0;   // the network has failed: no more insertions
    while (id < 0) {
        if (!mine) {
            if (++probes > L.mask) { atomicMax(L.err, 1); return 0; }                      // (table full: cannot happen below max_states)
            const unsigned long long old = atomicCAS(&L.keys[slot], 0ULL, key);
            if (old == 0ULL) {
                id = atomicAdd(L.n_states, 1);
                if (id < L.max_states) {
                    const float fc = L.cl_fin[cf & ~LZ_FLAG], fg = L.g_fin[g];
                    const bool fin = fc < __builtin_inff() && fg < __builtin_inff();
                    __hip_atomic_store(&L.st_c[id], (int)cf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&L.st_g[id], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lz_st8(&L.rows[id].x, 0, 0);
                    lz_st8(&L.rows[id].z, (int)LZ_UNKNOWN, __float_as_int(fin ? (L.push ? (fc + fg) - lz_paid(L, cf, g) : fc + fg) : __builtin_inff()));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else { atomicMax(L.err, 1); id = L.max_states - 1; }
                __hip_atomic_store(&L.vals[slot], id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else if (old == key) mine = true;
            else slot = (slot + 1) & L.mask;
        } else {
            id = __hip_atomic_load(&L.vals[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (id < 0) __builtin_amdgcn_s_sleep(1);
        }
    }
    return id;
}

__device__ __forceinline__ int lz_match(const LazyDev &L, int g, int x)   // WFSTOnTheFlyDecoder.cpp:3106-3159
{
    int lo = L.g_row[g], hi = L.g_row[g + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int l = L.g_arcs[mid].in;
        if (l == x) return mid;
        if (l < x) lo = mid + 1; else hi = mid;
    }
    return -1;
}
__device__ __forceinline__ bool lz_any_in(const LazyDev &L, int g, int lo_l, int hi_l)
{
    if (lo_l > hi_l) return false;
    int lo = L.g_row[g], hi = L.g_row[g + 1];
    const int end = hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.g_arcs[mid].in < lo_l) lo = mid + 1; else hi = mid;
    }
    return lo < end && L.g_arcs[lo].in <= hi_l;
}
__device__ __forceinline__ int lz_arc_kind(const LazyDev &L, const JdArc &ca, int g, int *ga)
{
    *ga = -1;
    if (ca.out == 0) {
        const int2 la = L.cl_la[ca.to];
        return (lz_any_in(L, g, la.x, la_hi(la)) || (la_mayfin(la) && L.g_fin[g] < __builtin_inff())) ? 1 : 0;
    }
    *ga = lz_match(L, g, ca.out);
    return *ga >= 0;
}

__device__ __forceinline__ int lz_status(const LazyDev &L, int s)
{
    return __hip_atomic_load(&L.rows[s].z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sticky: once a capacity has run out the network is of no use to anybody (a state may be left half expanded)
__device__ __forceinline__ bool lz_failed(const LazyDev &L)
{
    return __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// One wave gives composed state D (wave-uniform, claimed by the caller: status EXPANDING) its arcs.  Nothing in
// here waits for anybody.  Returns false when a capacity ran out (the state then stays EXPANDING: the network
// has failed, see lz_failed).
__device__ bool lz_expand(const LazyDev &L, const float *hmm_tee, int D)
{
    const int lane = threadIdx.x & 63;
    const unsigned cf = (unsigned)__hip_atomic_load(&L.st_c[D], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c = (int)(cf & ~LZ_FLAG), g = __hip_atomic_load(&L.st_g[D], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool flag = (cf & LZ_FLAG) != 0;
    bool bo = false;
    JdArc boa = {0, 0.0f, 0, 0};
    if (flag && L.g_row[g + 1] > L.g_row[g]) { boa = L.g_arcs[L.g_row[g]]; bo = boa.in == 0; }
    const int a0 = L.cl_row[c], a1 = L.cl_row[c + 1];
    const float p_src = lz_paid(L, cf, g);
    int mine = 0, ga;
    for (int a = a0 + lane; a < a1; a += 64) mine += lz_arc_kind(L, L.cl_arcs[a], g, &ga);
    int total = mine;
#pragma unroll
    for (int o = 32; o; o >>= 1) total += __shfl_xor(total, o);
    total += bo ? 1 : 0;
    int base = 0;
    if (lane == 0) {
        const unsigned long long b64 = atomicAdd(L.n_arcs, (unsigned long long)total);
        base = (b64 + (unsigned long long)total > (unsigned long long)L.max_arcs) ? -1 : (int)b64;
    }
    base = __shfl(base, 0);
    if (base < 0) { if (lane == 0) atomicMax(L.err, 2); return false; }
    int run = base;
    if (bo) {                                                          // the back-off arc first (an epsilon)
        if (lane == 0) lz_store_arc(&L.arcs[run], lz_state_id(L, cf, boa.to), boa.w, 0, boa.out);
        ++run;
    }
    for (int a = a0; a < a1; a += 64) {                                // then the C.L arcs in their order
        const bool on = a + lane < a1;
        JdArc ca = {0, 0.0f, 0, 0};
        int cnt = 0;
        ga = -1;
        if (on) { ca = L.cl_arcs[a + lane]; cnt = lz_arc_kind(L, ca, g, &ga); }
        int pre = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(pre, o); if (lane >= o) pre += y; }
        const int chunk_total = __shfl(pre, 63);
        if (cnt) {
            const int pos = run + pre - 1;
            const bool tee = ca.in > 0 && hmm_tee[ca.in - 1] > LZ;
            const int in = ca.in | (tee ? TEE_FLAG : 0);
            if (ca.out == 0) {
                float w = ca.w;
                if (L.push) { const int2 la = L.cl_la[ca.to]; w = (ca.w + lz_potential(L, g, la.x, la_hi(la))) - p_src; }
                lz_store_arc(&L.arcs[pos], lz_state_id(L, (unsigned)ca.to, g), w, in, 0);
            } else {
                const JdArc m = L.g_arcs[ga];
                lz_store_arc(&L.arcs[pos], lz_state_id(L, (unsigned)ca.to | LZ_FLAG, m.to), L.push ? (ca.w + m.w) - p_src : ca.w + m.w, in, m.out);
            }
        }
        run += chunk_total;
    }
    // the arcs are in memory (write-through stores, drained) before the row says so; {first, count} before the status
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        lz_st8(&L.rows[D].x, base, total);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&L.rows[D].z, (int)LZ_EXPANDED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return !lz_failed(L);
}

// One wave closes composed state D0 (wave-uniform): D0 and everything its epsilon / tee arcs lead to gets its
// arcs.  Depth first with an explicit stack of {state, next arc to look at} in the wave's LDS (stk[0 .. LZQ)): a
// state is marked CLOSED when its walk is through, so the stack is as deep as the closure is, not as broad (a
// back-off chain: a handful of entries).  A state another wave is expanding is waited for (expansions wait for
// nobody); one that somebody else expanded and has not closed yet is walked here too - never waited for, since
// two walkers could wait for each other.  Returns false when a capacity ran out.
__device__ bool lz_close(const LazyDev &L, const float *hmm_tee, int D0, int2 *stk, long long t_limit)
{
    const int lane = threadIdx.x & 63;
    int sp = 0;
    if (lane == 0) stk[0] = make_int2(D0, 0);
    sp = 1;
    int cyc = LZQ;                                                     // lowest stack index an epsilon CYCLE came back to (LZQ: none)
    unsigned spins = 0;
    while (sp > 0) {
        const int2 top = stk[sp - 1];
        const int S = top.x;
        int st = 0;
        if (lane == 0) st = atomicCAS(&L.rows[S].z, (int)LZ_UNKNOWN, (int)LZ_EXPANDING);
        st = __shfl(st, 0);
        if (st == LZ_CLOSED) { --sp; continue; }
        if (st == LZ_UNKNOWN) {                                        // claimed: give it its arcs, then walk them
            if (!lz_expand(L, hmm_tee, S)) return false;
            continue;
        }
        if (st == LZ_EXPANDING) {                                      // another wave's expansion: soon over
            __builtin_amdgcn_s_sleep(4);
            ++spins;
            if ((spins & 63u) == 0 && lz_failed(L)) return false;      // (an expansion that ran out of room never finishes)
            if ((spins & 4095u) == 0 && wall_clock64() > t_limit) { atomicMax(L.err, 3); return false; }
            continue;
        }
        // EXPANDED: from arc top.y on, the first epsilon / tee arc whose destination is not closed yet
        const unsigned long long fc = __hip_atomic_load((unsigned long long *)&L.rows[S].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int first = (int)(unsigned)fc, cnt = (int)(fc >> 32);
        bool descended = false;
        for (int a = top.y; a < cnt; a += 64) {
            int to = -1;
            bool need = false;
            if (a + lane < cnt) {
                const JdArc *pa = &L.arcs[first + a + lane];
                const int in = __hip_atomic_load(&pa->in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (in == 0 || (in & TEE_FLAG) != 0) {
                    to = __hip_atomic_load(&pa->to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    need = to != S && lz_status(L, to) != LZ_CLOSED;
                }
            }
            const unsigned long long bn = __ballot(need);
            if (bn) {
                const int j = __ffsll((long long)bn) - 1;
                const int T = __shfl(to, j);
                // a cycle of epsilons comes back to a state this walk is inside of: it is being taken care of - go on
                // behind the arc, and remember how far down the cycle reaches (see the marking below)
                const unsigned long long on0 = __ballot(lane < sp && stk[lane].x == T);
                const unsigned long long on1 = __ballot(lane + 64 < sp && stk[(lane + 64) & (LZQ - 1)].x == T);
                if (lane == 0) stk[sp - 1] = make_int2(S, a + j + 1);
                descended = true;
                if (on0 | on1) {
                    const int at = on0 ? __ffsll((long long)on0) - 1 : 64 + __ffsll((long long)on1) - 1;
                    cyc = at < cyc ? at : cyc;
                    break;
                }
                if (sp >= LZQ) { if (lane == 0) atomicMax(L.err, 3); return false; }   // epsilon / tee arcs more than LZQ states deep
                if (lane == 0) stk[sp] = make_int2(T, 0);
                ++sp;
                break;
            }
        }
        if (!descended) {
            // every epsilon / tee arc of S leads to a closed state - unless S sits on a cycle whose entry (further down
            // the stack) is not through yet: then S keeps "expanded" and whoever asks next finds the entry closed
            if (sp - 1 <= cyc) {
                if (lane == 0) __hip_atomic_store(&L.rows[S].z, (int)LZ_CLOSED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sp - 1 == cyc) cyc = LZQ;
            }
            --sp;
        }
    }
    return true;
}

#endif
