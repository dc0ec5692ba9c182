// jd_compose.hip - C.L o G on the device: first step of the dynamic-composition row (SURVEY.md §8 f3,
// BASELINE.json configs[4]; reference: WFSTOnTheFlyDecoder + WFSTSortedInLabelNetwork).
//
// The reference keeps C.L and G apart and composes while it searches: at a C.L arc with output word x
// the hypothesis' G state is advanced by a binary search among that state's input-sorted arcs
// (WFSTOnTheFlyDecoder::binarySearchGTrans, WFSTOnTheFlyDecoder.cpp:3106-3159); when a hypothesis
// enters a new prefix region it is duplicated along the back-off (epsilon) chain of its G state with the
// chain's weights added (getStatesOnEpsPath, WFSTNetwork.cpp:2605-2646; WFSTOnTheFlyDecoder.cpp:
// 1590-1622); label-set look-ahead keeps it out of lexicon-tree branches none of whose words the G state
// has an arc for (WFSTLabelPushingNetwork, WFSTNetwork.cpp:1505-2590).  That design exists because a
// composed trigram graph did not fit the hosts of 2006.  An MI355X has 288 GB: the composition is done
// here ON THE DEVICE, breadth-first from the start pair, with the reference's three ingredients as the
// expansion step, and the result is an ordinary jd_net for the static search (DESIGN.md "dynamic
// composition" has the note on the lazy, search-driven variant this grows into).
//
//   composed state  = (C.L state c, G state g, flag f), reachable triples only; f = 1 right after a word
//                     was matched (and at the start): the only place a back-off may be taken, which
//                     gives every path one canonical form (the epsilon-sequencing composition filter)
//   back-off        if f = 1 and the first arc of g is an epsilon g -eps/b-> g":  (c,g,1) -eps:eps/b-> (c,g",1)
//   C.L arc c -i:eps/w-> c'    gives (c,g,f) -i:eps/w-> (c',g,0), provided g has an arc for a label in
//                     [lo(c'), hi(c')], the interval of the first word labels reachable from c' through
//                     label-less arcs - OR a final C.L state is reachable from c' that way and g is final
//                     (the tail of the last word; the reference always follows the transitions before the
//                     C.L final states, WFSTOnTheFlyDecoder.cpp:2665-2697).  A superset test: it only
//                     ever drops dead ends
//   C.L arc c -i:x/w->  c', x != eps, and an arc g -x:y/v-> g'   gives (c,g,f) -i:y/(w + v)-> (c',g',1)
//   (c,g,f) is final iff c and g are: weight fin(c) + fin(g)
//   pushing (the reference's -pushing, doLabelAndWeightPushing, juicer.cpp:240, 931-935; weights only): let
//   P(c,g) = the best weight among the arcs of g with a label in [lo(c), hi(c)] - what entering c's part of
//   the lexicon will cost at least - and P = 0 at the states with f = 1.  Every arc gets + P(destination)
//   - P(source) (a matched arc: - P(source) only; a final weight likewise), so the grammar's weight is
//   paid as early as the tree allows and the beam sees it: (w + P(c',g)) - P(c,g), resp. (w + v) - P(c,g).
//   Path totals are unchanged up to float association.
//
// All weights are the ones the two networks carry after loading (each with its own scale, as
// juicer.cpp:933-970 loads them), i.e. log-domain scores that add, in float32.  States are numbered by
// (c, f, g) order, arcs of a state as listed above (C.L arcs in C.L order): the result does not depend
// on the order the device discovered things in.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "jd_internal.h"
#include "jd_lazy.h"        // LazyDev and the expansion step shared with the search kernel

#define CHK(expr)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { rc = jd_fail(JD_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

enum { JC_OK = 0, JC_ESTATES = 1, JC_EARCS = 2 };
#define JC_FLAG 0x80000000u   // the filter flag rides in the top bit of the stored C.L state

struct ComposeArgs {
    const int *cl_row; const JdArc *cl_arcs; const float *cl_fin; const int2 *cl_la;   // cl_la[c] = {lo, hi} look-ahead interval
    const int *g_row; const JdArc *g_arcs; const float *g_fin;
    unsigned long long *keys; int *vals; unsigned long long mask;        // open-addressing table (c,g) -> temporary id
    int *st_c, *st_g; int *n_states; int max_states;                    // states in discovery order (= the BFS queue)
    long long *arc_start; int *arc_cnt; JdArc *arcs; unsigned long long *n_arcs; long long max_arcs;
    float *fin; int *err;
    int push;                                                            // weight look-ahead pushing (see jd_net_compose)
};

__device__ __forceinline__ unsigned long long jc_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

// id of the triple (cf = c | flag, g); a triple seen for the first time is appended to the state list.
// Never blocks inside a branch: the lane that created an entry and a lane of the same wave that found
// it in the same step both come back round the loop (the id is published before the second one polls).
__device__ int jc_state_id(const ComposeArgs &A, unsigned cf, int g)
{
    const unsigned long long key = (((unsigned long long)cf << 32) | (unsigned)g) + 1ULL;
    unsigned long long slot = jc_hash(key) & A.mask;
    bool mine = false;                                                 // slot holds this key
    int id = -1;
    unsigned long long probes = 0;
    while (id < 0) {
        if (!mine) {
            // table full (a level far beyond max_states): give up, the host reports it
            if (++probes > A.mask) { atomicMax(A.err, (int)JC_ESTATES); return 0; }
            const unsigned long long old = atomicCAS(&A.keys[slot], 0ULL, key);
            if (old == 0ULL) {
                id = atomicAdd(A.n_states, 1);
                if (id < A.max_states) { A.st_c[id] = (int)cf; A.st_g[id] = g; }
                else atomicMax(A.err, (int)JC_ESTATES);
                __hip_atomic_store(&A.vals[slot], id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else if (old == key) mine = true;
            else slot = (slot + 1) & A.mask;
        } else {
            id = __hip_atomic_load(&A.vals[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (id < 0) __builtin_amdgcn_s_sleep(1);
        }
    }
    return id;
}

// WFSTOnTheFlyDecoder::binarySearchGTrans (WFSTOnTheFlyDecoder.cpp:3106-3159): the arc of G state g
// whose input label is x, among arcs sorted by input label; -1 if there is none
__device__ __forceinline__ int jc_match(const ComposeArgs &A, int g, int x)
{
    int lo = A.g_row[g], hi = A.g_row[g + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int l = A.g_arcs[mid].in;
        if (l == x) return mid;
        if (l < x) lo = mid + 1; else hi = mid;
    }
    return -1;
}

// look-ahead: does G state g have an arc whose input label lies in [lo, hi] ?
__device__ __forceinline__ bool jc_any_in(const ComposeArgs &A, int g, int lo_l, int hi_l)
{
    if (lo_l > hi_l) return false;
    int lo = A.g_row[g], hi = A.g_row[g + 1];
    const int end = hi;
    while (lo < hi) {                                                  // first arc with label >= lo_l
        const int mid = (lo + hi) >> 1;
        if (A.g_arcs[mid].in < lo_l) lo = mid + 1; else hi = mid;
    }
    return lo < end && A.g_arcs[lo].in <= hi_l;
}

// weight look-ahead: the best weight among the arcs of G state g whose input label lies in [lo, hi] (0 if none)
__device__ __forceinline__ float jc_potential(const ComposeArgs &A, int g, int lo_l, int hi_l)
{
    if (lo_l > hi_l) return 0.0f;
    int lo = A.g_row[g], hi = A.g_row[g + 1];
    const int end = hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A.g_arcs[mid].in < lo_l) lo = mid + 1; else hi = mid;
    }
    float best = 0.0f;
    bool have = false;
    for (; lo < end && A.g_arcs[lo].in <= hi_l; ++lo) {
        const float w = A.g_arcs[lo].w;
        if (!have || w > best) best = w;
        have = true;
    }
    return best;
}

// what C.L arc ca contributes at G state g: 0 = nothing, 1 = an arc; *ga = the matched G arc (index, or -1)
__device__ __forceinline__ int jc_arc_kind(const ComposeArgs &A, const JdArc &ca, int g, int *ga)
{
    *ga = -1;
    if (ca.out == 0) {                                                 // (see LA_MAYFIN, jd_lazy.h)
        const int2 la = A.cl_la[ca.to];
        return (jc_any_in(A, g, la.x, la_hi(la)) || (la_mayfin(la) && A.g_fin[g] < std::numeric_limits<float>::infinity())) ? 1 : 0;
    }
    *ga = jc_match(A, g, ca.out);
    return *ga >= 0;
}

// one wave per state of the current BFS level
__global__ __launch_bounds__(256) void jc_expand(ComposeArgs A, int begin, int end)
{
    const int lane = threadIdx.x & 63;
    const int s = begin + (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (s >= end) return;
    const unsigned cf = (unsigned)A.st_c[s];
    const int c = (int)(cf & ~JC_FLAG), g = A.st_g[s];
    const bool flag = (cf & JC_FLAG) != 0;
    // back-off (getStatesOnEpsPath, WFSTNetwork.cpp:2605-2646: the FIRST arc of the state, if it is an epsilon)
    bool bo = false;
    JdArc boa = {0, 0.0f, 0, 0};
    if (flag && A.g_row[g + 1] > A.g_row[g]) { boa = A.g_arcs[A.g_row[g]]; bo = boa.in == 0; }
    const int a0 = A.cl_row[c], a1 = A.cl_row[c + 1];
    // pushing: what this state's incoming arcs have already paid of the word that is under way
    float p_src = 0.0f;
    if (A.push && !flag) { const int2 la = A.cl_la[c]; p_src = jc_potential(A, g, la.x, la_hi(la)); }
    // pass 1: how many arcs this state gets
    int mine = 0, ga;
    for (int a = a0 + lane; a < a1; a += 64) mine += jc_arc_kind(A, A.cl_arcs[a], g, &ga);
    int total = mine;
#pragma unroll
    for (int o = 32; o; o >>= 1) total += __shfl_xor(total, o);
    total += bo ? 1 : 0;
    long long base = 0;
    if (lane == 0) {
        base = (long long)atomicAdd(A.n_arcs, (unsigned long long)total);
        if (base + total > A.max_arcs) atomicMax(A.err, (int)JC_EARCS);
        A.arc_start[s] = base; A.arc_cnt[s] = total;
        const float fc = A.cl_fin[c], fg = A.g_fin[g];
        const bool fin = fc < std::numeric_limits<float>::infinity() && fg < std::numeric_limits<float>::infinity();
        A.fin[s] = fin ? (A.push ? (fc + fg) - p_src : fc + fg) : std::numeric_limits<float>::infinity();
    }
    base = __shfl(base, 0);
    if (base + total > A.max_arcs) return;
    // pass 2: write them - the back-off arc first, then the C.L arcs in their order
    long long run = base;
    if (bo) {
        if (lane == 0) A.arcs[run] = JdArc{jc_state_id(A, cf, boa.to), boa.w, 0, boa.out};
        ++run;
    }
    for (int a = a0; a < a1; a += 64) {
        const bool on = a + lane < a1;
        JdArc ca = {0, 0.0f, 0, 0};
        int cnt = 0;
        ga = -1;
        if (on) { ca = A.cl_arcs[a + lane]; cnt = jc_arc_kind(A, ca, g, &ga); }
        int pre = cnt;                                                 // inclusive scan over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(pre, o); if (lane >= o) pre += y; }
        const int chunk_total = __shfl(pre, 63);
        if (cnt) {
            const long long pos = run + pre - 1;
            if (ca.out == 0) {
                float w = ca.w;
                if (A.push) { const int2 la = A.cl_la[ca.to]; w = (ca.w + jc_potential(A, g, la.x, la_hi(la))) - p_src; }
                A.arcs[pos] = JdArc{jc_state_id(A, (unsigned)ca.to, g), w, ca.in, 0};
            } else {
                const JdArc m = A.g_arcs[ga];
                const float w = A.push ? (ca.w + m.w) - p_src : ca.w + m.w;
                A.arcs[pos] = JdArc{jc_state_id(A, (unsigned)ca.to | JC_FLAG, m.to), w, ca.in, m.out};
            }
        }
        run += chunk_total;
    }
}

__global__ void jc_keys(const int *st_c, const int *st_g, int n, unsigned long long *key, int *id)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned cf = (unsigned)st_c[i];
        key[i] = ((unsigned long long)(cf & ~JC_FLAG) << 33) | ((unsigned long long)(cf >> 31) << 32) | (unsigned)st_g[i];
        id[i] = i;
    }
}

// sorted position -> old id  =>  old id -> new id, and the per-state arc counts / final weights in new order
__global__ void jc_rank(const int *sorted_id, int n, int *rank, const int *arc_cnt, int *cnt_new, const float *fin, float *fin_new)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int o = sorted_id[i]; rank[o] = i; cnt_new[i] = arc_cnt[o]; fin_new[i] = fin[o]; }
}

__global__ __launch_bounds__(256) void jc_place(const int *rank, int n, const long long *arc_start, const int *arc_cnt,
                                                const JdArc *arcs, const int *row_new, JdArc *out)
{
    const int lane = threadIdx.x & 63;
    const int s = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (s >= n) return;
    const long long src = arc_start[s];
    const int dst = row_new[rank[s]];
    for (int j = lane; j < arc_cnt[s]; j += 64) {
        JdArc a = arcs[src + j];
        a.to = rank[a.to];
        out[dst + j] = a;
    }
}

// G as WFSTSortedInLabelNetwork holds it (WFSTNetwork.cpp:2693-2710): the arcs of a state sorted by
// input label; a label may not occur twice at a state (binarySearchInLabel, :2653-2700, errors out)
static int sorted_g_arcs(const jd_net *g, std::vector<JdArc> &arcs)
{
    arcs = g->arcs;
    for (int32_t s = 0; s < g->n_states; ++s) {
        const auto b = arcs.begin() + g->row_ptr[(size_t)s], e = arcs.begin() + g->row_ptr[(size_t)s + 1];
        std::stable_sort(b, e, [](const JdArc &x, const JdArc &y) { return x.in < y.in; });
        for (auto it = b; it != e && it + 1 != e; ++it)
            if (it->in == (it + 1)->in)
                return jd_fail(JD_EINVAL, "jd_net_compose: G state %d has two arcs with input label %d "
                                          "(WFSTSortedInLabelNetwork::binarySearchInLabel - inLabel == inLabelInTransArray)", s, it->in);
    }
    return JD_OK;
}

// Look-ahead intervals of C.L: la[c] = {lo, hi} bounds the output labels of the first label-carrying arcs
// reachable from c through arcs without an output label (lo > hi: none - a dead end).  An interval is a
// superset of the label set, so the test built on it can only keep too much, never drop a path; it is
// tight when the words are numbered in the lexicon tree's depth-first order (label reachability as in
// the reference's label-set look-ahead, WFSTNetwork.cpp:1505-2590, with intervals for sets).  States on
// a cycle of label-less arcs get the full range.
static void cl_lookahead(const jd_net *cl, std::vector<int2> &la)
{
    const int S = cl->n_states;
    const int2 EMPTY = make_int2(0x7fffffff, 0), FULL = make_int2(1, 0x7fffffff);
    la.assign((size_t)S, EMPTY);
    std::vector<char> st((size_t)S, 0);                                // 0 new, 1 on the stack, 2 done
    std::vector<std::pair<int, int>> stack;                            // (state, next arc)
    for (int r = 0; r < S; ++r) {
        if (st[(size_t)r]) continue;
        stack.push_back({r, cl->row_ptr[(size_t)r]});
        st[(size_t)r] = 1;
        while (!stack.empty()) {
            const int c = stack.back().first;
            int &a = stack.back().second;
            if (a == cl->row_ptr[(size_t)c + 1]) { st[(size_t)c] = 2; stack.pop_back(); continue; }
            const JdArc &arc = cl->arcs[(size_t)a];
            int2 &I = la[(size_t)c];
            if (arc.out != 0) { I.x = std::min(I.x, arc.out); I.y = std::max(I.y, arc.out); ++a; }
            else if (st[(size_t)arc.to] == 2) { const int2 J = la[(size_t)arc.to]; if (J.x <= J.y) { I.x = std::min(I.x, J.x); I.y = std::max(I.y, J.y); } ++a; }
            else if (st[(size_t)arc.to] == 1) { I = FULL; ++a; }       // cycle of label-less arcs
            // (a stays: the arc is read again when the child is done)
            else { st[(size_t)arc.to] = 1; stack.push_back({arc.to, cl->row_ptr[(size_t)arc.to]}); }
        }
    }
    // a FULL interval below spreads upwards only through the pass above if it was set before the parent
    // finished; one more sweep makes every ancestor of a cycle state FULL as well.  The same sweep finds the
    // states that reach a FINAL C.L state through label-less arcs (LA_MAYFIN, jd_lazy.h).
    std::vector<char> mf((size_t)S, 0);
    for (int c = 0; c < S; ++c) mf[(size_t)c] = cl->fin_w[(size_t)c] < std::numeric_limits<float>::infinity();
    bool changed = true;
    while (changed) {
        changed = false;
        for (int c = S - 1; c >= 0; --c)
            for (int a = cl->row_ptr[(size_t)c]; a < cl->row_ptr[(size_t)c + 1]; ++a) {
                const JdArc &arc = cl->arcs[(size_t)a];
                if (arc.out != 0) continue;
                const int2 J = la[(size_t)arc.to];
                int2 &I = la[(size_t)c];
                if (J.x <= J.y && (J.x < I.x || J.y > I.y)) { I.x = std::min(I.x, J.x); I.y = std::max(I.y, J.y); changed = true; }
                if (mf[(size_t)arc.to] && !mf[(size_t)c]) { mf[(size_t)c] = 1; changed = true; }
            }
    }
    for (int c = 0; c < S; ++c)
        if (mf[(size_t)c]) la[(size_t)c].y = (int)((unsigned)la[(size_t)c].y | LA_MAYFIN);
}

// Label pushing, the other half of the reference's -pushing (doLabelAndWeightPushing, juicer.cpp:240, 931-935).  The
// reference gives every C.L transition the SET of output labels that can follow it (WFSTLabelPushingNetwork::
// assignOutlabsToTrans, WFSTNetwork.cpp:1643-1764, loops included) and its on-the-fly decoder takes the G transition
// as soon as that set has shrunk to one label - the word is known, and with it the grammar state, before its last
// model has been decoded.  Stated on the transducer: every output label moves towards the initial state, up to the
// first arc behind which it is the only label that can follow.  With the look-ahead intervals above: a state c is
// SINGLE when its interval is one label w(c) and no final state can be reached from it without a label; single
// states joined by label-less arcs (they share their label) form a region.  A label-less arc INTO a single state
// from outside such a run of "already emitted" states gets the label, the arcs that carried it lose it:
//   E(c)   single, entered by label-less arcs only (and not the initial state): w(c) has been emitted when c is reached
//   arc c0 -i:eps-> c, E(c), not E(c0)   becomes  -i:w(c)->        arc c0 -i:w-> c, E(c0)   becomes  -i:eps->
// A region in which some state is entered both ways (by a label-carrying arc, or as the initial state, AND by a
// label-less arc) could not say whether its label is out yet: it is left as it is.  Arcs, weights and states are
// unchanged and correspond one to one, every complete path keeps its label sequence (tests/test_compose_ref_cpu.py
// walks random paths through both); what moves is WHEN a word's label is passed - the times of a hypothesis' word
// ends become the frames in which the words were identified.
static void cl_push_labels(const jd_net *cl, std::vector<JdArc> &arcs, int64_t *n_moved)
{
    std::vector<int2> la;
    cl_lookahead(cl, la);
    const int S = cl->n_states;
    auto single = [&](int c) { const int2 I = la[(size_t)c]; return I.x == la_hi(I) && !la_mayfin(I); };
    std::vector<int> comp((size_t)S);
    for (int c = 0; c < S; ++c) comp[(size_t)c] = c;
    auto find = [&](int c) { while (comp[(size_t)c] != c) { comp[(size_t)c] = comp[(size_t)comp[(size_t)c]]; c = comp[(size_t)c]; } return c; };
    std::vector<char> pend((size_t)S, 0), lessin((size_t)S, 0);
    pend[(size_t)cl->init] = 1;
    for (int c0 = 0; c0 < S; ++c0)
        for (int a = cl->row_ptr[(size_t)c0]; a < cl->row_ptr[(size_t)c0 + 1]; ++a) {
            const JdArc &arc = cl->arcs[(size_t)a];
            if (arc.out != 0) { pend[(size_t)arc.to] = 1; continue; }
            lessin[(size_t)arc.to] = 1;
            if (single(c0) && single(arc.to)) { const int x = find(c0), y = find(arc.to); if (x != y) comp[(size_t)x] = y; }
        }
    std::vector<char> bad((size_t)S, 0);
    for (int c = 0; c < S; ++c)
        if (single(c) && pend[(size_t)c] && lessin[(size_t)c]) bad[(size_t)find(c)] = 1;
    auto emitted = [&](int c) { return single(c) && !pend[(size_t)c] && !bad[(size_t)find(c)]; };
    arcs = cl->arcs;
    int64_t moved = 0;
    for (int c0 = 0; c0 < S; ++c0)
        for (int a = cl->row_ptr[(size_t)c0]; a < cl->row_ptr[(size_t)c0 + 1]; ++a) {
            JdArc &arc = arcs[(size_t)a];
            if (arc.out == 0) {
                if (emitted(arc.to) && !emitted(c0)) { arc.out = la[(size_t)arc.to].x; ++moved; }
            } else if (emitted(c0)) arc.out = 0;
        }
    if (n_moved) *n_moved = moved;
}

static jd_net *net_with_arcs(const jd_net *n, std::vector<JdArc> &&arcs)
{
    jd_net *r = new jd_net();
    r->n_states = n->n_states; r->init = n->init; r->n_final = n->n_final; r->n_arcs = n->n_arcs;
    r->row_ptr = n->row_ptr; r->arcs = std::move(arcs); r->fin_w = n->fin_w; r->max_in = n->max_in;
    r->lm_scale = n->lm_scale; r->ins_penalty = n->ins_penalty;
    return r;
}

// the C.L transducer with its output labels pushed towards the initial state (host code: no device needed)
extern "C" int jd_net_push_labels(jd_net **out, const jd_net *cl, int64_t *n_moved)
{
    if (!out || !cl) return jd_fail(JD_EINVAL, "jd_net_push_labels: null argument");
    if (cl->lazy_dev) return jd_fail(JD_EINVAL, "jd_net_push_labels: the input must be an ordinary network");
    std::vector<JdArc> arcs;
    cl_push_labels(cl, arcs, n_moved);
    *out = net_with_arcs(cl, std::move(arcs));
    return JD_OK;
}

extern "C" int jd_net_compose(jd_net **out, const jd_net *cl, const jd_net *g, int32_t device, int64_t max_states, int64_t max_arcs,
                              int32_t pushing)
{
    if (!out || !cl || !g) return jd_fail(JD_EINVAL, "jd_net_compose: null argument");
    if (pushing < 0 || pushing > 3) return jd_fail(JD_EINVAL, "jd_net_compose: pushing is a combination of JD_PUSH_WEIGHTS and JD_PUSH_LABELS");
    std::vector<JdArc> g_sorted;
    int rc = sorted_g_arcs(g, g_sorted);
    if (rc) return rc;
    struct Owned { jd_net *p = nullptr; ~Owned() { delete p; } } pushed;   // C.L with its labels pushed (JD_PUSH_LABELS)
    if (pushing & 2) {
        if (cl->lazy_dev) return jd_fail(JD_EINVAL, "jd_net_compose: the inputs must be ordinary networks");
        std::vector<JdArc> pa;
        cl_push_labels(cl, pa, nullptr);
        pushed.p = net_with_arcs(cl, std::move(pa));
        cl = pushed.p;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return jd_fail(JD_ENODEV, "jd_net_compose: no HIP device %d (the composition runs on the GPU; there is no CPU path)", device);
    if (hipSetDevice(device) != hipSuccess) return jd_fail(JD_EHIP, "hipSetDevice(%d) failed", device);
    if (max_states <= 0) max_states = std::min<int64_t>(0x7ffffff0LL, std::max<int64_t>(1 << 20, 4 * ((int64_t)cl->n_states + g->n_states) + (int64_t)g->n_arcs * 8));
    if (max_arcs <= 0) max_arcs = std::min<int64_t>(0x7ffffff0LL, std::max<int64_t>(1 << 22, ((int64_t)cl->n_arcs + g->n_arcs) * 16));
    if (max_states > 0x7ffffff0LL || max_arcs > 0x7ffffff0LL) return jd_fail(JD_EINVAL, "jd_net_compose: capacities are limited to 2^31 states / arcs");

    ComposeArgs A;
    memset(&A, 0, sizeof A);
    std::vector<void *> allocs;
    auto dal = [&](size_t bytes) -> void * { void *p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; allocs.push_back(p); return p; };
    int *d_cnt_new = nullptr, *d_row_new = nullptr, *d_rank = nullptr, *d_id = nullptr, *d_id_sorted = nullptr;
    unsigned long long *d_key = nullptr, *d_key_sorted = nullptr;
    float *d_fin_new = nullptr;
    JdArc *d_out = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    jd_net *res = nullptr;
    int h_n = 1, h_err = 0, level = 0;
    unsigned long long h_arcs = 0ULL;
    {
        size_t cap = 1;
        while (cap < (size_t)max_states * 2) cap <<= 1;
        A.mask = cap - 1;
#define DAL(p, T, n) do { p = (T *)dal(sizeof(T) * (size_t)(n)); if (!p) { rc = jd_fail(JD_ENOMEM, "jd_net_compose: hipMalloc of %zu bytes failed", sizeof(T) * (size_t)(n)); goto done; } } while (0)
        int *cl_row, *g_row; JdArc *cl_arcs, *g_arcs; float *cl_fin, *g_fin; int2 *cl_la;
        std::vector<int2> la;
        cl_lookahead(cl, la);
        DAL(cl_row, int, cl->row_ptr.size()); DAL(cl_arcs, JdArc, cl->arcs.size()); DAL(cl_fin, float, cl->fin_w.size()); DAL(cl_la, int2, la.size());
        CHK(hipMemcpy(cl_la, la.data(), la.size() * sizeof(int2), hipMemcpyHostToDevice));
        DAL(g_row, int, g->row_ptr.size()); DAL(g_arcs, JdArc, g->arcs.size()); DAL(g_fin, float, g->fin_w.size());
        CHK(hipMemcpy(cl_row, cl->row_ptr.data(), cl->row_ptr.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(cl_arcs, cl->arcs.data(), cl->arcs.size() * sizeof(JdArc), hipMemcpyHostToDevice));
        CHK(hipMemcpy(cl_fin, cl->fin_w.data(), cl->fin_w.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(g_row, g->row_ptr.data(), g->row_ptr.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(g_arcs, g_sorted.data(), g_sorted.size() * sizeof(JdArc), hipMemcpyHostToDevice));
        CHK(hipMemcpy(g_fin, g->fin_w.data(), g->fin_w.size() * 4, hipMemcpyHostToDevice));
        A.cl_row = cl_row; A.cl_arcs = cl_arcs; A.cl_fin = cl_fin; A.cl_la = cl_la; A.g_row = g_row; A.g_arcs = g_arcs; A.g_fin = g_fin;
        DAL(A.keys, unsigned long long, cap); DAL(A.vals, int, cap);
        DAL(A.st_c, int, max_states); DAL(A.st_g, int, max_states); DAL(A.n_states, int, 1);
        DAL(A.arc_start, long long, max_states); DAL(A.arc_cnt, int, max_states); DAL(A.arcs, JdArc, max_arcs);
        DAL(A.n_arcs, unsigned long long, 1); DAL(A.fin, float, max_states); DAL(A.err, int, 1);
        A.max_states = (int)max_states; A.max_arcs = max_arcs; A.push = (pushing & 1) ? 1 : 0;
        CHK(hipMemset(A.keys, 0, cap * 8)); CHK(hipMemset(A.vals, 0xff, cap * 4));
        CHK(hipMemset(A.n_arcs, 0, 8)); CHK(hipMemset(A.err, 0, 4));
        // the start pair is state 0 of the discovery order
        {
            const unsigned cf0 = (unsigned)cl->init | JC_FLAG;              // the start triple has the flag set
            const unsigned long long key = (((unsigned long long)cf0 << 32) | (unsigned)g->init) + 1ULL;
            unsigned long long k = key; k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
            const unsigned long long slot = k & A.mask;
            const int zero = 0, one = 1, c0 = (int)cf0, g0 = g->init;
            CHK(hipMemcpy(A.keys + slot, &key, 8, hipMemcpyHostToDevice));
            CHK(hipMemcpy(A.vals + slot, &zero, 4, hipMemcpyHostToDevice));
            CHK(hipMemcpy(A.st_c, &c0, 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(A.st_g, &g0, 4, hipMemcpyHostToDevice));
            CHK(hipMemcpy(A.n_states, &one, 4, hipMemcpyHostToDevice));
        }
        // breadth-first: a level's states are expanded by one wave each; the states they reach form the next level
        int begin = 0;
        while (begin < h_n) {
            const int end = h_n;
            const long long waves = end - begin;
            hipLaunchKernelGGL(jc_expand, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, A, begin, end);
            CHK(hipGetLastError());
            CHK(hipMemcpy(&h_n, A.n_states, 4, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(&h_err, A.err, 4, hipMemcpyDeviceToHost));
            if (h_err || h_n > max_states) break;
            begin = end;
            ++level;
        }
        CHK(hipMemcpy(&h_arcs, A.n_arcs, 8, hipMemcpyDeviceToHost));
        if (h_err == JC_ESTATES || h_n > max_states) { rc = jd_fail(JD_ENOMEM, "jd_net_compose: more than %lld composed states (raise max_states)", (long long)max_states); goto done; }
        if (h_err == JC_EARCS || (long long)h_arcs > max_arcs) { rc = jd_fail(JD_ENOMEM, "jd_net_compose: more than %lld composed arcs (raise max_arcs)", (long long)max_arcs); goto done; }
        // canonical numbering: states by (c, g); arcs keep their per-state order
        const int N = h_n;
        const long long M = (long long)h_arcs;
        DAL(d_key, unsigned long long, N); DAL(d_key_sorted, unsigned long long, N); DAL(d_id, int, N); DAL(d_id_sorted, int, N);
        DAL(d_rank, int, N); DAL(d_cnt_new, int, N + 1); DAL(d_row_new, int, N + 1); DAL(d_fin_new, float, N); DAL(d_out, JdArc, std::max<long long>(M, 1));
        hipLaunchKernelGGL(jc_keys, dim3((N + 255) / 256), dim3(256), 0, 0, A.st_c, A.st_g, N, d_key, d_id);
        CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_key_sorted, d_id, d_id_sorted, N));
        d_tmp = dal(tmp_bytes);
        if (!d_tmp) { rc = jd_fail(JD_ENOMEM, "jd_net_compose: hipMalloc failed"); goto done; }
        CHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key_sorted, d_id, d_id_sorted, N));
        CHK(hipMemset(d_cnt_new, 0, (size_t)(N + 1) * 4));
        hipLaunchKernelGGL(jc_rank, dim3((N + 255) / 256), dim3(256), 0, 0, d_id_sorted, N, d_rank, A.arc_cnt, d_cnt_new, A.fin, d_fin_new);
        {
            size_t tb = 0;
            CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_cnt_new, d_row_new, N + 1));
            void *t2 = dal(tb);
            if (!t2) { rc = jd_fail(JD_ENOMEM, "jd_net_compose: hipMalloc failed"); goto done; }
            CHK(hipcub::DeviceScan::ExclusiveSum(t2, tb, d_cnt_new, d_row_new, N + 1));
        }
        hipLaunchKernelGGL(jc_place, dim3((unsigned)(((long long)N + 3) / 4)), dim3(256), 0, 0, d_rank, N, A.arc_start, A.arc_cnt, A.arcs, d_row_new, d_out);
        CHK(hipGetLastError());
        CHK(hipDeviceSynchronize());
        res = new jd_net();
        res->n_states = N; res->n_arcs = M;
        res->row_ptr.resize((size_t)N + 1); res->arcs.resize((size_t)M); res->fin_w.resize((size_t)N);
        CHK(hipMemcpy(res->row_ptr.data(), d_row_new, (size_t)(N + 1) * 4, hipMemcpyDeviceToHost));
        if (M) CHK(hipMemcpy(res->arcs.data(), d_out, (size_t)M * sizeof(JdArc), hipMemcpyDeviceToHost));
        CHK(hipMemcpy(res->fin_w.data(), d_fin_new, (size_t)N * 4, hipMemcpyDeviceToHost));
        int init_new = 0;
        CHK(hipMemcpy(&init_new, d_rank, 4, hipMemcpyDeviceToHost));
        res->init = init_new;
        res->n_final = 0; res->max_in = 0;
        for (float f : res->fin_w) res->n_final += f < std::numeric_limits<float>::infinity();
        for (const JdArc &a : res->arcs) res->max_in = std::max(res->max_in, a.in);
        // the arc weights carry both networks' scalings already; a binary dump of the composed graph
        // (jd_net_save_jwnt) stores them as they are
        res->lm_scale = 1.0f; res->ins_penalty = 0.0f;
        if (getenv("JD_VERBOSE")) fprintf(stderr, "jd_net_compose: %d states, %lld arcs, %d breadth-first levels\n", N, M, level);
        *out = res;
        res = nullptr;
    }
done:
    for (void *p : allocs) (void)hipFree(p);
    delete res;
    return rc;
}


// ------------------------------------------------------------------ search-driven composition
//
// jd_net_create_lazy: C.L and G stay apart on the device; what jd_dec_create gets is a network whose arc
// arena is empty but for the start state (and its epsilon closure).  The search kernel expands composed
// states as its tokens approach them (jd_lazy.h); the arena is shared by every decoder / stream that uses
// the network and lives as long as it does.

__global__ void jl_init(LazyDev L, const float *hmm_tee, unsigned cf0, int g0, int *ok)
{
    __shared__ int2 stk[LZQ];
    int s0 = 0;
    if (threadIdx.x == 0) s0 = lz_state_id(L, cf0, g0);
    s0 = __shfl(s0, 0);
    const bool good = lz_close(L, hmm_tee, s0, stk, wall_clock64() + 300000000LL);
    if (threadIdx.x == 0) { ok[0] = (good && !lz_failed(L)) ? 1 : 0; ok[1] = s0; }
}

static void lazy_free(jd_net *n)
{
    if (n->lazy_device >= 0) (void)hipSetDevice(n->lazy_device);
    for (void *p : n->lazy_allocs) (void)hipFree(p);
    n->lazy_allocs.clear();
    n->lazy_dev = nullptr;
}

// empties the graph (hash table, rows, arena counters, failure flag) and expands the start state with its closure
static int lazy_start(jd_net *n, const LazyDev &L, int *h_ok)
{
    int rc = JD_OK;
    CHK(hipMemset(L.keys, 0, (size_t)(L.mask + 1) * 8)); CHK(hipMemset(L.vals, 0xff, (size_t)(L.mask + 1) * 4));
    CHK(hipMemset(L.n_states, 0, 4)); CHK(hipMemset(L.n_arcs, 0, 8)); CHK(hipMemset(L.err, 0, 4));
    CHK(hipMemset(L.rows, 0, (size_t)L.max_states * sizeof(int4)));
    hipLaunchKernelGGL(jl_init, dim3(1), dim3(64), 0, 0, L, (const float *)n->lazy_tee, (unsigned)n->lazy_cf0, (int)n->lazy_g0, (int *)n->lazy_ok);
    CHK(hipGetLastError());
    CHK(hipMemcpy(h_ok, n->lazy_ok, 2 * sizeof(int), hipMemcpyDeviceToHost));
    if (!h_ok[0]) rc = jd_fail(JD_ENOMEM, "lazily composed network: capacities too small for the start state's closure");
done:
    return rc;
}

extern "C" int jd_net_create_lazy(jd_net **out, const jd_net *cl, const jd_net *g, const jd_am *am, int32_t device,
                                  int64_t max_states, int64_t max_arcs, int32_t pushing)
{
    if (!out || !cl || !g || !am) return jd_fail(JD_EINVAL, "jd_net_create_lazy: null argument");
    if (cl->lazy_dev || g->lazy_dev) return jd_fail(JD_EINVAL, "jd_net_create_lazy: the inputs must be ordinary networks");
    if (pushing < 0 || pushing > 3) return jd_fail(JD_EINVAL, "jd_net_create_lazy: pushing is a combination of JD_PUSH_WEIGHTS and JD_PUSH_LABELS");
    std::vector<JdArc> g_sorted;
    int rc = sorted_g_arcs(g, g_sorted);
    if (rc) return rc;
    struct Owned { jd_net *p = nullptr; ~Owned() { delete p; } } pushed;   // C.L with its labels pushed (JD_PUSH_LABELS)
    if (pushing & 2) {
        std::vector<JdArc> pa;
        cl_push_labels(cl, pa, nullptr);
        pushed.p = net_with_arcs(cl, std::move(pa));
        cl = pushed.p;
    }
    if (cl->max_in > am->n_hmm) return jd_fail(JD_EINVAL, "network input label %d exceeds the number of HMMs %d", cl->max_in, am->n_hmm);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return jd_fail(JD_ENODEV, "jd_net_create_lazy: no HIP device %d (there is no CPU path)", device);
    if (hipSetDevice(device) != hipSuccess) return jd_fail(JD_EHIP, "hipSetDevice(%d) failed", device);
    if (max_states <= 0) max_states = 1 << 22;
    if (max_arcs <= 0) max_arcs = 1 << 24;
    if (max_states > 0x0fffffffLL || max_arcs > 0x0fffffffLL) return jd_fail(JD_EINVAL, "jd_net_create_lazy: capacities are limited to 2^28 states / arcs");
    jd_net *n = new jd_net();
    n->lazy_device = device; n->lazy_free = lazy_free;
    LazyDev L;
    memset(&L, 0, sizeof L);
    auto dal = [&](size_t bytes) -> void * { void *p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; n->lazy_allocs.push_back(p); return p; };
    int *d_ok = nullptr;
    float *d_tee = nullptr;
    LazyDev *d_L = nullptr;
    int h_ok[2] = {0, 0};
    {
        size_t cap = 1;
        while (cap < (size_t)max_states * 2) cap <<= 1;
        L.mask = cap - 1;
#define LAL(p, T, cnt) do { p = (T *)dal(sizeof(T) * (size_t)(cnt)); if (!p) { rc = jd_fail(JD_ENOMEM, "jd_net_create_lazy: hipMalloc of %zu bytes failed", sizeof(T) * (size_t)(cnt)); goto done; } } while (0)
        int *cl_row, *g_row; JdArc *cl_arcs, *g_arcs; float *cl_fin, *g_fin; int2 *cl_la;
        std::vector<int2> la;
        cl_lookahead(cl, la);
        LAL(cl_row, int, cl->row_ptr.size()); LAL(cl_arcs, JdArc, cl->arcs.size()); LAL(cl_fin, float, cl->fin_w.size()); LAL(cl_la, int2, la.size());
        LAL(g_row, int, g->row_ptr.size()); LAL(g_arcs, JdArc, g_sorted.size()); LAL(g_fin, float, g->fin_w.size());
        CHK(hipMemcpy(cl_row, cl->row_ptr.data(), cl->row_ptr.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(cl_arcs, cl->arcs.data(), cl->arcs.size() * sizeof(JdArc), hipMemcpyHostToDevice));
        CHK(hipMemcpy(cl_fin, cl->fin_w.data(), cl->fin_w.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(cl_la, la.data(), la.size() * sizeof(int2), hipMemcpyHostToDevice));
        CHK(hipMemcpy(g_row, g->row_ptr.data(), g->row_ptr.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(g_arcs, g_sorted.data(), g_sorted.size() * sizeof(JdArc), hipMemcpyHostToDevice));
        CHK(hipMemcpy(g_fin, g->fin_w.data(), g->fin_w.size() * 4, hipMemcpyHostToDevice));
        L.cl_row = cl_row; L.cl_arcs = cl_arcs; L.cl_fin = cl_fin; L.cl_la = cl_la; L.g_row = g_row; L.g_arcs = g_arcs; L.g_fin = g_fin;
        LAL(L.keys, unsigned long long, cap); LAL(L.vals, int, cap);
        LAL(L.st_c, int, max_states); LAL(L.st_g, int, max_states); LAL(L.rows, int4, max_states);
        LAL(L.arcs, JdArc, max_arcs); LAL(L.n_states, int, 1); LAL(L.n_arcs, unsigned long long, 1); LAL(L.err, int, 1);
        LAL(d_ok, int, 2); LAL(d_tee, float, am->hmm_tee.size()); LAL(d_L, LazyDev, 1);
        L.max_states = (int)max_states; L.max_arcs = max_arcs; L.push = (pushing & 1) ? 1 : 0;
        CHK(hipMemcpy(d_tee, am->hmm_tee.data(), am->hmm_tee.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d_L, &L, sizeof L, hipMemcpyHostToDevice));
        n->lazy_tee = d_tee; n->lazy_ok = d_ok; n->lazy_cf0 = (uint32_t)cl->init | LZ_FLAG; n->lazy_g0 = g->init;
        rc = lazy_start(n, L, h_ok);                 // the start state and its closure
        if (rc) goto done;
        n->n_states = (int32_t)max_states; n->n_arcs = max_arcs; n->init = h_ok[1];
        n->n_final = 0; n->max_in = cl->max_in;
        n->lm_scale = 1.0f; n->ins_penalty = 0.0f;
        n->lazy_dev = d_L;
        *out = n;
        n = nullptr;
    }
done:
    if (n) { lazy_free(n); delete n; }
    return rc;
}

// Forget everything that has been expanded (the reference bounds its look-ahead memory with an LRU cache,
// WFSTOnTheFlyDecoder.h:210-371; here the arena simply starts again): for a long-running service whose network is
// about to run out of room, or has.  No stream of any decoder on the network may be inside an utterance.
extern "C" int jd_net_lazy_reset(jd_net *n)
{
    if (!n || !n->lazy_dev) return jd_fail(JD_EINVAL, "jd_net_lazy_reset: not a lazily composed network");
    std::lock_guard<std::mutex> lk(n->lazy_mu);
    if (n->lazy_busy > 0) return jd_fail(JD_ESTATE, "jd_net_lazy_reset: %d utterance(s) are being decoded on the network", n->lazy_busy);
    if (hipSetDevice(n->lazy_device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return jd_fail(JD_EHIP, "jd_net_lazy_reset: device %d", n->lazy_device);
    LazyDev L;
    if (hipMemcpy(&L, n->lazy_dev, sizeof L, hipMemcpyDeviceToHost) != hipSuccess) return jd_fail(JD_EHIP, "jd_net_lazy_reset: copy failed");
    int h_ok[2] = {0, 0};
    const int rc = lazy_start(n, L, h_ok);
    if (rc) return rc;
    if (h_ok[1] != n->init) return jd_fail(JD_ESTATE, "jd_net_lazy_reset: the start state moved");   // (first insertion: always 0)
    ++n->lazy_generation;
    return JD_OK;
}

// Bounded look-ahead memory.  The reference keeps what it composes in an LRU cache and evicts entry by entry
// (WFSTOnTheFlyDecoder.h:210-371) - its tokens hold pointers into cache entries that are pinned while in use.  Here
// instance records hold arc and state NUMBERS of one shared arena that thousands of lanes append to; evicting single
// states would need every stream's records re-validated.  So eviction is by GENERATION: the arena starts again, at a
// moment when no stream of any decoder is inside an utterance (records of a finished utterance are dead), when it is
// past its high-water mark (jd_net_lazy_set_high_water, default 0.9 of either capacity) or has run out of room.  A
// batch that runs out of room under way is decoded again on a fresh generation (jd_decode_batch_device), so running
// out is only an error when ONE batch needs more than the capacities - never a sticky state of the network.
static int lazy_fill(const jd_net *n, const LazyDev &L, int *ns, unsigned long long *na, int *err)
{
    int rc = JD_OK;
    CHK(hipMemcpy(ns, L.n_states, 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(na, L.n_arcs, 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(err, L.err, 4, hipMemcpyDeviceToHost));
done:
    (void)n;
    return rc;
}

int jd_lazy_enter(const jd_net *n, int n_utts, bool *failed)
{
    if (failed) *failed = false;
    if (!n || !n->lazy_dev) return JD_OK;
    std::lock_guard<std::mutex> lk(n->lazy_mu);
    if (hipSetDevice(n->lazy_device) != hipSuccess) return jd_fail(JD_EHIP, "hipSetDevice(%d) failed", n->lazy_device);
    LazyDev L;
    if (hipMemcpy(&L, n->lazy_dev, sizeof L, hipMemcpyDeviceToHost) != hipSuccess) return jd_fail(JD_EHIP, "lazily composed network: copy failed");
    int ns = 0, err = 0;
    unsigned long long na = 0;
    int rc = lazy_fill(n, L, &ns, &na, &err);
    if (rc) return rc;
    const bool past = (double)ns > n->lazy_high_water * (double)L.max_states || (double)na > n->lazy_high_water * (double)L.max_arcs;
    if ((err != 0 || past) && n->lazy_busy == 0) {
        if (hipDeviceSynchronize() != hipSuccess) return jd_fail(JD_EHIP, "lazily composed network: device %d", n->lazy_device);
        int h_ok[2] = {0, 0};
        rc = lazy_start(const_cast<jd_net *>(n), L, h_ok);
        if (rc) return rc;
        if (h_ok[1] != n->init) return jd_fail(JD_ESTATE, "lazily composed network: the start state moved");
        ++n->lazy_generation;
        err = 0;
        if (getenv("JD_VERBOSE")) fprintf(stderr, "lazy network: generation %lld starts (%d states, %llu arcs dropped)\n", (long long)n->lazy_generation, ns, na);
    }
    if (failed) *failed = err != 0;
    n->lazy_busy += n_utts;
    return JD_OK;
}

void jd_lazy_leave(const jd_net *n, int n_utts)
{
    if (!n || !n->lazy_dev) return;
    std::lock_guard<std::mutex> lk(n->lazy_mu);
    n->lazy_busy = n->lazy_busy > n_utts ? n->lazy_busy - n_utts : 0;
}

extern "C" int jd_net_lazy_set_high_water(jd_net *n, double fraction)
{
    if (!n || !n->lazy_dev) return jd_fail(JD_EINVAL, "jd_net_lazy_set_high_water: not a lazily composed network");
    if (!(fraction > 0.0 && fraction <= 1.0)) return jd_fail(JD_EINVAL, "jd_net_lazy_set_high_water: fraction in (0, 1]");
    std::lock_guard<std::mutex> lk(n->lazy_mu);
    n->lazy_high_water = fraction;
    return JD_OK;
}

extern "C" int jd_net_lazy_generation(const jd_net *n, int64_t *generation)
{
    if (!n || !n->lazy_dev || !generation) return jd_fail(JD_EINVAL, "jd_net_lazy_generation: not a lazily composed network");
    std::lock_guard<std::mutex> lk(n->lazy_mu);
    *generation = n->lazy_generation;
    return JD_OK;
}

// how far a lazily composed network has grown: composed states and arcs materialised so far
extern "C" int jd_net_lazy_size(const jd_net *n, int64_t *states, int64_t *arcs)
{
    if (!n || !n->lazy_dev) return jd_fail(JD_EINVAL, "jd_net_lazy_size: not a lazily composed network");
    if (hipSetDevice(n->lazy_device) != hipSuccess) return jd_fail(JD_EHIP, "hipSetDevice failed");
    LazyDev L;
    int ns = 0; unsigned long long na = 0;
    if (hipMemcpy(&L, n->lazy_dev, sizeof L, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&ns, L.n_states, 4, hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(&na, L.n_arcs, 8, hipMemcpyDeviceToHost) != hipSuccess) return jd_fail(JD_EHIP, "jd_net_lazy_size: copy failed");
    if (states) *states = ns;
    if (arcs) *arcs = (int64_t)na;
    return JD_OK;
}
