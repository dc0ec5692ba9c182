// jd_gc.h - what runs BETWEEN launches of the persistent search kernel (included by jd_device.hip; gfx950 only):
// the Path collection (collectPaths, src/WFSTDecoderLite.cpp:699-747) as a mark-compact of six kernels, the
// PARTIAL_DECODING trace (tracePartialPath, :824-868), recognitionFinish's walk (:230-309) and the small
// per-launch helper kernels.
#pragma once

// Path garbage collection = collectPaths (WFSTDecoderLite.cpp:699-747) as a mark-compact: records
// reachable from a live token, from a frontier item of the last processed frame (the pending
// entry-token candidates point at those) or from bestFinalToken are kept, everything else is
// dropped.  No effect on results.  What the REFERENCE keeps is a subset of that - the records the tokens of its
// active instances reach (:703-719): the emitting tokens of the instance records and the entry token of every
// state that received a token in the last frame and has an arc with a model; not the exit tokens (nulled, :964), not
// bestFinalToken.  Those roots mark with bit 1 as well, and their count is the reference's nPathNew (path_new_ref)
// while everything marked is kept.  Run between launches for the streams that stopped for it, G
// 1024-thread workgroups per stream; the steps are separate kernels (a kernel boundary is the
// barrier between them): begin (decide, clear the marks) - mark - sum (marks per workgroup range) -
// scan (new indices) - compact (into the second arena, predecessors remapped) - remap (tokens, items,
// bestFinalToken; swap the arenas).
#define GC_MAXG 32
struct GcState { int active, kept, kept_ref, by_rule; int part[GC_MAXG], part_ref[GC_MAXG]; };

struct GcCtx {
    int s, blk, G, np, nw, p;
    Geo g;
};
__device__ __forceinline__ bool gc_ctx(const DecConst &C, const StreamCtl *ctl, const int4 *work, int s_single, int G, GcCtx &x)
{
    const int wi = blockIdx.x / G;
    x.blk = blockIdx.x % G; x.G = G;
    x.s = work ? work[wi].x : s_single;
    const StreamCtl &c = ctl[x.s];
    x.np = c.n_paths; x.nw = c.lst_nw;
    x.p = c.frame & 1;                                // list the next frame reads; items of the last frame: parity p^1
    if (x.nw > 0) x.g = make_geo(C, x.nw);
    return true;
}
// the range of Path records workgroup blk of G looks after (multiples of 1024)
__device__ __forceinline__ void gc_range(const GcCtx &x, int &lo, int &hi)
{
    const long long per = ((((long long)x.np + x.G - 1) / x.G) + 1023) & ~1023LL;
    lo = (int)min((long long)x.np, per * x.blk);
    hi = (int)min((long long)x.np, per * (x.blk + 1));
}

__global__ __launch_bounds__(1024) void k_gc_begin(DecConst C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int s_single, int G)
{
    GcCtx x;
    gc_ctx(C, ctl, work, s_single, G, x);
    const StreamCtl &c = ctl[x.s];
    StreamDev &S = streams[x.s];
    // (gc_threshold < 0: the host asks for this collection - the frame rule, or a count rule that fires behind a chunk's
    // last frame; only the collections the reference runs too count as such and restart its counts)
    const bool rule = C.path_rule && (C.gc_threshold < 0 || path_rule_fires(C.pcount ? c.n_paths_ref : x.np, C.pcount ? c.path_new_ref : c.path_new));
    const bool active = c.started && !c.needs_init && c.error == 0 && x.nw > 0 && (x.np > C.gc_threshold || rule);
    GcState *gs = (GcState *)S.gc_state;
    if (x.blk == 0 && threadIdx.x == 0) { gs->active = active ? 1 : 0; gs->kept = 0; gs->kept_ref = 0; gs->by_rule = rule ? 1 : 0; }
    if (!active) return;
    int lo, hi;
    gc_range(x, lo, hi);
    // (ranges start at multiples of 1024 and the arena is 256-byte aligned: 16-byte stores, a scalar tail)
    const int hi4 = lo + ((hi - lo) & ~3);
    for (int q = lo + 4 * (int)threadIdx.x; q < hi4; q += 4 * (int)blockDim.x) *(int4 *)(S.gc_idx + q) = make_int4(0, 0, 0, 0);
    for (int q = hi4 + threadIdx.x; q < hi; q += blockDim.x) S.gc_idx[q] = 0;
}

template <int NE>
__global__ __launch_bounds__(1024) void k_gc_mark(DecConst C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int s_single, int G)
{
    typedef RecLayout<NE> RL;
    GcCtx x;
    gc_ctx(C, ctl, work, s_single, G, x);
    StreamDev &S = streams[x.s];
    if (!((const GcState *)S.gc_state)->active) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int *idx = S.gc_idx;
    // bit 0: kept; bit 1: the reference keeps it too.  (A walk goes on as long as it adds a bit: one that sets both passes
    // records that an "extra" walk marked before it.)
    auto mark = [&](int q) { while (q >= 0 && (atomicOr(&idx[q], 1) & 1) == 0) q = S.paths[q].prev; };
    auto mark_ref = [&](int q) { while (q >= 0 && (atomicOr(&idx[q], 3) & 2) == 0) q = S.paths[q].prev; };
    // tokens of the instance records (structure-of-arrays chunks of 64) ...
    for (int w = x.blk * 16 + wid; w < x.nw; w += G * 16) {
        const char *seg = (const char *)S.rec + (size_t)x.p * C.cap_slots * RL::REC_BYTES + (size_t)w * (x.g.seg_rec >> 6) * RL::CHUNK_BYTES;
        const int n_rec = min(S.tot[(size_t)(TOT_REC0 + x.p) * MAXW + w], (int)x.g.seg_rec);
        for (int k = lane; k < n_rec * NE; k += 64) {
            const int q = k / NE, j = k - q * NE + 1;
            const char *r = seg + (size_t)(q >> 6) * RL::CHUNK_BYTES + (size_t)(q & 63) * 16;
            const int n = ((const int4 *)r)->y & 0xff;
            if (j < n - 1) {
                const int4 t = *(const int4 *)(r + (size_t)(RL::HF + j - 1) * 1024);
                if (__int_as_float(t.x) > LZ) mark_ref(t.w);
            }
        }
        // ... and of the last frame's frontier items
        const int n_it = min(S.item_end[w], (int)x.g.seg_item);
        for (int k = lane; k < n_it; k += 64) mark(S.items[2 * ((size_t)(x.p ^ 1) * C.cap_items + (size_t)w * x.g.seg_item + k)].w);
    }
    if (x.blk == 0 && tid == 0) mark(ctl[x.s].best_final.path);
    if (C.pcount != nullptr) {
        // the entry tokens of the reference's instances: every arc with a model that leaves a state of the last frame's
        // dirty list holds the best token that arrived there (one Path for all of them)
        const StreamCtl &c = ctl[x.s];
        const int dn = c.dirty_nw[x.p ^ 1];
        const Geo gd = make_geo(C, dn > 0 ? dn : x.nw);
        const int *dl = S.dirtyl + (size_t)(x.p ^ 1) * C.cap_new;
        for (int w = x.blk * 16 + wid; w < gd.nw; w += G * 16) {
            const int n_d = min(S.tot[(size_t)(TOT_DIRTY0 + (x.p ^ 1)) * MAXW + w], (int)gd.seg_new);
            for (int q = lane; q < n_d; q += 64) {
                const int st = dl[(size_t)w * gd.seg_new + q];
                const unsigned long long kv = SREC_E(S.srec, C, st, x.p ^ 1);
                if (kv == 0ULL) continue;
                bool has_model = false;
                for (int a = C.row_ptr[st]; a < C.row_ptr[st + 1] && !has_model; ++a) has_model = (C.arcs[a].in & ~ARC_FLAGS) != 0;
                if (has_model) mark_ref(S.items[2 * ((size_t)(x.p ^ 1) * C.cap_items + (size_t)(kv & 0xffffffffULL))].w);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_gc_sum(DecConst C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int s_single, int G)
{
    GcCtx x;
    gc_ctx(C, ctl, work, s_single, G, x);
    StreamDev &S = streams[x.s];
    GcState *gs = (GcState *)S.gc_state;
    if (!gs->active) return;
    __shared__ int sh_sum, sh_ref;
    if (threadIdx.x == 0) { sh_sum = 0; sh_ref = 0; }
    __syncthreads();
    int lo, hi, mine = 0, ref = 0;
    gc_range(x, lo, hi);
    const int hi4 = lo + ((hi - lo) & ~3);
    for (int q = lo + 4 * (int)threadIdx.x; q < hi4; q += 4 * (int)blockDim.x) {
        const int4 m = *(const int4 *)(S.gc_idx + q);
        mine += (m.x & 1) + (m.y & 1) + (m.z & 1) + (m.w & 1);
        ref += (m.x >> 1) + (m.y >> 1) + (m.z >> 1) + (m.w >> 1);
    }
    for (int q = hi4 + threadIdx.x; q < hi; q += blockDim.x) { const int m = S.gc_idx[q]; mine += m & 1; ref += m >> 1; }
#pragma unroll
    for (int o = 32; o; o >>= 1) { mine += __shfl_xor(mine, o); ref += __shfl_xor(ref, o); }
    if ((threadIdx.x & 63) == 0 && mine) { atomicAdd(&sh_sum, mine); atomicAdd(&sh_ref, ref); }
    __syncthreads();
    if (threadIdx.x == 0) { gs->part[x.blk] = sh_sum; gs->part_ref[x.blk] = sh_ref; }
}

// exclusive scan of the marks -> new indices (idx[q] = new index, -1 if dropped)
__global__ __launch_bounds__(1024) void k_gc_scan(DecConst C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int s_single, int G)
{
    GcCtx x;
    gc_ctx(C, ctl, work, s_single, G, x);
    StreamDev &S = streams[x.s];
    GcState *gs = (GcState *)S.gc_state;
    if (!gs->active) return;
    __shared__ int sh_w[16];
    __shared__ int sh_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int *idx = S.gc_idx;
    if (tid == 0) {
        int base = 0, all = 0, ref = 0;
        for (int b = 0; b < G; ++b) { if (b < x.blk) base += gs->part[b]; all += gs->part[b]; ref += gs->part_ref[b]; }
        sh_carry = base;
        if (x.blk == 0) { gs->kept = all; gs->kept_ref = ref; }
    }
    __syncthreads();
    int lo, hi;
    gc_range(x, lo, hi);
    // four marks per thread and step (16-byte accesses; a step behind the range's end is done element by element)
    for (int b0 = lo; b0 < hi; b0 += 4096) {
        const int q = b0 + 4 * tid;
        int4 m = make_int4(0, 0, 0, 0);
        if (q + 3 < hi) m = *(const int4 *)(idx + q);
        else { if (q < hi) m.x = idx[q]; if (q + 1 < hi) m.y = idx[q + 1]; if (q + 2 < hi) m.z = idx[q + 2]; }
        m.x &= 1; m.y &= 1; m.z &= 1; m.w &= 1;                        // (bit 1: the reference's survivors, counted by k_gc_sum)
        const int mine = m.x + m.y + m.z + m.w;
        int v = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(v, o); if (lane >= o) v += y; }
        if (lane == 63) sh_w[wid] = v;
        __syncthreads();
        int base = sh_carry, tot = 0;
        for (int w = 0; w < 16; ++w) { const int u = sh_w[w]; if (w < wid) base += u; tot += u; }
        base += v - mine;                                              // marks before this thread's four
        int4 o4;
        o4.x = m.x ? base : -1; base += m.x;
        o4.y = m.y ? base : -1; base += m.y;
        o4.z = m.z ? base : -1; base += m.z;
        o4.w = m.w ? base : -1;
        if (q + 3 < hi) *(int4 *)(idx + q) = o4;
        else { if (q < hi) idx[q] = o4.x; if (q + 1 < hi) idx[q + 1] = o4.y; if (q + 2 < hi) idx[q + 2] = o4.z; }
        __syncthreads();
        if (tid == 0) sh_carry += tot;
        __syncthreads();
    }
}

// compact into the second arena, remapping prev (prev < q: its new index is final after the scan)
__global__ __launch_bounds__(1024) void k_gc_compact(DecConst C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int s_single, int G)
{
    GcCtx x;
    gc_ctx(C, ctl, work, s_single, G, x);
    StreamDev &S = streams[x.s];
    if (!((const GcState *)S.gc_state)->active) return;
    const int *idx = S.gc_idx;
    int lo, hi;
    gc_range(x, lo, hi);
    // four new indices per 16-byte load; the kept ones' records, then their predecessors' new indices, are requested
    // together (one chain of dependent round trips per four records, not four)
    const int hi4 = lo + ((hi - lo) & ~3);
    for (int q = lo + 4 * (int)threadIdx.x; q < hi4; q += 4 * (int)blockDim.x) {
        const int4 nv = *(const int4 *)(idx + q);
        const int ni[4] = {nv.x, nv.y, nv.z, nv.w};
        PathRec pr[4];
        int np[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (ni[k] >= 0) pr[k] = S.paths[q + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) np[k] = (ni[k] >= 0 && pr[k].prev >= 0) ? idx[pr[k].prev] : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (ni[k] >= 0) { pr[k].prev = np[k]; S.paths2[ni[k]] = pr[k]; }
    }
    for (int q = hi4 + threadIdx.x; q < hi; q += blockDim.x) {
        const int ni = idx[q];
        if (ni >= 0) {
            PathRec pr = S.paths[q];
            pr.prev = (pr.prev >= 0) ? idx[pr.prev] : -1;
            S.paths2[ni] = pr;
        }
    }
}

template <int NE>
__global__ __launch_bounds__(1024) void k_gc_remap(DecConst C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int s_single, int G)
{
    typedef RecLayout<NE> RL;
    GcCtx x;
    gc_ctx(C, ctl, work, s_single, G, x);
    StreamDev &S = streams[x.s];
    const GcState *gs = (const GcState *)S.gc_state;
    if (!gs->active) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *idx = S.gc_idx;
    for (int w = x.blk * 16 + wid; w < x.nw; w += G * 16) {
        char *seg = (char *)S.rec + (size_t)x.p * C.cap_slots * RL::REC_BYTES + (size_t)w * (x.g.seg_rec >> 6) * RL::CHUNK_BYTES;
        const int n_rec = min(S.tot[(size_t)(TOT_REC0 + x.p) * MAXW + w], (int)x.g.seg_rec);
        for (int k = lane; k < n_rec * NE; k += 64) {
            const int q = k / NE, j = k - q * NE + 1;
            char *r = seg + (size_t)(q >> 6) * RL::CHUNK_BYTES + (size_t)(q & 63) * 16;
            const int n = ((const int4 *)r)->y & 0xff;
            if (j < n - 1) {
                int4 *t = (int4 *)(r + (size_t)(RL::HF + j - 1) * 1024);
                if (t->w >= 0) t->w = (__int_as_float(t->x) > LZ) ? idx[t->w] : -1;
            }
        }
        const int n_it = min(S.item_end[w], (int)x.g.seg_item);
        for (int k = lane; k < n_it; k += 64) {
            int4 *t = S.items + 2 * ((size_t)(x.p ^ 1) * C.cap_items + (size_t)w * x.g.seg_item + k);   // token half of the item
            if (t->w >= 0) t->w = idx[t->w];
        }
    }
    if (x.blk == 0 && tid == 0) {                     // (no workgroup of this kernel reads the Path arenas)
        StreamCtl &c = ctl[x.s];
        if (c.best_final.path >= 0) c.best_final.path = idx[c.best_final.path];
        PathRec *tmp = S.paths; S.paths = S.paths2; S.paths2 = tmp;
        c.n_paths = gs->kept;
        c.path_new = gs->kept;                        // nPathNew = nPath (:745)
        if (C.pcount == nullptr) c.n_collect += 1;
        else if (gs->by_rule) {                       // a collection of the reference's: its counts start again from its survivors
            c.n_paths_ref = gs->kept_ref; c.path_new_ref = gs->kept_ref; c.n_collect += 1;
        }
    }
}

// the six steps for the streams of a work list (or one stream), on stream st
static void launch_gc(const DecConst &C, StreamCtl *ctl, StreamDev *streams, const int4 *work, int n_work, int s_single, bool ne3,
                      int n_cus, hipStream_t st)
{
    const int G = std::max(1, std::min(GC_MAXG, n_cus / std::max(1, n_work)));
    const dim3 grid((unsigned)(n_work * G)), blk(1024);
    hipLaunchKernelGGL(k_gc_begin, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    if (ne3) hipLaunchKernelGGL(k_gc_mark<3>, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    else hipLaunchKernelGGL(k_gc_mark<6>, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    hipLaunchKernelGGL(k_gc_sum, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    hipLaunchKernelGGL(k_gc_scan, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    hipLaunchKernelGGL(k_gc_compact, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    if (ne3) hipLaunchKernelGGL(k_gc_remap<3>, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
    else hipLaunchKernelGGL(k_gc_remap<6>, grid, blk, 0, st, C, ctl, streams, work, s_single, G);
}

// PARTIAL_DECODING: tracePartialPath (WFSTDecoderLite.cpp:824-868) on the state a launch left behind.
//
// The reference walks back from the first token with a Path of every active instance, counts the
// visits per Path record (records newer than the last traced one only) and stops at the first record
// that all nActiveInsts walks reach: the deepest record common to all of them.  Here: the instances
// are the records of the next frame's list plus the arcs entered in the last frame that have no
// record yet (new list, and the clean-up list of the "hopeless" ones - the reference attached an
// instance for those too); an instance's first token is its entry token - the candidate that won the
// arc's key in the last phase X - then its emitting states in order.  Path indices grow along a
// chain (a record is allocated after its predecessor, and the collection keeps the order), so the common
// record lies on the chain of ANY tip: the chain of the highest tip is written out, every other tip
// walks down until it meets it, and the shallowest meeting point is the answer.
// out[0] = found, out[1] = records on the chain from the root to the found one (oldest first in
// res_label / res_time, at most res_cap of them).
template <int NE, typename F>
__device__ __forceinline__ void jd_for_each_tip(const DecConst &C, const StreamCtl &c, const StreamDev &S, F &&f)
{
    typedef RecLayout<NE> RL;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = c.lst_nw;
    const Geo g = make_geo(C, nw);
    const int p = c.frame & 1;
    // the pending entry token of the arcs leaving state st: the best token that arrived there in the last frame
    auto entry_tip = [&](int st) -> int {
        const unsigned long long kv = SREC_E(S.srec, C, st, p ^ 1);
        if (kv == 0ULL) return -1;
        return S.items[2 * ((size_t)(p ^ 1) * C.cap_items + (size_t)(kv & 0xffffffffULL))].w;
    };
    for (int w = wid; w < nw; w += 16) {
        const char *seg = (const char *)S.rec + (size_t)p * C.cap_slots * RL::REC_BYTES + (size_t)w * (g.seg_rec >> 6) * RL::CHUNK_BYTES;
        const int n_rec = min(S.tot[(size_t)(TOT_REC0 + p) * MAXW + w], (int)g.seg_rec);
        for (int q = lane; q < n_rec; q += 64) {
            const char *r = seg + (size_t)(q >> 6) * RL::CHUNK_BYTES + (size_t)(q & 63) * 16;
            const int4 h0 = *(const int4 *)r;
            int tip = entry_tip(h0.z);                                  // (h0.z: the source state of the instance's arc)
            const int n = h0.y & 0xff;
            for (int j = 1; j <= NE && tip < 0; ++j)
                if (j < n - 1) tip = ((const int4 *)(r + (size_t)(RL::HF + j - 1) * 1024))->w;
            f(tip);
        }
    }
    // ... and the arcs entered in the last frame that have no record yet (the reference attached an instance to every
    // one of them, hopeless or not): all of them hold the token that arrived at their source state, so every state
    // of the last frame's dirty list that has an arc with a model is one tip
    const int dn = c.dirty_nw[p ^ 1];
    const Geo gd = make_geo(C, dn > 0 ? dn : nw);
    const int *dl = S.dirtyl + (size_t)(p ^ 1) * C.cap_new;
    for (int w = wid; w < gd.nw; w += 16) {
        const int n_d = min(S.tot[(size_t)(TOT_DIRTY0 + (p ^ 1)) * MAXW + w], (int)gd.seg_new);
        for (int q = lane; q < n_d; q += 64) {
            const int st = dl[(size_t)w * gd.seg_new + q];
            bool has_model = false;
            if (C.lazy) {
                const int4 row = C.lazy->rows[st];
                for (int a = row.x; a < row.x + row.y && !has_model; ++a) has_model = (C.lazy->arcs[a].in & ~ARC_FLAGS) != 0;
            } else
                for (int a = C.row_ptr[st]; a < C.row_ptr[st + 1] && !has_model; ++a) has_model = (C.arcs[a].in & ~ARC_FLAGS) != 0;
            if (has_model) f(entry_tip(st));
        }
    }
}

template <int NE>
__global__ __launch_bounds__(1024) void k_partial(DecConst C, StreamCtl *ctl, StreamDev *streams, int s, int last_frame, int *out)
{
    StreamCtl &c = ctl[s];
    StreamDev &S = streams[s];
    __shared__ int sh_max, sh_bad, sh_cnt, sh_depth, sh_D;
    const int tid = threadIdx.x;
    if (tid == 0) { sh_max = -1; sh_bad = 0; sh_cnt = 0; sh_depth = 0; sh_D = 0; out[0] = 0; out[1] = 0; }
    __syncthreads();
    if (!c.started || c.needs_init || c.error != 0 || c.lst_nw <= 0) return;
    {
        int mx = -1, bad = 0, cnt = 0;
        jd_for_each_tip<NE>(C, c, S, [&](int tip) { ++cnt; if (tip < 0) bad = 1; else mx = max(mx, tip); });
        if (mx >= 0) atomicMax(&sh_max, mx);
        if (bad) atomicOr(&sh_bad, 1);
        if (cnt) atomicAdd(&sh_cnt, cnt);
    }
    __syncthreads();
    // an instance none of whose tokens has a Path yet: nothing can be common to all (:850-854)
    if (sh_cnt == 0 || sh_bad || sh_max < 0) return;
    int *ch = S.gc_idx;                                                // the chain of the highest tip, newest first
    if (tid == 0) {
        int n = 0;
        for (int q = sh_max; q >= 0; q = S.paths[q].prev) ch[n++] = q;
        sh_depth = n;
    }
    __syncthreads();
    const int depth = sh_depth;
    {
        int dmax = 0;
        jd_for_each_tip<NE>(C, c, S, [&](int tip) {
            int i = 0, q = tip;
            for (;;) {
                while (i < depth && ch[i] > q) ++i;
                if (i == depth || ch[i] == q) break;
                q = S.paths[q].prev;
                if (q < 0) { i = depth; break; }
            }
            dmax = max(dmax, i);
        });
        if (dmax) atomicMax(&sh_D, dmax);
    }
    __syncthreads();
    const int D0 = sh_D;
    if (D0 >= depth) return;                                           // no record is common to all
    if (S.paths[ch[D0]].frame <= last_frame) return;                   // nothing newer than the last traced record (:858)
    const int n = depth - D0;
    for (int k = tid; k < n && k < S.res_cap; k += blockDim.x) {       // traceWinningPaths :874-890, oldest first
        const PathRec pr = S.paths[ch[depth - 1 - k]];
        S.res_label[k] = pr.label; S.res_time[k] = pr.frame;
    }
    if (tid == 0) { out[0] = 1; out[1] = n; }
}

// recognitionFinish (:230-309): walk the Path chain of bestFinalToken.
__global__ void jd_finish_kernel(StreamCtl *ctl, StreamDev *streams, int s0, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    StreamDev &S = streams[s0 + s];
    const StreamCtl &c = ctl[s0 + s];
    const Tok best = c.best_final;
    if (!(best.score > LZ) || c.frame == 0) { S.res_n = -1; return; }
    int k = 0;
    for (int p = best.path; p >= 0; p = S.paths[p].prev) {
        if (k < S.res_cap) {
            const PathRec pr = S.paths[p];
            S.res_label[k] = pr.label; S.res_time[k] = pr.frame;
            S.res_score[k] = pr.score; S.res_ac[k] = pr.ac; S.res_lm[k] = pr.lm;
            if (k == 0) { S.res_score[0] = best.score; S.res_ac[0] = best.ac; S.res_lm[0] = best.lm; }   // :293-300
        }
        ++k;
    }
    S.res_n = k;
}

// ... and the same for the batch pipeline (jd_pipe_*): the utterances of a list of streams, exported to VIRTUAL result slots -
// word count, the five result arrays, a copy of the control block (statistics, error, bestFinalToken) - so that the stream
// can take its next utterance before the batch this one belongs to is handed back.  One 64-thread block per utterance.
struct ExportList { int n; int slot[64]; int vslot[64]; };
__global__ void jd_finish_export_kernel(const StreamCtl *ctl, const StreamDev *streams, ExportList L, StreamCtl *vctl, int *vres_n, int *vres,
                                        int res_cap)
{
    const int i = blockIdx.x;
    if (i >= L.n) return;
    const int s = L.slot[i], v = L.vslot[i];
    const StreamDev &S = streams[s];
    const StreamCtl &c = ctl[s];
    {   // the control block, word by word
        const int *src = (const int *)&c;
        int *dst = (int *)&vctl[v];
        for (int k = threadIdx.x; k < (int)(sizeof(StreamCtl) / sizeof(int)); k += blockDim.x) dst[k] = src[k];
    }
    if (threadIdx.x != 0) return;
    const Tok best = c.best_final;
    if (!(best.score > LZ) || c.frame == 0) { vres_n[v] = -1; return; }
    int *lab = vres + (size_t)v * 5 * res_cap, *tim = lab + res_cap;
    float *sc = (float *)(lab + 2 * (size_t)res_cap), *ac = (float *)(lab + 3 * (size_t)res_cap), *lm = (float *)(lab + 4 * (size_t)res_cap);
    int k = 0;
    for (int p = best.path; p >= 0; p = S.paths[p].prev) {
        if (k < res_cap) {
            const PathRec pr = S.paths[p];
            lab[k] = pr.label; tim[k] = pr.frame; sc[k] = pr.score; ac[k] = pr.ac; lm[k] = pr.lm;
            if (k == 0) { sc[0] = best.score; ac[0] = best.ac; lm[0] = best.lm; }     // :293-300
        }
        ++k;
    }
    vres_n[v] = k;
}

__global__ void jd_mark_init_kernel(StreamCtl *ctl, int s0, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) { ctl[s0 + s].needs_init = 1; ctl[s0 + s].started = 1; ctl[s0 + s].error = 0; ctl[s0 + s].T = 0; }
}

__global__ void jd_set_T_kernel(StreamCtl *ctl, int s0, int n, const int *T)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) ctl[s0 + s].T = T[s];
}

// before every k_search launch: the cluster barriers of the streams it advances start at zero
__global__ void jd_zero_bar_kernel(StreamCtl *ctl, const int4 *work, int n, int *status, int scoring_ahead)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { StreamCtl &c = ctl[work[i].x]; c.bar = 0u; c.xbar = 0u; c.xmask = 0u; c.stop_req = 0; }
    if (i == 0) {
        status[0] = 0; status[1] = 0; status[2] = 0; status[3] = 0; status[5] = 0; status[6] = 0;
        if (scoring_ahead) status[4] = 1;      // (cleared on the scoring stream, behind the scoring kernel: pf_launch)
    }
}

