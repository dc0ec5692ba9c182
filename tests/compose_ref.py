"""Offline composition of C.L and G in plain Python - TEST INFRASTRUCTURE for jd_net_compose
(csrc/jd_compose.hip).  There is no oracle for the dynamic-composition row: the reference's
WFSTOnTheFlyDecoder is not built by either of its build systems and no longer compiles (SURVEY.md
section 2 row 15), so the device composition is checked against

  compose_filtered  the same definition (filter flag, back-off after a word only, interval
                    look-ahead that always lets the tail of the last word through to a final C.L
                    state, canonical numbering) written a second time with dictionaries - the
                    arrays must come out identical, bit for bit;
  compose_naive     textbook epsilon composition with nothing clever: every (C.L state, G state)
                    pair, G's epsilon taken anywhere, no filter, no look-ahead.  A different (larger,
                    redundant) graph with the same set of weighted paths: decoding it with the CPU
                    oracle must give the words, times and (up to float association) scores the GPU
                    static path gives on the device-composed graph.

Inputs are csr() dictionaries of capi.Network objects, i.e. the weights as the loaders scaled them.
"""
import numpy as np

INF = np.float32(np.inf)


def _sorted_g(g):
    """per state: arcs sorted by input label (WFSTSortedInLabelNetwork)"""
    rows = []
    for s in range(len(g["row_ptr"]) - 1):
        a0, a1 = int(g["row_ptr"][s]), int(g["row_ptr"][s + 1])
        idx = sorted(range(a0, a1), key=lambda a: int(g["ilab"][a]))
        rows.append([(int(g["ilab"][a]), int(g["olab"][a]), int(g["to"][a]), np.float32(g["w"][a])) for a in idx])
    return rows


def _lookahead(cl):
    S = len(cl["row_ptr"]) - 1
    lo = [2 ** 31 - 1] * S
    hi = [0] * S
    changed = True
    on_cycle = set()
    # label-less cycles: states reachable from themselves through arcs without an output label
    eps_next = [[int(cl["to"][a]) for a in range(int(cl["row_ptr"][c]), int(cl["row_ptr"][c + 1])) if int(cl["olab"][a]) == 0]
                for c in range(S)]
    for c in range(S):
        seen, stack = set(), list(eps_next[c])
        while stack:
            x = stack.pop()
            if x == c:
                on_cycle.add(c)
                break
            if x not in seen:
                seen.add(x)
                stack.extend(eps_next[x])
    for c in on_cycle:
        lo[c], hi[c] = 1, 2 ** 31 - 1
    while changed:
        changed = False
        for c in range(S):
            for a in range(int(cl["row_ptr"][c]), int(cl["row_ptr"][c + 1])):
                o, t = int(cl["olab"][a]), int(cl["to"][a])
                l, h = (o, o) if o else (lo[t], hi[t])
                if l <= h and (l < lo[c] or h > hi[c]):
                    lo[c], hi[c] = min(lo[c], l), max(hi[c], h)
                    changed = True
    # states that reach a FINAL C.L state through label-less arcs (the tail of the last word): LA_MAYFIN, csrc/jd_lazy.h
    mf = [bool(np.isfinite(cl["fin_w"][c])) for c in range(S)]
    changed = True
    while changed:
        changed = False
        for c in range(S):
            if not mf[c] and any(mf[t] for t in eps_next[c]):
                mf[c] = True
                changed = True
    return lo, hi, mf


def _finish(states, arcs_of, fin_of, init_key, order_key):
    """canonical numbering + CSR arrays"""
    keys = sorted(states, key=order_key)
    idx = {k: i for i, k in enumerate(keys)}
    row_ptr, to, w, il, ol = [0], [], [], [], []
    for k in keys:
        for (dk, ww, i, o) in arcs_of[k]:
            to.append(idx[dk]); w.append(ww); il.append(i); ol.append(o)
        row_ptr.append(len(to))
    fin = np.asarray([fin_of[k] for k in keys], np.float32)
    return dict(n_states=len(keys), init=idx[init_key], row_ptr=np.asarray(row_ptr, np.int32), to=np.asarray(to, np.int32),
                w=np.asarray(w, np.float32), ilab=np.asarray(il, np.int32), olab=np.asarray(ol, np.int32), fin_w=fin)


def compose_filtered(cl, cl_init, g, g_init, pushing=False):
    G = _sorted_g(g)
    lo, hi, mf = _lookahead(cl)
    glabels = [[a[0] for a in row] for row in G]

    def any_in(gs, l, h):
        return l <= h and any(l <= x <= h for x in glabels[gs])

    def potential(gs, l, h):
        """best weight among the arcs of gs with a label in [l, h] (0 if none): the weight look-ahead"""
        ws = [a[3] for a in G[gs] if l <= a[0] <= h]
        return max(ws) if (l <= h and ws) else np.float32(0.0)

    start = (cl_init, g_init, 1)
    states, arcs_of, fin_of = {start}, {}, {}
    queue = [start]
    while queue:
        c, gs, f = k = queue.pop()
        out = []
        p_src = np.float32(0.0)
        if pushing and not f:
            p_src = np.float32(potential(gs, lo[c], hi[c]))
        if f and G[gs] and G[gs][0][0] == 0:
            _, o, t, ww = G[gs][0]
            out.append(((c, t, 1), ww, 0, o))
        for a in range(int(cl["row_ptr"][c]), int(cl["row_ptr"][c + 1])):
            x, t, ww, i = int(cl["olab"][a]), int(cl["to"][a]), np.float32(cl["w"][a]), int(cl["ilab"][a])
            if x == 0:
                if any_in(gs, lo[t], hi[t]) or (mf[t] and np.isfinite(g["fin_w"][gs])):
                    if pushing:
                        ww = np.float32(np.float32(ww + np.float32(potential(gs, lo[t], hi[t]))) - p_src)
                    out.append(((t, gs, 0), ww, i, 0))
            else:
                for (l, o, t2, wg) in G[gs]:
                    if l == x:
                        w2 = np.float32(ww + wg)
                        if pushing:
                            w2 = np.float32(w2 - p_src)
                        out.append(((t, t2, 1), w2, i, o))
        arcs_of[k] = out
        fc, fg = np.float32(cl["fin_w"][c]), np.float32(g["fin_w"][gs])
        if np.isfinite(fc) and np.isfinite(fg):
            fin_of[k] = np.float32(np.float32(fc + fg) - p_src) if pushing else np.float32(fc + fg)
        else:
            fin_of[k] = INF
        for (dk, _, _, _) in out:
            if dk not in states:
                states.add(dk)
                queue.append(dk)
    return _finish(states, arcs_of, fin_of, start, lambda k: (k[0], k[2], k[1]))


def compose_naive(cl, cl_init, g, g_init):
    G = _sorted_g(g)
    start = (cl_init, g_init)
    states, arcs_of, fin_of = {start}, {}, {}
    queue = [start]
    while queue:
        c, gs = k = queue.pop()
        out = []
        for (l, o, t, ww) in G[gs]:
            if l == 0:
                out.append(((c, t), ww, 0, o))
        for a in range(int(cl["row_ptr"][c]), int(cl["row_ptr"][c + 1])):
            x, t, ww, i = int(cl["olab"][a]), int(cl["to"][a]), np.float32(cl["w"][a]), int(cl["ilab"][a])
            if x == 0:
                out.append(((t, gs), ww, i, 0))
            else:
                for (l, o, t2, wg) in G[gs]:
                    if l == x:
                        out.append(((t, t2), np.float32(ww + wg), i, o))
        arcs_of[k] = out
        fc, fg = np.float32(cl["fin_w"][c]), np.float32(g["fin_w"][gs])
        fin_of[k] = np.float32(fc + fg) if np.isfinite(fc) and np.isfinite(fg) else INF
        for (dk, _, _, _) in out:
            if dk not in states:
                states.add(dk)
                queue.append(dk)
    return _finish(states, arcs_of, fin_of, start, lambda k: k)


def push_labels(cl, cl_init):
    """Label pushing on C.L (csrc/jd_compose.hip, cl_push_labels; the reference: WFSTLabelPushingNetwork's label sets,
    WFSTNetwork.cpp:1643-1764), written a second time: every output label moves towards the initial state, up to the
    first arc behind which it is the only label that can follow.  Returns (new olab array, labels moved)."""
    lo, hi, mf = _lookahead(cl)
    S = len(cl["row_ptr"]) - 1
    single = [lo[c] == hi[c] and not mf[c] for c in range(S)]
    arcs = [(c, a, int(cl["to"][a]), int(cl["olab"][a])) for c in range(S) for a in range(int(cl["row_ptr"][c]), int(cl["row_ptr"][c + 1]))]
    lab_in, less_in = [False] * S, [False] * S
    lab_in[cl_init] = True                              # nothing has been emitted on the way to the initial state
    nbr = {c: [] for c in range(S)}
    for c, _, t, o in arcs:
        if o:
            lab_in[t] = True
        else:
            less_in[t] = True
            if single[c] and single[t]:
                nbr[c].append(t); nbr[t].append(c)
    region = [-1] * S
    for c in range(S):                                  # runs of single states joined by label-less arcs
        if single[c] and region[c] < 0:
            region[c] = c
            todo = [c]
            while todo:
                x = todo.pop()
                for y in nbr[x]:
                    if region[y] < 0:
                        region[y] = c
                        todo.append(y)
    bad = {region[c] for c in range(S) if single[c] and lab_in[c] and less_in[c]}
    emitted = [single[c] and not lab_in[c] and region[c] not in bad for c in range(S)]
    olab = np.array(cl["olab"], np.int32, copy=True)
    moved = 0
    for c, a, t, o in arcs:
        if o == 0 and emitted[t] and not emitted[c]:
            olab[a] = lo[t]
            moved += 1
        elif o != 0 and emitted[c]:
            olab[a] = 0
    return olab, moved
