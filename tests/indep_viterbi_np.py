"""A second, independent statement of what the decoder must find - test infrastructure, not product code.

Un-pruned best path through (graph x HMM x frames) in float64, written from the token-passing
recurrences with numpy arrays over ALL arcs of the graph at once: no active lists, no instances, no
normalisation, no pruning, no recursion - epsilon and tee arcs are relaxed to a fixed point per frame
(Bellman-Ford over the label-less part of the graph).  The acoustic scores come from its own float64
diagonal-GMM evaluation (gmm_loglik), so nothing of oracle/ and nothing of juicer_amd's arithmetic is
shared: a decoder with every beam disabled must return the same words at the same frames, and a total
(acoustic + language model) score equal to float32 accumulation accuracy.

Semantics it restates (reference file:line, behaviour only):
  * arc weight = -w_file * lmScale (+ insertion penalty on arcs with an output label), final weight
    = -w * lmScale (WFSTNetwork.cpp:441, :481-486); HMM of an arc = inLabel - 1 (WFSTDecoderLite.cpp:754)
  * a tee model is one whose 0 -> N-1 transition is not the FIRST successor of state 0 (HTKModels.cpp:581-593)
  * emitting states take the best predecessor of the PREVIOUS frame's tokens + log a_ij + b_j(x_t); the exit
    state the best of THIS frame's emitting tokens + log a_iN (WFSTDecoderLite.cpp:376-484)
  * a word is recorded when a token leaves an arc that carries an output label, with that frame as its time
    (:497-509); only tokens that enter a final state in the LAST frame can win (:513-520, :316)
It scales to graphs of 10^4 - 10^5 arcs (tests/indep_viterbi.py, the per-arc Python loops, is the small-case twin).
"""
import numpy as np

NEG = -1e300
LOG_2PI = 1.8378770664093453


def gmm_loglik(am, x):
    """log sum_m w_m N(x; mu_m, diag var_m) in float64: [T, n_gmm]."""
    x = np.asarray(x, dtype=np.float64)
    mean, var, wt = am.mean.astype(np.float64), am.var.astype(np.float64), am.weight.astype(np.float64)
    out = np.full((x.shape[0], am.n_gmm), NEG)
    const = -0.5 * (am.D * LOG_2PI + np.log(var).sum(axis=2)) + np.log(np.maximum(wt, 1e-300))   # [G, M]
    valid = np.arange(am.max_mix)[None, :] < am.n_mix[:, None]
    for g0 in range(0, am.n_gmm, 64):                                    # (blocks keep the temporaries small)
        g1 = min(am.n_gmm, g0 + 64)
        d = x[:, None, None, :] - mean[None, g0:g1]
        e = const[None, g0:g1] - 0.5 * (d * d / var[None, g0:g1]).sum(axis=3)                     # [T, g, M]
        e = np.where(valid[None, g0:g1], e, NEG)
        m = e.max(axis=2, keepdims=True)
        out[:, g0:g1] = (m + np.log(np.exp(e - m).sum(axis=2, keepdims=True)))[:, :, 0]
    return out


class _Hist:
    """word records {label, frame, previous record}, appended in blocks"""
    def __init__(self):
        self.label, self.time, self.prev, self.n = [], [], [], 0

    def add(self, label, time, prev):
        k = len(label)
        ids = np.arange(self.n, self.n + k, dtype=np.int64)
        self.label.append(np.asarray(label, np.int64)); self.time.append(np.full(k, time, np.int64)); self.prev.append(np.asarray(prev, np.int64))
        self.n += k
        return ids

    def chain(self, h):
        if self.n == 0:
            return []
        L, Tm, P = np.concatenate(self.label), np.concatenate(self.time), np.concatenate(self.prev)
        out = []
        while h >= 0:
            out.append((int(L[h]), int(Tm[h])))
            h = int(P[h])
        return out[::-1]


def _best_per_key(key, score):
    """indices of the best-scoring entry of every distinct key"""
    if key.shape[0] == 0:
        return np.zeros(0, np.int64)
    order = np.lexsort((-score, key))
    k = key[order]
    first = np.ones(k.shape[0], bool); first[1:] = k[1:] != k[:-1]
    return order[first]


def viterbi(net, am, ll, lm_scale=1.0, ins_penalty=0.0):
    """net: SynthNet, am: SynthAM, ll: [T, n_gmm] float64 log-likelihoods.
    Returns (total score, [(label, frame), ...] oldest first) or None when no token ends in a final state."""
    src, dst, il, ol = (np.asarray(a, np.int64) for a in (net.src, net.dst, net.ilab, net.olab))
    w = -net.w_file.astype(np.float64) * lm_scale + np.where(ol > 0, ins_penalty, 0.0)
    nS = int(max(net.n_states, src.max() + 1, dst.max() + 1))
    fin = np.full(nS, NEG)
    fin[np.asarray(net.fstate, np.int64)] = -np.asarray(net.fweight_file, np.float64) * lm_scale
    init = int(src[0])
    MN = am.max_n
    with np.errstate(divide="ignore"):
        logA = np.where(am.transp > 0, np.log(np.maximum(am.transp.astype(np.float64), 1e-300)), NEG)   # [n_tm, MN, MN]
    T = ll.shape[0]
    # per model arc
    marc = np.nonzero(il > 0)[0]
    earc = np.nonzero(il == 0)[0]
    hm = il[marc] - 1
    n_st = am.hmm_nstates[hm].astype(np.int64)
    A = logA[am.hmm_tm[hm]]                                               # [nM, MN, MN]
    jj = np.arange(MN)[None, :]
    emitting = (jj >= 1) & (jj <= n_st[:, None] - 2)                      # [nM, MN]
    gm = np.where(emitting, am.hmm_gmm[hm], 0).astype(np.int64)
    A_exit = np.take_along_axis(A, (n_st - 1)[:, None, None].repeat(MN, axis=1), axis=2)[:, :, 0]      # log a_{i, N-1}
    A_in = np.where(emitting[:, None, :] & (jj[:, :, None] <= n_st[:, None, None] - 2), A, NEG)          # preds 0 .. N-2 -> emitting j
    tee = np.full(am.n_hmm, NEG)
    for h in range(am.n_hmm):
        n = int(am.hmm_nstates[h]); a = am.transp[am.hmm_tm[h]]
        sucs = [j for j in range(n) if a[0, j] > 0]
        if (n - 1) in sucs[1:]:
            tee[h] = np.log(float(a[0, n - 1]))
    tee_arc = marc[tee[hm] > NEG / 2]                                     # arcs whose model can be skipped
    tee_w = tee[il[tee_arc] - 1]
    # label-less moves: epsilon arcs, and model arcs through their tee transition
    c_arc = np.concatenate([earc, tee_arc]); c_w = np.concatenate([w[earc], w[tee_arc] + tee_w])
    H = _Hist()

    def expand(a_state, a_score, a_hist, t):
        """arrivals (state, score, history) -> best score / history per state after the label-less closure"""
        sb = np.full(nS, NEG); sh = np.full(nS, -1, np.int64)
        k = _best_per_key(a_state, a_score)
        sb[a_state[k]] = a_score[k]; sh[a_state[k]] = a_hist[k]
        changed = np.zeros(nS, bool); changed[a_state[k]] = True
        for _ in range(nS + 1):
            use = changed[src[c_arc]]
            if not use.any():
                break
            ca, cw = c_arc[use], c_w[use]
            cand = sb[src[ca]] + cw
            k = _best_per_key(dst[ca], cand)
            k = k[cand[k] > sb[dst[ca[k]]]]
            changed[:] = False
            if k.shape[0] == 0:
                break
            win = ca[k]
            hist = sh[src[win]].copy()
            lab = ol[win] != 0
            if lab.any():
                hist[lab] = H.add(ol[win[lab]], t, hist[lab])
            sb[dst[win]] = cand[k]; sh[dst[win]] = hist
            changed[dst[win]] = True
        return sb, sh

    sb, sh = expand(np.array([init]), np.array([0.0]), np.array([-1], np.int64), 0)
    S = np.full((marc.shape[0], MN), NEG); Sh = np.full((marc.shape[0], MN), -1, np.int64)
    best = None
    for t in range(T):
        ok = sb[src[marc]] > NEG / 2
        S[:, 0] = np.where(ok, sb[src[marc]] + w[marc], NEG); Sh[:, 0] = np.where(ok, sh[src[marc]], -1)
        cand = S[:, :, None] + A_in                                       # [nM, i, j]
        bi = cand.argmax(axis=1)                                          # (the lowest predecessor wins ties)
        bs = np.take_along_axis(cand, bi[:, None, :], axis=1)[:, 0, :]
        alive = emitting & (bs > NEG / 2)
        new = np.where(alive, bs + ll[t][gm], NEG)
        newh = np.where(alive, np.take_along_axis(Sh, bi, axis=1), -1)
        exc = np.where(emitting, new + A_exit, NEG)
        ei = exc.argmax(axis=1)
        ex = exc[np.arange(exc.shape[0]), ei]; exh = newh[np.arange(exc.shape[0]), ei]
        S, Sh = new, newh
        out = np.nonzero(ex > NEG / 2)[0]
        a_arc = marc[out]
        hist = exh[out].copy()
        lab = ol[a_arc] != 0
        if lab.any():
            hist[lab] = H.add(ol[a_arc[lab]], t, hist[lab])
        sb, sh = expand(dst[a_arc], ex[out], hist, t)
        if t == T - 1:
            f = np.where((sb > NEG / 2) & (fin > NEG / 2), sb + fin, NEG)
            q = int(f.argmax())
            if f[q] > NEG / 2:
                best = (float(f[q]), H.chain(int(sh[q])))
    return best
