"""Independent cross-check of the oracle's *semantics*: an un-pruned full-trellis Viterbi
in float64, written from the textbook token-passing recurrences (no instance lists, no
normalisation, no pruning, no recombination order).  With every beam disabled the decoder
must find the same best word sequence / boundary frames, and its un-normalised score
(acoustic + LM) must agree to float32 accumulation accuracy.  Small graphs only.
"""
import sys

import numpy as np

NEG = -1e300


def viterbi(net, am, ll, lm_scale=1.0, ins_penalty=0.0):
    """net: SynthNet, am: SynthAM, ll: [T, n_gmm] float log-likelihoods.
    Returns (total_score, [(label, frame), ...]) or None."""
    sys.setrecursionlimit(10000)
    src, dst, il, ol = net.src, net.dst, net.ilab, net.olab
    w = -net.w_file.astype(np.float64) * lm_scale + np.where(ol > 0, ins_penalty, 0.0)
    n_arcs = len(src)
    out_arcs = {}
    for b in range(n_arcs):
        out_arcs.setdefault(int(src[b]), []).append(b)
    fin = {int(s): -float(fw) * lm_scale for s, fw in zip(net.fstate, net.fweight_file)}
    init = int(src[0])
    logA = np.where(am.transp > 0, np.log(np.maximum(am.transp.astype(np.float64), 1e-300)), NEG)
    hmm_of = il - 1
    T = ll.shape[0]

    def tee_of(h):
        n = int(am.hmm_nstates[h]); a = am.transp[am.hmm_tm[h]]
        # HTKModels::addHMM: the 0 -> N-1 transition, if it is not the first successor of state 0
        sucs = [j for j in range(n) if a[0, j] > 0]
        return float(np.log(a[0, n - 1])) if (n - 1) in sucs[1:] else None

    tee = {int(h): tee_of(int(h)) for h in set(hmm_of[hmm_of >= 0].tolist())}
    entry = {}
    best_final = [None]

    def arrive(score, hist, arc, t, last):
        if arc is not None:
            if ol[arc] != 0:
                hist = hist + ((int(ol[arc]), t),)
            q = int(dst[arc])
            if last and q in fin:
                cand = (score + fin[q], hist)
                if best_final[0] is None or cand[0] > best_final[0][0]:
                    best_final[0] = cand
        else:
            q = init
        for b in out_arcs.get(q, []):
            if il[b] == 0:
                arrive(score + w[b], hist, b, t, last)
            else:
                s = score + w[b]
                if b not in entry or s > entry[b][0]:
                    entry[b] = (s, hist)
                tw = tee[int(hmm_of[b])]
                if tw is not None:
                    arrive(s + tw, hist, b, t, last)

    arrive(0.0, (), None, 0, False)
    state = {}                                   # arc -> list over HMM states of (score, hist) or None
    for t in range(T):
        cur_entry, entry = entry, {}
        new_state = {}
        exits = []
        for b in set(cur_entry) | set(state):
            h = int(hmm_of[b]); n = int(am.hmm_nstates[h]); A = logA[am.hmm_tm[h]]
            old = list(state.get(b, [None] * n))
            old[0] = cur_entry.get(b)
            new = [None] * n
            for j in range(1, n - 1):
                best = None
                for i in range(0, n - 1):
                    if old[i] is not None and A[i, j] > NEG / 2:
                        c = old[i][0] + A[i, j]
                        if best is None or c > best[0]:
                            best = (c, old[i][1])
                if best is not None:
                    new[j] = (best[0] + float(ll[t, am.hmm_gmm[h, j]]), best[1])
            ex = None
            for i in range(1, n - 1):
                if new[i] is not None and A[i, n - 1] > NEG / 2:
                    c = new[i][0] + A[i, n - 1]
                    if ex is None or c > ex[0]:
                        ex = (c, new[i][1])
            if any(v is not None for v in new):
                new_state[b] = new
            if ex is not None:
                exits.append((ex, b))
        state = new_state
        best_final[0] = None
        for (sc, hist), b in exits:
            arrive(sc, hist, b, t, t == T - 1)
    return best_final[0]
