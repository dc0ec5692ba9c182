"""Deterministic synthetic decoding problems (the build's own generators).

The reference ships no data, so BASELINE.json's configs are synthesised:
a C.L.G-shaped transducer in AT&T arc-list form, an HTK-style HMM/GMM set,
and 39-dim feature streams sampled from the models along a random accepted
word sequence (unmatched random features leave no final-state token alive at
realistic beams - SURVEY.md Appendix C).

Rules forced by the reference's behaviour (cited so the generators stay legal):
  * arcs of one source state contiguous, first arc's source = initial state
    (WFSTNetwork.cpp:455-456, 709-721);
  * in-label i>0 selects HMM i-1 (WFSTDecoderLite.cpp:754), 0 = epsilon;
  * file weights are -log probabilities, negated at load (WFSTNetwork.cpp:481);
  * per-frame log-likelihoods must stay below the histogram's +200 ceiling
    relative to the best score (Histogram.cpp:78-79).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class SynthAM:
    """HTK-level acoustic model parameters (input of jd_am_create_htk)."""
    D: int
    n_gmm: int
    max_mix: int
    n_mix: np.ndarray        # int32 [n_gmm]
    weight: np.ndarray       # float32 [n_gmm, max_mix]
    mean: np.ndarray         # float32 [n_gmm, max_mix, D]
    var: np.ndarray          # float32 [n_gmm, max_mix, D]
    n_hmm: int
    max_n: int
    hmm_nstates: np.ndarray  # int32 [n_hmm]
    hmm_gmm: np.ndarray      # int32 [n_hmm, max_n]  (-1 for entry/exit)
    hmm_tm: np.ndarray       # int32 [n_hmm]
    n_tm: int
    tm_nstates: np.ndarray   # int32 [n_tm]
    transp: np.ndarray       # float32 [n_tm, max_n, max_n]
    sp_hmm: int = -1         # index of the tee "sp" model, -1 if none


@dataclass
class SynthNet:
    """Arc list in FSM file order (input of jd_net_create_arcs)."""
    n_states: int
    src: np.ndarray          # int32 [n_arcs]
    dst: np.ndarray          # int32
    ilab: np.ndarray         # int32
    olab: np.ndarray         # int32
    w_file: np.ndarray       # float32 (-log prob, as written in an FSM file)
    fstate: np.ndarray       # int32 [n_final]
    fweight_file: np.ndarray  # float32
    # generator bookkeeping used by sample_utterance()
    n_words: int = 0
    prons: List[np.ndarray] = field(default_factory=list)   # per word: 0-based hmm ids
    succ: Optional[np.ndarray] = None                       # int32 [n_words+1, K] bigram successors
    sp_hmm: int = -1
    # trigram-shaped graphs (make_wfst_trigram): per history state the successor words and the
    # history state each one leads to, CSR by history state; back-off target of each history state
    walk_off: Optional[np.ndarray] = None                   # int64 [n_hist+1]
    walk_word: Optional[np.ndarray] = None                  # int32
    walk_dst: Optional[np.ndarray] = None                   # int32
    walk_backoff: Optional[np.ndarray] = None               # int32 [n_hist] (-1: the unigram hub)

    @property
    def n_arcs(self) -> int:
        return int(self.src.shape[0])


def make_models(seed: int, n_gmm: int, n_hmm: int, n_mix: int, D: int = 39,
                n_tm: int = 16, sep: float = 1.0, with_tee: bool = False,
                with_skip: bool = True) -> SynthAM:
    """Left-to-right 5-state (3 emitting) HMMs over a pool of tied states."""
    rng = np.random.default_rng(seed)
    max_n = 5
    centre = rng.normal(0.0, sep, size=(n_gmm, 1, D))
    mean = (centre + rng.normal(0.0, 0.5 * sep, size=(n_gmm, n_mix, D))).astype(np.float32)
    var = rng.uniform(0.3, 3.0, size=(n_gmm, n_mix, D)).astype(np.float32)
    wraw = rng.gamma(2.0, 1.0, size=(n_gmm, n_mix))
    wraw = wraw / wraw.sum(axis=1, keepdims=True)
    wraw = np.maximum(wraw, 1e-3)
    weight = (wraw / wraw.sum(axis=1, keepdims=True)).astype(np.float32)
    if n_mix == 1:
        weight[:] = 1.0

    n_tm_tot = n_tm + (1 if with_tee else 0)
    transp = np.zeros((n_tm_tot, max_n, max_n), dtype=np.float32)
    tm_nstates = np.full(n_tm_tot, 5, dtype=np.int32)
    for t in range(n_tm):
        p = rng.uniform(0.55, 0.85, size=3)
        transp[t, 0, 1] = 1.0
        for k, j in enumerate((1, 2, 3)):
            transp[t, j, j] = p[k]
            transp[t, j, j + 1] = 1.0 - p[k]
        if with_skip and t % 4 == 3:
            # a skip 1->3 so that some states have three predecessors
            s = 0.25 * (1.0 - p[0])
            transp[t, 1, 2] = np.float32(1.0 - p[0] - s)
            transp[t, 1, 3] = np.float32(s)
    hmm_nstates = np.full(n_hmm, 5, dtype=np.int32)
    hmm_gmm = np.full((n_hmm, max_n), -1, dtype=np.int32)
    # every tied state is used at least once, the rest at random
    pool = np.concatenate([rng.permutation(n_gmm),
                           rng.integers(0, n_gmm, size=max(0, 3 * n_hmm - n_gmm))])[:3 * n_hmm]
    if pool.shape[0] < 3 * n_hmm:
        pool = np.concatenate([pool, rng.integers(0, n_gmm, size=3 * n_hmm - pool.shape[0])])
    pool = rng.permutation(pool)
    hmm_gmm[:, 1:4] = pool.reshape(n_hmm, 3)
    hmm_tm = rng.integers(0, n_tm, size=n_hmm).astype(np.int32)
    sp = -1
    if with_tee:
        # a 3-state tee model "sp": entry -> emitting (0.6) or straight to exit (0.4)
        t = n_tm
        tm_nstates[t] = 3
        transp[t, 0, 1] = 0.6
        transp[t, 0, 2] = 0.4
        transp[t, 1, 1] = 0.7
        transp[t, 1, 2] = 0.3
        sp = n_hmm - 1
        hmm_nstates[sp] = 3
        hmm_gmm[sp, :] = -1
        hmm_gmm[sp, 1] = int(rng.integers(0, n_gmm))
        hmm_tm[sp] = t
    return SynthAM(D=D, n_gmm=n_gmm, max_mix=n_mix,
                   n_mix=np.full(n_gmm, n_mix, dtype=np.int32),
                   weight=weight, mean=mean, var=var, n_hmm=n_hmm, max_n=max_n,
                   hmm_nstates=hmm_nstates, hmm_gmm=hmm_gmm, hmm_tm=hmm_tm,
                   n_tm=n_tm_tot, tm_nstates=tm_nstates, transp=transp, sp_hmm=sp)


def make_models_mixed(seed: int, n_gmm: int, n_hmm: int, n_mix: int, D: int = 39,
                      sizes=(3, 4, 5, 6, 7, 8), with_tee: bool = True, sep: float = 1.0) -> SynthAM:
    """HMMs of mixed topology: `sizes` = total state counts (entry + emitting + exit), i.e. 1 to 6
    emitting states, left-to-right with occasional skips; one transition matrix per HMM size and
    variant.  Exercises the 8-lane instance layout (more than 3 emitting states) and HMMs of
    different lengths side by side.  With with_tee the last HMM is a 3-state tee model."""
    rng = np.random.default_rng(seed)
    max_n = max(sizes)
    centre = rng.normal(0.0, sep, size=(n_gmm, 1, D))
    mean = (centre + rng.normal(0.0, 0.5 * sep, size=(n_gmm, n_mix, D))).astype(np.float32)
    var = rng.uniform(0.3, 3.0, size=(n_gmm, n_mix, D)).astype(np.float32)
    wraw = rng.gamma(2.0, 1.0, size=(n_gmm, n_mix))
    wraw = np.maximum(wraw / wraw.sum(axis=1, keepdims=True), 1e-3)
    weight = (wraw / wraw.sum(axis=1, keepdims=True)).astype(np.float32)
    if n_mix == 1:
        weight[:] = 1.0
    variants = 3
    tms, tm_n = [], []
    for n in sizes:
        for v in range(variants):
            a = np.zeros((max_n, max_n), np.float32)
            a[0, 1] = 1.0
            for j in range(1, n - 1):
                stay = rng.uniform(0.5, 0.8)
                a[j, j] = stay
                if v == 2 and j + 2 <= n - 1 and n > 3:            # a skip: two successors besides the loop
                    sk = np.float32(0.3 * (1.0 - stay))
                    a[j, j + 2] = sk
                    a[j, j + 1] = np.float32(1.0 - stay - sk)
                else:
                    a[j, j + 1] = np.float32(1.0 - stay)
            if v == 1 and n > 4:                                   # entry may also start in the second emitting state
                a[0, 1], a[0, 2] = 0.7, 0.3
            tms.append(a); tm_n.append(n)
    tee_tm = -1
    if with_tee:
        a = np.zeros((max_n, max_n), np.float32)
        a[0, 1], a[0, 2], a[1, 1], a[1, 2] = 0.6, 0.4, 0.7, 0.3
        tms.append(a); tm_n.append(3); tee_tm = len(tms) - 1
    hmm_nstates = np.zeros(n_hmm, np.int32)
    hmm_tm = np.zeros(n_hmm, np.int32)
    hmm_gmm = np.full((n_hmm, max_n), -1, np.int32)
    # tied states are handed out without replacement while they last: two adjacent phones sharing
    # a tied state AND a transition matrix tie exactly (stay-then-move == move-then-stay)
    pool = list(rng.permutation(n_gmm))
    for h in range(n_hmm):
        k = int(rng.integers(0, len(sizes) * variants))
        hmm_tm[h] = k; hmm_nstates[h] = tm_n[k]
        for j in range(1, tm_n[k] - 1):
            hmm_gmm[h, j] = pool.pop() if pool else int(rng.integers(0, n_gmm))
    sp = -1
    if with_tee:
        sp = n_hmm - 1
        hmm_tm[sp] = tee_tm; hmm_nstates[sp] = 3
        hmm_gmm[sp, :] = -1
        hmm_gmm[sp, 1] = int(rng.integers(0, n_gmm))
    return SynthAM(D=D, n_gmm=n_gmm, max_mix=n_mix, n_mix=np.full(n_gmm, n_mix, dtype=np.int32), weight=weight,
                   mean=mean, var=var, n_hmm=n_hmm, max_n=max_n, hmm_nstates=hmm_nstates, hmm_gmm=hmm_gmm,
                   hmm_tm=hmm_tm, n_tm=len(tms), tm_nstates=np.asarray(tm_n, np.int32),
                   transp=np.stack(tms).astype(np.float32), sp_hmm=sp)


def make_wfst(seed: int, am: SynthAM, n_words: int, n_succ: int,
              pron_len=(2, 5), with_sp: bool = False, hub: str = "flat",
              n_phones: int = 40, eps_word_frac: float = 0.02) -> SynthNet:
    """Bigram-shaped C.L.G: state 0 = <s> history (initial), states 1..V = word
    histories (all final), state V+1 = unigram hub reached by back-off epsilon
    arcs.  Every (history, successor word) pair owns an un-shared chain of phone
    arcs.  hub="flat": every hub word starts with an eps:word arc (word label on
    an epsilon input; a stress case).  hub="tree": the hub is a lexicon prefix
    tree with tropical weight pushing, as det(L.G) gives a back-off state; word
    labels sit on the per-word last phone arc (a fraction eps_word_frac of the
    words get an eps:word arc instead).
    """
    rng = np.random.default_rng(seed)
    V, K = n_words, min(n_succ, n_words)
    n_real_hmm = am.n_hmm - (1 if am.sp_hmm >= 0 else 0)
    plen = rng.integers(pron_len[0], pron_len[1] + 1, size=V)
    if hub == "tree":
        # triphone-like naming: word-initial model = base phone, later ones hashed from (prev, cur)
        P = min(n_phones, max(2, n_real_hmm // 2))
        prons = []
        for l in plen:
            ph = rng.integers(0, P, size=int(l))
            hm = [int(ph[0])]
            for k in range(1, int(l)):
                hm.append(P + (int(ph[k - 1]) * 7919 + int(ph[k]) * 104729 + k * 31) % (n_real_hmm - P))
            prons.append(np.asarray(hm, dtype=np.int32))
    else:
        prons = [rng.integers(0, n_real_hmm, size=int(l)).astype(np.int32) for l in plen]
    hub_kind = hub
    hub = V + 1
    nxt = V + 2                       # next free chain state id
    src, dst, il, ol, wf = [], [], [], [], []
    use_sp = with_sp and am.sp_hmm >= 0

    succ = np.zeros((V + 1, K), dtype=np.int32)
    for h in range(V + 1):
        succ[h] = rng.choice(V, size=K, replace=False)

    def add_chain(s0: int, w: int, cost: float, label_first: bool, eps_word: bool):
        nonlocal nxt
        pr = prons[w]
        cur = s0
        if eps_word:
            src.append(cur); dst.append(nxt); il.append(0); ol.append(w + 1); wf.append(cost)
            cur = nxt; nxt += 1
            cost = 0.0
        L = len(pr)
        for k in range(L):
            last = (k == L - 1)
            lab = 0
            if not eps_word and ((label_first and k == 0) or (not label_first and last)):
                lab = w + 1
            if last and not use_sp:
                to = 1 + w
            else:
                to = nxt; nxt += 1
            src.append(cur); dst.append(to); il.append(int(pr[k]) + 1); ol.append(lab)
            wf.append(cost if k == 0 else 0.0)
            cur = to
        if use_sp:
            src.append(cur); dst.append(1 + w); il.append(am.sp_hmm + 1); ol.append(0); wf.append(0.0)

    lm_cost = rng.uniform(0.5, 8.0, size=(V + 1, K))
    label_first = rng.random(size=(V + 1, K)) < 0.5
    for h in range(V + 1):
        for k in range(K):
            add_chain(h, int(succ[h, k]), float(lm_cost[h, k]), bool(label_first[h, k]), False)
        src.append(h); dst.append(hub); il.append(0); ol.append(0); wf.append(float(rng.uniform(1.0, 4.0)))
    uni = rng.uniform(4.0, 12.0, size=V)
    if hub_kind == "flat":
        for w in range(V):
            add_chain(hub, w, float(uni[w]), True, True)
    else:
        # prefix tree over the first L-1 models of each word; the last model arc is per word
        children = {}                     # (node, hmm) -> child node
        node_words = {hub: []}            # node -> words ending right below it
        node_min = {}
        def child(nd, hm):
            nonlocal nxt
            key = (nd, hm)
            if key not in children:
                children[key] = nxt; node_words[nxt] = []; nxt += 1
            return children[key]
        ends = []
        for w in range(V):
            nd = hub
            for hm in prons[w][:-1]:
                nd = child(nd, int(hm))
            node_words[nd].append(w)
        kids = {}
        for (nd, hm), ch in children.items():
            kids.setdefault(nd, []).append((hm, ch))
        def min_cost(nd):                 # tropical "distance to the cheapest word" below nd
            if nd in node_min:
                return node_min[nd]
            m = min([float(uni[w]) for w in node_words[nd]] + [min_cost(ch) for _, ch in kids.get(nd, [])])
            node_min[nd] = m
            return m
        import sys as _sys
        _sys.setrecursionlimit(max(10000, _sys.getrecursionlimit()))
        eps_w = rng.random(size=V) < eps_word_frac
        stack = [hub]
        while stack:
            nd = stack.pop()
            base = min_cost(nd)
            for hm, ch in sorted(kids.get(nd, [])):
                src.append(nd); dst.append(ch); il.append(hm + 1); ol.append(0); wf.append(min_cost(ch) - base)
                stack.append(ch)
            for w in node_words[nd]:
                cur, cost = nd, float(uni[w]) - base
                if eps_w[w]:
                    src.append(cur); dst.append(nxt); il.append(0); ol.append(w + 1); wf.append(cost)
                    cur = nxt; nxt += 1; cost = 0.0
                lab = 0 if eps_w[w] else w + 1
                if use_sp:
                    src.append(cur); dst.append(nxt); il.append(int(prons[w][-1]) + 1); ol.append(lab); wf.append(cost)
                    src.append(nxt); dst.append(1 + w); il.append(am.sp_hmm + 1); ol.append(0); wf.append(0.0)
                    nxt += 1
                else:
                    src.append(cur); dst.append(1 + w); il.append(int(prons[w][-1]) + 1); ol.append(lab); wf.append(cost)

    src = np.asarray(src, dtype=np.int32); dst = np.asarray(dst, dtype=np.int32)
    il = np.asarray(il, dtype=np.int32); ol = np.asarray(ol, dtype=np.int32)
    wf = np.asarray(wf, dtype=np.float32)
    order = np.argsort(src, kind="stable")          # group by source; state 0 (initial) first
    fstate = np.arange(1, V + 1, dtype=np.int32)
    fweight = rng.uniform(0.0, 2.0, size=V).astype(np.float32)
    return SynthNet(n_states=int(nxt), src=src[order], dst=dst[order], ilab=il[order],
                    olab=ol[order], w_file=wf[order], fstate=fstate, fweight_file=fweight,
                    n_words=V, prons=prons, succ=succ, sp_hmm=am.sp_hmm if use_sp else -1)


def make_cl_g(seed: int, am: SynthAM, n_words: int, n_succ: int, n_tri: int = 0, n_succ3: int = 3,
              pron_len=(2, 5), with_sp: bool = False, n_phones: int = 40, terminal: bool = False):
    """SEPARATE C.L and G (BASELINE.json configs[4], the inputs of jd_net_compose), returned as two
    SynthNet objects (cl, g) in FSM file form.

    C.L: lexicon prefix-tree transducer.  State 0 = root (initial and final).  The first L-1 models of
    a pronunciation are shared tree arcs, the last model arc is per word, carries the word label and
    leads back to the root (through the tee "sp" model if the model set has one).  Words are numbered
    in the depth-first order of the tree (= pronunciations in lexicographic order), which makes the
    label sets below a tree node contiguous intervals.  Pronunciation costs on the word arcs.
    G: back-off n-gram acceptor.  State 0 = <s> (initial), 1..V = one-word histories (final), V+1 =
    unigram state, then n_tri two-word histories (u,v) (final).  A history has arcs for its successor
    words and an epsilon arc (the back-off) to the next shorter history; the unigram state has an arc
    for every word.  Arcs are written in random order (a loader has to sort them).

    terminal=True: sentences END.  C.L gets a word `</s>` (label V+1, index V in prons) whose last model
    is a label-less tail arc into a terminal final state (no arcs out; the root is then not final), and G
    gets a terminal final state reached by `</s>` from every history and the unigram state (the histories
    are then not final): the shape `... -sil:eps-> final` / terminal `</s>` of real decoding graphs."""
    rng = np.random.default_rng(seed)
    V, K = n_words, min(n_succ, n_words)
    n_real_hmm = am.n_hmm - (1 if am.sp_hmm >= 0 else 0)
    use_sp = with_sp and am.sp_hmm >= 0
    P = min(n_phones, max(2, n_real_hmm // 2))
    prons = []
    for l in rng.integers(pron_len[0], pron_len[1] + 1, size=V):
        ph = rng.integers(0, P, size=int(l))
        hm = [int(ph[0])]
        for k in range(1, int(l)):
            hm.append(P + (int(ph[k - 1]) * 7919 + int(ph[k]) * 104729 + k * 31) % (n_real_hmm - P))
        prons.append(tuple(hm))
    prons.sort()                                        # depth-first order of the prefix tree
    # ---- C.L
    src, dst, il, ol, wf = [], [], [], [], []
    nxt = 1
    back = 0
    if use_sp:
        back = nxt; nxt += 1                            # word-end node: sp (tee) arc to the root
        src.append(back); dst.append(0); il.append(am.sp_hmm + 1); ol.append(0); wf.append(0.0)
    children = {}
    for w, pr in enumerate(prons):
        nd = 0
        for hm in pr[:-1]:
            key = (nd, hm)
            if key not in children:
                children[key] = nxt
                src.append(nd); dst.append(nxt); il.append(hm + 1); ol.append(0); wf.append(0.0)
                nxt += 1
            nd = children[key]
        src.append(nd); dst.append(back); il.append(pr[-1] + 1); ol.append(w + 1); wf.append(float(rng.uniform(0.0, 1.0)))
    cl_final = 0
    if terminal:                                        # root -m0:</s>-> x -m1:eps-> F (terminal, final)
        m0, m1 = int(rng.integers(0, P)), int(P + rng.integers(0, n_real_hmm - P))
        src.append(0); dst.append(nxt); il.append(m0 + 1); ol.append(V + 1); wf.append(float(rng.uniform(0.0, 1.0)))
        src.append(nxt); dst.append(nxt + 1); il.append(m1 + 1); ol.append(0); wf.append(0.0)
        cl_final = nxt + 1
        nxt += 2
        prons = prons + [(m0, m1)]
    order = np.argsort(np.asarray(src), kind="stable")
    A = lambda x, dt: np.asarray(x, dtype=dt)[order]
    cl = SynthNet(n_states=nxt, src=A(src, np.int32), dst=A(dst, np.int32), ilab=A(il, np.int32), olab=A(ol, np.int32),
                  w_file=A(wf, np.float32), fstate=np.asarray([cl_final], np.int32), fweight_file=np.asarray([0.0], np.float32),
                  n_words=V, prons=[np.asarray(p, np.int32) for p in prons], sp_hmm=am.sp_hmm if use_sp else -1)
    # ---- G
    succ = np.zeros((V + 1, K), dtype=np.int32)
    for h in range(V + 1):
        succ[h] = rng.choice(V, size=K, replace=False)
    uni = V + 1
    tri = {}                                            # (u, v) -> state, v a successor of u
    while len(tri) < min(n_tri, V * K):
        u = int(rng.integers(0, V)); v = int(succ[1 + u, rng.integers(0, K)])
        if (u, v) not in tri:
            tri[(u, v)] = V + 2 + len(tri)
    src, dst, il, ol, wf = [], [], [], [], []
    def arc(a, b, lab, cost):
        src.append(a); dst.append(b); il.append(lab); ol.append(lab); wf.append(cost)
    for h in range(V + 1):
        for w in succ[h]:
            w = int(w)
            arc(h, tri.get((h - 1, w), 1 + w) if h > 0 else 1 + w, w + 1, float(rng.uniform(0.5, 8.0)))
        arc(h, uni, 0, float(rng.uniform(1.0, 4.0)))
    for w in range(V):
        arc(uni, 1 + w, w + 1, float(rng.uniform(4.0, 12.0)))
    for (u, v), st in tri.items():
        for w in rng.choice(V, size=min(n_succ3, V), replace=False):
            w = int(w)
            arc(st, tri.get((v, w), 1 + w), w + 1, float(rng.uniform(0.3, 5.0)))
        arc(st, 1 + v, 0, float(rng.uniform(0.5, 3.0)))
    n_g = V + 2 + len(tri)
    fstate = np.concatenate([np.arange(1, V + 1), np.arange(V + 2, V + 2 + len(tri))]).astype(np.int32)
    if terminal:                                        # every history -</s>-> the terminal state, the only final one
        for h in list(range(1, V + 2)) + sorted(tri.values()):
            arc(h, n_g, V + 1, float(rng.uniform(0.5, 4.0)))
        fstate = np.asarray([n_g], np.int32)
        n_g += 1
    perm = rng.permutation(len(src))                    # file order: shuffled, then grouped by state
    order = perm[np.argsort(np.asarray(src)[perm], kind="stable")]
    g = SynthNet(n_states=n_g, src=A(src, np.int32), dst=A(dst, np.int32), ilab=A(il, np.int32),
                 olab=A(ol, np.int32), w_file=A(wf, np.float32), fstate=fstate,
                 fweight_file=rng.uniform(0.0, 2.0, size=fstate.shape[0]).astype(np.float32),
                 n_words=V, prons=cl.prons, succ=succ, sp_hmm=cl.sp_hmm)
    return cl, g


def make_wfst_sized(seed: int, am: SynthAM, target_arcs: int, n_words: int,
                    pron_len=(2, 5), with_sp: bool = False, hub: str = "tree") -> SynthNet:
    """Pick the bigram fan-out so that the graph has about target_arcs arcs."""
    mean_len = 0.5 * (pron_len[0] + pron_len[1]) + (1.0 if with_sp else 0.0)
    k = int(round((target_arcs - n_words * (mean_len + 2.0)) / ((n_words + 1) * mean_len)))
    k = max(1, min(k, n_words))
    return make_wfst(seed, am, n_words, k, pron_len=pron_len, with_sp=with_sp, hub=hub)


def sample_utterance(seed: int, net: SynthNet, am: SynthAM, n_words: int,
                     p_backoff: float = 0.2, noise: float = 1.0, end_word: int = -1):
    """Random accepted word sequence -> (features [T, D] float32, word labels).  end_word >= 0: that word
    (make_cl_g's terminal `</s>`, no pause model behind it) closes the sequence."""
    rng = np.random.default_rng(seed)
    words, gmm_seq = [], []
    h = 0
    for k in range(n_words + (1 if end_word >= 0 else 0)):
        last = end_word >= 0 and k == n_words
        if last:
            w = end_word
        elif rng.random() < p_backoff:
            w = int(rng.integers(0, net.n_words))
        else:
            w = int(net.succ[h, rng.integers(0, net.succ.shape[1])])
        words.append(w)
        models = list(net.prons[w])
        if net.sp_hmm >= 0 and not last:
            models.append(net.sp_hmm)
        for hm in models:
            n = int(am.hmm_nstates[hm]); tm = am.transp[am.hmm_tm[hm]]
            s = 0
            while True:
                p = tm[s, :n].astype(np.float64)
                s2 = int(rng.choice(n, p=p / p.sum()))
                if s2 == n - 1:
                    break
                gmm_seq.append(int(am.hmm_gmm[hm, s2]))
                s = s2
        h = 1 + w
    g = np.asarray(gmm_seq, dtype=np.int64)
    T = g.shape[0]
    cw = np.cumsum(am.weight[g].astype(np.float64), axis=1)
    u = rng.random(size=(T, 1)) * cw[:, -1:]
    m = np.minimum((u > cw).sum(axis=1), am.max_mix - 1)
    mu = am.mean[g, m]
    sd = np.sqrt(am.var[g, m])
    x = (mu + noise * sd * rng.normal(size=mu.shape)).astype(np.float32)
    return x, np.asarray(words, dtype=np.int32) + 1


def _hub_tree(rng, hub, nxt, prons, uni, eps_word_frac):
    """Lexicon prefix tree below state `hub` with tropical weight pushing (the back-off /
    unigram state of det(L.G)).  Returns (src, dst, il, ol, w, next free state)."""
    V = len(prons)
    src, dst, il, ol, wf = [], [], [], [], []
    children, node_words, node_min, kids = {}, {hub: []}, {}, {}
    for w in range(V):
        nd = hub
        for hm in prons[w][:-1]:
            key = (nd, int(hm))
            if key not in children:
                children[key] = nxt; node_words[nxt] = []; nxt += 1
            nd = children[key]
        node_words[nd].append(w)
    for (nd, hm), ch in children.items():
        kids.setdefault(nd, []).append((hm, ch))
    order = [hub]                      # parents before children; costs bottom-up without recursion
    for nd in order:
        order.extend(ch for _, ch in sorted(kids.get(nd, [])))
    for nd in reversed(order):
        node_min[nd] = min([float(uni[w]) for w in node_words[nd]] + [node_min[ch] for _, ch in kids.get(nd, [])])
    eps_w = rng.random(size=V) < eps_word_frac
    for nd in order:
        base = node_min[nd]
        for hm, ch in sorted(kids.get(nd, [])):
            src.append(nd); dst.append(ch); il.append(hm + 1); ol.append(0); wf.append(node_min[ch] - base)
        for w in node_words[nd]:
            cur, cost = nd, float(uni[w]) - base
            if eps_w[w]:
                src.append(cur); dst.append(nxt); il.append(0); ol.append(w + 1); wf.append(cost)
                cur = nxt; nxt += 1; cost = 0.0
            src.append(cur); dst.append(1 + w); il.append(int(prons[w][-1]) + 1)
            ol.append(0 if eps_w[w] else w + 1); wf.append(cost)
    return (np.asarray(src, dtype=np.int32), np.asarray(dst, dtype=np.int32), np.asarray(il, dtype=np.int32),
            np.asarray(ol, dtype=np.int32), np.asarray(wf, dtype=np.float32), nxt)


def make_wfst_trigram(seed: int, am: SynthAM, n_words: int, n_tri_hist: int, k2_mean: float,
                      k3_mean: float, big_frac: float = 0.01, big_range=(1000, 10000),
                      pron_len=(2, 5), n_phones: int = 40, eps_word_frac: float = 0.02) -> SynthNet:
    """Trigram-shaped C.L.G (BASELINE.json configs[3]), generated with numpy only so that
    tens of millions of arcs take seconds.

    States: 0 = <s> history (initial); 1..V = one-word histories; V+1 = unigram hub (lexicon
    prefix tree); V+2 .. V+1+H = two-word histories (u,v).  All history states are final.
    Every (history, successor word) pair owns an un-shared chain of phone arcs that ends in
    the two-word history (v,w) if the graph has it, else in the one-word history (w).
    Two-word histories back off (epsilon) to their one-word history, one-word histories to the
    hub.  A fraction big_frac of the one-word histories has 10^3..10^4 successors (the
    heavy-tailed out-degree the config asks for); the others are geometric around the mean.
    """
    rng = np.random.default_rng(seed)
    V = n_words
    n_real_hmm = am.n_hmm - (1 if am.sp_hmm >= 0 else 0)
    P = min(n_phones, max(2, n_real_hmm // 2))
    plen = rng.integers(pron_len[0], pron_len[1] + 1, size=V).astype(np.int64)
    poff = np.zeros(V + 1, dtype=np.int64); poff[1:] = np.cumsum(plen)
    ph = rng.integers(0, P, size=int(poff[-1]))
    kk = np.arange(int(poff[-1]), dtype=np.int64) - np.repeat(poff[:-1], plen)      # position in word
    prev = np.concatenate([[0], ph[:-1]])
    pflat = np.where(kk == 0, ph, P + (prev * 7919 + ph * 104729 + kk * 31) % (n_real_hmm - P)).astype(np.int32)
    prons = [pflat[poff[w]:poff[w + 1]] for w in range(V)]
    hub = V + 1

    # ---- one-word histories (and <s>): successor sets
    k2 = 1 + rng.geometric(1.0 / max(1.0, k2_mean - 1.0), size=V + 1)
    big = rng.random(size=V + 1) < big_frac
    k2[big] = rng.integers(big_range[0], big_range[1] + 1, size=int(big.sum()))
    k2 = np.minimum(k2, V)
    h2 = np.repeat(np.arange(V + 1, dtype=np.int64), k2)
    key2 = np.unique(h2 * V + rng.integers(0, V, size=h2.shape[0]))                 # (history, word), deduplicated
    h2, w2 = key2 // V, key2 % V
    # ---- two-word histories: a random subset of the (v, w) pairs seen above (v a real word)
    cand = key2[h2 > 0]
    H = min(n_tri_hist, cand.shape[0])
    tri_key = np.sort(rng.choice(cand, size=H, replace=False))                      # (1+v)*V + w
    tri_v, tri_w = tri_key // V - 1, tri_key % V
    tri_state = (V + 2 + np.arange(H)).astype(np.int64)
    k3 = np.minimum(1 + rng.geometric(1.0 / max(1.0, k3_mean - 1.0), size=H), V)
    i3 = np.repeat(np.arange(H, dtype=np.int64), k3)
    key3 = np.unique(i3 * V + rng.integers(0, V, size=i3.shape[0]))
    i3, w3 = key3 // V, key3 % V

    # ---- all (history state, last word of the history, successor word) triples
    t_h = np.concatenate([h2, tri_state[i3]])
    t_last = np.concatenate([h2 - 1, tri_w[i3]])                                    # -1 for <s>
    t_w = np.concatenate([w2, w3])
    q = (t_last + 1) * V + t_w
    pos_c = np.minimum(np.searchsorted(tri_key, q), H - 1)
    hit = (t_last >= 0) & (tri_key[pos_c] == q)
    t_dst = np.where(hit, tri_state[pos_c], 1 + t_w)
    nT = t_h.shape[0]
    t_cost = rng.uniform(0.5, 8.0, size=nT).astype(np.float32)
    t_first = rng.random(size=nT) < 0.5                                             # word label on first / last arc
    # ---- expand the triples into phone-arc chains
    L = plen[t_w]
    aoff = np.zeros(nT + 1, dtype=np.int64); aoff[1:] = np.cumsum(L)
    nA = int(aoff[-1])
    n_hist = V + 2 + H
    ibase = n_hist + (aoff[:-1] - np.arange(nT))                                    # first interior state of a chain
    trip = np.repeat(np.arange(nT, dtype=np.int64), L)
    k = np.arange(nA, dtype=np.int64) - aoff[trip]
    last = k == (L[trip] - 1)
    c_src = np.where(k == 0, t_h[trip], ibase[trip] + k - 1).astype(np.int32)
    c_dst = np.where(last, t_dst[trip], ibase[trip] + k).astype(np.int32)
    c_il = (pflat[poff[t_w[trip]] + k] + 1).astype(np.int32)
    c_ol = np.where((t_first[trip] & (k == 0)) | (~t_first[trip] & last), t_w[trip] + 1, 0).astype(np.int32)
    c_w = np.where(k == 0, t_cost[trip], np.float32(0.0)).astype(np.float32)
    del trip, k, last
    nxt = n_hist + (nA - nT)
    # ---- back-off arcs (stable sort below puts them after the history state's word chains)
    b_src = np.concatenate([np.arange(0, V + 1), tri_state]).astype(np.int32)
    b_dst = np.concatenate([np.full(V + 1, hub), 1 + tri_v]).astype(np.int32)
    b_w = rng.uniform(1.0, 4.0, size=b_src.shape[0]).astype(np.float32)
    uni = rng.uniform(4.0, 12.0, size=V)
    u_src, u_dst, u_il, u_ol, u_w, nxt = _hub_tree(rng, hub, nxt, prons, uni, eps_word_frac)

    src = np.concatenate([c_src, b_src, u_src]); dst = np.concatenate([c_dst, b_dst, u_dst])
    il = np.concatenate([c_il, np.zeros_like(b_src), u_il]); ol = np.concatenate([c_ol, np.zeros_like(b_src), u_ol])
    wf = np.concatenate([c_w, b_w, u_w])
    del c_src, c_dst, c_il, c_ol, c_w
    order = np.argsort(src, kind="stable")
    fstate = np.concatenate([np.arange(1, V + 1), tri_state]).astype(np.int32)
    fweight = rng.uniform(0.0, 2.0, size=fstate.shape[0]).astype(np.float32)
    # walk tables for sample_utterance_walk (the triples are sorted by history state id)
    walk_off = np.zeros(n_hist + 1, dtype=np.int64)
    np.add.at(walk_off, t_h + 1, 1)
    walk_off = np.cumsum(walk_off)
    walk_backoff = np.full(n_hist, -1, dtype=np.int32)
    walk_backoff[tri_state] = 1 + tri_v
    return SynthNet(n_states=int(nxt), src=src[order], dst=dst[order], ilab=il[order], olab=ol[order],
                    w_file=wf[order], fstate=fstate, fweight_file=fweight, n_words=V, prons=prons,
                    succ=None, sp_hmm=-1, walk_off=walk_off, walk_word=t_w.astype(np.int32),
                    walk_dst=t_dst.astype(np.int32), walk_backoff=walk_backoff)


def sample_utterance_walk(seed: int, net: SynthNet, am: SynthAM, n_words: int,
                          p_backoff: float = 0.2, noise: float = 1.0):
    """sample_utterance for make_wfst_trigram graphs: a random accepted word sequence that
    follows explicit n-gram arcs, backing off one level with probability p_backoff."""
    rng = np.random.default_rng(seed)
    words, gmm_seq = [], []
    h = 0
    for _ in range(n_words):
        while True:
            lo, hi = int(net.walk_off[h]), int(net.walk_off[h + 1])
            if hi > lo and rng.random() >= p_backoff:
                j = int(rng.integers(lo, hi))
                w, h = int(net.walk_word[j]), int(net.walk_dst[j])
                break
            b = int(net.walk_backoff[h])
            if b < 0:                                   # the unigram hub: any word
                w = int(rng.integers(0, net.n_words)); h = 1 + w
                break
            h = b
        words.append(w)
        for hm in net.prons[w]:
            n = int(am.hmm_nstates[hm]); tm = am.transp[am.hmm_tm[hm]]
            s = 0
            while True:
                p = tm[s, :n].astype(np.float64)
                s2 = int(rng.choice(n, p=p / p.sum()))
                if s2 == n - 1:
                    break
                gmm_seq.append(int(am.hmm_gmm[hm, s2]))
                s = s2
    g = np.asarray(gmm_seq, dtype=np.int64)
    cw = np.cumsum(am.weight[g].astype(np.float64), axis=1)
    u = rng.random(size=(g.shape[0], 1)) * cw[:, -1:]
    m = np.minimum((u > cw).sum(axis=1), am.max_mix - 1)
    mu = am.mean[g, m]
    x = (mu + noise * np.sqrt(am.var[g, m]) * rng.normal(size=mu.shape)).astype(np.float32)
    return x, np.asarray(words, dtype=np.int32) + 1


# ------------------------------------------------------------------ named configs

@dataclass
class HybridAM:
    """Hybrid ANN / HMM models (input of jd_am_create_hybrid): one HMM per phone, features = log posteriors."""
    priors: np.ndarray            # float32 [n_phones]
    states_per_model: int = 5
    sp_hmm: int = -1

    @property
    def n_hmm(self) -> int:
        return int(self.priors.shape[0])


def config_hybrid(seed: int = 3, n_phones: int = 30, states_per_model: int = 5, n_words: int = 50, n_succ: int = 5,
                  n_utts: int = 3, utt_words=(4, 9), hub: str = "tree"):
    """Hybrid-scoring regression case: bigram-shaped graph over n_phones one-HMM-per-phone models,
    utterances as log-posterior vectors (softmax of noisy logits that favour the phone being spoken)."""
    rng = np.random.default_rng(seed)
    pri = rng.uniform(0.5, 2.0, size=n_phones)
    am = HybridAM(priors=(pri / pri.sum()).astype(np.float32), states_per_model=states_per_model)
    net = make_wfst(seed + 100, am, n_words=n_words, n_succ=n_succ, with_sp=False, hub=hub, eps_word_frac=0.1)
    feats, words = [], []
    for u in range(n_utts):
        r = np.random.default_rng(seed + 1000 + u)
        h, rows, ws = 0, [], []
        for _ in range(int(r.integers(utt_words[0], utt_words[1] + 1))):
            w = int(net.succ[h, r.integers(0, net.succ.shape[1])])
            ws.append(w + 1)
            for ph in net.prons[w]:
                for _ in range(states_per_model - 2):               # self loop 0.5 / forward 0.5 per emitting state
                    for _ in range(int(r.geometric(0.5))):
                        logit = r.normal(size=n_phones)
                        logit[int(ph)] += 4.0
                        rows.append(logit - np.log(np.exp(logit).sum()))
            h = 1 + w
        feats.append(np.asarray(rows, dtype=np.float32))
        words.append(np.asarray(ws, dtype=np.int32))
    return am, net, feats, words


def config_toy(seed: int = 1):
    """BASELINE.json configs[0]: ~16-state 3-word toy, 10 tied states, M=2, one
    100-frame utterance (the look-ahead/plumbing case; includes the tee model)."""
    am = make_models(seed, n_gmm=10, n_hmm=5, n_mix=2, n_tm=3, sep=1.0, with_tee=True)
    net = make_wfst(seed + 100, am, n_words=3, n_succ=1, pron_len=(1, 2), with_sp=False)
    # grow the utterance until it has at least 100 frames, then keep the words
    for nw in range(4, 40):
        x, words = sample_utterance(seed + 200, net, am, nw)
        if x.shape[0] >= 100:
            break
    return am, net, [x], [words]


def config_small(seed: int = 7, n_utts: int = 4, with_sp: bool = True, sep: float = 0.6,
                 n_words: int = 60, n_succ: int = 6, n_gmm: int = 90, n_hmm: int = 41,
                 n_mix: int = 4, utt_words=(6, 14), hub: str = "flat"):
    """~10k-arc regression case with the tee model between words.  hub="flat": every back-off
    fans out into one eps:word arc per word (large epsilon closures); hub="tree": lexicon prefix
    tree below the back-off state (small closures, the shape of a determinised C.L.G)."""
    am = make_models(seed, n_gmm=n_gmm, n_hmm=n_hmm, n_mix=n_mix, n_tm=8, sep=sep, with_tee=with_sp)
    net = make_wfst(seed + 100, am, n_words=n_words, n_succ=n_succ, with_sp=with_sp, hub=hub,
                    eps_word_frac=0.02 if hub == "flat" else 0.2)
    rng = np.random.default_rng(seed + 300)
    feats, words = [], []
    for u in range(n_utts):
        x, w = sample_utterance(seed + 1000 + u, net, am, int(rng.integers(utt_words[0], utt_words[1] + 1)))
        feats.append(x); words.append(w)
    return am, net, feats, words


def config_mixed(seed: int = 11, n_utts: int = 4, sizes=(3, 4, 5, 6, 7, 8), with_sp: bool = True,
                 n_words: int = 40, n_succ: int = 5):
    """~10k-arc regression case with HMMs of 1..6 emitting states (8-lane instance records)."""
    am = make_models_mixed(seed, n_gmm=220, n_hmm=36, n_mix=3, sizes=sizes, with_tee=with_sp, sep=0.7)
    net = make_wfst(seed + 100, am, n_words=n_words, n_succ=n_succ, with_sp=with_sp)
    rng = np.random.default_rng(seed + 300)
    feats, words = [], []
    for u in range(n_utts):
        x, w = sample_utterance(seed + 1000 + u, net, am, int(rng.integers(5, 11)))
        feats.append(x); words.append(w)
    return am, net, feats, words


def config_c2(seed: int = 0, n_utts: int = 64, target_arcs: int = 1_000_000, n_gmm: int = 3000,
              n_hmm: int = 8000, n_mix: int = 16, n_words: int = 5000, sep: float = 1.0,
              utt_words=(9, 28), utt_offset: int = 0):
    """BASELINE.json configs[1]/[2]: ~1M-arc graph, 3k tied states x 16 mix.
    The graph and models depend on `seed` only; utterance u of the global batch
    is drawn from seed + 1000 + utt_offset + u, so ranks can own disjoint shards."""
    am = make_models(seed, n_gmm=n_gmm, n_hmm=n_hmm, n_mix=n_mix, n_tm=48, sep=sep)
    net = make_wfst_sized(seed + 100, am, target_arcs, n_words)
    feats, words = [], []
    for u in range(utt_offset, utt_offset + n_utts):
        rng = np.random.default_rng(seed + 300 + u)
        x, w = sample_utterance(seed + 1000 + u, net, am, int(rng.integers(utt_words[0], utt_words[1] + 1)))
        feats.append(x); words.append(w)
    return am, net, feats, words


def config_c2_mixed(seed: int = 0, n_utts: int = 64, target_arcs: int = 1_000_000, n_gmm: int = 3000, n_hmm: int = 8000, n_mix: int = 16,
                    n_words: int = 5000, utt_words=(9, 28)):
    """configs[1]'s graph size and model count with HMMs of 1 .. 6 emitting states side by side, skips included (make_models_mixed):
    the search's OTHER record layout (144-byte records, general predecessor loop) at bench size."""
    am = make_models_mixed(seed, n_gmm=n_gmm, n_hmm=n_hmm, n_mix=n_mix, with_tee=False)
    net = make_wfst_sized(seed + 100, am, target_arcs, n_words)
    feats, words = [], []
    for u in range(n_utts):
        rng = np.random.default_rng(seed + 300 + u)
        x, w = sample_utterance(seed + 1000 + u, net, am, int(rng.integers(utt_words[0], utt_words[1] + 1)))
        feats.append(x); words.append(w)
    return am, net, feats, words


def config_c4(seed: int = 0, n_utts: int = 8, n_words: int = 20000, n_tri_hist: int = 400_000,
              k2_mean: float = 110.0, k3_mean: float = 26.0, n_gmm: int = 5000, n_hmm: int = 12000,
              n_mix: int = 16, sep: float = 1.0, utt_words=(9, 28), utt_offset: int = 0):
    """BASELINE.json configs[3]: ~50M-arc trigram-shaped graph, 5k tied states x 16 mix.
    Smaller n_words / n_tri_hist / k*_mean give the same shape at test sizes."""
    am = make_models(seed, n_gmm=n_gmm, n_hmm=n_hmm, n_mix=n_mix, n_tm=48, sep=sep)
    net = make_wfst_trigram(seed + 100, am, n_words, n_tri_hist, k2_mean, k3_mean)
    feats, words = [], []
    for u in range(utt_offset, utt_offset + n_utts):
        rng = np.random.default_rng(seed + 300 + u)
        x, w = sample_utterance_walk(seed + 1000 + u, net, am, int(rng.integers(utt_words[0], utt_words[1] + 1)))
        feats.append(x); words.append(w)
    return am, net, feats, words
