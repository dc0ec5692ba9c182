"""CPU check of the offline-composition checkers themselves (tests/compose_ref.py, test infrastructure of the
dynamic-composition row): the filtered composition (back-off after a word only, interval look-ahead, optional
weight pushing) and textbook epsilon composition must accept the same word sequences with the same best
weights - they are two graphs with the same weighted paths."""
import numpy as np
import pytest

from compose_ref import compose_filtered, compose_naive


def _csr_of(net, scale):
    order = np.argsort(net.src, kind="stable")
    src = net.src[order]
    n = net.n_states
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, src + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    fin = np.full(n, np.inf, np.float32)
    fin[net.fstate] = (-(net.fweight_file.astype(np.float64) * scale)).astype(np.float32)
    return dict(row_ptr=rp, to=net.dst[order], w=(-(net.w_file[order].astype(np.float64) * scale)).astype(np.float32),
                ilab=net.ilab[order], olab=net.olab[order], fin_w=fin), int(net.src[0])


def _best(graph, words):
    """best total weight (scores: higher is better) of a path from the initial state to a final state whose
    output labels spell `words`; -inf if there is none.  Label-synchronous DP with epsilon-output relaxation."""
    S = graph["n_states"]
    NEG = -np.inf
    cur = np.full(S, NEG)
    cur[graph["init"]] = 0.0
    rp, to, w, ol = graph["row_ptr"], graph["to"], graph["w"].astype(np.float64), graph["olab"]

    def relax(v):
        changed = True
        while changed:                                  # epsilon-output arcs (acyclic in weight terms: they only add)
            changed = False
            for s in np.nonzero(v > NEG)[0]:
                for e in range(rp[s], rp[s + 1]):
                    if ol[e] == 0 and v[s] + w[e] > v[to[e]] + 1e-12:
                        v[to[e]] = v[s] + w[e]
                        changed = True
        return v

    cur = relax(cur)
    for x in words:
        nxt = np.full(S, NEG)
        for s in np.nonzero(cur > NEG)[0]:
            for e in range(rp[s], rp[s + 1]):
                if ol[e] == x and cur[s] + w[e] > nxt[to[e]]:
                    nxt[to[e]] = cur[s] + w[e]
        cur = relax(nxt)
    fin = graph["fin_w"].astype(np.float64)
    ok = np.isfinite(fin) & (cur > NEG)
    return float(np.max(cur[ok] + fin[ok])) if ok.any() else NEG


@pytest.mark.parametrize("seed,n_tri,with_sp", [(5, 30, True), (6, 0, False), (8, 40, True)])
def test_filtered_composition_equals_textbook_composition(seed, n_tri, with_sp):
    from juicer_amd import synth
    am = synth.make_models(seed, n_gmm=100, n_hmm=45, n_mix=2, n_tm=8, sep=0.6, with_tee=with_sp)
    cl, g = synth.make_cl_g(seed, am, n_words=25, n_succ=3, n_tri=n_tri, with_sp=with_sp)
    ccl, ci = _csr_of(cl, 1.0)
    cg, gi = _csr_of(g, 3.0)
    naive = compose_naive(ccl, ci, cg, gi)
    plain = compose_filtered(ccl, ci, cg, gi, pushing=False)
    pushed = compose_filtered(ccl, ci, cg, gi, pushing=True)
    assert plain["n_states"] == pushed["n_states"] and np.array_equal(plain["to"], pushed["to"])
    rng = np.random.default_rng(seed)
    checked = 0
    for _ in range(25):
        n = int(rng.integers(1, 5))
        h, ws = 0, []
        for _ in range(n):                              # successors of the history (explicit n-grams) or any word (back-off)
            wd = int(g.succ[h, rng.integers(0, g.succ.shape[1])]) if rng.random() < 0.6 else int(rng.integers(0, 25))
            ws.append(wd + 1)
            h = 1 + wd
        a, b, c = _best(naive, ws), _best(plain, ws), _best(pushed, ws)
        assert np.isfinite(a), ws                       # the unigram state accepts every word
        assert abs(a - b) <= 1e-3 * max(1.0, abs(a)) and abs(a - c) <= 1e-3 * max(1.0, abs(a)), (ws, a, b, c)
        checked += 1
    assert checked == 25


@pytest.mark.parametrize("seed,n_tri,with_sp", [(5, 30, True), (6, 0, False)])
def test_sentence_end_reaches_a_terminal_final_state(seed, n_tri, with_sp):
    """C.L ends in `root -m:</s>-> x -m:eps-> FINAL` (a terminal final state, no arcs out) and G in a terminal `</s>`
    state (final, no arcs, no back-off): the look-ahead interval of x is EMPTY and the G state has no arc at all, yet
    the label-less tail has to be followed (the reference always follows the transitions before the C.L final states,
    WFSTOnTheFlyDecoder.cpp:2665-2697) - else no composed state is final."""
    from juicer_amd import synth
    V = 25
    am = synth.make_models(seed, n_gmm=100, n_hmm=45, n_mix=2, n_tm=8, sep=0.6, with_tee=with_sp)
    cl, g = synth.make_cl_g(seed, am, n_words=V, n_succ=3, n_tri=n_tri, with_sp=with_sp, terminal=True)
    ccl, ci = _csr_of(cl, 1.0)
    cg, gi = _csr_of(g, 3.0)
    naive = compose_naive(ccl, ci, cg, gi)
    for pushing in (False, True):
        filt = compose_filtered(ccl, ci, cg, gi, pushing=pushing)
        assert np.isfinite(filt["fin_w"]).any()                         # (round 2: none - every sentence end was cut off)
        rng = np.random.default_rng(seed)
        for _ in range(15):
            h, ws = 0, []
            for _ in range(int(rng.integers(1, 5))):
                wd = int(g.succ[h, rng.integers(0, g.succ.shape[1])]) if rng.random() < 0.6 else int(rng.integers(0, V))
                ws.append(wd + 1)
                h = 1 + wd
            a, b = _best(naive, ws + [V + 1]), _best(filt, ws + [V + 1])
            assert np.isfinite(a), ws
            assert abs(a - b) <= 1e-3 * max(1.0, abs(a)), (ws, a, b)
            assert not np.isfinite(_best(naive, ws)) and not np.isfinite(_best(filt, ws))   # a sentence has to end


@pytest.mark.parametrize("seed,n_tri,with_sp,terminal", [(5, 30, True, False), (6, 0, False, False), (8, 40, True, True)])
def test_label_pushing_moves_labels_and_keeps_every_path(seed, n_tri, with_sp, terminal):
    """jd_net_push_labels (host code of the library, no device needed) against the same rule written in Python, and
    against what label pushing has to preserve: every complete path keeps its label sequence (arcs correspond one to
    one, so random walks through the original transducer are replayed on the pushed one), and the composition of the
    pushed C.L with G accepts the same word sequences at the same best weights as textbook composition of the
    original pair.  On a lexicon tree every label ends up on the first arc behind which its word is the only one left."""
    from juicer_amd import capi, synth
    from compose_ref import push_labels
    V = 25
    am = synth.make_models(seed, n_gmm=100, n_hmm=45, n_mix=2, n_tm=8, sep=0.6, with_tee=with_sp)
    cl, g = synth.make_cl_g(seed, am, n_words=V, n_succ=3, n_tri=n_tri, with_sp=with_sp, terminal=terminal)
    ncl = capi.Network.from_synth(cl, 1.0, 0.0)
    pushed_net, moved = ncl.push_labels()
    c0, c1 = ncl.csr(), pushed_net.csr()
    for k in ("row_ptr", "to", "ilab", "fin_w"):
        assert np.array_equal(c0[k], c1[k]), k
    assert np.array_equal(c0["w"].view(np.uint32), c1["w"].view(np.uint32))
    want, want_moved = push_labels(c0, ncl.init_state)
    assert np.array_equal(c1["olab"], want) and moved == want_moved
    assert moved > 0 and not np.array_equal(c0["olab"], c1["olab"])
    # every word still has exactly as many label-carrying arcs as it has (shared) emission points; labels are a
    # permutation-free move: the multiset of labels can only shrink where unique suffixes were shared
    assert set(np.unique(c1["olab"])) == set(np.unique(c0["olab"]))
    # random complete paths: same label sequence
    rp, to = c0["row_ptr"], c0["to"]
    fin = np.isfinite(c0["fin_w"])
    rng = np.random.default_rng(seed)
    done = 0
    for _ in range(400):
        s, a_seq = ncl.init_state, []
        for _ in range(60):
            if fin[s] and a_seq and rng.random() < 0.3:
                break
            if rp[s] == rp[s + 1]:
                break
            a = int(rng.integers(rp[s], rp[s + 1]))
            a_seq.append(a)
            s = int(to[a])
        if not fin[s]:
            continue
        l0 = [int(c0["olab"][a]) for a in a_seq if c0["olab"][a]]
        l1 = [int(c1["olab"][a]) for a in a_seq if c1["olab"][a]]
        assert l0 == l1
        done += 1
    assert done >= 20
    # a label never moves away from the initial state: along every arc sequence above the k-th label of the pushed
    # transducer is passed no later than the k-th label of the original (checked on prefixes of complete paths too)
    for _ in range(200):
        s, n0, n1 = ncl.init_state, 0, 0
        for _ in range(40):
            if rp[s] == rp[s + 1]:
                break
            a = int(rng.integers(rp[s], rp[s + 1]))
            n0 += int(c0["olab"][a] != 0); n1 += int(c1["olab"][a] != 0)
            assert n1 >= n0
            s = int(to[a])
    # composed with G: same weighted language as textbook composition of the original pair
    ccl, ci = _csr_of(cl, 1.0)
    cg, gi = _csr_of(g, 3.0)
    assert np.array_equal(ccl["olab"], c0["olab"])                      # (the loader keeps the file's arc order)
    pcl = dict(ccl, olab=want)
    naive = compose_naive(ccl, ci, cg, gi)
    for pushing in (False, True):
        filt = compose_filtered(pcl, ci, cg, gi, pushing=pushing)
        for _ in range(15):
            h, ws = 0, []
            for _ in range(int(rng.integers(1, 5))):
                wd = int(g.succ[h, rng.integers(0, g.succ.shape[1])]) if rng.random() < 0.6 else int(rng.integers(0, V))
                ws.append(wd + 1)
                h = 1 + wd
            if terminal:
                ws.append(V + 1)
            a, b = _best(naive, ws), _best(filt, ws)
            assert np.isfinite(a), ws
            assert abs(a - b) <= 1e-3 * max(1.0, abs(a)), (ws, a, b)
