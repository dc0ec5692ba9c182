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
