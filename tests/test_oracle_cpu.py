"""CPU tests of the oracle itself (no GPU): frozen regression vectors + an independent
cross-check of its semantics.  PARITY UNPINNED - see tests/golden/make_golden.py."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "oracle_golden.json")) as f:
        return json.load(f)


def _hex32(a):
    return [np.float32(v).tobytes().hex() for v in a]


@pytest.mark.parametrize("case", ["toy", "small", "mixed"])
def test_oracle_matches_golden(built, golden, case):
    import make_golden
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    mk, _ = make_golden.CASES[case]
    am, net, feats, _ = mk()
    g = golden[case]
    assert make_golden.input_digest(am, net, feats) == g["input_sha256"], "synthetic generator changed"
    onet, oam = OracleNet(net), OracleAM(am)
    ll = oam.score_frames(np.concatenate(feats)[:64])
    assert make_golden.digest(ll) == g["gmm_ll_sha256"]
    for run in g["runs"]:
        od = OracleDecoder(onet, oam, **run["beams"])
        for f, e in zip(feats, run["utts"]):
            h = od.decode(f)
            assert h.n == e["n"]
            assert h.label.tolist() == e["label"] and h.time.tolist() == e["time"]
            assert _hex32(h.score) == e["score_hex"] and _hex32(h.ac) == e["ac_hex"] and _hex32(h.lm) == e["lm_hex"]
            assert _hex32([h.tot_score, h.tot_ac, h.tot_lm]) == e["tot"]
            assert {k: int(v) for k, v in h.stats.items()} == e["stats"]


def _check_indep(am, net, x, **net_kw):
    from indep_viterbi import viterbi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    oam = OracleAM(am)
    ll = oam.score_frames(x).astype(np.float64)
    ref = viterbi(net, am, ll, **{k: v for k, v in net_kw.items()})
    h = OracleDecoder(OracleNet(net, **net_kw), oam).decode(x)        # every beam disabled
    if ref is None:
        assert h.n == -1
        return
    assert h.n == len(ref[1])
    assert list(zip(h.label[::-1].tolist(), h.time[::-1].tolist())) == list(ref[1])
    assert abs((h.tot_ac + h.tot_lm) - ref[0]) <= 2e-5 * abs(ref[0])


def test_oracle_vs_independent_viterbi_toy(built):
    """Un-pruned decode == textbook full-trellis Viterbi (float64, different formulation)."""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_toy()
    _check_indep(am, net, feats[0])
    _check_indep(am, net, feats[0][:37])                  # ends mid-word or not: both must agree
    _check_indep(am, net, feats[0], lm_scale=3.0, ins_penalty=-2.5)


def test_oracle_vs_independent_viterbi_tee(built):
    """Graph with the tee 'sp' model between words and eps:word hub arcs."""
    from juicer_amd import synth
    am = synth.make_models(11, n_gmm=12, n_hmm=6, n_mix=2, n_tm=3, sep=1.0, with_tee=True)
    net = synth.make_wfst(12, am, n_words=4, n_succ=2, pron_len=(1, 2), with_sp=True)
    x, _ = synth.sample_utterance(13, net, am, 5)
    _check_indep(am, net, x[:60])


def test_oracle_logadd_properties(built):
    """GMM scores: single-mixture GMM equals the closed form; logAdd is order sensitive only
    below float resolution (sanity of the restated arithmetic)."""
    from juicer_amd import synth
    from oracle.oracle import OracleAM
    am = synth.make_models(5, n_gmm=6, n_hmm=3, n_mix=1, D=7, n_tm=2)
    x = np.random.default_rng(1).normal(size=(9, 7)).astype(np.float32)
    ll = OracleAM(am).score_frames(x)
    mu, var = am.mean[:, 0].astype(np.float64), am.var[:, 0].astype(np.float64)
    ref = -0.5 * (7 * np.log(2 * np.pi) + np.log(var).sum(1)[None] + (((x[:, None] - mu[None]) ** 2) / var[None]).sum(2))
    assert np.allclose(ll, ref, rtol=2e-5, atol=2e-4)


def test_oracle_rejects_bad_input(built):
    from juicer_amd import synth
    from oracle.oracle import OracleNet
    am, net, _, _ = synth.config_toy()
    import copy
    bad = copy.deepcopy(net)
    for f in ("src", "dst", "ilab", "olab", "w_file"):    # first arc moved to the end: state 0 is split
        a = getattr(bad, f); setattr(bad, f, np.concatenate([a[1:], a[:1]]))
    with pytest.raises(RuntimeError):
        OracleNet(bad)


def test_oracle_partial_decoding(built):
    """PARTIAL_DECODING restatement (WFSTDecoderLite.cpp:822-896): every traced list is a prefix of the
    final result (a converged record can no longer change), the scheduled traces sit on path
    collection frames - BOTH of collectPaths' triggers are modelled (:362): the frame rule (the first frame f with
    f - lastPathCollectFrame > 100) and the count rule (nPath / nPathNew > 12 with nPath > 10000, on the live Path
    objects the reference's allocator would hold) - and respect the interval, and recognitionFinish completes the
    list to the whole best path."""
    from juicer_amd import synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small()
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=150.0)
    x = np.concatenate(feats)
    o = od.decode(x)
    best = list(zip(o.label.tolist()[::-1], o.time.tolist()[::-1]))
    snaps, final = od.decode_partial(x, interval=0, trace_at=range(5, x.shape[0], 17))
    assert final == (snaps[max(snaps)][1] if snaps else [])            # interval 0: finish adds nothing (:247)
    grew = 0
    prev = []
    for f in sorted(snaps):
        found, lst = snaps[f]
        assert best[:len(lst)] == lst and lst[:len(prev)] == prev
        assert found == (len(lst) > len(prev))
        assert all(t <= f for _, t in lst)
        grew += found
        prev = lst
    assert grew >= 3
    for interval in (1, 150, 250):
        snaps, final = od.decode_partial(x, interval=interval)
        assert final == best
        coll = od.collect_frames
        # this graph (every back-off fans out into one eps:word arc per word) creates ~170 Path objects per frame: the
        # count rule fires every few frames, long before the frame rule's 101
        assert coll[0] < 100 and len(coll) > 3 * (x.shape[0] // 101)
        assert all(b - a <= 101 for a, b in zip([-1] + coll, coll))
        want, last = [], -1                                             # a trace rides on a collection when the interval has passed
        for f in coll:
            if f - last > interval:
                want.append(f)
                last = f
        assert sorted(snaps) == want
    # the lexicon-tree shape of a determinised C.L.G creates few Path objects: there only the frame rule fires
    am2, net2, feats2, _ = synth.config_small(hub="tree")
    od2 = OracleDecoder(OracleNet(net2), OracleAM(am2), main_beam=150.0)
    x2 = np.concatenate(feats2)
    snaps2, _ = od2.decode_partial(x2, interval=150)
    assert od2.collect_frames == list(range(100, x2.shape[0], 101))     # 100, 201, 302, ...
    assert sorted(snaps2) == [f for f in od2.collect_frames if (f + 1) % 202 == 0]
    # the decoder is reusable afterwards and unaffected
    o2 = od.decode(x)
    assert o2.n == o.n and np.array_equal(o2.label, o.label)


def test_oracle_two_thread_core_gives_the_same_results(built):
    """WFSTDecoderLiteThreading + HTKFlatModelsThreading restated (search thread + scoring thread): results,
    scores and the reference's statistics equal the single-thread core's; models with a skip into the exit
    state are refused like the reference does (WFSTDecoderLiteThreading.cpp:45-60)."""
    from juicer_amd import synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small()
    for kw in (dict(main_beam=150.0, max_hyps=200), dict(main_beam=200.0, end_beam=120.0, word_beam=100.0, start_beam=150.0), dict()):
        od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
        for x in feats[:3]:
            a, b = od.decode(x), od.decode(x, threading=True)
            assert a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
            for f in ("score", "ac", "lm"):
                assert np.array_equal(getattr(a, f).view(np.uint32), getattr(b, f).view(np.uint32)), f
            for k in a.stats:
                if k != "ties":
                    assert a.stats[k] == b.stats[k], k
    am2, net2, feats2, _ = synth.config_mixed()
    od2 = OracleDecoder(OracleNet(net2), OracleAM(am2), main_beam=150.0)
    try:
        od2.decode(feats2[0], threading=True)
        refused = False
    except RuntimeError as e:
        refused = "to-exit transition" in str(e)
    skips = any(int((am2.transp[t, 1:n - 2, n - 1] > 0).any()) for t, n in enumerate(am2.tm_nstates))
    assert refused == skips


def test_oracle_hybrid_scoring(built):
    """Hybrid ANN / HMM models (HTKModels::Load(phones, priors, statesPerModel), HTKModels.cpp:74-218): an
    emitting state scores x[phone] - log(prior[phone]); decoding finds the spoken words."""
    from juicer_amd import synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, words = synth.config_hybrid()
    oam = OracleAM.from_hybrid(am.priors, am.states_per_model)
    x = feats[0][:7]
    got = oam.score_frames(x)
    want = x.astype(np.float64) - np.log(am.priors.astype(np.float64))[None, :]
    assert np.allclose(got, want, rtol=0, atol=2e-6)
    trP, se, tee = oam.trans()
    n = am.states_per_model
    assert trP.shape[0] == 1 and np.isclose(trP[0, 0, 1], 0.0) and np.isclose(trP[0, 1, 1], np.log(0.5)) and np.isclose(trP[0, n - 2, n - 1], np.log(0.5))
    od = OracleDecoder(OracleNet(net), oam, main_beam=200.0)
    hit = 0
    for x, w in zip(feats, words):
        o = od.decode(x)
        assert o.n > 0
        hit += int(np.array_equal(o.label[::-1], w))
    assert hit >= len(feats) - 1


@pytest.mark.parametrize("case", ["tree_hub_12k_arcs", "flat_hub", "mixed_topologies"])
def test_oracle_vs_vectorised_independent_viterbi(built, case):
    """The oracle, every beam off, against tests/indep_viterbi_np.py: float64, all arcs at once, its own GMM
    evaluation (nothing shared with oracle/).  The same anchor faces the HIP path in tests/test_gpu_indep.py."""
    import indep_cases
    import indep_viterbi_np as iv
    from juicer_amd import synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = {"tree_hub_12k_arcs": indep_cases.CASES["tree_hub"], "flat_hub": synth.config_small,
                         "mixed_topologies": synth.config_mixed}[case]()
    od = OracleDecoder(OracleNet(net), OracleAM(am))
    for u, x in enumerate(feats[:2]):
        indep_cases.check_against_viterbi(od.decode(x), iv.viterbi(net, am, iv.gmm_loglik(am, x)), "%s utt %d" % (case, u))
