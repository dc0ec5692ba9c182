"""The CPU oracle against the reference's OWN compiled classes, on every code path the GPU parity tests use (build container only).

`oracle/juicer_oracle.c` is what every `-m gpu` test holds the HIP path to; it is a restatement, and the reference ships no
vectors to pin it with.  What CAN be done in this container: compile the reference's hot-path translation units
from /root/reference/src where they lie (tools/refbase: against stand-ins for three absent third-party headers, which is why this
is a differential and not a pin - DESIGN.md 2) and demand that `WFSTDecoderLite` and the oracle agree, utterance by utterance, on
words, times, every score BIT FOR BIT, the reference's five statistics and the partial paths of PARTIAL_DECODING - over all six
pruning sets of tests/test_gpu_parity.py, the tee model, HMMs of 1-6 emitting states, histogram pruning, the epsilon back-off
graph shape, both network loaders (JWNT binary and the FSM TEXT constructor, src/WFSTNetwork.cpp:371-616, with a language-model
scale and an insertion penalty applied by the reference itself) and the two-thread decoder.

Skipped where /root/reference is absent (the GPU box).  The bench workloads at their own sizes: tools/refbase/run_refbase.py
-> profiles/cpu_reference_baseline.json.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "refbase"))
import refdiff  # noqa: E402

pytestmark = pytest.mark.skipif(not refdiff.available(), reason="/root/reference is not here: the differential runs in the build container only")

BEAMS = [                                                # tests/test_gpu_parity.py: BEAMS
    dict(),
    dict(main_beam=200.0),
    dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0),
    dict(main_beam=150.0, max_hyps=200),
    dict(max_hyps=300),
    dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150),
]


@pytest.fixture(scope="module")
def driver(built):
    return refdiff.build()


def _clean(r):
    assert r.get("error") is None, r
    n = r["utterances"]
    assert r["identical_hyps"] == n and r["identical_stats"] == n and r["ok"], r
    return r


@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_toy_all_pruning_sets(driver, bi):
    """configs[0] (tee model, look-ahead plumbing) under every pruning set, through both network loaders"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_toy()
    for loader in ("jwnt", "fsm"):
        r = _clean(refdiff.diff_case("toy", am, net, feats, BEAMS[bi], loader=loader))
        assert r["hyps_found"] == 1


@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_small_tee_model_all_pruning_sets(driver, bi):
    """the ~10k-arc graph with the tee `sp` model between words (tee recursion, src/WFSTDecoderLite.cpp:584-600)"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_small(n_utts=3)
    r = _clean(refdiff.diff_case("small", am, net, feats, BEAMS[bi]))
    assert r["hyps_found"] >= 2


@pytest.mark.parametrize("bi", [1, 2, 3, 5])
def test_mixed_topologies(driver, bi):
    """HMMs of 1 .. 6 emitting states with skips: the general predecessor loop (src/WFSTDecoderLite.cpp:387-424, :443-483)"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_mixed(n_utts=3)
    _clean(refdiff.diff_case("mixed", am, net, feats, BEAMS[bi]))


def test_fsm_text_loader_with_scale_and_penalty(driver):
    """the TEXT constructor applies lmScale and insPenalty itself (src/WFSTNetwork.cpp:371-616); the oracle's OracleNet restates that"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_small(n_utts=3)
    for scale, pen in ((7.5, -1.25), (0.5, 2.0)):
        _clean(refdiff.diff_case("small, scaled", am, net, feats, dict(main_beam=150.0, word_beam=90.0), loader="fsm", lm_scale=scale, ins_penalty=pen))


def test_partial_decoding_paths(driver):
    """PARTIAL_DECODING (src/WFSTDecoderLite.cpp:822-896): the partial paths recovered on the reference's own schedule (:358-368)
    and at recognitionFinish (:246-257), with and without end / word beams.  (Intervals of 30 frames and more: a trace in the first
    frames of an utterance, while some instance holds only tokens without a Path, walks the reference off that instance's token
    array - `while (path == NULL) { ++tok; ...}`, :848-852 - and the reference's own build segfaults on these inputs at an interval of 5.)"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_small(n_utts=4)
    some = 0
    for beams in (dict(main_beam=150.0), dict(main_beam=150.0, end_beam=100.0, word_beam=80.0), dict(main_beam=150.0, max_hyps=200)):
        for pti in (30, 50):
            r = _clean(refdiff.diff_case("small, traces", am, net, feats, beams, pti=pti))
            assert r["identical_partial"] == r["utterances"], r
    rows, _ = refdiff.run_reference(am, net, feats, dict(main_beam=150.0), pti=30)
    some = sum(len(r["partial"]) for r in rows)
    assert some > 0                                       # (the lists compared above are not all empty)


def test_configs1_shape_beam_and_histogram(driver):
    """configs[1]'s generator at a graph size the test can afford: beam 150 alone and with maxHyps (Histogram, src/Histogram.cpp)"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_c2(seed=0, n_utts=3, target_arcs=60_000, n_gmm=300, n_hmm=800, n_mix=8, n_words=500)
    _clean(refdiff.diff_case("c2 small", am, net, feats, dict(main_beam=150.0)))
    _clean(refdiff.diff_case("c2 small, maxHyps", am, net, feats, dict(main_beam=150.0, max_hyps=600)))


def test_trigram_backoff_shape_wide_beam(driver):
    """configs[3]'s generator (trigram-shaped, epsilon back-off arcs, lexicon-tree hubs) at a size the test can afford, beam 300"""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_c4(seed=0, n_utts=2, n_words=300, n_tri_hist=2000, k2_mean=12.0, k3_mean=4.0, n_gmm=200, n_hmm=500,
                                        n_mix=4, utt_words=(4, 7))
    _clean(refdiff.diff_case("c4 small", am, net, feats, dict(main_beam=300.0)))


def test_two_thread_decoder(driver):
    """WFSTDecoderLiteThreading + HTKFlatModelsThreading.  The reference's request queue is not synchronised
    (src/HTKFlatModelsThreading.cpp:100-133) and SURVEY.md 8d saw it abort: a run that hangs is reported, not failed."""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_c2(seed=0, n_utts=2, target_arcs=60_000, n_gmm=300, n_hmm=800, n_mix=8, n_words=500)
    r = refdiff.diff_case("c2 small, two threads", am, net, feats, dict(main_beam=150.0), threading=True, timeout=60)
    if r.get("error"):
        pytest.skip("the reference's two-thread decoder did not finish: %s" % r["error"])
    _clean(r)


@pytest.mark.parametrize("cfg", ["small", "mixed", "c2"])
def test_model_tables_prepared_by_the_reference(driver, cfg):
    """Model preparation, the other half of the path's set-up: the PRODUCT's tables (csrc/jd_host.cpp: jd_am_create_htk - what the HIP kernels
    read) and the oracle's (oracle/juicer_oracle.c) against what the reference's OWN code makes of the same models: HTKModels::readBinary +
    HTKFlatModels::init (src/HTKFlatModels.cpp:94-177: det = gconst + log weight, inverse variances by a double division) and its IModels view
    of the HMMs (log transition matrices, SEIndex ranges, tee log-probabilities, src/HTKModels.cpp:581-593).  Bit for bit.  (The JMBI file
    carries the HTK-level parameters and two derived arrays this build computed - sumLogVarPlusNObsLog2Pi, logCompWeights; the sum, the
    inverse variances and everything about transitions are the reference's arithmetic.)"""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM
    am = {"small": lambda: synth.config_small(n_utts=1)[0], "mixed": lambda: synth.config_mixed(n_utts=1)[0],
          "c2": lambda: synth.make_models(0, n_gmm=300, n_hmm=800, n_mix=8, n_tm=48, sep=1.0)}[cfg]()
    ref = refdiff.reference_model_tables(am)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    for who, M in (("product", capi.Models.from_htk(am)), ("oracle", OracleAM(am))):
        det, mean, ivar = M.flat()
        trP, se, tee = M.trans()
        assert ref["det"].shape == det.shape and np.array_equal(ref["n_mix"], np.asarray(am.n_mix, np.int32)), who
        for g in range(det.shape[0]):
            k = int(ref["n_mix"][g])                       # (beyond a state's mixtures the tables hold padding of each side's own)
            assert np.array_equal(bits(ref["det"][g, :k]), bits(det[g, :k])), (who, "det", g)
            assert np.array_equal(bits(ref["mean"][g, :k]), bits(mean[g, :k])), (who, "mean", g)
            assert np.array_equal(bits(ref["ivar"][g, :k]), bits(ivar[g, :k])), (who, "ivar", g)
        assert np.array_equal(bits(ref["tee"]), bits(tee)), who
        for h in range(len(ref["trans"])):
            n, tm = int(ref["hmm_n"][h]), int(am.hmm_tm[h])
            assert n == int(am.hmm_nstates[h])
            assert np.array_equal(bits(ref["trans"][h]), bits(trP[tm, :n, :n])), (who, "trP", h)
            # (state 0 has no predecessor and nobody asks for its range: the reference leaves whatever its loop computed last there)
            assert np.array_equal(ref["se"][h][1:], se[tm, 1:n]), (who, "SEIndex", h)


@pytest.mark.parametrize("cfg", ["small", "c2"])
def test_network_loaded_by_the_reference(driver, cfg):
    """The static graph, row A1 of the path's scope: the PRODUCT's loaders - jd_net_load_fsm on the very text + symbol files, jd_net_load_jwnt
    on the very binary file, jd_net_create_arcs from the arrays - against what the reference's own WFSTNetwork makes of them (text
    constructor src/WFSTNetwork.cpp:371-616 with lmScale / insPenalty applied by the reference; readBinary :1228-1365): states, the
    initial state, every arc {to, in, out} in the reference's order, every weight and final weight bit for bit."""
    from juicer_amd import capi, synth
    am, net, _, _ = synth.config_small(n_utts=1) if cfg == "small" else synth.config_c2(seed=0, n_utts=1, target_arcs=60_000, n_gmm=300, n_hmm=800,
                                                                                        n_mix=8, n_words=500)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    for loader, scale, pen in (("fsm", 1.0, 0.0), ("fsm", 7.5, -1.25), ("jwnt", 1.0, 0.0), ("jwnt", 7.5, -1.25)):
        ref, files = refdiff.reference_network_tables(am, net, loader, scale, pen)
        mine = [capi.Network.from_fsm_file(files[0], files[1], files[2], scale, pen) if loader == "fsm" else capi.Network.from_jwnt_file(files[0], scale, pen)]
        if loader == "fsm" or (scale == 1.0 and pen == 0.0):            # (a JWNT round trip divides by the scale and multiplies again: only the loader of that file is held to it)
            mine.append(capi.Network.from_synth(net, scale, pen))
        for gnet in mine:
            c = gnet.csr()
            assert gnet.n_states == ref["n_states"] and gnet.n_arcs == ref["to"].shape[0] and gnet.init_state == ref["init"], (loader, scale)
            assert np.array_equal(c["row_ptr"], ref["row_ptr"]) and np.array_equal(c["to"], ref["to"]), (loader, scale)
            assert np.array_equal(c["ilab"], ref["ilab"]) and np.array_equal(c["olab"], ref["olab"]), (loader, scale)
            assert np.array_equal(bits(c["w"]), bits(ref["w"])), (loader, scale, pen)
            fin = ~np.isnan(ref["fin_w"])
            assert np.array_equal(np.isfinite(c["fin_w"]), fin), (loader, scale)       # (this build marks "not final" with +inf)
            assert np.array_equal(bits(c["fin_w"][fin]), bits(ref["fin_w"][fin])), (loader, scale, pen)


@pytest.mark.parametrize("spm", [3, 5])
def test_hybrid_models(driver, spm):
    """Hybrid ANN / HMM scoring (row f4): HTKModels::Load(phonesList, priors, statesPerModel), src/HTKModels.cpp:74-218 - output = log posterior
    - log prior (:481-512), one shared transition matrix - through the reference's plain HTKModels class (its HTKFlatModels never sets the
    `currInput` its own hybrid branch reads, src/HTKFlatModels.cpp:196 / :295-306: with the flat class the reference itself cannot run this)."""
    from juicer_amd import synth
    am, net, feats, _ = synth.config_hybrid(seed=3 + spm, states_per_model=spm)
    for kw in (dict(main_beam=200.0), dict(main_beam=150.0, max_hyps=100)):
        r = _clean(refdiff.diff_case("hybrid", am, net, feats, kw, loader="fsm"))
        assert r["hyps_found"] >= 1


@pytest.mark.parametrize("block", range(6))
def test_random_topologies(driver, block):
    """Graphs of arbitrary shape (tests/random_topology.py: any in- and out-degree, parallel arcs, self loops, epsilon and tee arcs
    anywhere, labels and final weights anywhere, arcs into the initial state) - the cases tests/test_gpu_random_topology.py holds
    the HIP path to the oracle on: here the oracle is held to the reference's own classes on the same graphs, both loaders."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_random_topology as trt
    done = found = 0
    for seed in range(7000 + 8 * block, 7000 + 8 * (block + 1)):
        am, net, feats, kw = trt._case(seed)
        r = refdiff.diff_case("random topology %d" % seed, am, net, feats, kw, loader=("fsm" if seed % 2 else "jwnt"))
        if r.get("reference_crashed"):
            continue                                               # (a shape the reference itself does not survive: nothing to compare)
        _clean(r)
        done += r["utterances"]; found += r["hyps_found"]
    assert done >= 8 and found >= 4, (done, found)


def test_larger_random_topologies(driver):
    """... at 300-3000 states with a language-model scale and an insertion penalty, states of 700 / 3000 arcs, histogram pruning, end and
    word beams (tests/test_gpu_random_topology.py: _big_case), and the partial paths of PARTIAL_DECODING on the same graphs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_random_topology as trt
    done = found = 0
    for seed in range(8000, 8010):
        am, net, feats, kw, lm, pen = trt._big_case(seed)
        neutral = lm == 1.0 and pen == 0.0                             # (a scale / penalty: the reference applies it itself in its text constructor)
        r = refdiff.diff_case("larger random topology %d" % seed, am, net, feats, kw, loader=("jwnt" if neutral and seed % 2 == 0 else "fsm"), lm_scale=lm, ins_penalty=pen)
        if r.get("reference_crashed"):
            continue
        _clean(r)
        done += r["utterances"]; found += r["hyps_found"]
        if seed % 3 == 0 and kw.get("end_beam"):                       # (traces: with end / word beams on - the reference walks off its token array without)
            rp = refdiff.diff_case("larger random topology %d, traces" % seed, am, net, feats, kw, loader="fsm", lm_scale=lm, ins_penalty=pen, pti=40)
            if not rp.get("reference_crashed"):
                _clean(rp)
                assert rp["identical_partial"] == rp["utterances"], rp
    assert done >= 21 and found >= 12, (done, found)


def test_two_thread_decoder_on_random_topologies(driver):
    """WFSTDecoderLiteThreading (the organisation DESIGN.md 3.7 compares the pipeline with) on graphs of arbitrary shape: where its
    unsynchronised request queue lets a run finish, the result is the one-thread decoder's and the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_random_topology as trt
    done = skipped = 0
    for seed in (8002, 8005, 8008):
        am, net, feats, kw, lm, pen = trt._big_case(seed)
        r = refdiff.diff_case("larger random topology %d, two threads" % seed, am, net, feats[:2], kw, loader="fsm", lm_scale=lm, ins_penalty=pen, threading=True, timeout=60)
        if r.get("error"):
            skipped += 1
            continue
        _clean(r)
        done += r["utterances"]
    if done == 0:
        pytest.skip("the reference's two-thread decoder finished none of the %d runs" % skipped)
