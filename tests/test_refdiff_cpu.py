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
