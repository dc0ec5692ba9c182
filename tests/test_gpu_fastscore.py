"""The scoring OPTION jd_dec_set_scoring(JD_SCORE_FAST) (csrc/jd_gmm.h: jd_gmm_fast39 - fused multiply-add distance on pre-scaled
parameters, fp32 logAdd on the hardware's exp / log) held to what BASELINE.json's north_star asks of the path: 1-best words and
times IDENTICAL to the reference algorithm's (the CPU oracle), path / acoustic scores within 1e-4 relative - on every fixture the
exact path is tested on, through the launch path and through the resident slot pipeline.  The DEFAULT (JD_SCORE_EXACT) is the
bit-identical one and stays what every other test runs; the tolerance of this file is the one written here: RTOL = 1e-4."""
import numpy as np
import pytest

from helpers import bit_exact, oracle_certified_many

pytestmark = pytest.mark.gpu
RTOL = 1e-4                                               # north_star: "path/acoustic scores within 1e-4 relative"

BEAMS = [
    dict(),
    dict(main_beam=200.0),
    dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0),
    dict(main_beam=150.0, max_hyps=200),
    dict(max_hyps=300),
    dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150),
]


def _cfg(name):
    from juicer_amd import synth
    if name == "c2":
        return synth.config_c2(seed=0, n_utts=6, target_arcs=60_000, n_gmm=300, n_hmm=800, n_mix=8, n_words=500)
    return {"toy": synth.config_toy, "small": synth.config_small, "mixed": synth.config_mixed}[name]()


def _same_words_close_scores(g, o, what):
    assert g.n == o.n, "%s: %d words against the oracle's %d" % (what, g.n, o.n)
    if o.n <= 0:
        return
    assert np.array_equal(g.label, o.label) and np.array_equal(g.time, o.time), "%s: words / times differ" % (what,)
    for f in ("score", "ac", "lm"):
        a, b = np.asarray(getattr(g, f), np.float64), np.asarray(getattr(o, f), np.float64)
        assert np.all(np.abs(a - b) <= RTOL * np.maximum(1.0, np.abs(b))), "%s: %s beyond %g relative" % (what, f, RTOL)
    for f in ("tot_score", "tot_ac", "tot_lm"):
        assert abs(getattr(g, f) - getattr(o, f)) <= RTOL * max(1.0, abs(getattr(o, f))), "%s: %s" % (what, f)


@pytest.mark.parametrize("cfg", ["small", "c2"])
def test_fast_log_likelihoods_within_tolerance(built, cfg):
    """the table itself: every cell within 1e-4 relative of the exact kernel's (which equals the oracle's bit for bit) - and in fact
    within a few 1e-6: the headroom is what keeps pruning decisions, hence words and times, where they are"""
    from juicer_amd import capi
    from oracle.oracle import OracleAM
    am, net, feats, _ = _cfg(cfg)
    gam = capi.Models.from_htk(am)
    x = np.concatenate(feats)[:700]
    exact = gam.score_frames(x)
    assert np.array_equal(exact.view(np.uint32), OracleAM(am).score_frames(x).view(np.uint32))
    fast = gam.score_frames(x, mode=capi.SCORE_FAST)
    rel = np.abs(fast.astype(np.float64) - exact) / np.maximum(1.0, np.abs(exact))
    assert rel.max() <= RTOL, rel.max()
    assert rel.max() <= 2e-5, rel.max()                   # (measured: ~1e-6; a regression to "just inside 1e-4" would be a bug)
    assert not np.array_equal(fast.view(np.uint32), exact.view(np.uint32))       # (it IS the other kernel)


@pytest.mark.parametrize("cfg", ["toy", "small", "mixed", "c2"])
@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_fast_scoring_keeps_words_and_times(built, cfg, bi):
    from juicer_amd import capi
    am, net, feats, _ = _cfg(cfg)
    kw = BEAMS[bi]
    big = (1 << 25) if kw.get("main_beam", 0.0) in (0.0, 200.0) and not kw.get("max_hyps") else 0
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=len(feats), max_paths=big, **kw)
    gd.set_scoring(capi.SCORE_FAST)
    gs = gd.decode_batch(feats)
    want = oracle_certified_many(net, am, feats, **kw)
    for u, o in enumerate(want):
        _same_words_close_scores(gs[u], o, "%s utt %d %s" % (cfg, u, kw))
    # ... and back: the default is the bit-identical one
    gd.set_scoring(capi.SCORE_EXACT)
    gs = gd.decode_batch(feats)
    for u, o in enumerate(want):
        assert bit_exact(gs[u], o), (cfg, u, kw)
    gd.close()


def test_fast_scoring_through_the_resident_pipeline(built):
    """announced batches through the slots of the resident kernel, their tables scored by the fast kernel beside the search"""
    import torch
    from juicer_amd import capi
    am, net, feats, _ = _cfg("c2")
    kw = dict(main_beam=150.0)
    want = oracle_certified_many(net, am, feats, **kw)
    dev = torch.device("cuda", 0)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=8, **kw)
    gd.set_scoring(capi.SCORE_FAST)
    gd.set_pipeline(capi.FLOW_RESIDENT, 4, 8)
    offs = np.zeros(len(feats) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in feats])
    d_feats = torch.from_numpy(np.concatenate(feats)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        gd.prefetch_scores(d_feats.data_ptr(), offs, stream)
    for step in range(5):
        if step < 2:
            gd.prefetch_scores(d_feats.data_ptr(), offs, stream)
        gs = gd.decode_batch_device(d_feats.data_ptr(), offs, stream)
        for u, o in enumerate(want):
            _same_words_close_scores(gs[u], o, "pipeline step %d utt %d" % (step, u))
    assert gd.pipeline_stats()["utts_through"] >= 5 * len(feats)
    gd.close()


def test_fast_scoring_is_refused_where_it_does_not_apply(built):
    from juicer_amd import capi, synth
    am, net, _, _ = synth.config_hybrid()
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_hybrid(am.priors, am.states_per_model), max_streams=1, main_beam=150.0)
    with pytest.raises(capi.JuicerAmdError, match="39-dimensional GMM"):
        gd.set_scoring(capi.SCORE_FAST)
    with pytest.raises(capi.JuicerAmdError, match="JD_SCORE_EXACT or JD_SCORE_FAST"):
        gd.set_scoring(7)
    gd.close()
