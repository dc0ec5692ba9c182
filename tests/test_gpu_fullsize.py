"""GPU tests at BASELINE.json configs[1] size (~1M-arc graph, 3000 tied states x 16 mixtures):
oracle parity on a few utterances (the oracle needs ~1 s per utterance at beam 150) plus
size-independent properties on the batch."""
import numpy as np
import pytest

from helpers import LE_KEYS, STAT_KEYS, assert_hyp_matches, bit_exact, oracle_certified_many

pytestmark = pytest.mark.gpu

N_UTTS = 12


@pytest.fixture(scope="module")
def c2(built):
    from juicer_amd import capi, synth
    am, net, feats, words = synth.config_c2(n_utts=N_UTTS)
    return dict(am=am, net=net, feats=feats, words=words,
                gnet=capi.Network.from_synth(net), gam=capi.Models.from_htk(am))


def test_fullsize_oracle_parity(c2):
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    kw = dict(main_beam=150.0)
    gd = capi.Decoder(c2["gnet"], c2["gam"], max_streams=N_UTTS, **kw)
    gs = gd.decode_batch(c2["feats"])
    od = OracleDecoder(OracleNet(c2["net"]), OracleAM(c2["am"]), **kw)
    checked = exact = 0
    for u in range(6):
        # decode_certified: the oracle's result is verified not to depend on the visiting order
        # of equal-score tokens (it fails, never skips, if a fixture were order sensitive)
        o = od.decode_certified(c2["feats"][u])
        assert_hyp_matches(gs[u], o, "c2 utt %d" % u)
        checked += 1
        exact += bit_exact(gs[u], o)
    print("checked %d utterances, %d bit-exact incl. scores" % (checked, exact))
    assert checked == 6 and exact == 6


def test_fullsize_64_stream_batch_oracle_parity(built):
    """BASELINE.json configs[1] in the shape the headline is quoted on - ONE batch of 64 utterances on 64 streams,
    mainBeam 150, the next batch's table announced and scored beside the search - ALL 64 utterances checked against the
    certified oracle: words, times, the reference's statistics, scores bit for bit; then the same batches through the
    resident pipeline (the headline's own path), every result of every step against the oracle again."""
    import torch
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_c2(n_utts=64)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    gd = capi.Decoder(gnet, gam, max_streams=64, **kw)
    offs = np.zeros(65, dtype=np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in feats])
    d_feats = torch.from_numpy(np.concatenate(feats)).to("cuda:0")
    torch.cuda.synchronize()
    gd.decode_batch_device(d_feats.data_ptr(), offs, 0)                # (the decoder learns the load; plans as in the bench)
    gd.prefetch_scores(d_feats.data_ptr(), offs, 0)
    gd.decode_batch_device(d_feats.data_ptr(), offs, 0)
    gd.prefetch_scores(d_feats.data_ptr(), offs, 0)
    gs = gd.decode_batch_device(d_feats.data_ptr(), offs, 0)           # a step as bench.py times it
    assert gd.last_timing()["prefetched"] == 1
    # EVERY utterance of the batch against the certified oracle (all host cores: ~35 CPU-seconds at beam 150)
    want = oracle_certified_many(net, am, feats, **kw)
    for u in range(64):
        assert_hyp_matches(gs[u], want[u], "c2 batch of 64, utt %d" % u)
        assert bit_exact(gs[u], want[u]), u
    assert all(h.n > 0 for h in gs)
    gd.close()
    # ... and the same batches the way bench.py runs them: through the slot kernel's 256 one-workgroup slots (one per CU, beside the scoring),
    # announcements nine batches ahead (jd_dec_set_pipeline: JD_FLOW_RESIDENT, ten deep) - every utterance of every
    # batch DIRECTLY against the oracle (words, times, the reference's statistics, scores bit for bit), not against the launch above
    gp = capi.Decoder(gnet, gam, max_streams=256, **kw)
    gp.set_pipeline(capi.FLOW_RESIDENT, 10, 256)
    f0 = gp.pipeline_stats()["frames_searched"]
    for _ in range(9):
        gp.prefetch_scores(d_feats.data_ptr(), offs, 0)
    n_steps = 12
    for step in range(n_steps):
        if step < 3:
            gp.prefetch_scores(d_feats.data_ptr(), offs, 0)
        if step == 4:
            gp.quiesce()
            torch.cuda.synchronize()
        got = gp.decode_batch_device(d_feats.data_ptr(), offs, 0)
        assert gp.last_timing()["search_launches"] == 0               # (handed back by the pipeline)
        for u in range(64):
            assert_hyp_matches(got[u], want[u], "resident pipeline, step %d utt %d" % (step, u))
            assert bit_exact(got[u], want[u]), (step, u)
    ps = gp.pipeline_stats()
    assert ps["frames_searched"] - f0 == n_steps * int(offs[-1]) and ps["batches_back"] == n_steps, ps
    gp.close()


def test_fullsize_histogram_pruning_parity(c2):
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    kw = dict(main_beam=200.0, max_hyps=3000, end_beam=150.0, word_beam=120.0)
    gd = capi.Decoder(c2["gnet"], c2["gam"], max_streams=4, **kw)
    gs = gd.decode_batch(c2["feats"][:4])
    od = OracleDecoder(OracleNet(c2["net"]), OracleAM(c2["am"]), **kw)
    for u in range(4):
        o = od.decode_certified(c2["feats"][u])
        assert_hyp_matches(gs[u], o, "c2 hist utt %d" % u)
        assert bit_exact(gs[u], o)


def test_fullsize_properties(c2):
    """Decoding is deterministic, independent of batch composition / stream slot, and mostly finds
    the generating word sequence of the model-sampled utterances."""
    from juicer_amd import capi
    kw = dict(main_beam=150.0)
    gd = capi.Decoder(c2["gnet"], c2["gam"], max_streams=N_UTTS, **kw)
    a = gd.decode_batch(c2["feats"])
    b = gd.decode_batch(c2["feats"])                      # same decoder, second pass
    perm = list(range(N_UTTS))[::-1]
    c = gd.decode_batch([c2["feats"][i] for i in perm])   # different slots, different neighbours
    gd1 = capi.Decoder(c2["gnet"], c2["gam"], max_streams=1, **kw)
    d = gd1.decode_batch(c2["feats"][:3])                 # one stream at a time
    # the synthetic lexicon has homophones, so the generating words are only mostly recovered
    same = sum(int(np.array_equal(a[u].label[::-1], c2["words"][u])) for u in range(N_UTTS))
    assert same >= N_UTTS // 2
    for u in range(N_UTTS):
        assert a[u].n > 0
        for other in (b[u], c[perm.index(u)]):
            assert other.n == a[u].n and np.array_equal(other.label, a[u].label)
            assert np.array_equal(other.time, a[u].time)
            assert np.array_equal(other.score.view(np.uint32), a[u].score.view(np.uint32))
            # the reference's own statistics are reproducible; the two work counters of the build
            # (arcs visited / Path records) depend on the order in which the inline closure meets
            # competing tokens at a state and may differ by a few per mille between runs
            for k in STAT_KEYS:
                assert other.stats[k] == a[u].stats[k], k
            for k in LE_KEYS:
                assert abs(other.stats[k] - a[u].stats[k]) <= 0.01 * a[u].stats[k], k
    for u in range(3):
        assert np.array_equal(d[u].label, a[u].label) and np.array_equal(d[u].time, a[u].time)
        assert np.array_equal(d[u].ac.view(np.uint32), a[u].ac.view(np.uint32))
    # word-end frames are increasing and inside the utterance; totals are consistent
    for u in range(N_UTTS):
        t = a[u].time[::-1]
        assert np.all(np.diff(t) >= 0) and t[-1] <= c2["feats"][u].shape[0] - 1
        assert a[u].tot_ac == a[u].ac[0] and a[u].tot_lm == a[u].lm[0]
        assert np.all(np.diff(a[u].ac[::-1]) < 0)          # cumulative acoustic log-likelihood decreases


def test_fullsize_gmm_tile_parity(c2):
    """Companion kernel on the full 3000 x 16 x 39 model, ragged row count (not a multiple of 64)."""
    from oracle.oracle import OracleAM
    x = np.concatenate(c2["feats"][:2])[:333]
    g = c2["gam"].score_frames(x)
    o = OracleAM(c2["am"]).score_frames(x)
    diff = g.view(np.uint32) != o.view(np.uint32)
    print("gmm mismatches %d of %d" % (diff.sum(), diff.size))
    if diff.any():                                        # (what to look at when it fails: the first offending entries)
        r, c = np.nonzero(diff)
        print([(int(a), int(b), float(g[a, b]), float(o[a, b])) for a, b in list(zip(r, c))[:8]])
    assert not diff.any()                                 # bit for bit, 333 x 3000 x 16 evaluations


def test_fullsize_wide_beam(c2):
    """mainBeam 300 on the 1M-arc graph: ~180k live hypotheses per frame and stream (the default
    arenas must hold them) - parity with the oracle on a short utterance."""
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    kw = dict(main_beam=300.0)
    x = [f[:160] for f in c2["feats"][:3]]
    gd = capi.Decoder(c2["gnet"], c2["gam"], max_streams=3, **kw)
    gs = gd.decode_batch(x)
    od = OracleDecoder(OracleNet(c2["net"]), OracleAM(c2["am"]), **kw)
    o = od.decode_certified(x[0])
    print("beam 300: %.0f active emit hyps/frame, order-dependent ties %d" % (o.stats["tot_active_emit_hyps"] / o.stats["n_frames"], o.stats["ties"]))
    assert_hyp_matches(gs[0], o, "beam300")
    for k in ("n_frames", "tot_active_emit_hyps", "tot_proc_emit_hyps"):
        assert gs[0].stats[k] == o.stats[k]


def test_trigram_shaped_wide_beam_parity(built):
    """BASELINE.json configs[3] in shape (trigram-level + bigram-level histories, back-off
    epsilon arcs, history states with up to thousands of out-arcs, 5000 tied states, beam 300)
    at about a tenth of its size so that the oracle finishes in seconds; the full ~50M-arc
    case is tests/manual/run_c4.py (result recorded in profiles/)."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, words = synth.config_c4(n_utts=4, n_words=6000, n_tri_hist=40_000, utt_words=(3, 6))
    assert net.n_arcs > 4_000_000
    deg = np.bincount(net.src, minlength=net.n_states)
    assert deg.max() >= 1000
    kw = dict(main_beam=300.0)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=4, **kw)
    gs = gd.decode_batch(feats)
    od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
    for u in range(2):
        # with the whole graph inside the beam, float32 collisions between different tokens do
        # occur (utterance 1: 66 of ~2e8 recombinations); decode_certified proves the result does
        # not depend on which of the two the reference would have met first
        o = od.decode_certified(feats[u])
        assert_hyp_matches(gs[u], o, "c4-shaped utt %d" % u)
    for u in range(4):
        assert gs[u].n > 0


def test_fullsize_more_utterances_than_streams(c2):
    """configs[2] shape on one GPU: more utterances than streams are decoded as successive
    lock-step waves; every utterance gets the result it gets in a single wave."""
    from juicer_amd import capi, synth
    _, _, more, _ = synth.config_c2(n_utts=40, utt_offset=100)       # same graph / models (seed 0), other utterances
    kw = dict(main_beam=150.0)
    one = capi.Decoder(c2["gnet"], c2["gam"], max_streams=40, **kw).decode_batch(more)
    waves = capi.Decoder(c2["gnet"], c2["gam"], max_streams=16, **kw).decode_batch(more)   # 16 + 16 + 8
    for a, b in zip(one, waves):
        assert a.n == b.n and a.n > 0 and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
        assert np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32))
        for k in STAT_KEYS:
            assert a.stats[k] == b.stats[k]


def test_configs3_full_size(built):
    """BASELINE.json configs[3] at FULL size: ~48M-arc trigram-shaped graph (history states with up
    to 10^4 out-arcs, back-off epsilon arcs), 5000 tied states x 16 mixtures, mainBeam 300, 8
    utterances in one batch (~2 million live instances per stream-frame).  The oracle needs
    ~0.15 s per frame here: all 8 utterances are checked against it, one oracle decoder per host core - certified not to
    depend on the visiting order of equal-score tokens - plus the batch properties."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, words = synth.config_c4(n_utts=8, utt_words=(3, 6))
    assert net.n_arcs > 45_000_000 and am.n_gmm == 5000
    kw = dict(main_beam=300.0)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=8, **kw)
    gs = gd.decode_batch(feats)
    # ALL 8 utterances against the certified oracle (it manages ~7 frames/s on this graph: one decoder per host core)
    import time
    t0 = time.time()
    want = oracle_certified_many(net, am, feats, workers=8, **kw)
    for u, o in enumerate(want):
        print("configs[3]: utterance %d, %d frames, %.0f instances / frame, %d order-dependent ties, oracle %.1f frames/s"
              % (u, feats[u].shape[0], o.stats["tot_insts_in"] / o.stats["n_frames"], o.stats["ties"],
                 feats[u].shape[0] / o.cpu_seconds))
        assert_hyp_matches(gs[u], o, "configs[3] utt %d" % u)
        assert bit_exact(gs[u], o), u
    print("configs[3]: the oracle took %.0f s for the 8 utterances" % (time.time() - t0))
    assert max(o.stats["tot_insts_in"] / o.stats["n_frames"] for o in want) > 500_000
    for v in range(8):
        assert gs[v].n > 0
        t = gs[v].time[::-1]
        assert np.all(np.diff(t) >= 0) and t[-1] <= feats[v].shape[0] - 1


def test_north_star_workload_parity(built):
    """BASELINE.json north_star at its size: the 14.3 M-arc trigram-shaped composed graph bench.py's north_star leg
    decodes (5000 tied states x 16 mixtures), mainBeam 200 - all eight utterances against the CPU oracle, certified not
    to depend on the visiting order of equal-score tokens, the reference's statistics bit-equal, scores bit for bit."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, words = synth.config_c4(seed=0, n_utts=8, n_words=10000, n_tri_hist=100_000)
    assert net.n_arcs > 10_000_000 and am.n_gmm == 5000
    kw = dict(main_beam=200.0)
    gs = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=8, **kw).decode_batch(feats)
    want = oracle_certified_many(net, am, feats, workers=8, **kw)      # all 8 (the oracle manages ~100 frames / s here)
    for u, o in enumerate(want):
        print("north star: utterance %d, %d frames, %.0f instances / frame, %d order-dependent ties"
              % (u, feats[u].shape[0], o.stats["tot_insts_in"] / o.stats["n_frames"], o.stats["ties"]))
        assert_hyp_matches(gs[u], o, "north star utt %d" % u)
        assert bit_exact(gs[u], o), u
        assert o.n > 0 and o.stats["tot_insts_in"] / o.stats["n_frames"] > 50_000
    for v in range(8):
        assert gs[v].n > 0


@pytest.fixture(scope="module")
def clg_pair(built):
    """BASELINE.json configs[4] at bench.py's size: lexicon tree (20 k words) and back-off trigram G, apart."""
    from juicer_amd import capi, synth
    am = synth.make_models(0, n_gmm=3000, n_hmm=2000, n_mix=16, n_tm=8, sep=0.6, with_tee=True)
    cl, g = synth.make_cl_g(0, am, n_words=20000, n_succ=40, n_tri=200000, n_succ3=8, with_sp=True)
    ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, 10.0, 0.0)
    feats = [synth.sample_utterance(100 + u, g, am, 8)[0] for u in range(6)]
    return dict(am=am, g=g, ncl=ncl, ng=ng, feats=feats, gam=capi.Models.from_htk(am))


def test_configs4_bench_size_composed_graph_vs_oracle(clg_pair):
    """configs[4] at the size bench.py runs it: C.L o G composed ON THE DEVICE (8.4 M arcs), decoded at mainBeam 200 by
    the static search - and by the CPU oracle on the SAME device-composed CSR (read back through the C ABI): words,
    times, scores and the reference's statistics.  Then the search-driven composition (nothing composed beforehand)
    on the same pair: bit-identical hypotheses."""
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    p = clg_pair
    net = capi.Network.compose(p["ncl"], p["ng"], max_states=1 << 26, max_arcs=1 << 27)
    assert net.n_arcs > 8_000_000
    kw = dict(main_beam=200.0)
    gs = capi.Decoder(net, p["gam"], max_streams=len(p["feats"]), **kw).decode_batch(p["feats"])
    c = net.csr()
    fs = np.nonzero(np.isfinite(c["fin_w"]))[0].astype(np.int32)
    onet = OracleNet.from_csr(net.n_states, net.init_state, c["row_ptr"], c["to"], c["w"], c["ilab"], c["olab"], fs, c["fin_w"][fs])
    want = oracle_certified_many(onet, OracleAM(p["am"]), p["feats"], workers=6, **kw)   # all six utterances
    for u, o in enumerate(want):
        print("configs[4] graph: utterance %d, %d frames, %.0f instances / frame, %d order-dependent ties"
              % (u, p["feats"][u].shape[0], o.stats["tot_insts_in"] / o.stats["n_frames"], o.stats["ties"]))
        assert_hyp_matches(gs[u], o, "configs[4] utt %d" % u)
        assert bit_exact(gs[u], o), u
        assert o.n > 0
    lazy = capi.Network.lazy(p["ncl"], p["ng"], p["gam"], max_states=1 << 22, max_arcs=1 << 23)
    got = capi.Decoder(lazy, p["gam"], max_streams=len(p["feats"]), **kw).decode_batch(p["feats"])
    ns, na = lazy.lazy_size()
    assert 0 < ns < net.n_states
    for a, b in zip(got, gs):
        assert a.n == b.n and a.n > 0 and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
        for k in ("score", "ac", "lm"):
            assert np.array_equal(np.asarray(getattr(a, k), np.float32).view(np.uint32), np.asarray(getattr(b, k), np.float32).view(np.uint32)), k


def test_configs4_bench_size_lazy_generations(clg_pair):
    """The search-driven composition at bench size with BOUNDED memory (the reference: an LRU cache,
    WFSTOnTheFlyDecoder.h:210-371; here whole generations of the arena): six utterances in three waves of two on a
    network whose arena (a) starts a new generation in front of every wave that begins behind another one (high-water
    mark), (b) holds any one wave but not the three together, so that a wave runs out of room under way and is decoded
    again on a fresh generation - hypotheses bit-identical to the fully composed 8.4 M-arc graph's every time."""
    from juicer_amd import capi
    p = clg_pair
    kw = dict(main_beam=200.0, max_streams=2)
    net = capi.Network.compose(p["ncl"], p["ng"], max_states=1 << 26, max_arcs=1 << 27)
    want = capi.Decoder(net, p["gam"], **kw).decode_batch(p["feats"])
    assert all(h.n > 0 for h in want)

    def same(a, b):
        assert a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
        for k in ("score", "ac", "lm"):
            assert np.array_equal(np.asarray(getattr(a, k), np.float32).view(np.uint32), np.asarray(getattr(b, k), np.float32).view(np.uint32)), k

    big = capi.Network.lazy(p["ncl"], p["ng"], p["gam"], max_states=1 << 22, max_arcs=1 << 23)
    dec = capi.Decoder(big, p["gam"], **kw)
    start = big.lazy_size()
    alone = []
    for w in range(3):
        big.lazy_reset()
        got = dec.decode_batch(p["feats"][2 * w:2 * w + 2])
        same(got[0], want[2 * w]); same(got[1], want[2 * w + 1])
        alone.append(big.lazy_size())
    big.lazy_reset()
    dec.decode_batch(p["feats"])
    together = big.lazy_size()
    need_s, need_a = max(a[0] for a in alone), max(a[1] for a in alone)
    print("lazily composed configs[4] pair: a wave of two utterances builds %s states / %s arcs, the three waves together %d / %d (of %d / %d composed)"
          % ([a[0] for a in alone], [a[1] for a in alone], together[0], together[1], net.n_states, net.n_arcs))
    assert together[0] > need_s + 10000 and together[1] > need_a + 10000   # (the waves share most of what they build)
    del dec
    # (a) the high-water mark between the start state's closure and what a wave leaves behind
    lz = capi.Network.lazy(p["ncl"], p["ng"], p["gam"], max_states=1 << 22, max_arcs=1 << 23)
    lz.lazy_set_high_water(0.5 * (start[0] + min(a[0] for a in alone)) / float(1 << 22))
    got = capi.Decoder(lz, p["gam"], **kw).decode_batch(p["feats"])
    for a, b in zip(got, want):
        same(a, b)
    assert lz.lazy_generation() == 2                                   # before waves 2 and 3
    # (b) room for any one wave, not for the three: states first, then arcs
    cap_s, cap_a = (need_s + together[0]) // 2, (need_a + together[1]) // 2
    for cs, ca in ((cap_s, 1 << 23), (1 << 22, cap_a)):
        lz = capi.Network.lazy(p["ncl"], p["ng"], p["gam"], max_states=cs, max_arcs=ca)
        lz.lazy_set_high_water(1.0)                                    # (never ahead of time)
        d = capi.Decoder(lz, p["gam"], **kw)
        got = d.decode_batch(p["feats"])
        for a, b in zip(got, want):
            same(a, b)
        assert lz.lazy_generation() >= 1
        ns, na = lz.lazy_size()
        assert ns <= cs and na <= ca
        again = d.decode_batch(p["feats"][:2])                         # and the network goes on working
        same(again[0], want[0]); same(again[1], want[1])
        del d


def test_device_composition_at_100k_arcs_vs_python_reference(built):
    """jd_net_compose against the same definition written in Python dictionaries (tests/compose_ref.py) on a pair whose
    composition has > 100 k arcs (1500 words, 6000 trigram histories): identical arrays, bit for bit, with and without
    weight pushing; and the search-driven composition of that pair decodes like the composed graph."""
    from compose_ref import compose_filtered
    from juicer_amd import capi, synth
    am = synth.make_models(3, n_gmm=300, n_hmm=200, n_mix=2, n_tm=8, sep=0.6, with_tee=True)
    cl, g = synth.make_cl_g(3, am, n_words=1500, n_succ=12, n_tri=6000, n_succ3=4, with_sp=True)
    ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, 5.0, 0.0)
    models = capi.Models.from_htk(am)
    feats = [synth.sample_utterance(3000 + u, g, am, 5 + u)[0] for u in range(3)]
    for pushing in (False, True):
        dev = capi.Network.compose(ncl, ng, pushing=pushing)
        want = compose_filtered(ncl.csr(), ncl.init_state, ng.csr(), ng.init_state, pushing=pushing)
        got = dev.csr()
        assert dev.n_arcs > 100_000 and dev.n_states == want["n_states"] and dev.init_state == want["init"]
        for k in ("row_ptr", "to", "ilab", "olab"):
            assert np.array_equal(got[k], want[k]), k
        assert np.array_equal(got["w"].view(np.uint32), want["w"].view(np.uint32))
        assert np.array_equal(got["fin_w"].view(np.uint32), want["fin_w"].view(np.uint32))
        a = capi.Decoder(dev, models, max_streams=3, main_beam=250.0).decode_batch(feats)
        lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 18, max_arcs=1 << 19, pushing=pushing)
        b = capi.Decoder(lazy, models, max_streams=3, main_beam=250.0).decode_batch(feats)
        for x, y in zip(a, b):
            assert x.n == y.n and x.n > 0 and np.array_equal(x.label, y.label) and np.array_equal(x.time, y.time)
            assert np.array_equal(np.asarray(x.score, np.float32).view(np.uint32), np.asarray(y.score, np.float32).view(np.uint32))
