"""GPU parity on random WFSTs of arbitrary shape (tests/random_topology.py): the HIP path against the certified oracle on graphs no
lexicon / language-model generator makes - states of any in- and out-degree, parallel arcs, self loops, epsilon and tee arcs
anywhere (forward only), labels on any arc, final weights on any state, arcs into the initial state.  These are the shapes the
decoder's structural decisions look at (csrc/jd_search.h: REC_SOLE - exit tokens of an arc that is alone into its state skip the
recombination; jd_dec_create: the decoder's own state numbering and the layout of the per-state words), so each case runs through
both kernels and with every such decision forced on and off."""
import numpy as np
import pytest

from helpers import STAT_KEYS, assert_hyp_matches, bit_exact

pytestmark = pytest.mark.gpu

BEAMS = [dict(main_beam=150.0), dict(main_beam=200.0, end_beam=120.0, word_beam=90.0), dict(main_beam=120.0, start_beam=100.0, max_hyps=120),
         dict(max_hyps=300), dict(main_beam=250.0)]


def _case(seed):
    from juicer_amd import synth
    import random_topology as rt
    rng = np.random.default_rng(seed)
    with_tee = bool(rng.random() < 0.6)
    if rng.random() < 0.6:
        am = synth.make_models(seed, n_gmm=60, n_hmm=25, n_mix=2, n_tm=6, sep=0.7, with_tee=with_tee)
    else:
        am = synth.make_models_mixed(seed, n_gmm=120, n_hmm=25, n_mix=2, with_tee=with_tee, sep=0.7)
    net = rt.random_net(seed + 7, am, n_states=int(rng.integers(12, 70)), p_chain=float(rng.choice([0.2, 0.5, 0.8])))
    feats = [rt.random_walk_features(seed + 9 + u, net, am, n_arcs=int(rng.integers(4, 16))) for u in range(int(rng.integers(1, 4)))]
    return am, net, feats, dict(BEAMS[int(rng.integers(0, len(BEAMS)))])


@pytest.mark.parametrize("block", range(4))
def test_random_topologies_vs_oracle(built, block, monkeypatch):
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    monkeypatch.setenv("JD_DEV", "1")
    checked = exact = with_hyp = 0
    for seed in range(7000 + 12 * block, 7000 + 12 * (block + 1)):
        am, net, feats, kw = _case(seed)
        od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
        try:
            ora = [od.decode_certified(x) for x in feats]
        except AssertionError:
            continue                                                   # (a tie-order sensitive fixture: nothing to hold an implementation to)
        gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
        base = None
        # the cluster kernel as it decides for itself; the slot kernel; every structural decision forced the other way
        for env in (dict(), dict(JD_CW="1", JD_SLOT_BATCH="1"), dict(JD_NO_SOLE="1"), dict(JD_RENUMBER="1", JD_SREC_SPLIT="2"),
                    dict(JD_RENUMBER="0", JD_SREC_SPLIT="0", JD_CW="1", JD_SLOT_BATCH="1")):
            for k in ("JD_CW", "JD_SLOT_BATCH", "JD_NO_SOLE", "JD_RENUMBER", "JD_SREC_SPLIT"):
                if k in env: monkeypatch.setenv(k, env[k])
                else: monkeypatch.delenv(k, raising=False)
            gd = capi.Decoder(gnet, gam, max_streams=len(feats), **kw)
            gs = gd.decode_batch(feats)
            for u, g in enumerate(gs):
                what = "seed %d %s %s utt %d" % (seed, kw, env, u)
                assert_hyp_matches(g, ora[u], what, check_stats=False)
                for k in STAT_KEYS:
                    assert g.stats[k] == ora[u].stats[k], "%s: stat %s %d vs oracle %d" % (what, k, g.stats[k], ora[u].stats[k])
                if not env:
                    checked += 1; exact += bit_exact(g, ora[u]); with_hyp += ora[u].n > 0
            got = [(g.n, g.label.tobytes(), g.time.tobytes(), g.score.tobytes()) for g in gs]
            if base is None: base = got
            assert got == base, "seed %d %s: results differ with %s" % (seed, kw, env)
            gd.close()
    print("random topologies, block %d: %d utterances, %d bit-exact incl. scores, %d with a hypothesis" % (block, checked, exact, with_hyp))
    assert checked >= 12 and exact >= checked - 1 and with_hyp >= 4


def _big_case(seed):
    from juicer_amd import synth
    import random_topology as rt
    rng = np.random.default_rng(seed)
    am = synth.make_models(seed, n_gmm=150, n_hmm=60, n_mix=3, n_tm=8, sep=0.7, with_tee=bool(rng.random() < 0.5))
    net = rt.random_net(seed + 7, am, n_states=int(rng.integers(300, 3000)), arcs_per_state=float(rng.uniform(2.5, 6.0)), n_words=200,
                        p_chain=float(rng.choice([0.3, 0.7])), hub_fanout=int(rng.choice([0, 700, 3000])))
    feats = [rt.random_walk_features(seed + 9 + u, net, am, n_arcs=25) for u in range(3)]
    kw = [dict(main_beam=200.0, max_hyps=2000), dict(main_beam=180.0, end_beam=150.0, word_beam=120.0), dict(main_beam=220.0)][int(rng.integers(0, 3))]
    return am, net, feats, kw, float(rng.choice([1.0, 3.0])), float(rng.choice([0.0, -1.0]))


def test_larger_random_topologies_through_every_call(built, monkeypatch):
    """300-3000 states, 2.5-6 arcs per state, some with a state of 700 / 3000 arcs (rows the search hands on in slices), thousands of
    active models per frame, a language-model scale and an insertion penalty: whole batches against the certified oracle through the
    cluster kernel and the slot kernel, and - through the stream calls, pushed in ragged pieces, with a trace after every push -
    against the oracle's partial paths (PARTIAL_DECODING, WFSTDecoderLite.cpp:824-890)."""
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    monkeypatch.setenv("JD_DEV", "1")
    checked = found = traces = 0
    for seed in range(8000, 8010):
        am, net, feats, kw, lm, pen = _big_case(seed)
        od = OracleDecoder(OracleNet(net, lm, pen), OracleAM(am), **kw)
        try:
            ora = [od.decode_certified(x) for x in feats]
        except AssertionError:
            continue
        gnet, gam = capi.Network.from_synth(net, lm, pen), capi.Models.from_htk(am)
        base = None
        for env in (dict(), dict(JD_CW="1", JD_SLOT_BATCH="1"), dict(JD_NO_SOLE="1", JD_RENUMBER="1")):
            for k in ("JD_CW", "JD_SLOT_BATCH", "JD_NO_SOLE", "JD_RENUMBER"):
                if k in env: monkeypatch.setenv(k, env[k])
                else: monkeypatch.delenv(k, raising=False)
            gd = capi.Decoder(gnet, gam, max_streams=len(feats), **kw)
            gs = gd.decode_batch(feats)
            for u, g in enumerate(gs):
                what = "seed %d %s lm %g pen %g %s utt %d" % (seed, kw, lm, pen, env, u)
                assert_hyp_matches(g, ora[u], what, check_stats=False)
                for k in STAT_KEYS:
                    assert g.stats[k] == ora[u].stats[k], "%s: stat %s %d vs oracle %d" % (what, k, g.stats[k], ora[u].stats[k])
                assert bit_exact(g, ora[u]), what
                if not env:
                    checked += 1; found += ora[u].n > 0
            got = [(g.n, g.label.tobytes(), g.time.tobytes(), g.score.tobytes()) for g in gs]
            if base is None: base = got
            assert got == base, "seed %d: results differ with %s" % (seed, env)
            if not env and ora[0].n > 0:                               # the stream calls: ragged pushes, a trace behind each
                x = feats[0]
                at = sorted(set(int(v) for v in np.random.default_rng(seed).integers(3, x.shape[0], size=5)))
                snaps, _ = od.decode_partial(x, interval=0, trace_at=at)
                gd.stream_init(0)
                pos = 0
                for f in at:
                    gd.stream_push(0, x[pos:f + 1]); pos = f + 1
                    assert gd.stream_partial(0, trace_now=True) == snaps[f], "seed %d: trace after frame %d" % (seed, f)
                    traces += 1
                gd.stream_push(0, x[pos:])
                h = gd.stream_finish(0)
                assert_hyp_matches(h, ora[0], "seed %d stream" % seed, check_stats=False)
                assert bit_exact(h, ora[0])
            gd.close()
    print("larger random topologies: %d utterances, %d with a hypothesis, %d traces" % (checked, found, traces))
    assert checked >= 21 and found >= 12 and traces >= 15


@pytest.mark.parametrize("seed", [8001, 8004, 8007])
def test_larger_random_topology_through_the_slot_pipeline(built, seed, monkeypatch):
    """The headline's path - announced batches through the resident slot kernel (jd_dec_set_pipeline) - on a graph of arbitrary shape:
    three batches of seven utterances (ragged: concatenated walks), more utterances than slots, a Path arena so small that the slots
    stop for collections; every result the certified oracle's, bit for bit."""
    import torch
    from juicer_amd import capi
    import random_topology as rt
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, _, kw, lm, pen = _big_case(seed)
    kw = {k: v for k, v in kw.items() if k != "max_hyps"}              # (through the pipeline as the headline runs it)
    od = OracleDecoder(OracleNet(net, lm, pen), OracleAM(am), **kw)
    batches, want = [], []
    for b in range(3):
        fs = []
        for u in range(7):
            parts = [rt.random_walk_features(seed + 100 * b + 10 * u + j, net, am, n_arcs=10 + 3 * u) for j in range(1 + u % 3)]
            fs.append(np.concatenate(parts))
        try:
            w = [od.decode_certified(x) for x in fs]
        except AssertionError:
            pytest.skip("a tie-order sensitive fixture")
        batches.append(fs); want.append(w)
    dev = torch.device("cuda", 0)

    def resident(batch):
        offs = np.zeros(len(batch) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([x.shape[0] for x in batch])
        return torch.from_numpy(np.concatenate(batch)).to(dev), offs
    buf = [resident(b) for b in batches]
    monkeypatch.setenv("JD_DEV", "1"); monkeypatch.setenv("JD_PIPE_CHUNK", "50")
    gnet, gam = capi.Network.from_synth(net, lm, pen), capi.Models.from_htk(am)
    # (thousands of Path records per frame: 2^18 of them last a few dozen frames.  An arena of 2^13 - two frames' worth - is not an
    # error, it is a collection per frame: 5 frames a second, and the pipeline's 30 s watchdog ends the batch with JD_ESTATE)
    for streams, extra in ((4, {}), (6, dict(max_paths=1 << 18))):
        gd = capi.Decoder(gnet, gam, max_streams=streams, **kw, **extra)
        try:
            gd.set_pipeline(capi.FLOW_RESIDENT, 3)
            order = [0, 1, 2, 1, 0, 2]
            for n in order[:2]: gd.prefetch_scores(buf[n][0].data_ptr(), buf[n][1], 0)
            for i, n in enumerate(order):
                if i + 2 < len(order): gd.prefetch_scores(buf[order[i + 2]][0].data_ptr(), buf[order[i + 2]][1], 0)
                gs = gd.decode_batch_device(buf[n][0].data_ptr(), buf[n][1], 0)
                assert gd.last_timing()["search_launches"] == 0
                for u, g in enumerate(gs):
                    assert_hyp_matches(g, want[n][u], "seed %d streams %d step %d batch %d utt %d" % (seed, streams, i, n, u), check_stats=False)
                    assert bit_exact(g, want[n][u])
                    for k in STAT_KEYS: assert g.stats[k] == want[n][u].stats[k], (seed, streams, i, n, u, k)
            torch.cuda.synchronize()
            ps = gd.pipeline_stats()
            assert ps["batches_back"] == len(order) and ps["frames_searched"] == sum(sum(x.shape[0] for x in batches[n]) for n in order), ps
            assert (ps["collections"] > 0) == bool(extra), ps
        finally:
            gd.close()
