"""Dynamic-composition row, second step: C.L o G expanded BY THE SEARCH, where
its tokens go (jd_net_create_lazy, csrc/jd_lazy.h) - the network is never built as a whole.

The check is the static path on jd_net_compose's graph (itself checked against offline composition and
the CPU oracle in test_gpu_compose.py): same expansion step, so the same arcs in the same order within a
state, so bit-identical hypotheses - whatever the order in which states were discovered."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [dict(seed=5, n_words=40, n_succ=4, n_tri=30, with_sp=True, lm=1.0),
         dict(seed=6, n_words=60, n_succ=6, n_tri=0, with_sp=False, lm=7.5),
         dict(seed=8, n_words=25, n_succ=3, n_tri=40, with_sp=True, lm=3.0)]


def _case(c, n_gmm=100, n_hmm=45):
    from juicer_amd import capi, synth
    am = synth.make_models(c["seed"], n_gmm=n_gmm, n_hmm=n_hmm, n_mix=2, n_tm=8, sep=0.6, with_tee=c["with_sp"])
    cl, g = synth.make_cl_g(c["seed"], am, n_words=c["n_words"], n_succ=c["n_succ"], n_tri=c["n_tri"], with_sp=c["with_sp"])
    return am, g, capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, c["lm"], 0.0)


def _same(a, b):
    assert a.n == b.n
    assert np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
    for k in ("score", "ac", "lm", "tot_score", "tot_lm"):
        x, y = np.asarray(getattr(a, k), np.float32), np.asarray(getattr(b, k), np.float32)
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), k
    for k in ("tot_proc_end_hyps", "tot_paths", "tot_insts_in"):
        assert a.stats[k] == b.stats[k], k


@pytest.mark.parametrize("pushing", [False, True], ids=["plain", "pushing"])
@pytest.mark.parametrize("c", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_lazy_composition_decodes_like_the_composed_graph(built, c, pushing):
    from juicer_amd import capi, synth
    am, g, ncl, ng = _case(c)
    models = capi.Models.from_htk(am)
    static = capi.Network.compose(ncl, ng, pushing=pushing)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18, pushing=pushing)
    s0, a0 = lazy.lazy_size()
    assert 0 < s0 < static.n_states                    # only the start state's neighbourhood exists
    feats = [synth.sample_utterance(c["seed"] + 1000 + u, g, am, 6 + u)[0] for u in range(6)]
    kw = dict(main_beam=300.0, max_streams=3)
    want = capi.Decoder(static, models, **kw).decode_batch(feats)
    dec = capi.Decoder(lazy, models, **kw)
    got = dec.decode_batch(feats[:3])                  # three streams expand the shared graph at once
    s1, a1 = lazy.lazy_size()
    got += dec.decode_batch(feats[3:])                 # ... and later utterances reuse it
    s2, a2 = lazy.lazy_size()
    assert sum(h.n for h in want) > 0
    for u in range(len(feats)):
        _same(got[u], want[u])
    assert s0 < s1 <= s2 <= static.n_states            # never more than the reachable composition
    # a second decoder on the grown network, and the same utterances again: nothing new to expand
    again = capi.Decoder(lazy, models, **kw).decode_batch(feats[:3])
    for u in range(3):
        _same(again[u], want[u])
    assert lazy.lazy_size() == (s2, a2)


def test_lazy_composition_streaming_and_wide_beam(built):
    """An unpruned search expands (nearly) everything reachable, and never more than the full composition; the streaming interface (decoderInit / processFrame / decoderFinish) drives it too."""
    from juicer_amd import capi, synth
    c = CASES[2]
    am, g, ncl, ng = _case(c)
    models = capi.Models.from_htk(am)
    static = capi.Network.compose(ncl, ng)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    x = synth.sample_utterance(c["seed"] + 77, g, am, 12)[0]
    want = capi.Decoder(static, models, main_beam=0.0).decode_batch([x])[0]
    dec = capi.Decoder(lazy, models, main_beam=0.0)
    dec.stream_init(0)
    for i in range(0, x.shape[0], 7):
        dec.stream_push(0, x[i:i + 7])
    got = dec.stream_finish(0)
    _same(got, want)
    ns = lazy.lazy_size()[0]
    assert 0.9 * static.n_states <= ns <= static.n_states


def test_lazy_network_errors(built):
    from juicer_amd import capi, synth
    c = CASES[0]
    am, g, ncl, ng = _case(c)
    models = capi.Models.from_htk(am)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    with pytest.raises(capi.JuicerAmdError):
        lazy.csr()
    with pytest.raises(capi.JuicerAmdError):
        capi.Network.lazy(lazy, ng, models)
    # no room to grow into: creation fails if the start state's closure does not fit, decoding otherwise
    x = synth.sample_utterance(c["seed"] + 5, g, am, 8)[0]
    with pytest.raises(capi.JuicerAmdError) as ei:
        small = capi.Network.lazy(ncl, ng, models, max_states=256, max_arcs=1 << 18)
        capi.Decoder(small, models, main_beam=300.0).decode_batch([x])
    assert ei.value.code == capi.JD_ENOMEM and "out of states" in str(ei.value)
    with pytest.raises(capi.JuicerAmdError) as ei:
        small = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=512)
        capi.Decoder(small, models, main_beam=300.0).decode_batch([x])
    assert ei.value.code == capi.JD_ENOMEM and "out of arcs" in str(ei.value)
    for tiny in (4, 8, 32):                                        # (far fewer than one state's successors: the state table itself fills up)
        with pytest.raises(capi.JuicerAmdError) as ei:
            small = capi.Network.lazy(ncl, ng, models, max_states=tiny, max_arcs=1 << 18)
            capi.Decoder(small, models, main_beam=300.0).decode_batch([x])
        assert ei.value.code == capi.JD_ENOMEM
    # the decoder still works on a network with room
    assert capi.Decoder(lazy, models, main_beam=300.0).decode_batch([x])[0].n >= 0


def test_batch_test_cli_composes_lazily(built, tmp_path):
    """juicer's -gramFsmFName mode with the graph composed by the search (-lazy): same output as composing first."""
    import subprocess
    from juicer_amd import build as jbuild, capi, io as jio, synth
    c = CASES[0]
    am = synth.make_models(c["seed"], n_gmm=100, n_hmm=45, n_mix=2, n_tm=8, sep=0.6, with_tee=True)
    cl, g = synth.make_cl_g(c["seed"], am, n_words=c["n_words"], n_succ=c["n_succ"], n_tri=c["n_tri"], with_sp=True)
    jio.write_fsm(tmp_path / "cl.fsm", cl)
    jio.write_fsm(tmp_path / "g.fsm", g)
    jio.write_jdam(tmp_path / "m.jdam", am)
    with open(tmp_path / "list.txt", "w") as f:
        for u in range(3):
            jio.write_jdf(tmp_path / ("u%d.jdf" % u), synth.sample_utterance(c["seed"] + 2000 + u, g, am, 5 + u)[0])
            f.write("%s\n" % (tmp_path / ("u%d.jdf" % u)))
    base = [jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "cl.fsm"), "-gramFsmFName", str(tmp_path / "g.fsm"),
            "-modelsFName", str(tmp_path / "m.jdam"), "-inputFName", str(tmp_path / "list.txt"),
            "-mainBeam", "200", "-lmScaleFactor", str(c["lm"]), "-outputFormat", "ref"]
    first = subprocess.run(base, capture_output=True, text=True, timeout=240)
    lazy = subprocess.run(base + ["-lazy"], capture_output=True, text=True, timeout=240)
    assert first.returncode == 0 and lazy.returncode == 0, lazy.stderr
    assert "composed by the search" in lazy.stderr
    assert len(lazy.stdout.splitlines()) == 3 and lazy.stdout == first.stdout
    pushed = [subprocess.run(base + ["-pushing"] + x, capture_output=True, text=True, timeout=240) for x in ([], ["-lazy"])]
    assert pushed[1].returncode == 0 and pushed[1].stdout == pushed[0].stdout
    # the IDecoder mirror of WFSTOnTheFlyDecoder (include/juicer_amd_decoder.hpp), frame by frame
    adapter = subprocess.run(base + ["-lazy", "-perFrameAdapter"], capture_output=True, text=True, timeout=240)
    assert adapter.returncode == 0, adapter.stderr
    assert adapter.stdout == first.stdout
    # the multi-GPU C++ host with a lazily composed network per device (N = 1 on the one-GPU test box)
    multi = subprocess.run(base + ["-lazy", "-devices", "1"], capture_output=True, text=True, timeout=240)
    assert multi.returncode == 0, multi.stderr
    assert "on each of 1 devices" in multi.stderr and multi.stdout == first.stdout


def test_lazy_with_the_decoders_other_options(built):
    """Histogram pruning, PARTIAL_DECODING traces, a small Path arena (records collected and renumbered under
    way) and setMaxAllocModels behave on a lazily composed network as on the composed graph."""
    from juicer_amd import capi, synth
    c = CASES[0]
    am, g, ncl, ng = _case(c)
    models = capi.Models.from_htk(am)
    static = capi.Network.compose(ncl, ng)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    x = np.concatenate([synth.sample_utterance(c["seed"] + 300 + u, g, am, 9)[0] for u in range(3)])
    kw = dict(main_beam=250.0, max_hyps=400, max_streams=1, max_paths=1 << 12)
    ds, dl = capi.Decoder(static, models, **kw), capi.Decoder(lazy, models, **kw)
    dl.set_max_alloc_models(50)                       # (a percentage of the network's - here: the capacity's - transitions)
    traces = 0
    for d in (ds, dl):
        d.set_partial_interval(100)
        d.stream_init(0)
    for pos in range(0, x.shape[0], 61):
        for d in (ds, dl):
            d.stream_push(0, x[pos:pos + 61])
        a, b = ds.stream_partial(0, trace_now=True), dl.stream_partial(0, trace_now=True)
        assert a == b
        traces += int(a[0])
    hs, hl = ds.stream_finish(0), dl.stream_finish(0)
    _same(hl, hs)
    assert hs.n > 0 and traces > 0
    assert ds.stream_partial(0) == dl.stream_partial(0)


def test_lazy_reset(built):
    """jd_net_lazy_reset: the network forgets what it expanded and grows again to the same results; it also
    clears a capacity failure."""
    from juicer_amd import capi, synth
    c = CASES[1]
    am, g, ncl, ng = _case(c)
    models = capi.Models.from_htk(am)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    s0 = lazy.lazy_size()
    feats = [synth.sample_utterance(c["seed"] + 500 + u, g, am, 7)[0] for u in range(4)]
    dec = capi.Decoder(lazy, models, main_beam=250.0, max_streams=2)
    first = dec.decode_batch(feats)
    s1 = lazy.lazy_size()
    assert s1[0] > s0[0]
    lazy.lazy_reset()
    assert lazy.lazy_size() == s0
    again = dec.decode_batch(feats)                   # the same decoder, on the emptied network
    for a, b in zip(again, first):
        _same(a, b)
    assert lazy.lazy_size()[0] == s1[0]
    # a network too small for ONE utterance: every decode fails (after a second try on a fresh arena generation)
    small = capi.Network.lazy(ncl, ng, models, max_states=300, max_arcs=1 << 18)
    t0 = small.lazy_size()
    d2 = capi.Decoder(small, models, main_beam=250.0)
    for k in range(2):
        with pytest.raises(capi.JuicerAmdError) as ei:
            d2.decode_batch(feats[:1])
        assert ei.value.code == capi.JD_ENOMEM
        assert small.lazy_generation() == 2 * k + 1    # (the failed network started again before the retry, and before the next call)
    small.lazy_reset()
    assert small.lazy_size() == t0
    with pytest.raises(capi.JuicerAmdError):
        capi.Network.compose(ncl, ng).lazy_reset()


def test_lazy_broad_epsilon_closure(built):
    """A start state with 300 epsilon successors: closing a state is depth first, so the breadth of a closure is
    no limit (the stack is as deep as the epsilon chains are)."""
    from juicer_amd import capi, synth
    am = synth.make_models(3, n_gmm=60, n_hmm=30, n_mix=2, n_tm=8, sep=0.8, with_tee=False)
    n_br, n_words = 300, 20
    # C.L: 0 -eps-> i (i = 1..300) -phone:word-> 301 -eps-> 0 ; 301 final
    row_ptr = [0, n_br] + [n_br + i for i in range(1, n_br + 1)] + [2 * n_br + 1]
    to = list(range(1, n_br + 1)) + [n_br + 1] * n_br + [0]
    ilab = [0] * n_br + [1 + (i % am.n_hmm) for i in range(n_br)] + [0]
    olab = [0] * n_br + [1 + (i % n_words) for i in range(n_br)] + [0]
    rng = np.random.default_rng(11)
    w = (-rng.uniform(0.1, 3.0, size=len(to))).astype(np.float32)
    ncl = capi.Network.from_csr(n_states=n_br + 2, init_state=0, row_ptr=row_ptr, to=to, w=w, ilab=ilab, olab=olab,
                                fstate=[n_br + 1], fweight=[0.0])
    # G: one state, a loop per word
    gw = (-rng.uniform(0.1, 2.0, size=n_words)).astype(np.float32)
    ng = capi.Network.from_csr(n_states=1, init_state=0, row_ptr=[0, n_words], to=[0] * n_words, w=gw,
                               ilab=list(range(1, n_words + 1)), olab=list(range(1, n_words + 1)), fstate=[0], fweight=[0.0])
    models = capi.Models.from_htk(am)
    static = capi.Network.compose(ncl, ng)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 12, max_arcs=1 << 14)
    assert lazy.lazy_size()[0] >= n_br + 1           # the start state's closure: every branch, at creation
    x = np.random.default_rng(5).normal(size=(60, am.D)).astype(np.float32)
    want = capi.Decoder(static, models, main_beam=200.0).decode_batch([x])[0]
    got = capi.Decoder(lazy, models, main_beam=200.0).decode_batch([x])[0]
    assert want.n > 0
    _same(got, want)


def test_lazy_epsilon_cycle(built):
    """A cycle of epsilon arcs in C.L (A -eps-> B -eps-> A): the depth-first closing does not follow it for ever."""
    from juicer_amd import capi, synth
    am = synth.make_models(4, n_gmm=40, n_hmm=12, n_mix=2, n_tm=8, sep=0.8, with_tee=False)
    n_words = 6
    # states: 0 = A (init), 1 = B, 2 = word end (final, eps back to A).  A and B each have phone:word arcs to 2.
    arcs = ([(0, 1, -0.5, 0, 0)] + [(0, 2, -0.3 * (k + 1), 1 + k, 1 + k) for k in range(3)]
            + [(1, 0, -0.7, 0, 0)] + [(1, 2, -0.2 * (k + 1), 4 + k, 4 + k) for k in range(3)] + [(2, 0, -0.1, 0, 0)])
    row_ptr = [0, 4, 8, 9]
    ncl = capi.Network.from_csr(n_states=3, init_state=0, row_ptr=row_ptr, to=[a[1] for a in arcs], w=np.float32([a[2] for a in arcs]),
                                ilab=[a[3] for a in arcs], olab=[a[4] for a in arcs], fstate=[2], fweight=[0.0])
    ng = capi.Network.from_csr(n_states=1, init_state=0, row_ptr=[0, n_words], to=[0] * n_words, w=np.float32([-0.4] * n_words),
                               ilab=list(range(1, n_words + 1)), olab=list(range(1, n_words + 1)), fstate=[0], fweight=[0.0])
    models = capi.Models.from_htk(am)
    static = capi.Network.compose(ncl, ng)
    lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 10, max_arcs=1 << 12)
    x = np.random.default_rng(9).normal(size=(40, am.D)).astype(np.float32)
    want = capi.Decoder(static, models, main_beam=200.0).decode_batch([x])[0]
    got = capi.Decoder(lazy, models, main_beam=200.0).decode_batch([x])[0]
    assert want.n > 0
    _same(got, want)
    assert lazy.lazy_size()[0] <= static.n_states


def test_lazy_generations(built):
    """Bounded look-ahead memory (the reference: an LRU cache, WFSTOnTheFlyDecoder.h:210-371): the arena starts a new
    GENERATION between utterances when it is past its high-water mark, and a batch that runs out of room under way is
    decoded again on a fresh one - results are those of the composed graph, the network never stays failed."""
    from juicer_amd import capi, synth
    c = CASES[0]
    am, g, ncl, ng = _case(c)
    models = capi.Models.from_htk(am)
    feats = [synth.sample_utterance(c["seed"] + 700 + u, g, am, 9)[0] for u in range(8)]
    kw = dict(main_beam=300.0, max_streams=2)
    want = capi.Decoder(capi.Network.compose(ncl, ng), models, **kw).decode_batch(feats)
    assert sum(h.n for h in want) > 0
    # what one wave of two utterances needs on its own, and what the four waves need together
    big = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    dec = capi.Decoder(big, models, **kw)
    alone = []
    for w in range(4):
        big.lazy_reset()
        dec.decode_batch(feats[2 * w:2 * w + 2])
        alone.append(big.lazy_size())
    big.lazy_reset()
    dec.decode_batch(feats)
    together = big.lazy_size()
    gen0 = big.lazy_generation()
    assert gen0 == 5                                   # (the resets above; nothing was dropped by the network itself)
    need_s, need_a = max(a[0] for a in alone), max(a[1] for a in alone)
    assert together[0] > need_s and together[1] > need_a
    # 1. high-water mark (set between the start state's closure and what a wave leaves behind): every wave that
    # begins behind another one gets a new generation
    net = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    mark = 0.5 * (net.lazy_size()[0] + min(a[0] for a in alone)) / float(1 << 16)
    net.lazy_set_high_water(mark)
    got = capi.Decoder(net, models, **kw).decode_batch(feats)
    for a, b in zip(got, want):
        _same(a, b)
    assert net.lazy_generation() == 3                  # before waves 2, 3 and 4
    # 2. room for any one wave but not for all four: some wave runs out under way and is decoded again
    cap_s, cap_a = (need_s + together[0]) // 2, (need_a + together[1]) // 2
    for cs, ca in ((cap_s, 1 << 18), (1 << 16, cap_a)):
        net = capi.Network.lazy(ncl, ng, models, max_states=cs, max_arcs=ca)
        net.lazy_set_high_water(1.0)                   # (never ahead of time)
        d = capi.Decoder(net, models, **kw)
        got = d.decode_batch(feats)
        for a, b in zip(got, want):
            _same(a, b)
        assert net.lazy_generation() >= 1
        ns, na = net.lazy_size()
        assert ns <= cs and na <= ca
        again = d.decode_batch(feats[:2])              # and the network goes on working
        _same(again[0], want[0]); _same(again[1], want[1])
    # 3. the streaming interface: generations begin at jd_stream_init, never inside an utterance
    net = capi.Network.lazy(ncl, ng, models, max_states=1 << 16, max_arcs=1 << 18)
    net.lazy_set_high_water(mark)
    d = capi.Decoder(net, models, main_beam=300.0, max_streams=2)
    for u in range(3):
        d.stream_init(0)
        with pytest.raises(capi.JuicerAmdError):       # an utterance is inside the network: no reset by hand either
            net.lazy_reset()
        for k in range(0, feats[u].shape[0], 37):
            d.stream_push(0, feats[u][k:k + 37])
        _same(d.stream_finish(0), want[u])
    assert net.lazy_generation() == 2
