"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from helpers import assert_hyp_matches, bit_exact, rel_close

pytestmark = pytest.mark.gpu

BEAMS = [
    dict(),
    dict(main_beam=200.0),
    dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0),
    dict(main_beam=150.0, max_hyps=200),
    dict(max_hyps=300),
    dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150),
]


def _setup(cfg):
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleNet
    am, net, feats, words = cfg
    return (capi.Network.from_synth(net), capi.Models.from_htk(am), OracleNet(net), OracleAM(am), feats, words)


@pytest.fixture(scope="module")
def toy(built):
    from juicer_amd import synth
    return _setup(synth.config_toy())


@pytest.fixture(scope="module")
def small(built):
    from juicer_amd import synth
    return _setup(synth.config_small())


def test_gmm_kernel_matches_oracle(small):
    """Companion kernel vs HTKFlatModels::calcGMMOutput restatement: bit-exact."""
    gnet, gam, onet, oam, feats, _ = small
    x = np.concatenate(feats)[:700]
    g = gam.score_frames(x)
    o = oam.score_frames(x)
    diff = g.view(np.uint32) != o.view(np.uint32)
    frac = diff.mean()
    ulp = np.abs(g.view(np.int32).astype(np.int64) - o.view(np.int32).astype(np.int64)).max()
    print("gmm mismatches: %d of %d (max ulp %d)" % (diff.sum(), diff.size, ulp))
    assert ulp == 0 and frac == 0.0                       # bit for bit: a 1-ulp likelihood can flip a threshold decision


def test_gmm_kernel_generic_dim(built):
    """D != 39 takes the generic (LDS-resident feature) kernel variant."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM
    am = synth.make_models(3, n_gmm=20, n_hmm=8, n_mix=3, D=13, n_tm=2)
    x = np.random.default_rng(0).normal(size=(130, 13)).astype(np.float32)
    g = capi.Models.from_htk(am).score_frames(x)
    o = OracleAM(am).score_frames(x)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))


@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_toy_decode(toy, bi):
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = toy
    kw = BEAMS[bi]
    gd = capi.Decoder(gnet, gam, max_streams=1, **kw)
    od = OracleDecoder(onet, oam, **kw)
    g = gd.decode_batch(feats)[0]
    o = od.decode_certified(feats[0])
    assert o.stats["ties"] == 0
    assert_hyp_matches(g, o, "toy %s" % kw)
    assert bit_exact(g, o), "scores not bit-identical"


@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_small_decode_batch(small, bi):
    """~2k-arc graph with the tee model between words, all utterances in one lock-step batch."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, words = small
    kw = BEAMS[bi]
    # without a (tight) beam every word-end token writes a Path per hub word: tens of
    # millions of records (the reference garbage-collects them); size the arena for it
    big = (1 << 25) if kw.get("main_beam", 0.0) in (0.0, 200.0) and not kw.get("max_hyps") else 0
    gd = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=big, **kw)
    od = OracleDecoder(onet, oam, **kw)
    gs = gd.decode_batch(feats)
    nexact = 0
    for u, x in enumerate(feats):
        o = od.decode_certified(x)
        assert_hyp_matches(gs[u], o, "small utt %d %s" % (u, kw))
        nexact += bit_exact(gs[u], o)
    print("bit-exact utterances: %d / %d" % (nexact, len(feats)))


@pytest.fixture(scope="module")
def mixed(built):
    from juicer_amd import synth
    return _setup(synth.config_mixed())


@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_mixed_topology_decode(mixed, bi):
    """HMMs with 1 to 6 emitting states side by side (3..8 states incl. entry/exit), skip and
    double-entry transitions, the tee model between words: the 8-lane instance layout
    (k_phase_a<8>, 256-byte records) against the oracle."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, words = mixed
    assert gam.max_states == 8
    kw = BEAMS[bi]
    big = (1 << 25) if kw.get("main_beam", 0.0) in (0.0, 200.0) and not kw.get("max_hyps") else 0
    gd = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=big, **kw)
    od = OracleDecoder(onet, oam, **kw)
    gs = gd.decode_batch(feats)
    checked = 0
    for u, x in enumerate(feats):
        o = od.decode_certified(x)
        assert_hyp_matches(gs[u], o, "mixed utt %d %s" % (u, kw))
        checked += 1
    assert checked == len(feats)


def test_more_utts_than_streams(small):
    """decode_batch with n_utts > max_streams runs in waves and re-inits streams."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0)
    gd = capi.Decoder(gnet, gam, max_streams=3, **kw)
    od = OracleDecoder(onet, oam, **kw)
    order = [0, 1, 2, 3, 2, 0, 1]
    gs = gd.decode_batch([feats[i] for i in order])
    for k, i in enumerate(order):
        assert_hyp_matches(gs[k], od.decode_certified(feats[i]), "wave utt %d" % k)
    # and a second call on the same decoder (state fully reset)
    gs2 = gd.decode_batch([feats[3]])
    assert_hyp_matches(gs2[0], od.decode_certified(feats[3]), "second call")


def test_streaming_api(small):
    """IDecoder protocol: init / push in ragged pieces / finish, on stream 1 of 2."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0, max_hyps=200)
    gd = capi.Decoder(gnet, gam, max_streams=2, **kw)
    od = OracleDecoder(onet, oam, **kw)
    for u in (1, 2):
        x = feats[u]
        gd.stream_init(1)
        pos = 0
        for n in (1, 1, 7, 130, 64, 10 ** 6):
            gd.stream_push(1, x[pos:pos + n])
            pos = min(x.shape[0], pos + n)
        g = gd.stream_finish(1)
        assert_hyp_matches(g, od.decode_certified(x), "streaming utt %d" % u)


def test_no_survivor_returns_minus_one(small):
    """Truncated utterance: best path is mid-word at T-1 -> reference returns NULL."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=100.0)
    gd = capi.Decoder(gnet, gam, max_streams=4, **kw)
    od = OracleDecoder(onet, oam, **kw)
    cuts = [feats[0][:5], feats[1][:37], feats[2][:3], feats[3][:1]]
    gs = gd.decode_batch(cuts)
    for k, x in enumerate(cuts):
        o = od.decode_certified(x)
        assert_hyp_matches(gs[k], o, "cut %d" % k)


def test_empty_and_ragged_batch(small):
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    gd = capi.Decoder(gnet, gam, max_streams=4, main_beam=150.0)
    od = OracleDecoder(onet, oam, main_beam=150.0)
    assert gd.decode_batch([]) == []
    batch = [feats[0], feats[1][:0], feats[2][:129], feats[3][:128]]
    gs = gd.decode_batch(batch)
    assert gs[1].n == -1 and gs[1].stats["n_frames"] == 0
    for k in (0, 2, 3):
        assert_hyp_matches(gs[k], od.decode_certified(batch[k]), "ragged %d" % k)


def test_lm_scale_and_insertion_penalty(small):
    """Load-time weight arithmetic (WFSTNetwork.cpp:481-486) flows through to the search."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleDecoder, OracleNet
    gnet, gam, onet, oam, feats, _ = small
    am, net, _, _ = synth.config_small()
    g2 = capi.Network.from_synth(net, lm_scale=7.5, ins_penalty=-3.25)
    o2 = OracleNet(net, lm_scale=7.5, ins_penalty=-3.25)
    gd = capi.Decoder(g2, gam, max_streams=2, main_beam=180.0)
    od = OracleDecoder(o2, oam, main_beam=180.0)
    gs = gd.decode_batch(feats[:2])
    for u in range(2):
        assert_hyp_matches(gs[u], od.decode_certified(feats[u]), "lmscale utt %d" % u)


@pytest.mark.parametrize("arena", ["instance slots", "frontier items", "Path records"])
def test_arena_overflow_is_reported(small, arena):
    """Every device arena overflows into a recoverable JD_ENOMEM that names it (never a fault, never
    a silent truncation); small arenas also shrink the stream's workgroup cluster."""
    from juicer_amd import capi
    gnet, gam, onet, oam, feats, _ = small
    kw = {"instance slots": dict(max_slots=512), "frontier items": dict(max_items=512),
          "Path records": dict(max_paths=256)}[arena]
    gd = capi.Decoder(gnet, gam, max_streams=1, **kw)               # no beam: thousands of instances per frame
    with pytest.raises(capi.JuicerAmdError) as ei:
        gd.decode_batch(feats[:1])
    assert ei.value.code == capi.JD_ENOMEM and arena in str(ei.value), str(ei.value)
    # the same decoder object is usable again (its arenas are wiped) - still too small for this input
    with pytest.raises(capi.JuicerAmdError):
        gd.decode_batch(feats[1:2])
    # and a decoder with sufficient capacity is unaffected
    gd2 = capi.Decoder(gnet, gam, max_streams=1, main_beam=150.0)
    assert gd2.decode_batch(feats[:1])[0].n > 0


def test_small_arenas_still_decode(small):
    """Capacities just large enough: tiny wave segments, one workgroup per stream - same results."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0, max_hyps=200)
    gd = capi.Decoder(gnet, gam, max_streams=2, max_slots=4096, max_items=4096, max_paths=1 << 15, **kw)
    gs = gd.decode_batch(feats[:2])
    assert gd.last_timing()["cluster_wgs"] <= 8
    od = OracleDecoder(onet, oam, **kw)
    for u in range(2):
        assert_hyp_matches(gs[u], od.decode_certified(feats[u]), "small arenas utt %d" % u)


def test_randomised_sweep_slice(built):
    """A seeded slice of tests/manual/fuzz_parity.py: small random problems - both hub shapes (flat:
    thousands of epsilon closure items per word end), with / without the tee model, 5-state and
    mixed-topology HMMs, LM scale / insertion penalty, every pruning combination, 1-6 utterances
    per batch - each against the certified oracle, statistics included."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    beams = [dict(main_beam=150.0), dict(main_beam=200.0), dict(main_beam=100.0, end_beam=70.0, word_beam=50.0),
             dict(main_beam=150.0, max_hyps=200), dict(max_hyps=400), dict(main_beam=120.0, start_beam=100.0, max_hyps=150),
             dict(main_beam=250.0, end_beam=200.0)]
    s0 = 5000
    rng = np.random.default_rng(s0)
    checked = exact = 0
    for case in range(64):
        seed = s0 + case
        hub = "tree" if rng.random() < 0.6 else "flat"
        with_sp = bool(rng.random() < 0.7)
        if rng.random() < 0.35:
            am = synth.make_models_mixed(seed, n_gmm=260, n_hmm=40, n_mix=int(rng.integers(1, 5)), with_tee=with_sp, sep=0.7)
            kind = "mixed"
        else:
            am = synth.make_models(seed, n_gmm=100, n_hmm=45, n_mix=int(rng.integers(1, 6)), n_tm=8, sep=0.6, with_tee=with_sp,
                                   with_skip=bool(rng.random() < 0.5))
            kind = "5state"
        net = synth.make_wfst(seed + 100, am, n_words=int(rng.integers(20, 90)), n_succ=int(rng.integers(2, 9)),
                              with_sp=with_sp, hub=hub, eps_word_frac=float(rng.choice([0.0, 0.05, 0.3])))
        feats = [synth.sample_utterance(seed + 1000 + u, net, am, int(rng.integers(3, 12)))[0] for u in range(int(rng.integers(1, 7)))]
        kw = dict(beams[int(rng.integers(0, len(beams)))])
        lm = float(rng.choice([1.0, 7.5])); pen = float(rng.choice([0.0, -2.0]))
        gd = capi.Decoder(capi.Network.from_synth(net, lm, pen), capi.Models.from_htk(am), max_streams=len(feats), **kw)
        gs = gd.decode_batch(feats)
        od = OracleDecoder(OracleNet(net, lm, pen), OracleAM(am), **kw)
        for u, x in enumerate(feats):
            o = od.decode_certified(x)
            what = "case %d (%s hub=%s sp=%s %s lm=%g pen=%g) utt %d" % (seed, kind, hub, with_sp, kw, lm, pen, u)
            assert_hyp_matches(gs[u], o, what)
            checked += 1
            exact += bit_exact(gs[u], o)
    print("sweep: %d utterances, %d bit-exact incl. scores" % (checked, exact))
    assert checked >= 150 and exact >= checked - 2


def test_histogram_ceiling_is_an_error(built):
    """Histogram::addScore aborts the reference when a score exceeds +201 (Histogram.cpp:78-79)."""
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_toy()
    am.var[:] = 1e-6                     # sharp densities: log-likelihoods above +201 at the mean
    hmm = int(net.ilab[0]) - 1           # first arc out of the initial state
    x = am.mean[am.hmm_gmm[hmm, 1], 0][None, :].repeat(30, axis=0).astype(np.float32)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=1, max_hyps=100)
    with pytest.raises(capi.JuicerAmdError) as ei:
        gd.decode_batch([x])
    assert ei.value.code == capi.JD_EHIST


def test_batch_test_cli(small, tmp_path):
    """DecoderBatchTest counterpart (C++ host over the C ABI): list file in, the reference's output
    formats out (DecoderBatchTest.cpp:339-430) + RT-factor log lines; batched path, MMF models and
    the frame-by-frame IDecoder adapter."""
    import subprocess
    from juicer_amd import build as jbuild, io as jio, synth
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    am, net, _, _ = synth.config_small()
    jio.write_fsm(tmp_path / "g.fsm", net)
    jio.write_jdam(tmp_path / "m.jdam", am)
    jio.write_mmf(tmp_path / "m.mmf", am)
    (tmp_path / "out.syms").write_text("<eps> 0\n" + "".join("W%d %d\n" % (i, i) for i in range(1, net.n_words + 1)))
    lst = tmp_path / "list.txt"
    with open(lst, "w") as f:
        f.write("# comment line\n\n")
        for u, x in enumerate(feats):
            jio.write_jdf(tmp_path / ("u%d.jdf" % u), x)
            f.write("%s\n" % (tmp_path / ("u%d.jdf" % u)))
    od = OracleDecoder(onet, oam, main_beam=150.0, max_hyps=200)
    want = [od.decode_certified(x) for x in feats]

    def run(*extra):
        out = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "g.fsm"), "-inputFName", str(lst),
                              "-mainBeam", "150", "-maxHyps", "200"] + list(extra), capture_output=True, text=True,
                             timeout=240)
        assert out.returncode == 0, out.stderr
        assert out.stderr.count("RT factor") == len(feats) + 1 and out.stderr.count("File: ") == len(feats)
        return out.stdout.splitlines()

    # verbose: words (integers without a symbol table) + word-end frames, three front-ends
    for extra in (["-modelsFName", str(tmp_path / "m.jdam")], ["-htkModelsFName", str(tmp_path / "m.mmf")],
                  ["-htkModelsFName", str(tmp_path / "m.mmf"), "-perFrameAdapter"]):
        lines = run("-outputFormat", "verbose", *extra)
        assert len(lines) == 2 * len(feats)
        for u in range(len(feats)):
            assert lines[2 * u].endswith("u%d.jdf" % u)
            body, times = lines[2 * u + 1].split("[")
            assert [int(w) for w in body.replace("Actual :", "").split()] == (want[u].label[::-1] - 1).tolist()
            t = times.replace("]", "").replace("(", "").replace(")", "").split()
            assert [int(v) for v in t[:-1]] == (want[u].time[::-1] + 1).tolist() and int(t[-1]) == feats[u].shape[0]
    mm = ["-htkModelsFName", str(tmp_path / "m.mmf"), "-outSymsFName", str(tmp_path / "out.syms")]
    # ref / trans with word strings from the output symbol table
    lines = run("-outputFormat", "ref", *mm)
    for u in range(len(feats)):
        assert lines[u].split() == ["W%d" % l for l in want[u].label[::-1]]
    # -devices N: the C++ multi-GPU path (jd_multi_*: one decoder + thread per device, one RCCL
    # all-gather of the 1-best records).  This box has one GPU, so N = 1 - the communicator, the
    # packing, the collective and the unpacking all run; the xmlf lines carry times and scores.
    assert run("-outputFormat", "xmlf", "-devices", "1", *mm) == run("-outputFormat", "xmlf", *mm)
    # -residentSlots N: the list through N one-workgroup slots of the search kernel that stays (jd_dec_set_pipeline, JD_FLOW_RESIDENT:
    # a slot takes the next utterance the moment its own is through) - more utterances than slots, the same output
    assert run("-outputFormat", "xmlf", "-residentSlots", "3", "-batch", "3", *mm) == run("-outputFormat", "xmlf", *mm)
    lines = run("-outputFormat", "trans", *mm)
    assert all(lines[u].endswith("(trans-%d)" % want[u].n) for u in range(len(feats)))
    # mlf / xmlf
    lines = run("-outputFormat", "mlf", *mm)
    assert lines[0] == "#!MLF!#" and lines[1] == '"*/u0.rec"' and lines.count(".") == len(feats)
    lines = run("-outputFormat", "xmlf", *mm)
    k = lines.index('"*/u1.rec"') + 1
    w = want[1]
    ends = w.time[::-1]
    for j in range(w.n):
        st, et, name, sc = lines[k + j].split()
        s0 = 0 if j == 0 else int(ends[j - 1])
        e0 = int(ends[j])                   # HTK units; the reference adds one frame only to non-zero times
        assert int(st) == (0 if s0 == 0 else (s0 + 1) * 100000) and int(et) == (0 if e0 == 0 else (e0 + 1) * 100000)
        assert name == "W%d" % w.label[::-1][j]
        dac = w.ac[::-1][j] - (w.ac[::-1][j - 1] if j else 0.0)
        dlm = w.lm[::-1][j] - (w.lm[::-1][j - 1] if j else 0.0)
        assert abs(float(sc) - (dac + dlm)) <= 1e-3 * max(1.0, abs(dac + dlm))
    # -outputFName: results go to the named file (stdout stays empty), "stderr" to the log stream
    ref_lines = run("-outputFormat", "ref", *mm)
    out = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "g.fsm"), "-inputFName", str(lst), "-mainBeam", "150",
                          "-maxHyps", "200", "-outputFormat", "ref", "-outputFName", str(tmp_path / "res.txt")] + mm,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and out.stdout == ""
    assert (tmp_path / "res.txt").read_text().splitlines() == ref_lines
    # PartialTraceInterval (WFSTDecoderLite.cpp:116-119): the adapter traces on the reference's schedule
    # and the log gets recognitionFinish's line (:252-256) - the frames of the whole best path
    import os
    out = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "g.fsm"), "-inputFName", str(lst), "-mainBeam", "150",
                          "-maxHyps", "200", "-outputFormat", "ref", "-perFrameAdapter"] + mm,
                         capture_output=True, text=True, timeout=240, env=dict(os.environ, PartialTraceInterval="50"))
    assert out.returncode == 0 and out.stdout.splitlines() == ref_lines
    got = [l for l in out.stderr.splitlines() if l.startswith("Partial paths recovered at frames:")]
    assert len(got) == len(feats)
    for u in range(len(feats)):
        assert [int(v) for v in got[u].split(":")[1].split()] == want[u].time[::-1].tolist()


def test_path_garbage_collection(small):
    """A Path arena far smaller than the number of records an utterance writes: decoding only
    succeeds because unreachable records are collected between frames (collectPaths,
    WFSTDecoderLite.cpp:699-747), and the results do not change."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0, max_hyps=200)
    od = OracleDecoder(onet, oam, **kw)
    big = capi.Decoder(gnet, gam, max_streams=len(feats), **kw).decode_batch(feats)
    need = max(h.stats["tot_paths"] for h in big)
    cap = 1 << 14
    assert need > 2 * cap, "test is vacuous: %d records fit" % need
    gd = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=cap, **kw)
    gs = gd.decode_batch(feats)
    for u, x in enumerate(feats):
        assert_hyp_matches(gs[u], od.decode_certified(x), "gc utt %d" % u)
        assert np.array_equal(gs[u].score.view(np.uint32), big[u].score.view(np.uint32))
    # streaming API takes the same path
    gd.stream_init(0)
    for pos in range(0, feats[0].shape[0], 50):
        gd.stream_push(0, feats[0][pos:pos + 50])
    assert_hyp_matches(gd.stream_finish(0), od.decode_certified(feats[0]), "gc streaming")


def test_binary_caches_and_htk_feature_files(small, tmp_path):
    """The reference's other on-disk formats drive the same decode: "<fsm>.bin" (JWNT) and
    "<mmf>.bin" (JMBI) caches are preferred when present (juicer.cpp:854-866, :778-784),
    -writeBinaryFiles creates them, utterances come from HTK parameter files; the result lines
    equal the text-FSM / MMF / .jdf run's and the oracle's."""
    import subprocess
    from juicer_amd import build as jbuild, capi, io as jio, synth
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    am, net, _, _ = synth.config_small()
    d_txt, d_bin = tmp_path / "txt", tmp_path / "bin"
    d_txt.mkdir(); d_bin.mkdir()
    for d in (d_txt, d_bin):
        jio.write_fsm(d / "g.fsm", net)
        jio.write_mmf(d / "m.mmf", am)
    lists = {}
    for kind, d, writer in (("jdf", d_txt, jio.write_jdf), ("htk", d_bin, jio.write_htk)):
        lists[kind] = d / "list.txt"
        with open(lists[kind], "w") as f:
            for u, x in enumerate(feats):
                writer(d / ("u%d.%s" % (u, kind)), x)
                f.write("%s\n" % (d / ("u%d.%s" % (u, kind))))

    def run(d, lst, *extra):
        out = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(d / "g.fsm"), "-htkModelsFName", str(d / "m.mmf"),
                              "-inputFName", str(lst), "-mainBeam", "150", "-outputFormat", "verbose"] + list(extra),
                             capture_output=True, text=True, timeout=240)
        assert out.returncode == 0, out.stderr
        return [l for l in out.stdout.splitlines() if "Actual" in l], out.stderr

    base, err = run(d_txt, lists["jdf"])
    assert "pre-existing" not in err
    first, err = run(d_bin, lists["htk"], "-writeBinaryFiles")          # text load, caches written
    assert "pre-existing" not in err and (d_bin / "g.fsm.bin").exists() and (d_bin / "m.mmf.bin").exists()
    again, err = run(d_bin, lists["htk"])                               # now served from the caches
    assert err.count("pre-existing binary file") == 2
    assert base == first == again and len(base) == len(feats)
    od = OracleDecoder(onet, oam, main_beam=150.0)
    for u, x in enumerate(feats):
        o = od.decode_certified(x)
        body = base[u].split("[")[0].replace("Actual :", "").split()
        assert [int(w) for w in body] == (o.label[::-1] - 1).tolist()
    # HTK _C parameter files (16-bit samples with per-component scale and offset, what HTK corpora usually hold): the
    # harness decodes what the file says - the words the library gives on the de-quantised vectors
    d_c = tmp_path / "c"
    d_c.mkdir()
    deq = []
    with open(d_c / "list.txt", "w") as f:
        for u, x in enumerate(feats):
            deq.append(jio.write_htk_compressed(d_c / ("u%d.htk" % u), x))
            f.write("%s\n" % (d_c / ("u%d.htk" % u)))
    comp, _ = run(d_txt, d_c / "list.txt")
    want = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=len(feats)).decode_batch(deq)
    assert len(comp) == len(feats)
    for u, h in enumerate(want):
        body = comp[u].split("[")[0].replace("Actual :", "").split()
        assert h.n > 0 and [int(w) for w in body] == (h.label[::-1] - 1).tolist()
        assert_hyp_matches(h, od.decode_certified(deq[u]), "compressed utt %d" % u)
    # library level: binary-loaded handles decode bit-identically to the originals
    capi.Network.from_synth(net).save_jwnt(str(tmp_path / "n.bin"))
    capi.Models.from_htk(am).save_jmbi(str(tmp_path / "a.bin"))
    kw = dict(main_beam=150.0, max_streams=len(feats))
    a = capi.Decoder(gnet, gam, **kw).decode_batch(feats)
    b = capi.Decoder(capi.Network.from_jwnt_file(str(tmp_path / "n.bin")),
                     capi.Models.from_jmbi_file(str(tmp_path / "a.bin")), **kw).decode_batch(feats)
    for x, y in zip(a, b):
        assert x.n == y.n and np.array_equal(x.label, y.label) and np.array_equal(x.time, y.time)
        assert np.array_equal(x.score.view(np.uint32), y.score.view(np.uint32))


def _align_counts(actual, expected, ci=7, cd=7, cs=10):
    """Independent minimum-cost alignment (same tie order as the harness: match/sub, then ins, then del)."""
    A, E = len(actual), len(expected)
    cost = [[0] * (E + 1) for _ in range(A + 1)]
    op = [[0] * (E + 1) for _ in range(A + 1)]
    for i in range(A + 1):
        for j in range(E + 1):
            if i == 0 and j == 0:
                continue
            best, o = None, 0
            if i and j:
                eq = actual[i - 1] == expected[j - 1]
                best, o = cost[i - 1][j - 1] + (0 if eq else cs), (0 if eq else 1)
            if i and (best is None or cost[i - 1][j] + ci < best):
                best, o = cost[i - 1][j] + ci, 2
            if j and (best is None or cost[i][j - 1] + cd < best):
                best, o = cost[i][j - 1] + cd, 3
            cost[i][j], op[i][j] = best, o
    i, j, ins, dele, sub = A, E, 0, 0, 0
    while i or j:
        o = op[i][j]
        if o == 2:
            ins += 1; i -= 1
        elif o == 3:
            dele += 1; j -= 1
        else:
            sub += o; i -= 1; j -= 1
    return ins, dele, sub


def test_batch_test_expected_results_and_error_totals(small, tmp_path):
    """-refFName (DecoderBatchTest.cpp:804-939): reference transcripts as an MLF or one line per
    file, 'Expected :' lines in verbose output, and the closing insertion/deletion/substitution
    totals at HTK costs 7/7/10 (:145-201, :251)."""
    import subprocess
    from juicer_amd import build as jbuild, io as jio, synth
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    am, net, _, words = synth.config_small()
    jio.write_fsm(tmp_path / "g.fsm", net)
    jio.write_mmf(tmp_path / "m.mmf", am)
    (tmp_path / "out.syms").write_text("<eps> 0\n" + "".join("W%d %d\n" % (i, i) for i in range(1, net.n_words + 1)))
    lst = tmp_path / "list.txt"
    with open(lst, "w") as f:
        for u, x in enumerate(feats):
            jio.write_htk(tmp_path / ("utt%02d.htk" % u), x)
            f.write("%s\n" % (tmp_path / ("utt%02d.htk" % u)))
    # the truth: generating words, with one word changed, one dropped and an out-of-vocabulary one
    truth = [["W%d" % w for w in ws] for ws in words]
    truth[0][1] = "W1" if truth[0][1] != "W1" else "W2"
    truth[1] = truth[1][:-1]
    truth[2] = truth[2] + ["NOTAWORD"]
    (tmp_path / "ref.txt").write_text("".join(" ".join(t) + "\n" for t in truth))
    with open(tmp_path / "ref.mlf", "w") as f:
        f.write("#!MLF!#\n")
        for u in reversed(range(len(feats))):                      # MLF entries are matched by name, any order
            f.write('"*/utt%02d.lab"\n' % u + "".join(w + "\n" for w in truth[u]) + ".\n")
    od = OracleDecoder(onet, oam, main_beam=150.0)
    hyp = [(od.decode_certified(x).label[::-1] - 1).tolist() for x in feats]
    ids = lambda t, keep_oov: [int(w[1:]) - 1 if w[1:].isdigit() else -1 for w in t if keep_oov or w[1:].isdigit()]

    def run(ref):
        out = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "g.fsm"), "-htkModelsFName", str(tmp_path / "m.mmf"),
                              "-outSymsFName", str(tmp_path / "out.syms"), "-inputFName", str(lst), "-mainBeam", "150",
                              "-outputFormat", "verbose", "-refFName", str(ref)], capture_output=True, text=True, timeout=240)
        assert out.returncode == 0, out.stderr
        return out.stdout.splitlines(), out.stderr

    for ref, keep_oov in ((tmp_path / "ref.txt", True), (tmp_path / "ref.mlf", False)):
        lines, err = run(ref)
        exp_lines = [l for l in lines if "Expected :" in l]
        assert len(exp_lines) == len(feats)
        for u, l in enumerate(exp_lines):
            want = [w if w[1:].isdigit() else "<OOV>" for w in truth[u] if keep_oov or w[1:].isdigit()]
            assert l.split(":", 1)[1].split() == want
        if not keep_oov:
            assert "Unknown word in ground truth" in err            # MLF: dropped with a warning (:876-880)
        else:
            assert any("result word NOTAWORD not in vocab" in l for l in lines)   # ref format: kept as <OOV> (:905-906)
        tot = [0, 0, 0]
        n_ref = 0
        for u in range(len(feats)):
            e = ids(truth[u], keep_oov)
            c = _align_counts(hyp[u], e)
            tot = [a + b for a, b in zip(tot, c)]
            n_ref += len(e)
        assert sum(tot) >= 2
        line = [l for l in lines if l.startswith("total ")][0]
        assert line.startswith("total %d: insert %d / delete %d / subst %d / n_seq %d" % (n_ref, tot[0], tot[1], tot[2], len(feats)))
        assert any(l.startswith("Real-time (RT) factor") for l in lines)
    # a transcript missing for one input file is an error before any decoding (:925-931)
    (tmp_path / "short.mlf").write_text('#!MLF!#\n"*/utt00.lab"\nW1\n.\n')
    out = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "g.fsm"), "-htkModelsFName", str(tmp_path / "m.mmf"),
                          "-outSymsFName", str(tmp_path / "out.syms"), "-inputFName", str(lst), "-refFName", str(tmp_path / "short.mlf")],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "ref transcription not found" in out.stderr


@pytest.mark.parametrize("case", ["toy", "small", "mixed"])
def test_hip_path_matches_committed_golden_vectors(built, case):
    """The HIP path against tests/golden/oracle_golden.json directly (no oracle at run time):
    labels and word-end frames identical, scores within 1e-4 relative (bit-exact counted),
    the reference's statistics identical on tie-free utterances."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    from helpers import STAT_KEYS, rel_close
    from juicer_amd import capi
    g = json.load(open(os.path.join(here, "golden", "oracle_golden.json")))[case]
    am, net, feats, _ = make_golden.CASES[case][0]()
    assert make_golden.input_digest(am, net, feats) == g["input_sha256"], "synthetic generator changed"
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    unhex = lambda hs: np.frombuffer(bytes.fromhex("".join(hs)), np.float32) if hs else np.zeros(0, np.float32)
    exact = total = 0
    for run in g["runs"]:
        hyps = capi.Decoder(gnet, gam, max_streams=len(feats), **run["beams"]).decode_batch(feats)
        for u, (h, want) in enumerate(zip(hyps, run["utts"])):
            what = "%s %s utt %d" % (case, run["beams"], u)
            assert h.n == want["n"], what
            assert h.label.tolist() == want["label"] and h.time.tolist() == want["time"], what
            for f in ("score", "ac", "lm"):
                w = unhex(want[f + "_hex"])
                assert rel_close(getattr(h, f), w), what + " " + f
                exact += int(np.array_equal(getattr(h, f).view(np.uint32), w.view(np.uint32))); total += 1
            tot = unhex(want["tot"])
            assert rel_close([h.tot_score, h.tot_ac, h.tot_lm], tot), what
            assert want["stats"]["ties"] == 0, what          # golden vectors are order independent
            for k in STAT_KEYS:
                assert h.stats[k] == want["stats"][k], "%s stat %s" % (what, k)
    print("bit-exact score arrays: %d / %d" % (exact, total))
    assert exact >= total * 0.9


def test_ragged_mixture_counts(built):
    """GMMs with different numbers of mixture components (the flat arrays are padded to the
    largest, HTKFlatModels.cpp:113,145): scores bit-exact, decode identical to the oracle."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small()
    rng = np.random.default_rng(5)
    am.n_mix = rng.integers(1, am.max_mix + 1, size=am.n_gmm).astype(np.int32)
    for g in range(am.n_gmm):                                   # weights of the kept components sum to one
        k = int(am.n_mix[g])
        am.weight[g, k:] = 0.0
        am.weight[g, :k] = am.weight[g, :k] / am.weight[g, :k].sum() if k > 1 else 1.0
    gam, oam = capi.Models.from_htk(am), OracleAM(am)
    for a, b in zip(gam.flat(), oam.flat()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    x = np.concatenate(feats)[:300]
    assert np.array_equal(gam.score_frames(x).view(np.uint32), oam.score_frames(x).view(np.uint32))
    kw = dict(main_beam=150.0)
    gs = capi.Decoder(capi.Network.from_synth(net), gam, max_streams=len(feats), **kw).decode_batch(feats)
    od = OracleDecoder(OracleNet(net), oam, **kw)
    for u, f in enumerate(feats):
        o = od.decode_certified(f)
        assert_hyp_matches(gs[u], o, "ragged utt %d" % u)


def test_histogram_too_wide_is_refused(small):
    """The histogram spans [-(beam+800)-1, 201] in unit bins (WFSTDecoderLite.cpp:76-82); the
    kernel keeps it in LDS, so a beam that needs more bins than fit is an error at creation."""
    from juicer_amd import capi
    gnet, gam = small[0], small[1]
    capi.Decoder(gnet, gam, main_beam=1000.0, max_hyps=100, max_streams=1)
    with pytest.raises(capi.JuicerAmdError) as e:
        capi.Decoder(gnet, gam, main_beam=1200.0, max_hyps=100, max_streams=1)
    assert "histogram" in str(e.value).lower()


@pytest.fixture(scope="module")
def small_tree(built):
    from juicer_amd import synth
    return _setup(synth.config_small(hub="tree"))


@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_tree_hub_closure_rounds(small_tree, bi):
    """A determinised-C.L.G-shaped graph (prefix-tree back-off state, tee model between words,
    eps:word arcs inside the tree): every frame runs epsilon / tee closure rounds behind the exit
    tokens (one cluster barrier per non-empty round).  Same results as the oracle, and the same
    whatever the cluster size (1 workgroup: no inter-workgroup traffic at all; 7: ragged)."""
    import os
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, words = small_tree
    kw = BEAMS[bi]
    big = (1 << 25) if kw.get("main_beam", 0.0) in (0.0, 200.0) and not kw.get("max_hyps") else 0
    gd = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=big, **kw)
    gs = gd.decode_batch(feats)
    od = OracleDecoder(onet, oam, **kw)
    for u, x in enumerate(feats):
        o = od.decode_certified(x)
        assert_hyp_matches(gs[u], o, "tree utt %d %s" % (u, kw))
    for cw in ("1", "7"):
        os.environ["JD_CW"] = cw                                  # development knob: workgroups per stream cluster
        try:
            gd2 = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=big, **kw)
            g2 = gd2.decode_batch(feats)
            assert gd2.last_timing()["cluster_wgs"] == int(cw)
        finally:
            del os.environ["JD_CW"]
        for a, b in zip(gs, g2):
            assert a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
            assert np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32))
            for k in ("tot_active_models", "tot_proc_emit_hyps", "tot_proc_end_hyps", "tot_insts_in"):
                assert a.stats[k] == b.stats[k]


@pytest.mark.parametrize("cfg_name", ["small", "small_tree", "mixed"])
def test_partial_decoding_traces(cfg_name, request):
    """PARTIAL_DECODING, tracePartialPath / traceWinningPaths (WFSTDecoderLite.cpp:824-890): a trace
    asked for after the same frames finds the same converged Path record on the GPU as the restated
    reference, and the accumulated partialPaths lists are identical throughout - also with a Path
    arena so small that records are collected (and renumbered) between the traces."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = request.getfixturevalue(cfg_name)
    kw = dict(main_beam=150.0, max_hyps=200) if cfg_name != "mixed" else dict(main_beam=200.0)
    od = OracleDecoder(onet, oam, **kw)
    n_found = n_traces = 0
    for gd in (capi.Decoder(gnet, gam, max_streams=2, **kw), capi.Decoder(gnet, gam, max_streams=2, max_paths=1 << 12, **kw)):
        for u in range(min(3, len(feats))):
            x = feats[u]
            at = list(range(7, x.shape[0], 23))
            snaps, _ = od.decode_partial(x, interval=0, trace_at=at)
            gd.stream_init(1)
            pos = 0
            for f in at:
                gd.stream_push(1, x[pos:f + 1])
                pos = f + 1
                found, lst = gd.stream_partial(1, trace_now=True)
                assert (found, lst) == snaps[f], "%s utt %d frame %d: %r vs %r" % (cfg_name, u, f, (found, lst[-3:]), (snaps[f][0], snaps[f][1][-3:]))
                n_found += found
                n_traces += 1
            gd.stream_push(1, x[pos:])
            g = gd.stream_finish(1)
            assert_hyp_matches(g, od.decode_certified(x), "partial %s utt %d" % (cfg_name, u))
            # every traced record is part of the final result (the prefix that could no longer change)
            final = list(zip(g.label.tolist()[::-1], g.time.tolist()[::-1]))
            assert final[:len(lst)] == lst
    assert n_found >= 4 and n_found < n_traces, "vacuous: %d of %d traces found a record" % (n_found, n_traces)


def test_partial_decoding_schedule(small_tree):
    """setPartialDecodeOptions(interval): traces ride on the path collections (:362-368) whatever the push sizes
    are, and finish() completes the list from the best token (:245-251).  On this graph (lexicon-tree hub: few Path
    objects per frame) only collectPaths' frame rule fires in the reference - asserted on the oracle, which models
    both triggers - and the schedule here is the reference's exactly."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small_tree
    kw = dict(main_beam=150.0)
    od = OracleDecoder(onet, oam, **kw)
    gd = capi.Decoder(gnet, gam, max_streams=1, **kw)
    x = np.concatenate([feats[0], feats[1], feats[2], feats[0]])       # long enough for several collections
    assert x.shape[0] > 450
    for interval, step in ((1, 37), (150, 64), (150, 1000)):
        snaps, final = od.decode_partial(x, interval=interval)
        assert len(snaps) >= 2
        assert od.collect_frames == list(range(100, x.shape[0], 101))  # the frame rule alone: 100, 201, 302, ...
        gd.set_partial_interval(interval)
        gd.stream_init(0)
        seen = {}
        for pos in range(0, x.shape[0], step):
            gd.stream_push(0, x[pos:pos + step])
            last = min(x.shape[0], pos + step) - 1
            # the list as it stands after this push = the reference's after its last trace up to here
            due = [f for f in sorted(snaps) if f <= last]
            _, lst = gd.stream_partial(0)
            assert lst == (snaps[due[-1]][1] if due else []), "interval %d, after frame %d" % (interval, last)
            done = [f for f in od.collect_frames if f <= last]
            assert gd.stream_collect_info(0) == (len(done), done[-1] if done else -1)
            seen[last] = lst
        g = gd.stream_finish(0)
        _, lst = gd.stream_partial(0)
        assert lst == final == list(zip(g.label.tolist()[::-1], g.time.tolist()[::-1]))
    gd.set_partial_interval(0)


def test_partial_decoding_count_rule(built):
    """collectPaths' other trigger (:360-362): nPath / nPathNew > 12 with nPath > 10000.  A back-off state that fans out
    into one eps:word arc per word of a 400-word vocabulary makes hundreds of Path objects per frame: collections
    (and the traces that ride on them) come every few dozen frames, long before the frame rule's 101.  The reference
    counts a Path for every labelled propagateToken call, winner or not; with the end and word beams off the decoder
    counts the same objects without writing them (the closure's share is a static property of the state), and its
    collections keep count of what the reference's would keep: nPath and nPathNew equal the oracle's behind every
    frame, the collections run after the same frames, every traced list is a prefix of the final result, which is
    the oracle's."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small(seed=31, n_utts=2, n_words=400, n_succ=5, n_gmm=120, n_hmm=45, hub="flat")
    x = np.concatenate(feats)[:330]
    kw = dict(main_beam=250.0)
    od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
    snaps, final = od.decode_partial(x, interval=1)
    assert od.collect_frames[0] < 100 and len(od.collect_frames) >= 4
    # HMMs of 1..6 emitting states, a lexicon tree below the back-off state (closures several arcs deep, through tee models)
    am2, net2, feats2, _ = synth.config_mixed(seed=12, n_utts=3, n_words=300, n_succ=6)
    x2 = np.concatenate(feats2)[:260]
    od_2 = OracleDecoder(OracleNet(net2), OracleAM(am2), **kw)
    snaps2, final2 = od_2.decode_partial(x2, interval=1)
    cases = [(am, net, x, od, snaps, final, p, a) for p, a in ((1, 0), (16, 0), (330, 0), (7, 1 << 15))]   # (a small arena: collections of its own in between)
    cases += [(am2, net2, x2, od_2, snaps2, final2, p, a) for p, a in ((1, 0), (23, 1 << 15))]
    # a small vocabulary: the rule fires now and then (after frames 9, 11, 12, ..., 18, 20, 23, ...), not after every frame
    am3, net3, feats3, _ = synth.config_small(seed=5, n_utts=4, n_words=60)
    x3 = np.concatenate(feats3)[:600]
    od_3 = OracleDecoder(OracleNet(net3), OracleAM(am3), **kw)
    snaps3, final3 = od_3.decode_partial(x3, interval=1)
    gaps = np.diff(od_3.collect_frames)
    assert gaps.min() == 1 and gaps.max() >= 3
    cases += [(am3, net3, x3, od_3, snaps3, final3, p, a) for p, a in ((1, 0), (37, 0))]
    for am, net, x, od, snaps, final, push, arena in cases:
        args = dict(max_paths=arena) if arena else {}
        gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=1, **kw, **args)
        gd.set_partial_interval(1)
        gd.stream_init(0)
        prev, colls = [], []
        for pos in range(0, x.shape[0], push):
            gd.stream_push(0, x[pos:pos + push])
            at = min(x.shape[0], pos + push) - 1
            n, last = gd.stream_collect_info(0)
            colls.append((n, last))
            done = [f for f in od.collect_frames if f <= at]
            assert (n, last) == (len(done), done[-1] if done else -1), (push, at, n, last, done)
            npath, nnew, exact = gd.stream_path_counts(0)
            assert exact and (npath, nnew) == od.path_counts[at], (push, at, npath, nnew, od.path_counts[at])
            _, lst = gd.stream_partial(0)
            assert lst[:len(prev)] == prev and all(t <= at for _, t in lst)
            if at in snaps and push == 1:
                assert lst == snaps[at][1]
            prev = lst
        g = gd.stream_finish(0)
        _, lst = gd.stream_partial(0)
        assert lst == final == list(zip(g.label.tolist()[::-1], g.time.tolist()[::-1])) and lst[:len(prev)] == prev
        assert_hyp_matches(g, od.decode_certified(x), "count rule")
        gd.set_partial_interval(0)
        gd.close()
    am, net, x = cases[0][:3]
    # with an end beam the closure is pruned by score: the rule runs on this build's own records (documented approximation)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=1, end_beam=200.0, **kw)
    gd.set_partial_interval(1)
    gd.stream_init(0)
    gd.stream_push(0, x[:120])
    assert gd.stream_path_counts(0)[2] is False
    od2 = OracleDecoder(OracleNet(net), OracleAM(am), end_beam=200.0, **kw)
    od2.decode_partial(x[:120], interval=1)
    assert gd.stream_collect_info(0)[0] <= len(od2.collect_frames)      # never more often than the reference
    gd.close()


@pytest.mark.parametrize("cfg_name", ["small", "small_tree", "mixed"])
def test_decoders_own_state_numbering_and_layouts(cfg_name, request, monkeypatch, capfd):
    """Where a state's words sit and what number it has are the decoder's own business (jd_dec_create: the numbering along the
    chains behind a state's arcs, the per-state words joint or split): forced on and off on graphs with tee models, epsilon
    arcs inside a lexicon tree and HMMs of 1-6 emitting states, every combination gives the oracle's hypotheses, the
    reference's statistics and the same bits as every other - through a whole-batch decode, through the stream calls with a
    trace after every push (the per-state Path counts of the count rule are permuted with the states), and with final
    weights on states that moved."""
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = request.getfixturevalue(cfg_name)
    kw = dict(main_beam=150.0, max_hyps=200) if cfg_name != "mixed" else dict(main_beam=200.0)
    od = OracleDecoder(onet, oam, **kw)
    ora = [od.decode_certified(x) for x in feats]
    x = np.concatenate([feats[0], feats[1]])
    monkeypatch.setenv("JD_DEV", "1")
    monkeypatch.setenv("JD_VERBOSE", "1")
    base = None
    said_own = 0
    for renumber in ("0", "1"):
        for split in ("0", "1", "2"):
            monkeypatch.setenv("JD_RENUMBER", renumber)
            monkeypatch.setenv("JD_SREC_SPLIT", split)
            capfd.readouterr()
            gd = capi.Decoder(gnet, gam, max_streams=len(feats), **kw)
            err = capfd.readouterr().err
            said_own += "state numbers: the decoder's own" in err
            assert ("per-state words: " + ("joint", "split (bids | arrival keys)", "split (bids | arrival keys of either")[int(split)]) in err, err[-600:]
            gs = gd.decode_batch(feats)
            for u, g in enumerate(gs):
                assert_hyp_matches(g, ora[u], "%s renumber %s split %s utt %d" % (cfg_name, renumber, split, u))
            gd.set_partial_interval(1)
            gd.stream_init(0)
            trace = []
            for pos in range(0, x.shape[0], 29):
                gd.stream_push(0, x[pos:pos + 29])
                pc = gd.stream_path_counts(0)
                assert pc[2], "the count rule's exact counts are expected here (no end / word beam)"
                trace.append((gd.stream_partial(0), gd.stream_collect_info(0), pc))
            h = gd.stream_finish(0)
            trace.append(gd.stream_partial(0))
            gd.set_partial_interval(0)
            # (tot_arcs_visited / tot_paths are left out: how many arcs a state's losers walked before the winner arrived is a matter of timing)
            got = dict(hyps=[(g.n, g.label.tobytes(), g.time.tobytes(), g.score.tobytes()) for g in gs],
                       stats=[[g.stats[k] for k in ("n_frames", "tot_active_emit_hyps", "tot_active_end_hyps", "tot_active_models", "tot_proc_emit_hyps", "tot_proc_end_hyps", "tot_insts_in")] for g in gs],
                       stream=(h.n, h.label.tobytes(), h.time.tobytes(), h.score.tobytes()), trace=trace)
            if base is None: base = got
            for k in got:
                assert got[k] == base[k], "%s: renumber %s split %s: %s differ from the network's numbers in joint records" % (cfg_name, renumber, split, k)
            gd.close()
    assert said_own == 3, "the decoder's own numbering never took effect on %s" % cfg_name


@pytest.mark.parametrize("cfg_name", ["small", "small_tree", "mixed"])
def test_exit_tokens_that_recombine_with_nobody(cfg_name, request, monkeypatch, capfd):
    """REC_SOLE (jd_search.h): the exit tokens of an arc that is the only arc into its destination place no bid for the state and
    read, win and reset none - the reference compares the tokens that arrive at a state (WFSTDecoderLite.cpp:560-582), and here
    there is one.  Both kernels, with the shortcut and without it (JD_NO_SOLE): the oracle's hypotheses, the reference's
    statistics, identical bits; the fixtures have arcs of both kinds, and tee arcs never take the shortcut (their pass-through
    arrives beside their exit token)."""
    import re
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = request.getfixturevalue(cfg_name)
    kw = dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0) if cfg_name != "mixed" else dict(main_beam=200.0)
    od = OracleDecoder(onet, oam, **kw)
    ora = [od.decode_certified(x) for x in feats]
    monkeypatch.setenv("JD_DEV", "1")
    monkeypatch.setenv("JD_VERBOSE", "1")
    base = None
    for slot in (False, True):
        if slot and gam.max_states > 8: continue
        monkeypatch.setenv("JD_SLOT_BATCH", "1" if slot else "0")
        monkeypatch.setenv("JD_CW", "1" if slot else "64")
        for no_sole in (False, True):
            if no_sole: monkeypatch.setenv("JD_NO_SOLE", "1")
            else: monkeypatch.delenv("JD_NO_SOLE", raising=False)
            capfd.readouterr()
            gd = capi.Decoder(gnet, gam, max_streams=len(feats), **kw)
            m = re.search(r"recombine with nobody: those of (\d+) of (\d+) model arcs", capfd.readouterr().err)
            assert m, "jd_dec_create did not say how many arcs take the shortcut"
            n_sole, n_model = int(m.group(1)), int(m.group(2))
            assert (n_sole == 0) if no_sole else (0 < n_sole < n_model), (n_sole, n_model)
            gs = gd.decode_batch(feats)
            for u, g in enumerate(gs):
                assert_hyp_matches(g, ora[u], "%s slot %d no_sole %d utt %d" % (cfg_name, slot, no_sole, u))
            got = [(g.n, g.label.tobytes(), g.time.tobytes(), g.score.tobytes(), [g.stats[k] for k in ("tot_active_models", "tot_proc_emit_hyps", "tot_proc_end_hyps", "tot_insts_in")]) for g in gs]
            if base is None: base = got
            assert got == base, "%s: slot kernel %d, shortcut off %d" % (cfg_name, slot, no_sole)
            gd.close()


def test_max_alloc_models(small):
    """setMaxAllocModels (WFSTDecoderLite.cpp:807-820) is a SOFT limit in the reference (it decides whether cached
    NetInst objects are dropped between utterances, :164-169): whatever its value - percentage / MB / count form,
    generous or tiny - results do not depend on it and no decode fails because of it."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    n_arcs = synth.config_small()[1].n_arcs
    kw = dict(main_beam=150.0, max_hyps=200)
    od = OracleDecoder(onet, oam, **kw)
    want = od.decode_certified(feats[0])
    for v in (100, 20000):                                              # 100 MB, 20000 instances
        gd = capi.Decoder(gnet, gam, max_streams=1, **kw)
        gd.set_max_alloc_models(v)
        assert_hyp_matches(gd.decode_batch(feats[:1])[0], want, "MaxAllocModels %d" % v)
        with pytest.raises(capi.JuicerAmdError):                        # only before the arenas exist
            gd.set_max_alloc_models(v)
    # the reference's default (MaxAllocModels = 10: 10 % of the transitions, WFSTDecoderLite.cpp:73) with NO beam - an
    # instance on most arcs, far more than the limit: the reference decodes that, so does this
    od0 = OracleDecoder(onet, oam)
    want0 = od0.decode_certified(feats[0])
    for v in (10, 1):
        gd = capi.Decoder(gnet, gam, max_streams=1)
        gd.set_max_alloc_models(v)
        assert_hyp_matches(gd.decode_batch(feats[:1])[0], want0, "MaxAllocModels %d, no beam" % v)
    assert n_arcs > 0
    with pytest.raises(capi.JuicerAmdError):
        capi.Decoder(gnet, gam, max_streams=1).set_max_alloc_models(0)  # assert(maxAllocModels_ > 0) :808


def test_stress_many_streams_repeated(built):
    """Guards the inter-workgroup protocol of the persistent kernel (write-through stores, agent-scope
    loads, counter barriers across non-coherent XCD L2s): 64 streams of short utterances, decoded over
    and over with different cluster shapes (1, 2, 3 and 4+ workgroups per stream, uniform and weighted) -
    every run must reproduce the oracle's result for every utterance, scores bit for bit."""
    import os
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small(seed=21, n_utts=64, utt_words=(1, 6), hub="tree")
    kw = dict(main_beam=150.0, max_hyps=300)
    od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
    want = [od.decode_certified(x) for x in feats]
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    runs = 0
    for env in ({}, {"JD_CW": "1"}, {"JD_CW": "2"}, {"JD_CW": "3"}, {"JD_WEIGHTED": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            gd = capi.Decoder(gnet, gam, max_streams=64, **kw)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        for rep in range(6):
            order = np.random.default_rng(rep).permutation(64) if rep else np.arange(64)
            gs = gd.decode_batch([feats[i] for i in order])
            for j, i in enumerate(order):
                assert_hyp_matches(gs[j], want[i], "stress %r rep %d utt %d" % (env, rep, i))
                assert bit_exact(gs[j], want[i]), "stress %r rep %d utt %d: scores" % (env, rep, i)
            runs += 1
    assert runs == 30


def test_states_with_many_arcs_are_sliced(built):
    """A back-off state with 700 eps:word arcs and word histories with 300 successors: phase X walks
    such states in slices of 256 arcs handed to the next round - same results, same statistics."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small(seed=31, n_utts=3, n_words=700, n_succ=300, hub="flat", utt_words=(3, 6))
    deg = np.bincount(net.src, minlength=net.n_states)
    assert deg.max() >= 700 and (deg > 256).sum() > 100
    for kw in (dict(main_beam=150.0), dict(main_beam=200.0, max_hyps=3000)):
        od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
        gs = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=3, **kw).decode_batch(feats)
        for u, x in enumerate(feats):
            assert_hyp_matches(gs[u], od.decode_certified(x), "sliced %r utt %d" % (kw, u))


def test_xcd_local_launch_and_its_fallback(small, capfd):
    """Clusters are launched XCD-local (plain stores, workgroup-scope atomics) when they fit an eighth of the
    grid; a cluster that finds itself on several XCDs must leave its stream untouched and the agent-scope
    kernel take over.  JD_XL_SELFTEST numbers the workgroups in dispatch order, which spreads every cluster
    over XCDs: results must be unaffected and the fallback must have happened."""
    import os
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0, max_hyps=200)
    od = OracleDecoder(onet, oam, **kw)
    want = [od.decode_certified(x) for x in feats]
    batch = [feats[i % len(feats)] for i in range(16)]
    for selftest in (False, True):
        env = {"JD_VERBOSE": "1"}
        if selftest:
            env["JD_XL_SELFTEST"] = "1"
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            gd = capi.Decoder(gnet, gam, max_streams=16, **kw)
            for rep in range(3):
                gs = gd.decode_batch(batch)
                for i, g in enumerate(gs):
                    assert_hyp_matches(g, want[i % len(feats)], "xcd-local selftest=%s rep %d utt %d" % (selftest, rep, i))
                    assert bit_exact(g, want[i % len(feats)])
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        err = capfd.readouterr().err
        assert "XCD-local" in err, err[-400:]
        if selftest:          # (without the knob the check passes wherever workgroup b runs on XCD b % 8 - observed, not
            # promised: on a device that places them otherwise the fallback is simply what runs)
            assert "not on one XCD" in err, err[-600:]


def test_launch_is_replanned_under_way(small, capfd):
    """When part of the grid has run out of work a launch is cut short and the rest planned anew (SearchArgs::
    rebalance_at).  Forced here at toy size - utterances of very different lengths, every launch eligible, a
    twentieth of the grid as the mark: many short launches, also together with Path collections (small arena) -
    the results are those of the oracle, bit for bit."""
    import os
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0)
    od = OracleDecoder(onet, oam, **kw)
    batch = []
    for i in range(12):                                 # lengths from one utterance to four in a row
        batch.append(np.concatenate([feats[(i + j) % len(feats)] for j in range(1 + i % 4)]))
    want = [od.decode_certified(x) for x in batch]
    env = {"JD_VERBOSE": "1", "JD_REBALANCE_MIN_US": "0", "JD_REBALANCE_FRAC": "0.05"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for extra in (dict(), dict(max_paths=1 << 12)):
            gd = capi.Decoder(gnet, gam, max_streams=12, **kw, **extra)
            for rep in range(2):
                gs = gd.decode_batch(batch)
                for i, g in enumerate(gs):
                    assert_hyp_matches(g, want[i], "re-plan %r rep %d utt %d" % (extra, rep, i))
                    assert bit_exact(g, want[i])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    err = capfd.readouterr().err
    assert err.count("cut short for a re-plan") >= 4, err[-600:]


def test_hybrid_models(built):
    """Hybrid ANN / HMM scoring (HTKModels::Load(phones, priors, statesPerModel); calcOutput,
    HTKFlatModels.cpp:190-222): the feature vector is one log posterior per phone and the scoring kernel
    is a subtraction - scores bit for bit, decoding as with Gaussians (batch, streaming, three- and
    five-state models)."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    for spm in (5, 3):
        am, net, feats, words = synth.config_hybrid(seed=3 + spm, states_per_model=spm)
        gam = capi.Models.from_hybrid(am.priors, am.states_per_model)
        oam = OracleAM.from_hybrid(am.priors, am.states_per_model)
        a, b = gam.score_frames(feats[0][:50]), oam.score_frames(feats[0][:50])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for kw in (dict(main_beam=200.0), dict(main_beam=150.0, max_hyps=100)):
            od = OracleDecoder(OracleNet(net), oam, **kw)
            gd = capi.Decoder(capi.Network.from_synth(net), gam, max_streams=len(feats), **kw)
            gs = gd.decode_batch(feats)
            for u, x in enumerate(feats):
                o = od.decode_certified(x)
                assert_hyp_matches(gs[u], o, "hybrid spm %d %r utt %d" % (spm, kw, u))
                assert bit_exact(gs[u], o)
            gd.stream_init(0)
            for pos in range(0, feats[0].shape[0], 37):
                gd.stream_push(0, feats[0][pos:pos + 37])
            assert_hyp_matches(gd.stream_finish(0), od.decode_certified(feats[0]), "hybrid streaming")
    with pytest.raises(capi.JuicerAmdError):
        capi.Models.from_hybrid(am.priors, 2)                        # statesPerModel <= 2 (HTKModels.cpp:82-83)


def test_exact_signature_constructor_decodes(small, tmp_path):
    """The drop-in line itself: `decoder = new GpuWFSTDecoder(network, models, phoneStartBeam, mainBeam, phoneEndBeam,
    wordEmitBeam, maxHyps)` with a Juicer::WFSTNetwork* and a Juicer::IModels* (WFSTDecoderLite.h:81-89,
    juicer.cpp:582-586), driven frame by frame with the look-ahead protocol of DecoderSingleTest.cpp:267-295, compiled
    against mocks of the reference's declarations (tests/mock_juicer/): the DecHyp chain equals the oracle's."""
    import subprocess
    from bridge_helper import build_program, read_arrays, write_case
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    exe = build_program(tmp_path)
    beams = dict(start_beam=120.0, main_beam=150.0, end_beam=100.0, word_beam=80.0, max_hyps=200)
    od = OracleDecoder(onet, oam, **beams)
    for u in (0, 1):
        write_case(tmp_path / "case.bin", gnet, gam, feats[u], (beams["start_beam"], beams["main_beam"], beams["end_beam"],
                                                               beams["word_beam"], beams["max_hyps"]), pad=u)
        subprocess.check_call([exe, str(tmp_path / "case.bin"), str(tmp_path / "out.bin"), "decode"], timeout=240)
        lab, tim, sc, tot = read_arrays(tmp_path / "out.bin", [np.int32, np.int32, np.float32, np.float32])
        o = od.decode(feats[u])
        assert o.n > 0 and lab.tolist() == o.label.tolist() and tim.tolist() == o.time.tolist()
        assert rel_close(sc, o.score) and rel_close(tot, [o.tot_score, o.tot_ac, o.tot_lm])


@pytest.mark.parametrize("hold", ["auto", "0", "1"])
def test_scores_ahead_of_the_search(small, hold):
    """jd_dec_prefetch_scores: the NEXT batch's likelihood table is scored beside the current batch's search (on the
    CUs its clusters leave), into the decoder's second table.  Results never depend on it: batches decoded from a
    table scored ahead equal the oracle bit for bit, a decode that was not the announced one drops the table, and so
    does an empty announcement; with re-planning forced at toy size the launch beside the scoring is held (0),
    re-planned at will (1), or either by the measured ratio (auto)."""
    import os
    import torch
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0)
    od = OracleDecoder(onet, oam, **kw)
    A = [np.concatenate([feats[(i + j) % len(feats)] for j in range(1 + i % 3)]) for i in range(8)]
    B = [np.concatenate([feats[(3 * i + j + 1) % len(feats)] for j in range(1 + (i + 1) % 4)]) for i in range(8)]
    want = {"A": [od.decode_certified(x) for x in A], "B": [od.decode_certified(x) for x in B]}
    dev = torch.device("cuda", 0)

    def resident(batch):
        offs = np.zeros(len(batch) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([x.shape[0] for x in batch])
        return torch.from_numpy(np.concatenate(batch)).to(dev), offs
    buf = {"A": resident(A), "B": resident(B)}
    env = {"JD_REBALANCE_MIN_US": "0", "JD_REBALANCE_FRAC": "0.05"}
    if hold != "auto":
        env["JD_PF_REBALANCE"] = hold
    old = {k: os.environ.get(k) for k in list(env) + ["JD_PF_REBALANCE"]}
    os.environ.pop("JD_PF_REBALANCE", None)
    os.environ.update(env)
    try:
        gd = capi.Decoder(gnet, gam, max_streams=8, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def announce(name):
        gd.prefetch_scores(buf[name][0].data_ptr(), buf[name][1], 0)

    def decode(name, ahead):
        gs = gd.decode_batch_device(buf[name][0].data_ptr(), buf[name][1], 0)
        assert gd.last_timing()["prefetched"] == (1 if ahead else 0), (name, ahead, gd.last_timing())
        for i, g in enumerate(gs):
            assert_hyp_matches(g, want[name][i], "scored ahead=%r batch %s utt %d" % (ahead, name, i))
            assert bit_exact(g, want[name][i])
    announce("B"); decode("A", False)                   # B's table is scored beside A's search
    announce("A"); decode("B", True)
    announce("A"); decode("A", True)                    # (the bench's pattern: the same batch again)
    decode("A", True)                                   # nothing announced beside it: the next one scores its own
    decode("B", False)
    announce("B"); decode("A", False)
    decode("A", False)                                  # not the announced batch: B's table is dropped
    announce("B"); decode("A", False)
    gd.prefetch_scores(0, None)                         # an empty announcement drops it as well
    decode("B", False)
    announce("A"); decode("B", False)
    gs = gd.decode_batch(A)                             # the host-buffer entry has its own device copy: not the announced buffer
    for i, g in enumerate(gs):
        assert bit_exact(g, want["A"][i])
    # more utterances than streams are decoded in waves formed by length: such a batch cannot be announced, but every
    # wave of it is scored beside the wave before it - and a batch the caller announced rides on the last wave
    gd3 = capi.Decoder(gnet, gam, max_streams=3, **kw)
    gd3.prefetch_scores(buf["A"][0].data_ptr(), buf["A"][1], 0)
    gs = gd3.decode_batch_device(buf["A"][0].data_ptr(), buf["A"][1], 0)
    assert gd3.last_timing()["prefetched"] == 2                # waves 2 and 3 of 3
    for i, g in enumerate(gs):
        assert bit_exact(g, want["A"][i])
    small3, offs3 = buf["B"][0], buf["B"][1][:4]                # three utterances: one wave
    gd3.prefetch_scores(small3.data_ptr(), offs3, 0)
    gs = gd3.decode_batch_device(buf["A"][0].data_ptr(), buf["A"][1], 0)
    for i, g in enumerate(gs):
        assert bit_exact(g, want["A"][i])
    gs = gd3.decode_batch_device(small3.data_ptr(), offs3, 0)
    assert gd3.last_timing()["prefetched"] == 1
    for i, g in enumerate(gs):
        assert bit_exact(g, want["B"][i])
    gd3.close(); gd.close()


def test_scores_ahead_with_collections_and_replans(small):
    """Scoring ahead together with everything that cuts a launch short: a Path arena so small that streams stop for
    collections all the time, re-planning forced at toy size (the launches beside a scoring are then held or cut by
    the measured ratio), waves inside one call.  Every batch equals the oracle bit for bit."""
    import os
    import torch
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0)
    od = OracleDecoder(onet, oam, **kw)
    batches = [[np.concatenate([feats[(b + 2 * i + j) % len(feats)] for j in range(1 + (i + b) % 4)]) for i in range(10)] for b in range(3)]
    want = [[od.decode_certified(x) for x in bt] for bt in batches]
    dev = torch.device("cuda", 0)
    bufs = []
    for bt in batches:
        offs = np.zeros(len(bt) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([x.shape[0] for x in bt])
        bufs.append((torch.from_numpy(np.concatenate(bt)).to(dev), offs))
    env = {"JD_REBALANCE_MIN_US": "0", "JD_REBALANCE_FRAC": "0.05"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for streams in (10, 4):                                       # one wave per batch / three waves per batch
            gd = capi.Decoder(gnet, gam, max_streams=streams, max_paths=1 << 12, **kw)
            for rep in range(7):
                b, nxt = rep % 3, (rep + 1) % 3
                gd.prefetch_scores(bufs[nxt][0].data_ptr(), bufs[nxt][1], 0)
                gs = gd.decode_batch_device(bufs[b][0].data_ptr(), bufs[b][1], 0)
                tm = gd.last_timing()
                if streams == 10:
                    assert tm["prefetched"] == (1 if rep > 0 else 0), tm
                else:
                    assert tm["prefetched"] == 2, tm              # waves 2 and 3 of the call (its first wave is another batch's size)
                for i, g in enumerate(gs):
                    assert_hyp_matches(g, want[b][i], "streams %d rep %d utt %d" % (streams, rep, i))
                    assert bit_exact(g, want[b][i])
            gd.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_batches_through_the_resident_kernel(small, monkeypatch):
    """JD_PIPELINE=3: announced batches (up to JD_PIPE_DEPTH of them) are scored whole and their utterances go through the
    one-workgroup slots of a search kernel that stays, a slot taking the next queued utterance the moment its own is through;
    a decode hands back the oldest batch.  Results are the oracle's bit for bit - with fewer slots than a batch has
    utterances, with Path arenas so small that the slots stop for collections, across jd_dec_quiesce (the kernel leaves
    for a device-wide synchronisation and comes back), when a decode is not the announced one (everything under way is
    dropped), and when an utterance fails (the error is its batch's; the batches behind it are untouched)."""
    import torch
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0)
    od = OracleDecoder(onet, oam, **kw)
    mk = lambda k: [np.concatenate([feats[(k * i + j + k) % len(feats)] for j in range(1 + (i + k) % 3)]) for i in range(6)]
    batches = {n: mk(k) for n, k in (("A", 1), ("B", 2), ("C", 3))}
    want = {n: [od.decode_certified(x) for x in b] for n, b in batches.items()}
    dev = torch.device("cuda", 0)

    def resident(batch):
        offs = np.zeros(len(batch) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([x.shape[0] for x in batch])
        return torch.from_numpy(np.concatenate(batch)).to(dev), offs
    buf = {n: resident(b) for n, b in batches.items()}
    monkeypatch.setenv("JD_DEV", "1")                                  # (development knob: 50-frame commands instead of 128)
    monkeypatch.setenv("JD_PIPE_CHUNK", "50")
    # (keep_se: the development knob that deals the slots two per CU on that many CUs of every shader engine and keeps the other CUs
    # for the scoring - jd_park_kernel; the default deals them one per CU beside the scoring)
    for streams, extra, keep_se in ((10, {}, 0), (4, {}, 0), (8, dict(max_paths=1 << 12), 0), (10, {}, 1)):
        if keep_se:
            monkeypatch.setenv("JD_SLOT_KEEP_SE", str(keep_se))
        else:
            monkeypatch.delenv("JD_SLOT_KEEP_SE", raising=False)
        gd = capi.Decoder(gnet, gam, max_streams=streams, **kw, **extra)
        gd.set_pipeline(capi.FLOW_RESIDENT, 4)                         # the interface: jd_dec_set_pipeline, four batches deep
        assert gd.pipeline_stats()["mode"] == capi.FLOW_RESIDENT and gd.pipeline_stats()["frames_searched"] == 0
        order = ["A", "B", "C", "A", "C", "B", "B", "A", "C"]
        ahead = 3
        for n in order[:ahead]:
            gd.prefetch_scores(buf[n][0].data_ptr(), buf[n][1], 0)
        for i, n in enumerate(order):
            if i + ahead < len(order):
                nx = order[i + ahead]
                gd.prefetch_scores(buf[nx][0].data_ptr(), buf[nx][1], 0)
            if i == 4:
                gd.quiesce()
                torch.cuda.synchronize()                               # (returns: the kernel has left)
            if i == 6 and streams == 10 and not keep_se:
                import time
                time.sleep(5.6)                                        # a caller that is away: the kernel leaves by itself after 5 s, and comes back
            gs = gd.decode_batch_device(buf[n][0].data_ptr(), buf[n][1], 0)
            tm = gd.last_timing()
            for u, g in enumerate(gs):
                assert_hyp_matches(g, want[n][u], "streams %d step %d batch %s utt %d" % (streams, i, n, u))
                assert bit_exact(g, want[n][u])
            assert tm["search_launches"] == 0, (streams, i, n, tm)      # (handed back by the pipeline, not by a launch of its own)
        torch.cuda.synchronize()                                       # nothing announced is left: the kernel has gone
        ps = gd.pipeline_stats()                                       # every frame of every batch went through the slots, once
        assert ps["frames_searched"] == sum(sum(x.shape[0] for x in batches[n]) for n in order), ps
        assert ps["batches_back"] == len(order) and ps["resident"] == 0 and ps["slots"] == streams
        assert (ps["collections"] > 0) == bool(extra)
        # a fifth announcement does not fit a pipeline four deep; a decode that is not the announced one drops everything
        for n in ("A", "B", "C", "A"):
            gd.prefetch_scores(buf[n][0].data_ptr(), buf[n][1], 0)
        with pytest.raises(capi.JuicerAmdError):
            gd.prefetch_scores(buf["B"][0].data_ptr(), buf["B"][1], 0)
        gs = gd.decode_batch_device(buf["C"][0].data_ptr(), buf["C"][1], 0)
        # (a launch of its own - unless the batch has more utterances than the decoder has streams: those go through the slots anyway)
        assert (gd.last_timing()["search_launches"] > 0) == (streams >= 6)
        for u, g in enumerate(gs):
            assert bit_exact(g, want["C"][u])
        gs = gd.decode_batch_device(buf["A"][0].data_ptr(), buf["A"][1], 0)
        for u, g in enumerate(gs):
            assert bit_exact(g, want["A"][u])
        gd.close()
    monkeypatch.delenv("JD_SLOT_KEEP_SE", raising=False)
    # more utterances than streams in ONE call, nothing announced: through the slots as well
    gd = capi.Decoder(gnet, gam, max_streams=4, **kw)
    gd.set_pipeline(capi.FLOW_RESIDENT, 4)
    allf = batches["A"] + batches["B"] + batches["C"]
    for rep in range(2):
        gs = gd.decode_batch(allf)
        assert gd.last_timing()["search_launches"] == 0
        for g, w in zip(gs, want["A"] + want["B"] + want["C"]):
            assert bit_exact(g, w)
    gd.close()
    # an utterance that fails (Histogram::addScore's ceiling): its batch's decode raises, the batches behind it are the oracle's
    am, net, f2, _ = synth.config_small()
    g_sharp = int(am.hmm_gmm[int(net.ilab[0]) - 1, 1])
    am.var[g_sharp] = 1e-6
    poison = am.mean[g_sharp, 0][None, :].repeat(30, axis=0).astype(np.float32)
    gnet2, gam2 = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw2 = dict(main_beam=150.0, max_hyps=100)
    od2 = OracleDecoder(OracleNet(net), OracleAM(am), **kw2)
    good, bad = [f2[i] for i in range(4)], [f2[0], poison, f2[2], f2[3]]
    wg = [od2.decode_certified(x) for x in good]
    bg, bb = resident(good), resident(bad)
    gd = capi.Decoder(gnet2, gam2, max_streams=3, **kw2)
    gd.set_pipeline(capi.FLOW_RESIDENT, 4)
    for rep in range(2):
        gd.prefetch_scores(bg[0].data_ptr(), bg[1], 0); gd.prefetch_scores(bb[0].data_ptr(), bb[1], 0); gd.prefetch_scores(bg[0].data_ptr(), bg[1], 0)
        for u, g in enumerate(gd.decode_batch_device(bg[0].data_ptr(), bg[1], 0)):
            assert bit_exact(g, wg[u])
        with pytest.raises(capi.JuicerAmdError) as ei:
            gd.decode_batch_device(bb[0].data_ptr(), bb[1], 0)
        assert ei.value.code == capi.JD_EHIST, str(ei.value)
        for u, g in enumerate(gd.decode_batch_device(bg[0].data_ptr(), bg[1], 0)):
            assert bit_exact(g, wg[u])
    gd.close()


def test_error_in_the_batch_behind(built):
    """A stream of the batch that is searched AHEAD runs into an error (Histogram::addScore's ceiling, Histogram.cpp:78-79:
    a log-likelihood above +201 at the mean of a sharp density): the error belongs to that batch's decode - not to the
    decode it ran beside, whose results are the oracle's - and the decoder goes on: the batches behind the failed one
    are the oracle's bit for bit."""
    import torch
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small()
    hmm = int(net.ilab[0]) - 1                                         # first arc out of the initial state
    g_sharp = int(am.hmm_gmm[hmm, 1])
    am.var[g_sharp] = 1e-6
    poison = am.mean[g_sharp, 0][None, :].repeat(30, axis=0).astype(np.float32)
    gnet, gam, onet, oam = capi.Network.from_synth(net), capi.Models.from_htk(am), OracleNet(net), OracleAM(am)
    kw = dict(main_beam=150.0, max_hyps=100)
    od = OracleDecoder(onet, oam, **kw)
    mk = lambda k: [np.concatenate([feats[(k * i + j + k) % len(feats)] for j in range(1 + (i + k) % 3)]) for i in range(6)]
    batches = {"A": mk(1), "B": mk(2)[:3] + [poison] + mk(2)[4:], "C": mk(3)}
    want = {n: [od.decode_certified(x) for x in batches[n]] for n in ("A", "C")}
    dev = torch.device("cuda", 0)

    def resident(batch):
        offs = np.zeros(len(batch) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([x.shape[0] for x in batch])
        return torch.from_numpy(np.concatenate(batch)).to(dev), offs
    buf = {n: resident(b) for n, b in batches.items()}
    for streams in (6, 12):                                            # one batch in flight / two
        gd = capi.Decoder(gnet, gam, max_streams=streams, **kw)

        def decode(n):
            gs = gd.decode_batch_device(buf[n][0].data_ptr(), buf[n][1], 0)
            for i, g in enumerate(gs):
                assert_hyp_matches(g, want[n][i], "streams %d batch %s utt %d" % (streams, n, i))
                assert bit_exact(g, want[n][i])
            return gd.last_timing()
        ahead = 0
        for rep in range(3):
            gd.prefetch_scores(buf["B"][0].data_ptr(), buf["B"][1], 0)
            if rep == 0:
                gd.prefetch_scores(buf["C"][0].data_ptr(), buf["C"][1], 0)
            decode("A")                                                # (B is started beside it when there are two banks)
            gd.prefetch_scores(buf["C"][0].data_ptr(), buf["C"][1], 0) if rep else None
            with pytest.raises(capi.JuicerAmdError) as ei:
                gd.decode_batch_device(buf["B"][0].data_ptr(), buf["B"][1], 0)
            assert ei.value.code == capi.JD_EHIST, str(ei.value)
            gd.prefetch_scores(buf["A"][0].data_ptr(), buf["A"][1], 0)
            ahead += 1 if decode("C")["ahead_frames"] > 0 else 0
        if streams == 12:
            assert ahead >= 1                                          # C had been started beside the failing batch
        gd.close()


def test_two_batches_in_flight(small):
    """Announcements that run two batches ahead on a decoder whose streams hold two batches: the utterances of the batch
    behind the running one are started beside it (one workgroup each, the other bank of streams) and are frames in when
    their turn comes.  Results are those of the oracle bit for bit, whatever was announced, dropped or re-ordered."""
    import torch
    from juicer_amd import capi
    from oracle.oracle import OracleDecoder
    gnet, gam, onet, oam, feats, _ = small
    kw = dict(main_beam=150.0)
    od = OracleDecoder(onet, oam, **kw)
    mk = lambda k: [np.concatenate([feats[(k * i + j + k) % len(feats)] for j in range(1 + (i + k) % 3)]) for i in range(6)]
    batches = {n: mk(k) for n, k in (("A", 1), ("B", 2), ("C", 3))}
    want = {n: [od.decode_certified(x) for x in b] for n, b in batches.items()}
    dev = torch.device("cuda", 0)

    def resident(batch):
        offs = np.zeros(len(batch) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([x.shape[0] for x in batch])
        return torch.from_numpy(np.concatenate(batch)).to(dev), offs
    buf = {n: resident(b) for n, b in batches.items()}
    gd = capi.Decoder(gnet, gam, max_streams=12, **kw)                 # two banks of six streams

    def announce(n):
        gd.prefetch_scores(buf[n][0].data_ptr(), buf[n][1], 0)

    def decode(n, scored=None, ahead=None):
        gs = gd.decode_batch_device(buf[n][0].data_ptr(), buf[n][1], 0)
        tm = gd.last_timing()
        for i, g in enumerate(gs):
            assert_hyp_matches(g, want[n][i], "batch %s utt %d (%r)" % (n, i, tm))
            assert bit_exact(g, want[n][i])
        if scored is not None:
            assert tm["prefetched"] == (1 if scored else 0), (n, tm)
        if ahead is not None:
            assert (tm["ahead_frames"] > 0) == ahead, (n, tm)
        return tm
    announce("B"); announce("C")                                      # two ahead of the first decode ...
    decode("A", scored=False, ahead=False)                            # ... both scored beside it
    started = 0
    for n, nxt in (("B", "A"), ("C", "B"), ("A", "C"), ("B", "A"), ("C", "B")):
        announce(nxt)                                                  # ... and one before every later one
        tm = decode(n, scored=True)
        started += 1 if tm["ahead_frames"] > 0 else 0
    assert started >= 3, started                                      # (the first of them may find its table still being scored)
    # a decode that is not the announced one drops what was worked ahead; the decoder goes on as if nothing had been
    decode("C", scored=False, ahead=False)                            # (A is at the head of the queue, started; B behind it)
    decode("A", scored=False, ahead=False)
    # a batch that does not fit a bank takes every stream: the batch behind it starts again when its turn comes
    big = batches["A"] + batches["B"][:4]
    announce("B"); announce("C"); decode("A")
    bb, bo = resident(big)
    gs = gd.decode_batch_device(bb.data_ptr(), bo, 0)
    for i, g in enumerate(gs):
        assert bit_exact(g, (want["A"] + want["B"][:4])[i])
    decode("B", ahead=False); decode("C")
    # small Path arenas: streams of both batches stop for collections inside the common launches, and go on
    gs_ = capi.Decoder(gnet, gam, max_streams=12, max_paths=1 << 14, **kw)
    gs_.prefetch_scores(buf["B"][0].data_ptr(), buf["B"][1], 0); gs_.prefetch_scores(buf["C"][0].data_ptr(), buf["C"][1], 0)
    ahead_seen = 0
    for n, nxt in (("A", "A"), ("B", "B"), ("C", "C"), ("A", "A"), ("B", None), ("C", None)):
        if nxt:
            gs_.prefetch_scores(buf[nxt][0].data_ptr(), buf[nxt][1], 0)
        got = gs_.decode_batch_device(buf[n][0].data_ptr(), buf[n][1], 0)
        tm = gs_.last_timing()
        ahead_seen += 1 if tm["ahead_frames"] > 0 else 0
        for i, g in enumerate(got):
            assert bit_exact(g, want[n][i]), (n, i, tm)
    assert ahead_seen >= 2
    gs_.close()
    # switched off: same results, nothing ahead
    g0 = capi.Decoder(gnet, gam, max_streams=12, **kw)
    g0.set_pipeline(capi.FLOW_SERIAL)
    g0.prefetch_scores(buf["B"][0].data_ptr(), buf["B"][1], 0); g0.prefetch_scores(buf["C"][0].data_ptr(), buf["C"][1], 0)
    for n in ("A", "B", "C"):
        gs = g0.decode_batch_device(buf[n][0].data_ptr(), buf[n][1], 0)
        assert g0.last_timing()["ahead_frames"] == 0
        for i, g in enumerate(gs):
            assert bit_exact(g, want[n][i])
    g0.close(); gd.close()
