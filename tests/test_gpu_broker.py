"""Many serial IDecoder callers on one decoder (jd_broker_*, jd_streams_push): results are those of a batch decode,
bit for bit, whatever the interleaving; pushes are coalesced into common launches."""
import threading
import time

import numpy as np
import pytest

from helpers import bit_exact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(built):
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=12)
    return capi.Network.from_synth(net), capi.Models.from_htk(am), feats


KW = dict(main_beam=150.0, end_beam=100.0, word_beam=80.0)


def test_streams_push_matches_batch(small):
    """Streams that sit at different frames share one scoring launch and one search launch."""
    from juicer_amd import capi
    gnet, gam, feats = small
    for kw in (KW, dict(main_beam=150.0, max_hyps=200)):
        want = capi.Decoder(gnet, gam, max_streams=4, **kw).decode_batch(feats[:4])
        dec = capi.Decoder(gnet, gam, max_streams=4, **kw)
        for s in range(4):
            dec.stream_init(s)
        pos = [0, 0, 0, 0]
        step = [17, 64, 5, 130]                                        # every stream at its own pace
        while any(pos[s] < feats[s].shape[0] for s in range(4)):
            ss = [s for s in range(4) if pos[s] < feats[s].shape[0]]
            dec.streams_push(ss, [feats[s][pos[s]:pos[s] + step[s]] for s in ss])
            for s in ss:
                pos[s] += step[s]
        for s in range(4):
            assert bit_exact(dec.stream_finish(s), want[s]), s
        # a second utterance on the same streams, one of them idle
        for s in (0, 2):
            dec.stream_init(s)
        dec.streams_push([2, 0], [feats[1], feats[3]])
        assert bit_exact(dec.stream_finish(0), want[3]) and bit_exact(dec.stream_finish(2), want[1])
        dec.close()


def _drive(broker, utts, out, chunk):
    c = broker.open()
    for u, x in utts:
        broker.init(c)
        for i in range(0, x.shape[0], chunk):
            broker.push(c, x[i:i + chunk])
        out[u] = broker.finish(c)
    broker.close_client(c)


def test_broker_threads_match_batch(small):
    from juicer_amd import capi
    gnet, gam, feats = small
    want = capi.Decoder(gnet, gam, max_streams=len(feats), **KW).decode_batch(feats)
    dec = capi.Decoder(gnet, gam, max_streams=6, **KW)
    broker = capi.Broker(dec)
    out = [None] * len(feats)
    threads = [threading.Thread(daemon=True, target=_drive, args=(broker, [(u, feats[u]) for u in range(t, len(feats), 6)], out, 23 + 7 * t))
               for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for u in range(len(feats)):
        assert out[u] is not None and bit_exact(out[u], want[u]), u
    st = broker.stats()
    pushes = sum((feats[u].shape[0] + (23 + 7 * (u % 6)) - 1) // (23 + 7 * (u % 6)) for u in range(len(feats)))
    print("broker: %d ticks for %d pushes, %.1f streams per tick" % (st["ticks"], pushes, st["stream_ticks"] / max(st["ticks"], 1)))
    assert st["frames"] == sum(f.shape[0] for f in feats)
    assert st["ticks"] < pushes                                        # pushes were coalesced
    if not st["resident"]:
        assert st["stream_ticks"] > st["ticks"]                        # launches were shared
    # errors come back through the client that caused them: a push outside init .. finish
    c = broker.open()
    with pytest.raises(capi.JuicerAmdError):
        broker.push(c, feats[0][:3])
    broker.init(c)
    broker.push(c, feats[0])
    assert bit_exact(broker.finish(c), want[0])
    broker.close()
    dec.close()


def test_broker_error_stays_with_its_client(built):
    """One caller's utterance fails (Histogram::addScore's ceiling, Histogram.cpp:78-79: a log-likelihood above +201) in the
    middle of common launches: the error comes back through THAT caller's finish; its next utterance and everybody else's
    are those of a batch decode, bit for bit."""
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=12)
    hmm = int(net.ilab[0]) - 1                                         # first arc out of the initial state
    g_sharp = int(am.hmm_gmm[hmm, 1])
    am.var[g_sharp] = 1e-6
    poison = am.mean[g_sharp, 0][None, :].repeat(40, axis=0).astype(np.float32)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0, max_hyps=100)
    want = capi.Decoder(gnet, gam, max_streams=len(feats), **kw).decode_batch(feats)
    dec = capi.Decoder(gnet, gam, max_streams=4, **kw)
    broker = capi.Broker(dec)
    out, errs = [None] * len(feats), []

    def drive(t):
        c = broker.open()
        for k, u in enumerate(range(t, len(feats), 4)):
            if t == 2 and k == 1:                                      # this caller's second utterance is the one that fails
                broker.init(c)
                try:
                    for i in range(0, poison.shape[0], 9):
                        broker.push(c, poison[i:i + 9])
                    broker.finish(c)
                    errs.append(None)
                except capi.JuicerAmdError as e:
                    errs.append(e.code)
            broker.init(c)
            for i in range(0, feats[u].shape[0], 31 + 5 * t):
                broker.push(c, feats[u][i:i + 31 + 5 * t])
            out[u] = broker.finish(c)
        broker.close_client(c)
    threads = [threading.Thread(daemon=True, target=drive, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errs == [capi.JD_EHIST], errs
    for u in range(len(feats)):
        assert out[u] is not None and bit_exact(out[u], want[u]), u
    broker.close()
    dec.close()


@pytest.mark.parametrize("resident", ["1", "0"])
def test_broker_corner_cases(built, resident, monkeypatch):
    """Both workers of the broker - the resident search kernel (default) and the ticks - on what the plain path never
    meets: Path arenas so small that the streams stop for collections all the time, an utterance without frames, init()
    in the middle of an utterance, HMMs of 1-6 emitting states (the other record layout).  Results are those of the
    streaming API / a batch decode, bit for bit."""
    from juicer_amd import capi, synth
    monkeypatch.setenv("JD_BROKER_RESIDENT", resident)
    am, net, feats, _ = synth.config_small(n_utts=8)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    want = capi.Decoder(gnet, gam, max_streams=len(feats), **kw).decode_batch(feats)
    ref = capi.Decoder(gnet, gam, max_streams=1, **kw)
    ref.stream_init(0)
    empty = ref.stream_finish(0)                                       # init() directly followed by finish()
    ref.close()
    # small Path arenas: collections between the chunks and inside them
    dec = capi.Decoder(gnet, gam, max_streams=4, max_paths=1 << 12, **kw)
    broker = capi.Broker(dec)
    assert bool(broker.stats()["resident"]) == (resident == "1")
    out = [None] * len(feats)
    threads = [threading.Thread(daemon=True, target=_drive, args=(broker, [(u, feats[u]) for u in range(t, len(feats), 4)], out, 50 + 11 * t)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for u in range(len(feats)):
        assert out[u] is not None and bit_exact(out[u], want[u]), u
    if resident == "1":
        assert broker.stats()["us_init"] > 0                          # (resident worker: the number of collections between chunks)
    # an utterance without frames; init() in the middle of an utterance drops it
    c = broker.open()
    broker.init(c)
    h = broker.finish(c)
    assert h.n == empty.n and bit_exact(h, empty)
    broker.init(c)
    broker.push(c, feats[0][:100])
    broker.init(c)
    broker.push(c, feats[1])
    assert bit_exact(broker.finish(c), want[1])
    broker.init(c)
    broker.push(c, feats[2][:37])
    broker.close_client(c)                                             # (an utterance that was never finished)
    c = broker.open()
    broker.init(c)
    broker.push(c, feats[3])
    assert bit_exact(broker.finish(c), want[3])
    broker.close()
    dec.close()
    # HMMs of 1-6 emitting states
    am2, net2, feats2, _ = synth.config_mixed(n_utts=6)
    gnet2, gam2 = capi.Network.from_synth(net2), capi.Models.from_htk(am2)
    want2 = capi.Decoder(gnet2, gam2, max_streams=6, **kw).decode_batch(feats2)
    dec = capi.Decoder(gnet2, gam2, max_streams=3, **kw)
    broker = capi.Broker(dec)
    out = [None] * 6
    threads = [threading.Thread(daemon=True, target=_drive, args=(broker, [(u, feats2[u]) for u in range(t, 6, 3)], out, 40 + 9 * t)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for u in range(6):
        assert out[u] is not None and bit_exact(out[u], want2[u]), u
    broker.close()
    dec.close()


def test_two_brokers_and_a_batch_decoder_share_the_device(built):
    """A resident search kernel holds the device while it has work; when another decoder of the process waits for the device
    - a second broker's kernel, a batch decode - it lets its running chunks run out and makes room: two busy brokers and a
    batch decoder, all on one GPU from different threads, all finish, results bit for bit."""
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=12)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    want = capi.Decoder(gnet, gam, max_streams=len(feats), **kw).decode_batch(feats)
    decs = [capi.Decoder(gnet, gam, max_streams=3, **kw) for _ in range(2)]
    brokers = [capi.Broker(d) for d in decs]
    assert all(b.stats()["resident"] for b in brokers)
    bd = capi.Decoder(gnet, gam, max_streams=4, **kw)
    outs = [[None] * len(feats) for _ in range(2)]
    batch_out = []
    order = [(u % len(feats), feats[u % len(feats)]) for u in range(3 * len(feats))]        # every caller: the list three times

    def drive(k, t):
        mine = {}
        _drive(brokers[k], [(i, x) for i, (u, x) in enumerate(order) if i % 3 == t], mine, 64)
        for i, h in mine.items():
            assert bit_exact(h, want[order[i][0]]), (k, t, i)
        outs[k][t] = len(mine)

    def batch():
        for rep in range(6):
            got = bd.decode_batch(feats[:4])
            batch_out.append(all(bit_exact(g, w) for g, w in zip(got, want[:4])))
    threads = [threading.Thread(daemon=True, target=drive, args=(k, t)) for k in range(2) for t in range(3)] + [threading.Thread(daemon=True, target=batch)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "somebody starved"
    assert batch_out == [True] * 6
    assert all(outs[k][t] == len(feats) for k in range(2) for t in range(3))
    for b in brokers:
        b.close()
    for d in decs + [bd]:
        d.close()


def test_broker_throughput_at_configs1(built):
    """16 serial callers (threads) on the configs[1] graph against ONE batch of the same 64 utterances."""
    import torch
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_c2(n_utts=64)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    frames = sum(f.shape[0] for f in feats)
    bd = capi.Decoder(gnet, gam, max_streams=64, **kw)
    want = bd.decode_batch(feats)
    t0 = time.perf_counter()
    bd.decode_batch(feats)
    t_batch = time.perf_counter() - t0
    bd.close()
    dec = capi.Decoder(gnet, gam, max_streams=16, **kw)
    broker = capi.Broker(dec)
    out = [None] * 64
    threads = [threading.Thread(daemon=True, target=_drive, args=(broker, [(u, feats[u]) for u in range(t, 64, 16)], out, 64)) for t in range(16)]
    for t in threads:                                                  # (this pass warms the decoder up)
        t.start()
    for t in threads:
        t.join()
    for u in range(64):
        assert bit_exact(out[u], want[u]), u
    # the steady state: every caller decodes the whole list, each from another starting point - equal work per caller, so
    # that the rate is not the tail of the caller that drew the longest utterances
    outs = [[None] * 64 for _ in range(16)]
    order = [[(u % 64, feats[u % 64]) for u in range(4 * t, 4 * t + 64)] for t in range(16)]
    threads = [threading.Thread(daemon=True, target=_drive, args=(broker, order[t], outs[t], 64)) for t in range(16)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    t_broker = time.perf_counter() - t0
    torch.cuda.synchronize()
    for t in range(16):
        for u in range(64):
            assert bit_exact(outs[t][u], want[u]), (t, u)
    st = broker.stats()
    ratio = (16 * frames / t_broker) / (frames / t_batch)
    print("16 callers through the broker: %.0f frames/s, one batch of 64: %.0f frames/s (host-inclusive) - ratio %.2f; %.1f streams, %.0f frames per tick"
          % (16 * frames / t_broker, frames / t_batch, ratio, st["stream_ticks"] / st["ticks"], st["frames"] / st["ticks"]))
    assert ratio >= (0.4 if st["resident"] else 0.3)                  # (16 streams are latency-bound: docs/DESIGN_HISTORY.md 6, "the drop-in seam's own throughput")
    broker.close()
    dec.close()


def test_batch_test_cli_harness_threads(built, tmp_path):
    """jd_batch_test -threads N: N serial harness threads (the reference's init / processFrame / finish loop each, with
    its 20-row look-ahead) over their shares of the list through ONE decoder - GpuDecoderPool + GpuWFSTPooledDecoder
    (include/juicer_amd_decoder.hpp) - give the output of the batched path, in list order."""
    import subprocess
    from juicer_amd import build as jbuild, io as jio, synth
    am, net, feats, _ = synth.config_small(n_utts=7)
    jio.write_fsm(tmp_path / "g.fsm", net)
    jio.write_jdam(tmp_path / "m.jdam", am)
    with open(tmp_path / "list.txt", "w") as f:
        for u, x in enumerate(feats):
            jio.write_jdf(tmp_path / ("u%d.jdf" % u), x)
            f.write("%s\n" % (tmp_path / ("u%d.jdf" % u)))
    base = [jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "g.fsm"), "-modelsFName", str(tmp_path / "m.jdam"),
            "-inputFName", str(tmp_path / "list.txt"), "-mainBeam", "150", "-maxHyps", "200", "-outputFormat", "xmlf"]
    plain = subprocess.run(base, capture_output=True, text=True, timeout=240)
    assert plain.returncode == 0, plain.stderr
    for n in (1, 3, 7):
        th = subprocess.run(base + ["-threads", str(n)], capture_output=True, text=True, timeout=240)
        assert th.returncode == 0, th.stderr
        assert th.stdout == plain.stdout, n
        assert th.stderr.count("RT factor") == len(feats) + 1


def test_two_processes_share_one_gpu(built, tmp_path):
    """The reference's way of using several cores - several processes over split file lists (userman :584) - pointed at ONE
    GPU: persistent search launches of the two processes take turns (the per-GPU file lock of launch_search) and both
    finish, with the results of a process that has the GPU to itself."""
    import subprocess
    import sys
    import os
    code = (
        "import sys, numpy as np\\n"
        "sys.path.insert(0, %r)\\n"
        "from juicer_amd import capi, synth\\n"
        "am, net, feats, _ = synth.config_small(n_utts=8)\\n"
        "dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=150.0, max_streams=8)\\n"
        "sig = None\\n"
        "for it in range(int(sys.argv[1])):\\n"
        "    hy = dec.decode_batch(feats)\\n"
        "    s = [(h.n, h.label.tobytes(), h.time.tobytes(), np.asarray(h.score, np.float32).tobytes()) for h in hy]\\n"
        "    assert sig is None or s == sig\\n"
        "    sig = s\\n"
        "import hashlib; print(hashlib.sha256(repr(sig).encode()).hexdigest())\\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JD_GPU_LOCK_DIR=str(tmp_path))
    alone = subprocess.run([sys.executable, "-c", code.replace("\\n", "\n"), "2"], capture_output=True, text=True, timeout=300, env=env)
    assert alone.returncode == 0, alone.stderr
    procs = [subprocess.Popen([sys.executable, "-c", code.replace("\\n", "\n"), "40"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
        assert o.strip().splitlines()[-1] == alone.stdout.strip().splitlines()[-1]
    assert any(f.startswith("juicer_amd.gpu-") for f in os.listdir(tmp_path))


def test_paused_caller_keeps_its_cluster(built):
    """One caller pauses for longer than the resident kernel's idle limit (5 s) in the middle of an utterance, and another opens its
    client that late, while two others keep the kernel busy: the paused stream's cluster must still be there (round 4: a
    cluster left after 5 s without a command for ITS stream, the paused caller's next chunk was never served and every
    client's utterance was dropped with an error).  The kernel now leaves only when the HOST shows no sign of life."""
    import time
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=6)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    want = capi.Decoder(gnet, gam, max_streams=len(feats), **kw).decode_batch(feats)
    dec = capi.Decoder(gnet, gam, max_streams=4, **kw)
    broker = capi.Broker(dec)
    assert broker.stats()["resident"]
    stop = threading.Event()
    errs, got, n_busy = [], {}, [0, 0]

    def busy(t):                                                       # utterance after utterance, in small pushes: the kernel never idles
        try:
            c = broker.open()
            k = 0
            while not stop.is_set():
                u = 2 + (k + t) % 4
                broker.init(c)
                for i in range(0, feats[u].shape[0], 23):
                    broker.push(c, feats[u][i:i + 23])
                    time.sleep(0.0005)
                h = broker.finish(c)
                assert bit_exact(h, want[u]), ("busy", t, u)
                k += 1
            n_busy[t] = k
            broker.close_client(c)
        except Exception as e:                                         # noqa: BLE001
            errs.append(("busy %d" % t, repr(e)))

    def paused():
        try:
            c = broker.open()
            broker.init(c)
            half = feats[0].shape[0] // 2
            broker.push(c, feats[0][:half])
            time.sleep(6.5)                                            # longer than RES_IDLE_TICKS
            broker.push(c, feats[0][half:])
            got[0] = broker.finish(c)
            broker.close_client(c)
        except Exception as e:                                         # noqa: BLE001
            errs.append(("paused", repr(e)))

    def late():
        try:
            time.sleep(6.0)                                            # a client slot that is first used when the kernel is 6 s old
            c = broker.open()
            broker.init(c)
            for i in range(0, feats[1].shape[0], 40):
                broker.push(c, feats[1][i:i + 40])
            got[1] = broker.finish(c)
            broker.close_client(c)
        except Exception as e:                                         # noqa: BLE001
            errs.append(("late", repr(e)))
    ths = [threading.Thread(daemon=True, target=busy, args=(t,)) for t in range(2)] + [threading.Thread(daemon=True, target=paused),
                                                                                        threading.Thread(daemon=True, target=late)]
    for t in ths:
        t.start()
    ths[2].join(60); ths[3].join(60)
    stop.set()
    ths[0].join(60); ths[1].join(60)
    assert not any(t.is_alive() for t in ths), "a caller hangs"
    assert not errs, errs
    assert bit_exact(got[0], want[0]) and bit_exact(got[1], want[1])
    assert min(n_busy) > 3
    broker.close()
    dec.close()


@pytest.mark.parametrize("resident", ["1", "0"])
def test_failed_init_wakes_its_callers(built, resident, monkeypatch):
    """An init that fails (injected: JD_BROKER_FAIL_INIT) must come back as THE error of the calls behind it - a push that waits for
    room and a finish used to wait for ever, and the error itself was masked by "not between init and finish"."""
    from juicer_amd import capi, synth
    monkeypatch.setenv("JD_BROKER_RESIDENT", resident)
    monkeypatch.setenv("JD_BROKER_FAIL_INIT", "0")
    am, net, feats, _ = synth.config_small(n_utts=2)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    want = capi.Decoder(gnet, gam, max_streams=2, **kw).decode_batch(feats)
    dec = capi.Decoder(gnet, gam, max_streams=2, **kw)
    broker = capi.Broker(dec)
    res = {}

    c_doomed, c_fine = broker.open(), broker.open()                   # (both open from the start: the injection goes by client number)
    assert c_doomed == 0

    def doomed():
        c = c_doomed
        codes = []
        broker.init(c)
        try:
            for _ in range(40):                                        # far more than a client may have pending: a push has to wait for room
                broker.push(c, feats[0][:100])
            codes.append(None)
        except capi.JuicerAmdError as e:
            codes.append((e.code, "injected" in str(e)))
        try:
            broker.finish(c)
            codes.append(None)
        except capi.JuicerAmdError as e:
            codes.append((e.code, "injected" in str(e)))
        res["doomed"] = codes
        broker.close_client(c)

    def fine():
        c = c_fine
        broker.init(c)
        broker.push(c, feats[1])
        res["fine"] = broker.finish(c)
        broker.close_client(c)
    ths = [threading.Thread(daemon=True, target=doomed), threading.Thread(daemon=True, target=fine)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(60)
    assert not any(t.is_alive() for t in ths), "a caller hangs behind the failed init"
    assert res["doomed"] == [(capi.JD_EHIP, True), (capi.JD_EHIP, True)], res["doomed"]
    assert bit_exact(res["fine"], want[1])
    broker.close()
    dec.close()
