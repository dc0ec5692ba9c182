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
    threads = [threading.Thread(target=_drive, args=(broker, [(u, feats[u]) for u in range(t, len(feats), 6)], out, 23 + 7 * t))
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
    assert st["ticks"] < pushes and st["stream_ticks"] > st["ticks"]   # launches were shared
    # errors come back through the client that caused them: a push outside init .. finish
    c = broker.open()
    with pytest.raises(capi.JuicerAmdError):
        broker.push(c, feats[0][:3])
    broker.init(c)
    broker.push(c, feats[0])
    assert bit_exact(broker.finish(c), want[0])
    broker.close()
    dec.close()


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
    for rep in range(2):                                               # (the first pass warms the decoder up)
        threads = [threading.Thread(target=_drive, args=(broker, [(u, feats[u]) for u in range(t, 64, 16)], out, 64)) for t in range(16)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        t_broker = time.perf_counter() - t0
    torch.cuda.synchronize()
    for u in range(64):
        assert bit_exact(out[u], want[u]), u
    st = broker.stats()
    ratio = (frames / t_broker) / (frames / t_batch)
    print("16 callers through the broker: %.0f frames/s, one batch of 64: %.0f frames/s (host-inclusive) - ratio %.2f; %.1f streams, %.0f frames per tick"
          % (frames / t_broker, frames / t_batch, ratio, st["stream_ticks"] / st["ticks"], st["frames"] / st["ticks"]))
    assert ratio >= 0.2                                               # (16 streams are latency-bound: DESIGN.md, "the drop-in seam's own throughput")
    broker.close()
    dec.close()
