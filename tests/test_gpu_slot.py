"""The slot kernel (csrc/jd_slot.h: one workgroup per stream, every per-frame word in LDS, compiled for two workgroups per CU)
against the CPU oracle.  The batch pipeline runs it under a mailbox (tests/test_gpu_parity.py::test_batches_through_the_resident_kernel,
tests/test_gpu_fullsize.py); here it is the plain launch k_slot_batch that launch_search takes for batches of more streams than
the chip has CUs - forced at toy size with the development knobs JD_CW=1 (one workgroup per stream) + JD_SLOT_BATCH=1 - so that
every corner the cluster kernel is tested on is tested on this one too: all pruning combinations, the tee model, HMMs of 1-6
emitting states (the other record layout), Path collections inside an utterance, the streaming calls with pushes of odd sizes,
partial traces on the reference's count rule, and a batch of more utterances than CUs."""
import numpy as np
import pytest

from helpers import assert_hyp_matches, bit_exact, oracle_certified_many

pytestmark = pytest.mark.gpu

BEAMS = [
    dict(),
    dict(main_beam=200.0),
    dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0),
    dict(main_beam=150.0, max_hyps=200),
    dict(max_hyps=300),
    dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150),
]


@pytest.fixture()
def slot_kernel(monkeypatch, built):
    monkeypatch.setenv("JD_DEV", "1")
    monkeypatch.setenv("JD_CW", "1")
    monkeypatch.setenv("JD_SLOT_BATCH", "1")
    return True


def _cfg(name):
    from juicer_amd import synth
    return {"toy": synth.config_toy, "small": synth.config_small, "mixed": synth.config_mixed}[name]()


@pytest.mark.parametrize("cfg", ["toy", "small", "mixed"])
@pytest.mark.parametrize("bi", range(len(BEAMS)))
def test_slot_kernel_decode(slot_kernel, cfg, bi):
    from juicer_amd import capi
    am, net, feats, _ = _cfg(cfg)
    kw = BEAMS[bi]
    big = (1 << 25) if kw.get("main_beam", 0.0) in (0.0, 200.0) and not kw.get("max_hyps") else 0
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=len(feats), max_paths=big, **kw)
    gs = gd.decode_batch(feats)
    tm = gd.last_timing()
    assert tm["slot_launches"] == tm["search_launches"] > 0, tm        # (every launch was the slot kernel's)
    want = oracle_certified_many(net, am, feats, **kw)
    for u, o in enumerate(want):
        assert_hyp_matches(gs[u], o, "%s utt %d %s" % (cfg, u, kw))
        assert bit_exact(gs[u], o), (cfg, u, kw)
    gd.close()


def test_slot_kernel_path_collections_and_streaming(slot_kernel):
    """Path arenas so small that a stream stops for collections inside its utterance (the launch is repeated for what is left),
    and the streaming calls: init / push x n / finish with pushes of odd sizes - the lists a command leaves in HBM are what
    the next one finds."""
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=6)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    kw = dict(main_beam=150.0)
    want = oracle_certified_many(net, am, feats, **kw)
    gd = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=1 << 12, **kw)
    gs = gd.decode_batch(feats)
    tm = gd.last_timing()
    assert tm["relaunches"] > 0 and tm["slot_launches"] == tm["search_launches"], tm
    for u, o in enumerate(want):
        assert bit_exact(gs[u], o), u
    gd.close()
    gd = capi.Decoder(gnet, gam, max_streams=2, **kw)
    for u, step in ((0, 1), (1, 7), (2, 37), (3, 128), (4, 1000)):
        s = u % 2
        gd.stream_init(s)
        for i in range(0, feats[u].shape[0], step):
            gd.stream_push(s, feats[u][i:i + step])
        assert bit_exact(gd.stream_finish(s), want[u]), (u, step)
    gd.close()


def test_slot_kernel_more_utterances_than_cus(built):
    """What launch_search takes the slot kernel for WITHOUT any knob: one call with more utterances (and streams) than the chip
    has CUs - a workgroup per stream, two per CU, the dispatcher deals the next one when one leaves."""
    import torch
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=24)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    many = [feats[u % len(feats)] for u in range(n_cus + 40)]
    kw = dict(main_beam=150.0, max_hyps=200)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=len(many), **kw)
    gs = gd.decode_batch(many)
    tm = gd.last_timing()
    assert tm["slot_launches"] > 0 and tm["slot_launches"] == tm["search_launches"], tm
    want = oracle_certified_many(net, am, feats, **kw)
    for u, g in enumerate(gs):
        assert bit_exact(g, want[u % len(feats)]), u
    gd.close()


def test_slot_kernel_partial_traces(slot_kernel):
    """PARTIAL_DECODING through the slot kernel: the collections run after the oracle's own frames (frame rule and the count
    rule on the reference's Path counts) and every trace is the oracle's."""
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small(n_utts=2)
    kw = dict(main_beam=150.0)
    od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=1, **kw)
    gd.set_partial_interval(30)
    for u in range(2):
        snaps, final = od.decode_partial(feats[u], interval=30)
        gd.stream_init(0)
        for i in range(0, feats[u].shape[0], 16):
            gd.stream_push(0, feats[u][i:i + 16])
        n_coll, last = gd.stream_collect_info(0)
        assert n_coll == len(od.collect_frames) and (last == od.collect_frames[-1] if od.collect_frames else last == -1), (n_coll, od.collect_frames)
        h = gd.stream_finish(0)
        _, got = gd.stream_partial(0)
        assert got == final and h.n > 0, u
    gd.close()


def test_pipeline_api_arguments(built):
    """jd_dec_set_pipeline / jd_dec_pipeline_stats: the modes, the defaults derived from the slots, and what is refused."""
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_toy()
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=4, main_beam=150.0)
    assert gd.pipeline_stats()["mode"] == capi.FLOW_TWO_IN_FLIGHT                     # the default
    gd.set_pipeline(capi.FLOW_SERIAL)
    assert gd.pipeline_stats()["mode"] == capi.FLOW_SERIAL
    gd.set_pipeline(capi.FLOW_RESIDENT)                                              # depth and slots by default
    ps = gd.pipeline_stats()
    assert ps["mode"] == capi.FLOW_RESIDENT and ps["slots"] == 4 and ps["depth"] == 8 and ps["resident"] == 0
    for bad in ((2, 0, 0), (capi.FLOW_RESIDENT, 1, 0), (capi.FLOW_RESIDENT, 33, 0), (capi.FLOW_RESIDENT, 4, 5), (capi.FLOW_RESIDENT, 4, -1)):
        with pytest.raises(capi.JuicerAmdError) as ei:
            gd.set_pipeline(*bad)
        assert ei.value.code == capi.JD_EINVAL, bad
    assert gd.pipeline_stats()["mode"] == capi.FLOW_RESIDENT                          # (a refused call changes nothing)
    h = gd.decode_batch(feats)[0]                                                    # one utterance, nothing announced: the usual way
    gd.set_pipeline(capi.FLOW_TWO_IN_FLIGHT)
    h2 = gd.decode_batch(feats)[0]
    assert bit_exact(h, h2) and h.n > 0
    gd.close()


def test_slot_kernel_rows_of_every_length(monkeypatch, built):
    """Phase X's prefix walk (XState, jd_slot.h) decides per row how it is walked: rows of up to 57 arcs by the sorted prefix and
    their instance flags in one or two batches of loads, longer rows whole.  A trigram-shaped graph has rows of every length - from
    one arc to thousands, epsilon back-off arcs among them - on both sides of every one of those limits: the slot kernel's results
    AND the reference's statistics on it must be those of k_search (which the full-size tests hold against the oracle)."""
    from helpers import STAT_KEYS
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_c4(seed=0, n_utts=4, n_words=10000, n_tri_hist=100_000)
    deg = np.bincount(net.src, minlength=net.n_states)
    for lo, hi in ((1, 24), (25, 32), (33, 57), (58, 64), (65, 256), (257, 1 << 30)):      # the limits of jd_slot.h / X_SLICE
        assert np.any((deg >= lo) & (deg <= hi)), (lo, hi)
    models, network = capi.Models.from_htk(am), capi.Network.from_synth(net)
    monkeypatch.setenv("JD_DEV", "1")
    outs = []
    for slot in (False, True):
        monkeypatch.setenv("JD_SLOT_BATCH", "1" if slot else "0")
        monkeypatch.setenv("JD_CW", "1" if slot else "64")
        gd = capi.Decoder(network, models, main_beam=200.0, max_streams=len(feats))
        outs.append(gd.decode_batch(feats))
        assert (gd.last_timing()["slot_launches"] > 0) == slot
        gd.close()
    for u, (a, b) in enumerate(zip(*outs)):
        assert a.n > 0 and bit_exact(b, a), u
        for k in STAT_KEYS:
            assert a.stats[k] == b.stats[k], (u, k, a.stats[k], b.stats[k])


def test_slot_launch_is_not_chosen_over_streams_with_cluster_lists(built):
    """launch_search took k_slot_batch by the launch's shape alone (more streams than CUs, one workgroup each).  Streams in
    the middle of an utterance whose lists were written by clusters of SEVERAL workgroups - 100 callers served with clusters of two,
    then 200 more join - are not the slot kernel's to read (eight wave segments, JDE_GEOM): such a launch stays with k_search,
    which reads lists of any geometry; nobody's utterance dies.  The next utterances, started together, do take the slot kernel."""
    import torch
    from juicer_amd import capi, synth
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    n1, n2 = max(2, n_cus * 100 // 256), n_cus + 44                     # (100 and 300 on an MI355X)
    am, net, feats, _ = synth.config_small(n_utts=4)
    kw = dict(main_beam=150.0)
    want = oracle_certified_many(net, am, feats, **kw)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=n2, **kw)
    for s in range(n2):
        gd.stream_init(s)
    gd.streams_push(list(range(n1)), [feats[s % 4][:60] for s in range(n1)])
    tm = gd.last_timing()                                               # (the streaming calls add to the decoder's timing record)
    assert tm["cluster_wgs"] >= 2 and tm["slot_launches"] == 0, tm      # clusters of several workgroups wrote these lists
    gd.streams_push(list(range(n2)), [feats[s % 4][60:] if s < n1 else feats[s % 4] for s in range(n2)])
    tm1 = gd.last_timing()
    assert tm1["slot_launches"] == 0 and tm1["search_launches"] > tm["search_launches"], tm1   # by shape a slot launch; by the streams' lists not
    for s in range(n2):
        assert bit_exact(gd.stream_finish(s), want[s % 4]), s
    for s in range(n2):                                                 # the next utterances start together: the slot kernel's
        gd.stream_init(s)
    gd.streams_push(list(range(n2)), [feats[(s + 1) % 4] for s in range(n2)])
    tm2 = gd.last_timing()
    assert tm2["slot_launches"] == tm2["search_launches"] - tm1["search_launches"] > 0, tm2
    for s in range(n2):
        assert bit_exact(gd.stream_finish(s), want[(s + 1) % 4]), s
    gd.close()


def test_counters_of_what_the_kernels_touch(built, monkeypatch):
    """jd_stats' tot_recs_read .. tot_closure_items (bench.py's design-bytes roofline): what the kernels really took up, against the
    reference's figures beside them - records read + arcs attached never exceed the reference's instance count (hopeless candidates
    are counted there and never become a record), arcs walked never exceed arcs visited (a prefix walk accounts for the rest).
    Phase A's counters are the algorithm's, not a kernel's: the cluster kernel and the slot kernel must report the same."""
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_small(n_utts=4)
    seen = {}
    for slot in (False, True):
        if slot:
            monkeypatch.setenv("JD_DEV", "1"); monkeypatch.setenv("JD_CW", "1"); monkeypatch.setenv("JD_SLOT_BATCH", "1")
        gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=4, main_beam=150.0)
        gs = gd.decode_batch(feats)
        assert (gd.last_timing()["slot_launches"] > 0) == slot
        for g in gs:
            st = g.stats
            assert 0 < st["tot_recs_read"] and 0 < st["tot_new_attached"]
            assert st["tot_recs_read"] + st["tot_new_attached"] <= st["tot_insts_in"], st
            assert 0 < st["tot_recs_written"] <= st["tot_recs_read"] + st["tot_new_attached"], st
            assert 0 < st["tot_entry_items"] <= st["tot_recs_read"] + st["tot_new_attached"], st
            assert st["tot_new_attached"] <= st["tot_entry_items"], st      # (a newly attached instance has an entry token by construction)
            assert 0 < st["tot_arcs_walked"] <= st["tot_arcs_visited"], st
            assert st["tot_items_expanded"] >= st["tot_proc_end_hyps"] > 0, st
            assert st["tot_closure_items"] > 0, st                          # (the tee model between words: closure items every word end)
            assert 0 < st["tot_bids_placed"] < st["tot_active_end_hyps"], st  # (REC_SOLE: the inside of a word's chain places none)
        seen[slot] = [{k: g.stats[k] for k in ("tot_recs_read", "tot_new_attached", "tot_recs_written", "tot_entry_items", "tot_bids_placed")} for g in gs]
        gd.close()
    assert seen[False] == seen[True], (seen[False], seen[True])
