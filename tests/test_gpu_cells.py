"""jd_dec_debug_cells: the part of a batch's likelihood table the search reads (SURVEY.md 8d's Ug)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cells_read_by_the_search(built):
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small(n_utts=3)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    frames = sum(f.shape[0] for f in feats)
    got = {}
    for name, kw in (("beam", dict(main_beam=120.0)), ("wide", dict())):
        dec = capi.Decoder(gnet, gam, max_streams=3, **kw)
        want = dec.decode_batch(feats)
        dec.debug_cells(True)
        hyps = dec.decode_batch(feats)
        read, total = dec.debug_cells(False)
        assert total == frames * am.n_gmm and 0 < read <= total
        # every emitting hypothesis that passed the threshold reads one cell; several may share it
        emit = sum(h.stats["tot_proc_emit_hyps"] for h in hyps)
        assert read <= emit
        for a, b in zip(hyps, want):                                   # marking changes nothing
            assert a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32))
        assert dec.debug_cells(False) == (0, total)                    # (switched off: nothing marked)
        got[name] = read / total
        dec.close()
    assert got["beam"] < got["wide"]                                   # a beam leaves tied states unasked for
    # the un-pruned search asks for every tied state some arc of the graph can reach in a frame it can be in
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=120.0)
    o = od.decode(feats[0])
    assert o.stats["tot_proc_emit_hyps"] > 0
