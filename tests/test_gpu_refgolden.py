"""The HIP path held DIRECTLY to what the reference's own compiled classes produced (tests/golden/refbase_golden.json: WFSTDecoderLite
from /root/reference/src, built against stand-ins in the build container - tests/golden/make_refbase_golden.py): words, times, every
score bit for bit, the reference's five statistics - no oracle in between.  (Still "parity unpinned": a stand-in build is not a
reference build.  But the kernels, the oracle and the reference's object code now meet pairwise.)"""
import pytest

from test_refgolden_cpu import REF_STATS, load_case, same_as_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("slot", [False, True])
@pytest.mark.parametrize("case", ["toy", "small", "mixed", "c2_small", "configs1_first8"])
def test_hip_path_equals_the_reference_built_vectors(built, monkeypatch, case, slot):
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    if slot:                                                # ... through the slot kernel as well as through clusters of k_search
        monkeypatch.setenv("JD_DEV", "1"); monkeypatch.setenv("JD_CW", "1"); monkeypatch.setenv("JD_SLOT_BATCH", "1")
    g, am, net, feats = load_case(case)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    checked = order_dependent = 0
    for run in g["runs"]:
        kw = run["beams"]
        big = (1 << 25) if kw.get("main_beam", 0.0) in (0.0, 200.0) and not kw.get("max_hyps") else 0
        gd = capi.Decoder(gnet, gam, max_streams=len(feats), max_paths=big, **kw)
        hyps = gd.decode_batch(feats)
        assert (gd.last_timing()["slot_launches"] > 0) == slot
        for u, (h, want) in enumerate(zip(hyps, run["utts"])):
            if not same_as_golden(h, want):
                # the one licence: two EQUAL-scored tokens met at a state and the reference kept the one its list order visited first
                # (oracle: stats["ties"], decode_certified) - an artefact no other traversal can be held to
                od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
                assert od.decode(feats[u]).stats["ties"] > 0, (case, kw, u)
                with pytest.raises(AssertionError):
                    od.decode_certified(feats[u])
                order_dependent += 1
                continue
            for k in REF_STATS:
                assert int(h.stats[k]) == want["stats"][k], (case, kw, u, k)
            checked += 1
        gd.close()
    assert checked >= 6 and order_dependent == 0, (checked, order_dependent)


@pytest.mark.parametrize("case", ["toy", "small", "mixed", "c2_small", "configs1_first8"])
def test_scoring_kernel_equals_the_reference_built_log_likelihoods(built, case):
    """the companion kernel's table cells against the reference's own HTKFlatModels::calcOutput (digest of the float bits in the golden file)"""
    import numpy as np
    import make_golden
    from juicer_amd import capi
    g, am, net, feats = load_case(case)
    ll = capi.Models.from_htk(am).score_frames(feats[0][:g["ll_frames"]])
    assert [np.float32(v).tobytes().hex() for v in ll[0, :8]] == g["ll_first"]
    assert make_golden.digest(ll) == g["ll_sha256"]


@pytest.mark.parametrize("slot", [False, True])
def test_hip_path_equals_the_reference_built_vectors_on_random_topologies(built, monkeypatch, slot):
    """tests/golden/refbase_random_golden.json: graphs of arbitrary shape (any in- and out-degree, parallel arcs, self loops, epsilon and
    tee arcs anywhere, labels and final weights anywhere, arcs into the initial state), decoded by the reference's own compiled
    WFSTDecoderLite in the build container; here by the HIP path, through both kernels: words, times, every score bit for bit, the
    reference's five statistics."""
    from juicer_amd import capi
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    from test_refgolden_cpu import load_random_cases
    if slot:
        monkeypatch.setenv("JD_DEV", "1"); monkeypatch.setenv("JD_CW", "1"); monkeypatch.setenv("JD_SLOT_BATCH", "1")
    checked = found = order_dependent = 0
    for c, am, net, feats in load_random_cases():
        kw = c["beams"]
        gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=len(feats), **kw)
        hyps = gd.decode_batch(feats)
        assert (gd.last_timing()["slot_launches"] > 0) == slot
        for u, (h, want) in enumerate(zip(hyps, c["utts"])):
            if not same_as_golden(h, want):                            # (the one licence, as above: an order-dependent tie inside the reference)
                od = OracleDecoder(OracleNet(net), OracleAM(am), **kw)
                assert od.decode(feats[u]).stats["ties"] > 0, (c["seed"], kw, u)
                with pytest.raises(AssertionError):
                    od.decode_certified(feats[u])
                order_dependent += 1
                continue
            for k in REF_STATS:
                assert int(h.stats[k]) == want["stats"][k], (c["seed"], kw, u, k)
            checked += 1; found += want["n"] > 0
        gd.close()
    assert checked >= 40 and found >= 30 and order_dependent == 0, (checked, found, order_dependent)
