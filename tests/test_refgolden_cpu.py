"""The CPU oracle against tests/golden/refbase_golden.json - outputs of the REFERENCE's own compiled WFSTDecoderLite (tools/refbase,
stand-in build, generated in the build container by tests/golden/make_refbase_golden.py).  Unlike tests/test_refdiff_cpu.py this
needs no /root/reference: it runs on the GPU box too, so the checker that the `-m gpu` tests use is held to the reference's object
code wherever those tests run."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
REF_STATS = ("tot_active_emit_hyps", "tot_active_end_hyps", "tot_active_models", "tot_proc_emit_hyps", "tot_proc_end_hyps")


def load_case(case):
    import make_golden
    import make_refbase_golden
    g = json.load(open(os.path.join(HERE, "golden", "refbase_golden.json")))[case]
    am, net, feats, _ = make_refbase_golden.CASES[case][0]()
    assert make_golden.input_digest(am, net, feats) == g["input_sha256"], "synthetic generator changed: regenerate the golden file"
    return g, am, net, feats


def same_as_golden(h, want):
    """words, times, every score and the totals bit for bit"""
    if h.n != want["n"]:
        return False
    if want["n"] <= 0:
        return True
    unhex = lambda hs: np.frombuffer(bytes.fromhex("".join(hs)), np.float32)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    return bool(h.label.tolist() == want["label"] and h.time.tolist() == want["time"]
                and all(np.array_equal(bits(getattr(h, f)), bits(unhex(want[f + "_hex"]))) for f in ("score", "ac", "lm"))
                and np.array_equal(bits([h.tot_score, h.tot_ac, h.tot_lm]), bits(unhex(want["tot"]))))


@pytest.mark.parametrize("case", ["toy", "small", "mixed", "c2_small", "configs1_first8"])
def test_oracle_equals_the_reference_built_vectors(built, case):
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    g, am, net, feats = load_case(case)
    onet, oam = OracleNet(net), OracleAM(am)
    n = 0
    for run in g["runs"]:
        od = OracleDecoder(onet, oam, **run["beams"])
        for u, (f, want) in enumerate(zip(feats, run["utts"])):
            o = od.decode(f)
            assert same_as_golden(o, want), (case, run["beams"], u)
            for k in REF_STATS:
                assert int(o.stats[k]) == want["stats"][k], (case, run["beams"], u, k)
            n += 1
    assert n >= 6


@pytest.mark.parametrize("case", ["toy", "small", "mixed", "c2_small", "configs1_first8"])
def test_oracle_scoring_equals_the_reference_built_log_likelihoods(built, case):
    """HTKFlatModels::calcOutput (calcGMMOutput + logAdd, src/HTKFlatModels.cpp:202-293) run by the reference's own object code on the first
    frames of utterance 0, every tied state: the golden file holds a digest of the float bits"""
    import make_golden
    from oracle.oracle import OracleAM
    g, am, net, feats = load_case(case)
    ll = OracleAM(am).score_frames(feats[0][:g["ll_frames"]])
    assert [np.float32(v).tobytes().hex() for v in ll[0, :8]] == g["ll_first"]
    assert make_golden.digest(ll) == g["ll_sha256"]


def load_random_cases():
    """tests/golden/refbase_random_golden.json: the reference's own outputs on random graphs of arbitrary shape (make_refbase_random_golden.py)"""
    import make_golden
    sys.path.insert(0, HERE)
    import test_gpu_random_topology as trt
    g = json.load(open(os.path.join(HERE, "golden", "refbase_random_golden.json")))
    out = []
    for c in g["cases"]:
        am, net, feats, kw = trt._case(c["seed"])
        assert make_golden.input_digest(am, net, feats) == c["input_sha256"] and kw == c["beams"], "random generator changed: regenerate the golden file"
        out.append((c, am, net, feats))
    return out


def test_oracle_equals_the_reference_built_vectors_on_random_topologies(built):
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    n = found = 0
    for c, am, net, feats in load_random_cases():
        od = OracleDecoder(OracleNet(net), OracleAM(am), **c["beams"])
        for u, (f, want) in enumerate(zip(feats, c["utts"])):
            o = od.decode(f)
            assert same_as_golden(o, want), (c["seed"], c["beams"], u)
            for k in REF_STATS:
                assert int(o.stats[k]) == want["stats"][k], (c["seed"], u, k)
            n += 1; found += want["n"] > 0
    assert n >= 40 and found >= 30
