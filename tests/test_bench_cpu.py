"""bench.py's host-side arithmetic (no GPU): the edit distance behind `wer_vs_oracle`, the design-bytes formula, the quoted reference
baselines, and the contract that every string inside config / roofline / cpu_baseline stays short enough for the driver's record."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_word_errors_is_levenshtein():
    assert bench.word_errors([], []) == 0
    assert bench.word_errors([1, 2, 3], [1, 2, 3]) == 0
    assert bench.word_errors([1, 2, 3], []) == 3 and bench.word_errors([], [4, 5]) == 2
    assert bench.word_errors([1, 2, 3], [1, 3]) == 1            # one insertion
    assert bench.word_errors([1, 3], [1, 2, 3]) == 1            # one deletion
    assert bench.word_errors([1, 9, 3], [1, 2, 3]) == 1         # one substitution
    assert bench.word_errors([7, 8], [1, 2, 3]) == 3
    rng = np.random.default_rng(0)
    for _ in range(50):                                         # symmetric, bounded by the longer sequence, zero only when equal
        a, b = rng.integers(0, 4, rng.integers(0, 8)).tolist(), rng.integers(0, 4, rng.integers(0, 8)).tolist()
        d = bench.word_errors(a, b)
        assert d == bench.word_errors(b, a) and abs(len(a) - len(b)) <= d <= max(len(a), len(b)) and (d == 0) == (a == b)


def test_wer_vs_oracle_against_itself(built):
    """the oracle's own hypotheses stand in for the GPU's: WER 0, everything identical - and one changed word shows"""
    from juicer_amd import synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, refs = synth.config_small(n_utts=3)
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=150.0)
    hyps = [od.decode(x) for x in feats]
    w = bench.wer_vs_oracle(net, am, feats, hyps, 150.0, 0, workers=2)
    assert w["wer"] == 0.0 and w["identical_1best"] == 3 and w["identical_scores_bitwise"] == 3 and w["utterances"] == 3 and w["ref_words"] > 0
    hyps[1].label = hyps[1].label.copy(); hyps[1].label[0] += 1
    w = bench.wer_vs_oracle(net, am, feats, hyps, 150.0, 0, workers=2)
    assert w["word_errors"] == 1 and w["identical_1best"] == 2 and 0.0 < w["wer"] < 0.2


def test_design_bytes_prices_every_counter():
    keys = ["tot_recs_read", "tot_new_attached", "tot_entry_items", "tot_recs_written", "tot_active_end_hyps", "tot_proc_emit_hyps",
            "tot_items_expanded", "tot_proc_end_hyps", "tot_arcs_walked", "tot_closure_items", "tot_paths", "tot_bids_placed"]
    zero = {k: 0 for k in keys}
    assert bench.design_bytes(zero, 5, 3000, 10, row_in_lds=False) == 0.0
    assert bench.design_bytes(zero, 5, 3000, 10, row_in_lds=True) == 10 * 3000 * 4.0          # the likelihood row, once per frame
    for k in keys:                                               # every counter moves the figure (k_search prices P, the slot kernel the row)
        one = dict(zero, **{k: 1})
        assert bench.design_bytes(one, 5, 3000, 0, row_in_lds=False) > 0.0 or k == "tot_proc_emit_hyps", k
    assert bench.design_bytes(dict(zero, tot_recs_read=1), 8, 10, 0, False) - bench.design_bytes(dict(zero, tot_recs_read=1), 5, 10, 0, False) == 64.0


def test_reference_cpu_constants_come_from_the_committed_file():
    r = bench.reference_cpu()
    f = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_baseline.json")))
    assert r["B1_WFSTDecoderLite_fps"] == f["WFSTDecoderLite"]["frames_per_s"] and r["B2_WFSTDecoderLiteThreading_fps"] == f["WFSTDecoderLiteThreading"]["frames_per_s"]
    assert r["B1_identical_to_oracle"] == "64/64" and r["B2_identical_to_oracle"] == "64/64"
    assert all(len(v) < 128 for v in r.values() if isinstance(v, str)), r      # (the driver's record cuts longer strings)
