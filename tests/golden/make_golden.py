"""Regenerates tests/golden/oracle_golden.json.

PARITY UNPINNED: idiap/juicer ships no golden vectors for this path and cannot be
built in this image, so these vectors are outputs of the CPU oracle (the C
restatement under oracle/) on the deterministic synthetic configs, frozen as
regression pins for both the oracle and the HIP path.  Inputs are regenerated
from seeds; their SHA-256 digests are stored so a generator change is detected.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from juicer_amd import synth                                    # noqa: E402
from oracle.oracle import OracleAM, OracleDecoder, OracleNet    # noqa: E402

CASES = {
    "toy": (lambda: synth.config_toy(), [
        dict(), dict(main_beam=200.0),
        dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0, max_hyps=50)]),
    "small": (lambda: synth.config_small(), [
        dict(main_beam=150.0), dict(main_beam=150.0, max_hyps=200),
        dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150)]),
    # HMMs with 1..6 emitting states, skips, double entries, tee model (the 8-lane instance layout)
    "mixed": (lambda: synth.config_mixed(), [
        dict(main_beam=150.0), dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0),
        dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150)]),
}


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def input_digest(am, net, feats):
    return digest(am.weight, am.mean, am.var, am.transp, am.hmm_gmm, am.hmm_tm, net.src, net.dst, net.ilab,
                  net.olab, net.w_file, net.fstate, net.fweight_file, *feats)


def main():
    out = {"_note": "oracle-generated regression pins; parity unpinned (see make_golden.py)"}
    for name, (mk, beams) in CASES.items():
        am, net, feats, words = mk()
        onet, oam = OracleNet(net), OracleAM(am)
        case = {"input_sha256": input_digest(am, net, feats), "n_arcs": net.n_arcs, "runs": []}
        x = np.concatenate(feats)[:64]
        ll = oam.score_frames(x)
        case["gmm_ll_sha256"] = digest(ll)
        case["gmm_ll_first"] = [float(v) for v in ll[0, :8]]
        for kw in beams:
            od = OracleDecoder(onet, oam, **kw)
            utts = []
            for f in feats:
                h = od.decode_certified(f)          # result independent of the visiting order of equal-score tokens
                utts.append({"n": h.n, "label": h.label.tolist(), "time": h.time.tolist(),
                             "score_hex": [np.float32(v).tobytes().hex() for v in h.score],
                             "ac_hex": [np.float32(v).tobytes().hex() for v in h.ac],
                             "lm_hex": [np.float32(v).tobytes().hex() for v in h.lm],
                             "tot": [np.float32(v).tobytes().hex() for v in (h.tot_score, h.tot_ac, h.tot_lm)],
                             "stats": {k: int(v) for k, v in h.stats.items()}})
            case["runs"].append({"beams": kw, "utts": utts})
        out[name] = case
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
