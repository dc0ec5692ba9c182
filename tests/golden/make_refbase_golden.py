"""Regenerates tests/golden/refbase_golden.json: outputs of the REFERENCE's OWN WFSTDecoderLite (tools/refbase: its translation units
compiled from /root/reference/src where they lie, against stand-ins for three absent third-party headers) on the deterministic
synthetic configs.  Run in the build container only (the GPU box has no /root/reference); what is committed is DATA - inputs are
regenerated from seeds (their SHA-256 digests are stored), expected outputs are the driver's JSON.

These vectors let the GPU box hold the HIP path DIRECTLY to what the reference's compiled classes produced here, with no oracle in
between (tests/test_gpu_refgolden.py), and the oracle to the same files where /root/reference is absent (tests/test_oracle_cpu.py).
A build against stand-ins is not a reference build: `parity` stays "unpinned"  - but a shared misreading of the
reference by the oracle and the kernels would have to be shared by the reference's own object code too to pass these.

    python tests/golden/make_refbase_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "refbase"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refdiff                                                   # noqa: E402
from juicer_amd import synth                                    # noqa: E402
from make_golden import digest, input_digest                    # noqa: E402

BEAMS = [dict(), dict(main_beam=200.0), dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0), dict(main_beam=150.0, max_hyps=200),
         dict(max_hyps=300), dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150)]
CASES = {
    "toy": (lambda: synth.config_toy(), BEAMS),
    "small": (lambda: synth.config_small(), BEAMS),
    "mixed": (lambda: synth.config_mixed(), [BEAMS[1], BEAMS[2], BEAMS[3], BEAMS[5]]),
    "c2_small": (lambda: synth.config_c2(seed=0, n_utts=6, target_arcs=60_000, n_gmm=300, n_hmm=800, n_mix=8, n_words=500),
                 [dict(main_beam=150.0), dict(main_beam=150.0, max_hyps=600)]),
    # BASELINE.json configs[1] itself (991,848 arcs, 3000 x 16 mixtures): the first eight utterances of the bench's batch
    "configs1_first8": (lambda: synth.config_c2(seed=0, n_utts=8), [dict(main_beam=150.0), dict(main_beam=150.0, max_hyps=6000)]),
}
f32hex = lambda v: np.float32(v).tobytes().hex()
LL_FRAMES = 48


def main():
    assert refdiff.available(), "the build container only"
    out = {"_note": "outputs of the reference's own WFSTDecoderLite, compiled against stand-ins (tools/refbase): see make_refbase_golden.py"}
    for name, (mk, beams) in CASES.items():
        am, net, feats, _ = mk()
        case = {"input_sha256": input_digest(am, net, feats), "n_arcs": int(net.n_arcs), "runs": []}
        # the reference's own HTKFlatModels::calcOutput for every tied state of the first frames of utterance 0 (digest of the float bits)
        ll = refdiff.reference_log_likelihoods(am, feats[0], LL_FRAMES)
        case["ll_frames"] = int(ll.shape[0]); case["ll_sha256"] = digest(ll); case["ll_first"] = [f32hex(v) for v in ll[0, :8]]
        for kw in beams:
            rows, (rc, err, _) = refdiff.run_reference(am, net, feats, kw)
            assert rc == 0 and len(rows) == len(feats), (name, kw, rc, err)
            utts = []
            for r in rows:
                u = {"n": r["n"], "stats": {k: int(v) for k, v in r["stats"].items()}}
                if r["n"] > 0:
                    u.update(label=r["label"], time=r["time"], score_hex=[f32hex(v) for v in r["score"]], ac_hex=[f32hex(v) for v in r["ac"]],
                             lm_hex=[f32hex(v) for v in r["lm"]], tot=[f32hex(v) for v in r["tot"]])
                utts.append(u)
            case["runs"].append({"beams": kw, "utts": utts})
        out[name] = case
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refbase_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
