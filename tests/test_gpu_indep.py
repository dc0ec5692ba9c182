"""The HIP path against an anchor that shares nothing with the oracle: tests/indep_viterbi_np.py, a float64
full-trellis Viterbi over all arcs with its own GMM evaluation.  Every beam is off, so the decoder must return
the best path there is - words, frames and total score - on graphs of more than 10^4 arcs with epsilon closures,
tee models, a lexicon-tree hub and mixed topologies.  Nothing under oracle/ is imported here."""
import numpy as np
import pytest

import indep_cases
import indep_viterbi_np as iv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", sorted(indep_cases.CASES))
def test_hip_path_vs_independent_viterbi(built, case):
    from juicer_amd import capi
    am, net, feats, _ = indep_cases.CASES[case]()
    assert net.n_arcs >= 10000
    dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=len(feats))   # every beam disabled
    hyps = dec.decode_batch(feats)
    for u, x in enumerate(feats):
        ref = iv.viterbi(net, am, iv.gmm_loglik(am, x))
        assert ref is not None
        indep_cases.check_against_viterbi(hyps[u], ref, "%s utt %d" % (case, u))
    # ... and through the streaming interface (IDecoder's init / processFrame / finish), one utterance
    dec.stream_init(0)
    for i in range(0, feats[0].shape[0], 37):
        dec.stream_push(0, feats[0][i:i + 37])
    indep_cases.check_against_viterbi(dec.stream_finish(0), iv.viterbi(net, am, iv.gmm_loglik(am, feats[0])), case + " streamed")
    dec.close()


def test_scaled_graph_vs_independent_viterbi(built):
    """lmScale / insertion penalty applied by the network loader (WFSTNetwork.cpp:481-486)"""
    from juicer_amd import capi
    am, net, feats, _ = indep_cases.CASES["tree_hub"]()
    dec = capi.Decoder(capi.Network.from_synth(net, 2.5, -1.5), capi.Models.from_htk(am), max_streams=1)
    h = dec.decode_batch(feats[:1])[0]
    indep_cases.check_against_viterbi(h, iv.viterbi(net, am, iv.gmm_loglik(am, feats[0]), lm_scale=2.5, ins_penalty=-1.5), "scaled")
    dec.close()
