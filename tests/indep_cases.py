"""Graphs of 10^4 arcs and more for the independent anchor (tests/indep_viterbi_np.py): epsilon closures (one eps:word arc
per hub word), the tee model between words, a lexicon-tree hub, HMMs of 1 to 6 emitting states with skips."""
from juicer_amd import synth

CASES = {
    "flat_hub": lambda: synth.config_small(seed=21, n_utts=3, n_words=400, n_succ=6, n_gmm=120, n_hmm=61, utt_words=(4, 8)),
    "tree_hub": lambda: synth.config_small(seed=22, n_utts=3, n_words=400, n_succ=6, n_gmm=120, n_hmm=61, utt_words=(4, 8), hub="tree"),
    "mixed_topologies": lambda: synth.config_mixed(seed=23, n_utts=3, n_words=350, n_succ=6),
}


def check_against_viterbi(hyp, ref, what=""):
    """hyp: a decoder's result (n, label, time newest first, tot_ac, tot_lm); ref: viterbi()'s (score, [(label, frame)])"""
    if ref is None:
        assert hyp.n == -1, what
        return
    assert hyp.n == len(ref[1]), "%s: %d words, independent Viterbi %d" % (what, hyp.n, len(ref[1]))
    got = list(zip(hyp.label[::-1].tolist(), hyp.time[::-1].tolist()))
    assert got == ref[1], "%s: words / times differ\n%s\n%s" % (what, got, ref[1])
    tot = float(hyp.tot_ac) + float(hyp.tot_lm)
    assert abs(tot - ref[0]) <= 1e-5 * abs(ref[0]), "%s: total score %.6f, independent Viterbi %.6f" % (what, tot, ref[0])
