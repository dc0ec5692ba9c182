"""Random WFSTs of ARBITRARY shape for the parity tests (no lexicon, no language model: states with any number of arcs in and out,
parallel arcs, self loops, epsilon arcs with and without labels, tee models anywhere, final weights anywhere, an initial state
with arcs into it) - the shapes the generators of juicer_amd/synth.py never make, and the ones the decoder's structural
decisions look at: which arcs are alone into their state (csrc/jd_search.h: REC_SOLE), how the states are numbered and where
their words sit (jd_dec_create).  Epsilon arcs only lead to higher state numbers (an epsilon cycle would be an endless
recursion in the reference's propagateToken, WFSTDecoderLite.cpp:533-540; arcs with the tee model likewise, :584-600)."""
import numpy as np

from juicer_amd import synth


def random_net(seed, am, n_states=40, arcs_per_state=2.2, p_eps=0.18, p_label=0.3, p_final=0.3, n_words=50, p_chain=0.5, hub_fanout=0):
    rng = np.random.default_rng(seed)
    n_model = am.n_hmm
    src, dst, il, ol, w = [], [], [], [], []

    def arc(s, d, eps):
        src.append(s); dst.append(d)
        lab = 0 if eps else int(rng.integers(1, n_model + 1))
        if lab - 1 == am.sp_hmm and s >= d:                            # (a tee model passes tokens on within the frame: forward only, like epsilon)
            lab = 1 + (am.sp_hmm + 1) % n_model
        il.append(lab)
        ol.append(int(rng.integers(1, n_words + 1)) if rng.random() < p_label else 0)
        w.append(float(rng.choice([0.0, 0.0, rng.uniform(0.1, 6.0)])))

    # a spine that makes every state reachable; half of it chains (one arc in, one arc out), the rest joins other states
    order = rng.permutation(n_states)
    init = int(order[0])
    for k in range(1, n_states):
        d = int(order[k])
        s = int(order[k - 1]) if rng.random() < p_chain else int(order[rng.integers(0, k)])
        arc(s, d, eps=bool(rng.random() < p_eps and s < d))
    n_extra = int(n_states * arcs_per_state) - (n_states - 1)
    for _ in range(max(0, n_extra)):
        s, d = int(rng.integers(0, n_states)), int(rng.integers(0, n_states))
        arc(s, d, eps=bool(rng.random() < p_eps and s < d))
    if hub_fanout:                                                     # one state with thousands of arcs (the search slices such rows)
        hub = int(order[min(2, n_states - 1)])
        for _ in range(hub_fanout): arc(hub, int(rng.integers(0, n_states)), eps=False)
    # FSM convention: the arcs of a state stand together, and the source of the first arc is the initial state
    if init not in src: arc(init, int(order[1]), eps=False)
    sa = np.asarray(src)
    idx = np.lexsort((np.arange(sa.size), sa, sa != init))
    fin = np.flatnonzero(rng.random(n_states) < p_final).astype(np.int32)
    if fin.size == 0: fin = np.asarray([int(order[-1])], np.int32)
    return synth.SynthNet(n_states=n_states, src=np.asarray(src, np.int32)[idx], dst=np.asarray(dst, np.int32)[idx],
                          ilab=np.asarray(il, np.int32)[idx], olab=np.asarray(ol, np.int32)[idx], w_file=np.asarray(w, np.float32)[idx],
                          fstate=fin, fweight_file=rng.uniform(0.0, 3.0, size=fin.size).astype(np.float32), n_words=n_words)


def random_walk_features(seed, net, am, n_arcs=12, noise=1.0):
    """Frames along a random walk through the network's model arcs (so that something survives the beams)."""
    rng = np.random.default_rng(seed)
    out = {}
    for i, s in enumerate(net.src.tolist()): out.setdefault(s, []).append(i)
    s = int(net.src[0]); g = []
    for _ in range(n_arcs * 3):
        if s not in out: break
        a = int(rng.choice(out[s]))
        hm = int(net.ilab[a]) - 1
        if hm >= 0 and hm != am.sp_hmm:
            n = int(am.hmm_nstates[hm])
            for j in range(1, n - 1):
                for _ in range(int(rng.integers(1, 4))): g.append(int(am.hmm_gmm[hm, j]))
        s = int(net.dst[a])
        if len(g) > 8 * n_arcs: break
    if not g: g = [0] * 10
    g = np.asarray(g, np.int64)
    mu = am.mean[g, 0]; sd = np.sqrt(am.var[g, 0])
    return (mu + noise * sd * rng.normal(size=mu.shape)).astype(np.float32)
