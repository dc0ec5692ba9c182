"""Shared comparison helpers for the parity tests."""
import numpy as np

# north_star: labels/times bit-exact, path/acoustic scores within 1e-4 relative
SCORE_RTOL = 1e-4

# the reference's own per-utterance statistics (WFSTDecoderLite.cpp:231-241) + instances processed
STAT_KEYS = ["n_frames", "tot_active_emit_hyps", "tot_active_end_hyps", "tot_active_models",
             "tot_proc_emit_hyps", "tot_proc_end_hyps", "tot_insts_in"]
# build counters where the GPU path may do LESS work than the reference: it expands only the
# best token per destination state (result-equivalent), so it visits fewer arcs / writes fewer Paths
LE_KEYS = ["tot_arcs_visited", "tot_paths"]


def rel_close(a, b, rtol=SCORE_RTOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b)))


def assert_hyp_matches(gpu, ora, what="", check_stats=True):
    """gpu: juicer_amd.capi.Hyp, ora: oracle.OracleHyp (or anything with the same fields)."""
    assert gpu.n == ora.n, "%s: n %d vs oracle %d" % (what, gpu.n, ora.n)
    if ora.n > 0:
        assert np.array_equal(gpu.label, ora.label), "%s: labels differ\n%s\n%s" % (what, gpu.label, ora.label)
        assert np.array_equal(gpu.time, ora.time), "%s: times differ\n%s\n%s" % (what, gpu.time, ora.time)
        for f in ("score", "ac", "lm"):
            assert rel_close(getattr(gpu, f), getattr(ora, f)), "%s: %s differ" % (what, f)
        for f in ("tot_score", "tot_ac", "tot_lm"):
            assert rel_close(getattr(gpu, f), getattr(ora, f)), "%s: %s differ" % (what, f)
    if check_stats:
        for k in STAT_KEYS:
            assert gpu.stats[k] == ora.stats[k], "%s: stat %s %d vs oracle %d" % (what, k, gpu.stats[k], ora.stats[k])
        for k in LE_KEYS:
            assert 0 < gpu.stats[k] <= ora.stats[k] or ora.stats[k] == 0, \
                "%s: stat %s %d vs oracle %d" % (what, k, gpu.stats[k], ora.stats[k])


def bit_exact(gpu, ora):
    if gpu.n != ora.n:
        return False
    if ora.n <= 0:
        return True
    ok = np.array_equal(gpu.label, ora.label) and np.array_equal(gpu.time, ora.time)
    for f in ("score", "ac", "lm"):
        ok = ok and np.array_equal(np.asarray(getattr(gpu, f), np.float32).view(np.uint32),
                                   np.asarray(getattr(ora, f), np.float32).view(np.uint32))
    return bool(ok)


def oracle_certified_many(net, am, feats, workers=None, **kw):
    """decode_certified of every utterance with the CPU oracle, on several host cores: one OracleDecoder per thread over
    the shared network / model handles (the C library keeps all mutable state in the decoder; ctypes releases the GIL
    while it runs).  A whole 64-utterance configs[1] batch costs: about 35 CPU-seconds at beam 150."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    onet, oam = (net if isinstance(net, OracleNet) else OracleNet(net)), (am if isinstance(am, OracleAM) else OracleAM(am))
    n = len(feats)
    workers = max(1, min(workers or (os.cpu_count() or 1), 16, n))
    order = sorted(range(n), key=lambda u: -feats[u].shape[0])          # longest first: the pool ends together
    out = [None] * n

    def work(k):
        od = OracleDecoder(onet, oam, **kw)
        for u in order[k::workers]:
            out[u] = od.decode_certified(feats[u])
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(work, range(workers)))
    return out
