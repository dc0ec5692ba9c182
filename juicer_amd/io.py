"""Writers for the build's own on-disk containers consumed by jd_batch_test
(the DecoderBatchTest counterpart): .jdam model dumps, .jdf feature files, AT&T text FSMs."""
from __future__ import annotations

import numpy as np

JDAM_MAGIC = 0x4D41444A  # 'JDAM'


def write_fsm(path, net):
    """AT&T text FSM in the format WFSTNetwork::WFSTNetwork(text) parses (WFSTNetwork.cpp:414-447)."""
    with open(path, "w") as f:
        for i in range(net.n_arcs):
            f.write("%d %d %d %d %.9g\n" % (net.src[i], net.dst[i], net.ilab[i], net.olab[i], net.w_file[i]))
        for s, w in zip(net.fstate, net.fweight_file):
            f.write("%d %.9g\n" % (s, w))


def write_jdam(path, am):
    with open(path, "wb") as f:
        np.asarray([JDAM_MAGIC, am.D, am.n_gmm, am.max_mix, am.n_hmm, am.max_n, am.n_tm], np.int32).tofile(f)
        for a, dt in ((am.n_mix, np.int32), (am.weight, np.float32), (am.mean, np.float32), (am.var, np.float32),
                      (am.hmm_nstates, np.int32), (am.hmm_gmm, np.int32), (am.hmm_tm, np.int32),
                      (am.tm_nstates, np.int32), (am.transp, np.float32)):
            np.ascontiguousarray(a, dtype=dt).tofile(f)


def write_jdf(path, feats):
    x = np.ascontiguousarray(feats, dtype=np.float32)
    with open(path, "wb") as f:
        np.asarray([x.shape[0], x.shape[1]], np.int32).tofile(f)
        x.tofile(f)


def write_mmf(path, am, inline_every: int = 0):
    """HTK MMF text with shared states (~s) and transition matrices (~t).  inline_every > 0 writes
    every inline_every-th HMM with inline states / <TRANSP> instead (other grammar branch)."""
    def vec(v):
        return " ".join("%.9g" % float(x) for x in v)
    with open(path, "w") as f:
        f.write("~o\n<STREAMINFO> 1 %d\n<VECSIZE> %d<NULLD><MFCC_E_D_A><DIAGC>\n" % (am.D, am.D))
        f.write('~v "varFloor1"\n<VARIANCE> %d\n %s\n' % (am.D, vec(np.full(am.D, 0.01))))
        for t in range(am.n_tm):
            n = int(am.tm_nstates[t])
            f.write('~t "T_%d"\n<TRANSP> %d\n' % (t, n))
            for i in range(n):
                f.write(" " + vec(am.transp[t, i, :n]) + "\n")

        def state_body(g):
            nm = int(am.n_mix[g])
            out = []
            if nm > 1:
                out.append("<NUMMIXES> %d" % nm)
            for m in range(nm):
                if nm > 1:
                    out.append("<MIXTURE> %d %.9g" % (m + 1, float(am.weight[g, m])))
                out.append("<MEAN> %d\n %s" % (am.D, vec(am.mean[g, m])))
                out.append("<VARIANCE> %d\n %s" % (am.D, vec(am.var[g, m])))
                out.append("<GCONST> %.6e" % 0.0)
            return "\n".join(out) + "\n"
        for g in range(am.n_gmm):
            f.write('~s "s_%d"\n' % g + state_body(g))
        for h in range(am.n_hmm):
            n = int(am.hmm_nstates[h])
            inline = inline_every > 0 and h % inline_every == inline_every - 1
            f.write('~h "h_%d"\n<BEGINHMM>\n<NUMSTATES> %d\n' % (h, n))
            for j in range(1, n - 1):
                g = int(am.hmm_gmm[h, j])
                f.write("<STATE> %d\n" % (j + 1))
                f.write(state_body(g) if inline else '~s "s_%d"\n' % g)
            t = int(am.hmm_tm[h])
            if inline:
                f.write("<TRANSP> %d\n" % n)
                for i in range(n):
                    f.write(" " + vec(am.transp[t, i, :n]) + "\n")
            else:
                f.write('~t "T_%d"\n' % t)
            f.write("<ENDHMM>\n")


# ---------------------------------------------------------------- Juicer's binary caches
# Independent writers of the layouts WFSTNetwork::writeBinary (WFSTNetwork.cpp:1106-1225),
# WFSTAlphabet::writeBinary (:213-247) and HTKModels::output(.., true) (HTKModels.cpp:1044-1105)
# produce - used to feed jd_net_load_jwnt / jd_am_load_jmbi with files the library did not
# write itself (alphabets, names, shared vector pools, arcs in file order).

def _name(f, s):
    if s is None:
        np.asarray([0], "<i4").tofile(f)
    else:
        b = s.encode() + b"\0"
        np.asarray([len(b)], "<i4").tofile(f)
        f.write(b)


def _alphabet(f, labels, aux=()):
    """labels: list indexed by label id (None = unused id)."""
    f.write(b"JWAL")
    np.asarray([len(labels) - 1, sum(l is not None for l in labels)], "<i4").tofile(f)
    if labels:
        for l in labels:
            _name(f, l)
        is_aux = np.zeros(len(labels), np.uint8)
        for a in aux:
            is_aux[a] = 1
        np.asarray([int(is_aux.sum())], "<i4").tofile(f)
        is_aux.tofile(f)


def write_jwnt(path, net, in_labels=None, out_labels=None, aux_in=(), stored_w=None):
    """JWNT from an arc list in FSM file order (SynthNet).  Stored arc weights are the negated
    file weights (no scale, no penalty), final weights likewise - what writeBinary leaves after
    a text load with lmScale 1.  stored_w overrides the arc weights."""
    src = np.asarray(net.src, np.int32)
    S = int(max(src.max(), np.asarray(net.dst).max(), np.asarray(net.fstate).max(initial=0))) + 1
    first = np.full(S, -1, np.int64); cnt = np.zeros(S, np.int64)
    for i, s in enumerate(src):
        if first[s] < 0:
            first[s] = i
        cnt[s] += 1
    used = np.zeros(S, bool); used[src] = True; used[np.asarray(net.dst)] = True; used[np.asarray(net.fstate)] = True
    fin_ind = np.full(S, -1, np.int32)
    for i, s in enumerate(net.fstate):
        fin_ind[s] = i
    w = (-np.asarray(net.w_file, np.float32)).astype(np.float32) if stored_w is None else np.asarray(stored_w, np.float32)
    with open(path, "wb") as f:
        f.write(b"JWNT")
        np.asarray([src[0], S - 1, int(used.sum()), int(cnt.max()),
                    int(max(np.asarray(net.ilab).max(), np.asarray(net.olab).max())) + 1, -1, -1], "<i4").tofile(f)
        for s in range(S):
            np.asarray([s if used[s] else -1, fin_ind[s], cnt[s]], "<i4").tofile(f)
            if cnt[s]:
                np.arange(first[s], first[s] + cnt[s], dtype="<i4").tofile(f)
        np.asarray([len(net.fstate)], "<i4").tofile(f)
        for s, fw in zip(net.fstate, net.fweight_file):
            np.asarray([s], "<i4").tofile(f)
            np.asarray([-np.float32(fw)], "<f4").tofile(f)
        n = src.shape[0]
        np.asarray([n], "<i4").tofile(f)
        rec = np.zeros(n, np.dtype([("id", "<i4"), ("to", "<i4"), ("w", "<f4"), ("in", "<i4"), ("out", "<i4")]))
        rec["id"] = np.arange(n); rec["to"] = net.dst; rec["w"] = w; rec["in"] = net.ilab; rec["out"] = net.olab
        rec.tofile(f)
        for labels, aux in ((in_labels, aux_in), (out_labels, ())):
            if labels is None:
                f.write(b"\0")
            else:
                f.write(b"\1")
                _alphabet(f, labels, aux)
        f.write(b"JWNT")


def write_jmbi(path, am, derived, share_vars: bool = False, hmm_names=None):
    """JMBI from HTK-level parameters (SynthAM) plus the derived values a loader computed
    (`derived` = dict(sum_log_var[g,m], log_weight[g,m], trP[t,i,j]) - the reference stores
    them in the file).  share_vars pools identical variance vectors (shared ~v macros)."""
    D, G = am.D, am.n_gmm
    with open(path, "wb") as f:
        means, vars_, var_slv, mix = [], [], [], []
        var_key = {}
        for g in range(G):
            mi, vi = [], []
            for m in range(int(am.n_mix[g])):
                means.append(am.mean[g, m]); mi.append(len(means) - 1)
                key = am.var[g, m].tobytes() if share_vars else (g, m)
                if key not in var_key:
                    var_key[key] = len(vars_)
                    vars_.append(am.var[g, m]); var_slv.append(derived["sum_log_var"][g, m])
                vi.append(var_key[key])
            mix.append((mi, vi))
        f.write(b"JMBI")
        np.asarray([D, len(means), len(vars_), G, G, am.n_tm, am.n_hmm], "<i4").tofile(f)
        for i, mu in enumerate(means):
            f.write(b"JMMN"); _name(f, "mu%d" % i if i % 3 == 0 else None); np.asarray(mu, "<f4").tofile(f)
        for i, v in enumerate(vars_):
            f.write(b"JMVR"); _name(f, None)
            v32 = np.asarray(v, np.float32)
            v32.astype("<f4").tofile(f)
            (np.float32(-0.5) / v32).astype("<f4").tofile(f)
            np.asarray([var_slv[i]], "<f4").tofile(f)
        for g, (mi, vi) in enumerate(mix):
            f.write(b"JMMX"); _name(f, None)
            np.asarray([len(mi)], "<i4").tofile(f); np.asarray(mi, "<i4").tofile(f); np.asarray(vi, "<i4").tofile(f)
        for g in range(G):
            nc = int(am.n_mix[g])
            f.write(b"JMGM"); _name(f, "st_%d" % g)
            np.asarray([g, nc], "<i4").tofile(f)
            np.asarray(am.weight[g, :nc], "<f4").tofile(f)
            np.asarray(derived["log_weight"][g, :nc], "<f4").tofile(f)
        for t in range(am.n_tm):
            n = int(am.tm_nstates[t])
            f.write(b"JMTM"); _name(f, "T_%d" % t)
            np.asarray([n], "<i4").tofile(f)
            a = np.asarray(am.transp[t, :n, :n], np.float32)
            np.asarray((a > 0).sum(axis=1), "<i4").tofile(f)
            for i in range(n):
                np.asarray(np.nonzero(a[i] > 0)[0], "<i4").tofile(f)
            for i in range(n):
                a[i][a[i] > 0].astype("<f4").tofile(f)
            for i in range(n):
                np.asarray(derived["trP"][t, i, :n][a[i] > 0], "<f4").tofile(f)
        for h in range(am.n_hmm):
            n = int(am.hmm_nstates[h])
            f.write(b"JMHM"); _name(f, hmm_names[h] if hmm_names else "hmm%d" % h)
            np.asarray([n], "<i4").tofile(f)
            np.asarray(am.hmm_gmm[h, :n], "<i4").tofile(f)
            np.asarray([am.hmm_tm[h]], "<i4").tofile(f)
        f.write(b"\0")


def write_htk(path, feats, samp_period: int = 100000, parm_kind: int = 6 | 0x100 | 0x200 | 0x2000):
    """Uncompressed HTK parameter file: 12-byte big-endian header (nSamples, sampPeriod in 100 ns,
    sampSize in bytes, parmKind - default MFCC_D_A_0) + big-endian float32 vectors."""
    x = np.ascontiguousarray(feats, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(np.asarray([x.shape[0], samp_period], ">i4").tobytes())
        f.write(np.asarray([x.shape[1] * 4, parm_kind], ">i2").tobytes())
        f.write(x.astype(">f4").tobytes())


def write_htk_compressed(path, feats, samp_period: int = 100000, parm_kind: int = 6 | 0x100 | 0x200 | 0x2000):
    """HTK _C parameter file (HTK Book, "Storage of parameter files"): per component A = 2 * 32767 / (max - min),
    B = (max + min) * 32767 / (max - min); samples stored as big-endian int16 x * A - B; the A and B vectors
    (big-endian float32) come first and count as 4 samples in the header.  Returns what a reader gets back:
    (sample + B) / A in float32."""
    x = np.ascontiguousarray(feats, dtype=np.float32)
    lo, hi = x.min(axis=0).astype(np.float64), x.max(axis=0).astype(np.float64)
    span = np.where(hi > lo, hi - lo, 1.0)
    A = (2.0 * 32767.0 / span).astype(np.float32)
    B = ((hi + lo) * 32767.0 / span).astype(np.float32)
    q = np.clip(np.rint(x.astype(np.float64) * A - B), -32767, 32767).astype(np.int16)
    with open(path, "wb") as f:
        f.write(np.asarray([x.shape[0] + 4, samp_period], ">i4").tobytes())
        f.write(np.asarray([x.shape[1] * 2, parm_kind | 0x400], ">i2").tobytes())
        f.write(A.astype(">f4").tobytes()); f.write(B.astype(">f4").tobytes())
        f.write(q.astype(">i2").tobytes())
    return (q.astype(np.float32) + B) / A
