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
