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
