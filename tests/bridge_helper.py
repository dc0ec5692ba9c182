"""Drives tests/mock_juicer/bridge_roundtrip.cpp: the exact-signature seam of include/juicer_amd_decoder.hpp
(`GpuWFSTDecoder(Juicer::WFSTNetwork*, Juicer::IModels*, real, real, real, real, int)`, WFSTDecoderLite.h:81-89)
run against own-written mocks of the reference's WFSTNetwork / HTKFlatModels declarations."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_juicer")


def build_program(tmp_path):
    exe = os.path.join(str(tmp_path), "bridge_rt")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-maybe-uninitialized", "-I", MOCK, "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(MOCK, "bridge_roundtrip.cpp"), "-L", os.path.join(ROOT, "juicer_amd"), "-ljuicer_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "juicer_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def _w(f, a, dt):
    a = np.ascontiguousarray(np.asarray(a, dt)).reshape(-1)
    f.write(struct.pack("i", a.size)); f.write(a.tobytes())


def write_case(path, net, models, feats, beams, pad=0):
    """net / models: capi.Network / capi.Models (their prepared arrays are what a loaded WFSTNetwork / HTKFlatModels
    hold); beams = (start, main, end, word, maxHyps)."""
    c = net.csr()
    rp = c["row_ptr"]
    frm = np.repeat(np.arange(net.n_states, dtype=np.int32), np.diff(rp))
    fs = np.nonzero(np.isfinite(c["fin_w"]))[0].astype(np.int32)
    hn, hg, ht, nm = models.topology()
    det, mean, ivar = models.flat()
    trP, se, tee = models.trans()
    tmn = np.zeros(models.n_tm, np.int32)
    tmn[ht] = hn
    tmn[tmn == 0] = 3                                # (a transition matrix no HMM uses: any legal size)
    with open(path, "wb") as f:
        _w(f, [net.n_states, net.init_state, models.vec_size, models.n_gmms, models.max_mix, models.n_hmms, models.max_states,
               models.n_tm, pad], np.int32)
        _w(f, frm, np.int32); _w(f, c["to"], np.int32); _w(f, c["ilab"], np.int32); _w(f, c["olab"], np.int32); _w(f, c["w"], np.float32)
        _w(f, fs, np.int32); _w(f, c["fin_w"][fs], np.float32)
        _w(f, nm, np.int32); _w(f, det, np.float32); _w(f, mean, np.float32); _w(f, ivar, np.float32)
        _w(f, hn, np.int32); _w(f, hg, np.int32); _w(f, ht, np.int32); _w(f, tee, np.float32)
        _w(f, tmn, np.int32); _w(f, trP, np.float32); _w(f, se, np.int16)
        _w(f, feats, np.float32); _w(f, beams, np.float32)


def read_arrays(path, dtypes):
    out = []
    with open(path, "rb") as f:
        for dt in dtypes:
            n = struct.unpack("i", f.read(4))[0]
            out.append(np.frombuffer(f.read(n * np.dtype(dt).itemsize), dtype=dt).copy())
    return out
