"""ctypes wrapper of the CPU oracle (oracle/juicer_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg - never by juicer_amd/.  PARITY UNPINNED (see the
header of juicer_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjuicer_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "juicer_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


class _Stats(C.Structure):
    _fields_ = [("n_frames", C.c_int32),
                ("tot_active_emit_hyps", C.c_int64), ("tot_active_end_hyps", C.c_int64),
                ("tot_active_models", C.c_int64), ("tot_proc_emit_hyps", C.c_int64),
                ("tot_proc_end_hyps", C.c_int64), ("tot_arcs_visited", C.c_int64),
                ("tot_paths", C.c_int64), ("tot_insts_in", C.c_int64), ("ties", C.c_int64)]


class _Hyp(C.Structure):
    _fields_ = [("n", C.c_int32),
                ("label", C.POINTER(C.c_int32)), ("time", C.POINTER(C.c_int32)),
                ("score", C.POINTER(C.c_float)), ("ac", C.POINTER(C.c_float)), ("lm", C.POINTER(C.c_float)),
                ("tot_score", C.c_float), ("tot_ac", C.c_float), ("tot_lm", C.c_float),
                ("stats", _Stats)]


@dataclass
class OracleHyp:
    n: int                      # -1 = no surviving token
    label: np.ndarray           # chain order: newest first
    time: np.ndarray
    score: np.ndarray
    ac: np.ndarray
    lm: np.ndarray
    tot_score: float
    tot_ac: float
    tot_lm: float
    stats: dict
    cpu_seconds: float = 0.0
    tie_kinds: tuple = (0, 0, 0, 0)   # bestFinal, entry, HMM-internal, order-dependent entry ties


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.jo_last_error.restype = C.c_char_p
        L.jo_net_num_arcs.restype = C.c_int64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle error %d: %s" % (rc, lib().jo_last_error().decode()))


class OracleNet:
    def __init__(self, net, lm_scale: float = 1.0, ins_penalty: float = 0.0):
        L = lib()
        self.h = C.c_void_p()
        src, dst, il, ol = _i32(net.src), _i32(net.dst), _i32(net.ilab), _i32(net.olab)
        wf, fs, fw = _f32(net.w_file), _i32(net.fstate), _f32(net.fweight_file)
        _check(L.jo_net_create_arcs(C.byref(self.h), C.c_int64(src.shape[0]), _p(src, C.c_int32),
                                    _p(dst, C.c_int32), _p(il, C.c_int32), _p(ol, C.c_int32),
                                    _p(wf, C.c_float), C.c_int32(fs.shape[0]), _p(fs, C.c_int32),
                                    _p(fw, C.c_float), C.c_float(lm_scale), C.c_float(ins_penalty)))
        self.n_arcs = int(L.jo_net_num_arcs(self.h))
        self.n_states = int(L.jo_net_num_states(self.h))

    @classmethod
    def from_csr(cls, n_states, init_state, row_ptr, to, w, ilab, olab, fstate, fweight):
        """weights taken as they are (already scaled): jo_net_create_csr"""
        L = lib()
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        rp, to_, il, ol = _i32(row_ptr), _i32(to), _i32(ilab), _i32(olab)
        w_, fs, fw = _f32(w), _i32(fstate), _f32(fweight)
        _check(L.jo_net_create_csr(C.byref(self.h), C.c_int32(n_states), C.c_int32(init_state), _p(rp, C.c_int32),
                                   _p(to_, C.c_int32), _p(w_, C.c_float), _p(il, C.c_int32), _p(ol, C.c_int32),
                                   C.c_int32(fs.shape[0]), _p(fs, C.c_int32), _p(fw, C.c_float)))
        self.n_arcs = int(L.jo_net_num_arcs(self.h))
        self.n_states = int(L.jo_net_num_states(self.h))
        return self

    def arrays(self):
        L = lib()
        first = np.zeros(self.n_states, np.int32); cnt = np.zeros(self.n_states, np.int32)
        to = np.zeros(self.n_arcs, np.int32); w = np.zeros(self.n_arcs, np.float32)
        il = np.zeros(self.n_arcs, np.int32); ol = np.zeros(self.n_arcs, np.int32)
        fi = np.zeros(self.n_states, np.int32)
        nfin = int((np.unique(fi).shape[0]))  # placeholder, final_w sized generously below
        fw = np.zeros(self.n_states, np.float32)
        L.jo_net_get(self.h, _p(first, C.c_int32), _p(cnt, C.c_int32), _p(to, C.c_int32), _p(w, C.c_float),
                     _p(il, C.c_int32), _p(ol, C.c_int32), _p(fi, C.c_int32), None)
        del nfin, fw
        return dict(first=first, cnt=cnt, to=to, w=w, ilab=il, olab=ol, final_ind=fi)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jo_net_destroy(self.h)
            self.h = None


class OracleAM:
    def __init__(self, am):
        L = lib()
        self.h = C.c_void_p()
        self.D, self.n_gmm, self.max_mix = am.D, am.n_gmm, am.max_mix
        self.n_hmm, self.max_n, self.n_tm = am.n_hmm, am.max_n, am.n_tm
        nm, wt, mu, var = _i32(am.n_mix), _f32(am.weight), _f32(am.mean), _f32(am.var)
        hn, hg, ht = _i32(am.hmm_nstates), _i32(am.hmm_gmm), _i32(am.hmm_tm)
        tn, tp = _i32(am.tm_nstates), _f32(am.transp)
        _check(L.jo_am_create_htk(C.byref(self.h), C.c_int32(am.D), C.c_int32(am.n_gmm), C.c_int32(am.max_mix),
                                  _p(nm, C.c_int32), _p(wt, C.c_float), _p(mu, C.c_float), _p(var, C.c_float),
                                  C.c_int32(am.n_hmm), C.c_int32(am.max_n), _p(hn, C.c_int32),
                                  _p(hg, C.c_int32), _p(ht, C.c_int32), C.c_int32(am.n_tm),
                                  _p(tn, C.c_int32), _p(tp, C.c_float)))

    @classmethod
    def from_hybrid(cls, priors, states_per_model: int = 5):
        """hybrid ANN / HMM models: HTKModels::Load(phones, priors, statesPerModel)"""
        self = cls.__new__(cls)
        pr = _f32(priors)
        self.h = C.c_void_p()
        P = pr.shape[0]
        self.D, self.n_gmm, self.max_mix = P, P, 1
        self.n_hmm, self.max_n, self.n_tm = P, states_per_model, 1
        _check(lib().jo_am_create_hybrid(C.byref(self.h), C.c_int32(P), _p(pr, C.c_float), C.c_int32(states_per_model)))
        return self

    def flat(self):
        det = np.zeros((self.n_gmm, self.max_mix), np.float32)
        mean = np.zeros((self.n_gmm, self.max_mix, self.D), np.float32)
        ivar = np.zeros_like(mean)
        lib().jo_am_get_flat(self.h, _p(det, C.c_float), _p(mean, C.c_float), _p(ivar, C.c_float))
        return det, mean, ivar

    def trans(self):
        trP = np.zeros((self.n_tm, self.max_n, self.max_n), np.float32)
        se = np.zeros((self.n_tm, self.max_n, 2), np.int16)
        tee = np.zeros(self.n_hmm, np.float32)
        lib().jo_am_get_trans(self.h, _p(trP, C.c_float), _p(se, C.c_int16), _p(tee, C.c_float))
        return trP, se, tee

    def score_frames(self, frames):
        x = _f32(frames)
        out = np.zeros((x.shape[0], self.n_gmm), np.float32)
        lib().jo_am_score_frames(self.h, _p(x, C.c_float), C.c_int32(x.shape[0]), _p(out, C.c_float))
        return out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jo_am_destroy(self.h)
            self.h = None


class OracleDecoder:
    """Restated WFSTDecoderLite; decode() follows DecoderSingleTest's frame loop."""

    def __init__(self, net: OracleNet, am: OracleAM, start_beam=0.0, main_beam=0.0, end_beam=0.0,
                 word_beam=0.0, max_hyps=0, block_size=5):
        L = lib()
        self.net, self.am = net, am
        self.h = C.c_void_p()
        _check(L.jo_dec_create(C.byref(self.h), net.h, am.h, C.c_float(start_beam), C.c_float(main_beam),
                               C.c_float(end_beam), C.c_float(word_beam), C.c_int32(max_hyps),
                               C.c_int32(block_size)))

    def decode(self, feats, trace: Optional[np.ndarray] = None, tie_mode: int = 0, threading: bool = False) -> OracleHyp:
        """threading=True: the reference's two-thread organisation (WFSTDecoderLiteThreading +
        HTKFlatModelsThreading: search thread + scoring thread); same results, cpu_seconds is then WALL time
        with two busy cores."""
        L = lib()
        x = _f32(feats)
        L.jo_dec_set_tie_mode(self.h, C.c_int(tie_mode))
        if trace is not None:
            L.jo_set_trace(self.h, _p(trace, C.c_float), C.c_int32(trace.shape[0]))
        else:
            L.jo_set_trace(self.h, None, C.c_int32(0))
        hyp = _Hyp()
        secs = C.c_double(0.0)
        fn = L.jo_decode_utt_threading if threading else L.jo_decode_utt
        _check(fn(self.h, _p(x, C.c_float), C.c_int32(x.shape[0]), C.byref(hyp), C.byref(secs)))
        n = hyp.n
        k = max(n, 0)

        def arr(ptr, dt):
            return np.array([ptr[i] for i in range(k)], dtype=dt)
        st = {f: getattr(hyp.stats, f) for f, _ in _Stats._fields_}
        kinds = (C.c_int64 * 4)()
        L.jo_tie_breakdown(self.h, kinds)
        return OracleHyp(n=n, label=arr(hyp.label, np.int32), time=arr(hyp.time, np.int32),
                         score=arr(hyp.score, np.float32), ac=arr(hyp.ac, np.float32),
                         lm=arr(hyp.lm, np.float32), tot_score=float(hyp.tot_score),
                         tot_ac=float(hyp.tot_ac), tot_lm=float(hyp.tot_lm), stats=st,
                         cpu_seconds=float(secs.value), tie_kinds=tuple(int(k) for k in kinds))

    def _partial_list(self):
        L = lib()
        n = C.c_int32(0)
        lab = C.POINTER(C.c_int32)()
        tim = C.POINTER(C.c_int32)()
        _check(L.jo_partial_get(self.h, C.byref(n), C.byref(lab), C.byref(tim)))
        return [(int(lab[i]), int(tim[i])) for i in range(n.value)]

    def decode_partial(self, feats, interval: int = 0, trace_at=()):
        """PARTIAL_DECODING (WFSTDecoderLite.cpp:822-896).  Runs DecoderSingleTest's frame loop with
        setPartialDecodeOptions(interval) (0: the reference's own schedule is off) and, as a test aid,
        calls tracePartialPath after every frame listed in trace_at.  Returns (snapshots, final):
        snapshots[f] = (found, partialPaths as [(label, frame)], oldest first) for every frame f after
        which a trace ran (explicit ones: found is the return value; scheduled ones: whether the list
        grew), final = partialPaths after recognitionFinish."""
        L = lib()
        x = _f32(feats)
        T, D = x.shape[0], self.am.D
        L.jo_dec_set_tie_mode(self.h, C.c_int(0))
        L.jo_set_trace(self.h, None, C.c_int32(0))
        _check(L.jo_set_partial_interval(self.h, C.c_int32(interval)))
        rows = (C.POINTER(C.c_float) * max(T, 1))()
        base = x.ctypes.data
        for t in range(T):
            rows[t] = C.cast(base + t * D * 4, C.POINTER(C.c_float))
        rows_addr = C.addressof(rows)
        psz = C.sizeof(C.POINTER(C.c_float))
        trace_at = set(int(f) for f in trace_at)
        snaps = {}
        _check(L.jo_init(self.h))
        n_frames, n_data = 0, min(20, T)                                # DecoderSingleTest.cpp:267-277
        last_trace, before = -1, 0
        counts = (C.c_int64 * 4)()
        n_coll = 0
        self.collect_frames = []                                        # frames after which collectPaths ran (:362)
        self.path_counts = []                                           # (nPath, nPathNew) behind every frame (:360, :745)
        while n_data > 0:                                               # :280-295
            _check(L.jo_process_frame(self.h, C.c_void_p(rows_addr + n_frames * psz), C.c_int32(n_frames), C.c_int32(n_data)))
            f = n_frames
            _check(L.jo_path_counts(self.h, counts))
            self.path_counts.append((int(counts[2]), int(counts[3])))
            if counts[0] > n_coll:                                      # jo_process_frame collected after this frame
                n_coll = int(counts[0])
                self.collect_frames.append(f)
                if interval > 0 and f - last_trace > interval:
                    lst = self._partial_list()
                    snaps[f] = (len(lst) > before, lst)
                    before, last_trace = len(lst), f
            if f in trace_at:
                found = L.jo_trace_partial(self.h)
                lst = self._partial_list()
                snaps[f] = (bool(found), lst)
                before, last_trace = len(lst), f
            n_frames += 1
            if not (n_frames + n_data - 1 < T):
                n_data -= 1
        hyp = _Hyp()
        _check(L.jo_finish(self.h, C.byref(hyp)))
        final = self._partial_list()
        _check(L.jo_set_partial_interval(self.h, C.c_int32(0)))
        return snaps, final

    def decode_certified(self, feats) -> OracleHyp:
        """Reference-rule decode whose result is certified not to depend on visiting order: when
        order-dependent equal-score recombinations occurred (stats['ties'] > 0: float32 collisions
        between different tokens), the utterance is decoded again with the opposite rule (last
        visited wins) and both results must coincide.  Raises AssertionError otherwise - pick
        another fixture, there is nothing a different implementation could be held to."""
        o = self.decode(feats)
        if o.stats["ties"]:
            f = self.decode(feats, tie_mode=1)
            same = (f.n == o.n and np.array_equal(f.label, o.label) and np.array_equal(f.time, o.time)
                    and np.array_equal(f.score.view(np.uint32), o.score.view(np.uint32))
                    and np.array_equal(f.ac.view(np.uint32), o.ac.view(np.uint32))
                    and np.array_equal(f.lm.view(np.uint32), o.lm.view(np.uint32)))
            assert same, "fixture is tie-order sensitive (%d order-dependent ties)" % o.stats["ties"]
            for k in o.stats:
                if k not in ("ties", "tot_arcs_visited", "tot_paths"):
                    assert f.stats[k] == o.stats[k], k
        return o

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jo_dec_destroy(self.h)
            self.h = None
