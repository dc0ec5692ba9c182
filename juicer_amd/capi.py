"""ctypes binding of the C ABI in include/juicer_amd.h (libjuicer_amd.so).

The library is the product; this module only marshals numpy arrays and raw
device pointers into it.  There is no CPU fallback anywhere: if the shared
library is missing or no HIP device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjuicer_amd.so")

JD_OK, JD_EINVAL, JD_ENODEV, JD_EHIP, JD_ENOMEM, JD_EHIST, JD_ESTATE, JD_EFORMAT = 0, -1, -2, -3, -4, -5, -6, -7
LOG_ZERO = float(np.float32(-3.402823466e+38))


class JuicerAmdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("juicer_amd error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("n_frames", C.c_int32),
                ("tot_active_emit_hyps", C.c_int64), ("tot_active_end_hyps", C.c_int64),
                ("tot_active_models", C.c_int64), ("tot_proc_emit_hyps", C.c_int64),
                ("tot_proc_end_hyps", C.c_int64), ("tot_arcs_visited", C.c_int64),
                ("tot_paths", C.c_int64), ("tot_insts_in", C.c_int64), ("ties", C.c_int64),
                # what the kernels really touched (include/juicer_amd.h: jd_stats)
                ("tot_recs_read", C.c_int64), ("tot_new_attached", C.c_int64), ("tot_recs_written", C.c_int64), ("tot_entry_items", C.c_int64),
                ("tot_items_expanded", C.c_int64), ("tot_arcs_walked", C.c_int64), ("tot_closure_items", C.c_int64), ("tot_bids_placed", C.c_int64)]


class CHyp(C.Structure):
    _fields_ = [("n", C.c_int32),
                ("label", C.POINTER(C.c_int32)), ("time", C.POINTER(C.c_int32)),
                ("score", C.POINTER(C.c_float)), ("ac", C.POINTER(C.c_float)), ("lm", C.POINTER(C.c_float)),
                ("tot_score", C.c_float), ("tot_ac", C.c_float), ("tot_lm", C.c_float),
                ("stats", Stats)]


class Timing(C.Structure):
    _fields_ = [("gmm_ms", C.c_double), ("search_ms", C.c_double), ("total_ms", C.c_double),
                ("gmm_wait_ms", C.c_double),
                ("gmm_launches", C.c_int32), ("search_launches", C.c_int32),
                ("relaunches", C.c_int32), ("cluster_wgs", C.c_int32),
                ("gmm_frames", C.c_int64), ("gmm_states", C.c_int64), ("search_frames", C.c_int64),
                ("prefetched", C.c_int32), ("ahead_frames", C.c_int32), ("slot_launches", C.c_int32), ("pad0", C.c_int32)]


FLOW_SERIAL, FLOW_TWO_IN_FLIGHT, FLOW_RESIDENT = 0, 1, 3      # jd_dec_set_pipeline
SCORE_EXACT, SCORE_FAST = 0, 1                                # jd_dec_set_scoring


class PipeStats(C.Structure):
    _fields_ = [("mode", C.c_int32), ("depth", C.c_int32), ("slots", C.c_int32), ("resident", C.c_int32),
                ("batches_announced", C.c_int32), ("pad0", C.c_int32),
                ("frames_searched", C.c_int64), ("utts_through", C.c_int64), ("rows_scored", C.c_int64),
                ("batches_back", C.c_int64), ("collections", C.c_int64),
                ("slot_busy_us", C.c_double), ("on_us", C.c_double)]


@dataclass
class Hyp:
    """1-best in DecHyp chain order (index 0 = newest word)."""
    n: int
    label: np.ndarray
    time: np.ndarray
    score: np.ndarray
    ac: np.ndarray
    lm: np.ndarray
    tot_score: float
    tot_ac: float
    tot_lm: float
    stats: dict


# every symbol include/juicer_amd.h declares
EXPORTS = [
    "jd_net_create_arcs", "jd_net_create_csr", "jd_net_load_fsm", "jd_net_num_arcs", "jd_net_num_states",
    "jd_net_init_state", "jd_net_destroy", "jd_net_get_csr", "jd_net_load_jwnt", "jd_net_save_jwnt",
    "jd_am_load_jmbi", "jd_am_save_jmbi", "jd_am_create_htk", "jd_am_create_flat", "jd_am_num_hmms", "jd_am_num_gmms",
    "jd_am_vec_size", "jd_am_max_states", "jd_am_max_mix", "jd_am_num_transmats", "jd_am_get_topology", "jd_am_load_mmf", "jd_am_get_flat", "jd_am_get_trans", "jd_am_destroy",
    "jd_dec_create", "jd_dec_destroy", "jd_dec_set_capacity", "jd_stream_init", "jd_stream_push",
    "jd_stream_finish", "jd_decode_batch", "jd_decode_batch_device", "jd_dec_last_timing",
    "jd_am_score_frames", "jd_last_error", "jd_version", "jd_dec_debug_trace", "jd_debug_expf",
    "jd_multi_create", "jd_multi_create_lazy", "jd_multi_decode_batch", "jd_multi_destroy",
    "jd_dec_set_partial_interval", "jd_stream_partial", "jd_stream_collect_info", "jd_stream_path_counts", "jd_dec_quiesce", "jd_debug_closure_path_counts", "jd_dec_set_max_alloc_models", "jd_net_compose", "jd_am_create_hybrid", "jd_net_create_lazy", "jd_net_lazy_size", "jd_net_lazy_reset",
    "jd_net_lazy_set_high_water", "jd_net_lazy_generation", "jd_release_cached_memory", "jd_net_push_labels",
    "jd_dec_prefetch_scores", "jd_streams_push", "jd_dec_info",
    "jd_broker_create", "jd_broker_destroy", "jd_broker_open", "jd_broker_close", "jd_broker_init", "jd_broker_push",
    "jd_broker_finish", "jd_broker_get_stats", "jd_dec_debug_cells", "jd_dec_set_pipeline", "jd_dec_pipeline_stats",
    "jd_dec_set_scoring", "jd_am_score_frames_mode",
]

_lib = None


def lib():
    """Load libjuicer_amd.so; fail loudly when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JuicerAmdError(JD_ENODEV, "HIP extension %s is not built (run python -m juicer_amd.build); "
                                 "there is no CPU fallback" % LIB_PATH)
        try:                      # share torch's HIP runtime when torch is in the process
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.jd_last_error.restype = C.c_char_p
        L.jd_version.restype = C.c_char_p
        L.jd_net_num_arcs.restype = C.c_int64
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise JuicerAmdError(rc, lib().jd_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _hyp_from_c(h: CHyp) -> Hyp:
    k = max(int(h.n), 0)

    def arr(ptr, dt):
        if k == 0:
            return np.zeros(0, dtype=dt)
        # (string_at + frombuffer: a tenth of what np.ctypeslib.as_array costs per array - 64 hypotheses x 5 arrays per step)
        return np.frombuffer(C.string_at(ptr, 4 * k), dtype=dt).copy()
    hs = h.stats                                                   # (one ctypes sub-object, not one per field)
    st = {f: int(getattr(hs, f)) for f, _ in Stats._fields_}
    return Hyp(n=int(h.n), label=arr(h.label, np.int32), time=arr(h.time, np.int32),
               score=arr(h.score, np.float32), ac=arr(h.ac, np.float32), lm=arr(h.lm, np.float32),
               tot_score=float(h.tot_score), tot_ac=float(h.tot_ac), tot_lm=float(h.tot_lm), stats=st)


class Network:
    """WFSTNetwork counterpart: CSR arc table (16-byte arc records) for HBM."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_arcs(cls, src, dst, ilab, olab, w_file, fstate, fweight_file, lm_scale=1.0, ins_penalty=0.0):
        L = lib()
        h = C.c_void_p()
        src, dst, il, ol = _i32(src), _i32(dst), _i32(ilab), _i32(olab)
        wf, fs, fw = _f32(w_file), _i32(fstate), _f32(fweight_file)
        _check(L.jd_net_create_arcs(C.byref(h), C.c_int64(src.shape[0]), _p(src, C.c_int32), _p(dst, C.c_int32),
                                    _p(il, C.c_int32), _p(ol, C.c_int32), _p(wf, C.c_float),
                                    C.c_int32(fs.shape[0]), _p(fs, C.c_int32), _p(fw, C.c_float),
                                    C.c_float(lm_scale), C.c_float(ins_penalty)))
        return cls(h)

    @classmethod
    def from_synth(cls, net, lm_scale=1.0, ins_penalty=0.0):
        return cls.from_arcs(net.src, net.dst, net.ilab, net.olab, net.w_file, net.fstate, net.fweight_file,
                             lm_scale, ins_penalty)

    @classmethod
    def from_csr(cls, n_states, init_state, row_ptr, to, w, ilab, olab, fstate, fweight):
        L = lib()
        h = C.c_void_p()
        rp, to_, il, ol = _i32(row_ptr), _i32(to), _i32(ilab), _i32(olab)
        w_, fs, fw = _f32(w), _i32(fstate), _f32(fweight)
        _check(L.jd_net_create_csr(C.byref(h), C.c_int32(n_states), C.c_int32(init_state), _p(rp, C.c_int32),
                                   _p(to_, C.c_int32), _p(w_, C.c_float), _p(il, C.c_int32), _p(ol, C.c_int32),
                                   C.c_int32(fs.shape[0]), _p(fs, C.c_int32), _p(fw, C.c_float)))
        return cls(h)

    @classmethod
    def from_fsm_file(cls, fsm_path, insyms_path=None, outsyms_path=None, lm_scale=1.0, ins_penalty=0.0):
        L = lib()
        h = C.c_void_p()
        enc = lambda s: None if s is None else os.fsencode(s)
        _check(L.jd_net_load_fsm(C.byref(h), enc(fsm_path), enc(insyms_path), enc(outsyms_path),
                                 C.c_float(lm_scale), C.c_float(ins_penalty)))
        return cls(h)

    @classmethod
    def from_jwnt_file(cls, path, lm_scale=1.0, ins_penalty=0.0):
        """Juicer's binary network cache (<fsm>.bin, WFSTNetwork::readBinary)."""
        h = C.c_void_p()
        _check(lib().jd_net_load_jwnt(C.byref(h), os.fsencode(path), C.c_float(lm_scale), C.c_float(ins_penalty)))
        return cls(h)

    @classmethod
    def compose(cls, cl: "Network", g: "Network", device: int = 0, max_states: int = 0, max_arcs: int = 0, pushing: bool = False,
                push_labels: bool = False):
        """C.L o G on the device (jd_net_compose): the dynamic-composition row's first step."""
        h = C.c_void_p()
        _check(lib().jd_net_compose(C.byref(h), cl.h, g.h, C.c_int32(device), C.c_int64(max_states), C.c_int64(max_arcs),
                                    C.c_int32((1 if pushing else 0) | (2 if push_labels else 0))))
        return cls(h)

    @classmethod
    def lazy(cls, cl: "Network", g: "Network", am: "Models", device: int = 0, max_states: int = 0, max_arcs: int = 0, pushing: bool = False,
             push_labels: bool = False):
        """C.L o G expanded by the search, where it goes (jd_net_create_lazy)."""
        h = C.c_void_p()
        _check(lib().jd_net_create_lazy(C.byref(h), cl.h, g.h, am.h, C.c_int32(device), C.c_int64(max_states), C.c_int64(max_arcs),
                                        C.c_int32((1 if pushing else 0) | (2 if push_labels else 0))))
        return cls(h)

    def push_labels(self):
        """C.L with its output labels pushed towards the initial state (jd_net_push_labels): (network, labels moved)."""
        h, n = C.c_void_p(), C.c_int64(0)
        _check(lib().jd_net_push_labels(C.byref(h), self.h, C.byref(n)))
        return type(self)(h), n.value

    def lazy_reset(self):
        _check(lib().jd_net_lazy_reset(self.h))

    def lazy_set_high_water(self, fraction: float):
        """The fill beyond which the arena starts a new generation between utterances (jd_net_lazy_set_high_water)."""
        _check(lib().jd_net_lazy_set_high_water(self.h, C.c_double(fraction)))

    def lazy_generation(self) -> int:
        g = C.c_int64(0)
        _check(lib().jd_net_lazy_generation(self.h, C.byref(g)))
        return g.value

    def lazy_size(self):
        ns, na = C.c_int64(0), C.c_int64(0)
        _check(lib().jd_net_lazy_size(self.h, C.byref(ns), C.byref(na)))
        return ns.value, na.value

    def save_jwnt(self, path):
        """WFSTNetwork::writeBinary counterpart (-writeBinaryFiles)."""
        _check(lib().jd_net_save_jwnt(self.h, os.fsencode(path)))

    def csr(self):
        ns, na = self.n_states, self.n_arcs
        rp = np.zeros(ns + 1, np.int32); to = np.zeros(na, np.int32); w = np.zeros(na, np.float32)
        il = np.zeros(na, np.int32); ol = np.zeros(na, np.int32); fw = np.zeros(ns, np.float32)
        _check(lib().jd_net_get_csr(self.h, _p(rp, C.c_int32), _p(to, C.c_int32), _p(w, C.c_float),
                                    _p(il, C.c_int32), _p(ol, C.c_int32), _p(fw, C.c_float)))
        return dict(row_ptr=rp, to=to, w=w, ilab=il, olab=ol, fin_w=fw)

    def closure_path_counts(self, models):
        """(per-state Path counts of the reference's propagateToken closure, acyclic) - jd_debug_closure_path_counts (host only)."""
        out = np.zeros(self.n_states, np.int32)
        ok = C.c_int32(0)
        _check(lib().jd_debug_closure_path_counts(self.h, models.h, _p(out, C.c_int32), C.byref(ok)))
        return out, bool(ok.value)

    @property
    def n_arcs(self):
        return int(lib().jd_net_num_arcs(self.h))

    @property
    def n_states(self):
        return int(lib().jd_net_num_states(self.h))

    @property
    def init_state(self):
        return int(lib().jd_net_init_state(self.h))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jd_net_destroy(self.h)
            self.h = None


class Models:
    """HTKFlatModels counterpart (IModels, Models.h:29-67)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_htk(cls, am):
        L = lib()
        h = C.c_void_p()
        nm, wt, mu, var = _i32(am.n_mix), _f32(am.weight), _f32(am.mean), _f32(am.var)
        hn, hg, ht = _i32(am.hmm_nstates), _i32(am.hmm_gmm), _i32(am.hmm_tm)
        tn, tp = _i32(am.tm_nstates), _f32(am.transp)
        _check(L.jd_am_create_htk(C.byref(h), C.c_int32(am.D), C.c_int32(am.n_gmm), C.c_int32(am.max_mix),
                                  _p(nm, C.c_int32), _p(wt, C.c_float), _p(mu, C.c_float), _p(var, C.c_float),
                                  C.c_int32(am.n_hmm), C.c_int32(am.max_n), _p(hn, C.c_int32), _p(hg, C.c_int32),
                                  _p(ht, C.c_int32), C.c_int32(am.n_tm), _p(tn, C.c_int32), _p(tp, C.c_float)))
        return cls(h)

    @classmethod
    def from_mmf_file(cls, path):
        L = lib()
        h = C.c_void_p()
        _check(L.jd_am_load_mmf(C.byref(h), os.fsencode(path)))
        return cls(h)

    @classmethod
    def from_jmbi_file(cls, path):
        """Juicer's binary model cache (<mmf>.bin, HTKModels::readBinary)."""
        h = C.c_void_p()
        _check(lib().jd_am_load_jmbi(C.byref(h), os.fsencode(path)))
        return cls(h)

    def save_jmbi(self, path):
        """HTKModels::output(path, true) counterpart (-writeBinaryFiles)."""
        _check(lib().jd_am_save_jmbi(self.h, os.fsencode(path)))

    @property
    def n_hmms(self):
        return int(lib().jd_am_num_hmms(self.h))

    @property
    def n_gmms(self):
        return int(lib().jd_am_num_gmms(self.h))

    @property
    def vec_size(self):
        return int(lib().jd_am_vec_size(self.h))

    @property
    def max_states(self):
        return int(lib().jd_am_max_states(self.h))

    @property
    def max_mix(self):
        return int(lib().jd_am_max_mix(self.h))

    @property
    def n_tm(self):
        return int(lib().jd_am_num_transmats(self.h))

    def topology(self):
        H, MN, G = self.n_hmms, self.max_states, self.n_gmms
        hn = np.zeros(H, np.int32); hg = np.zeros((H, MN), np.int32); ht = np.zeros(H, np.int32)
        nm = np.zeros(G, np.int32)
        _check(lib().jd_am_get_topology(self.h, _p(hn, C.c_int32), _p(hg, C.c_int32), _p(ht, C.c_int32),
                                        _p(nm, C.c_int32)))
        return hn, hg, ht, nm

    def flat(self):
        G, M, D = self.n_gmms, self.max_mix, self.vec_size
        det = np.zeros((G, M), np.float32)
        mean = np.zeros((G, M, D), np.float32)
        ivar = np.zeros((G, M, D), np.float32)
        _check(lib().jd_am_get_flat(self.h, _p(det, C.c_float), _p(mean, C.c_float), _p(ivar, C.c_float)))
        return det, mean, ivar

    def trans(self):
        MN = self.max_states
        trP = np.zeros((self.n_tm, MN, MN), np.float32)
        se = np.zeros((self.n_tm, MN, 2), np.int16)
        tee = np.zeros(self.n_hmms, np.float32)
        _check(lib().jd_am_get_trans(self.h, _p(trP, C.c_float), _p(se, C.c_int16), _p(tee, C.c_float)))
        return trP, se, tee

    @classmethod
    def from_hybrid(cls, priors, states_per_model: int = 5):
        """Hybrid ANN / HMM models (HTKModels::Load(phones, priors, statesPerModel)): features are log posteriors."""
        pr = _f32(priors)
        h = C.c_void_p()
        _check(lib().jd_am_create_hybrid(C.byref(h), C.c_int32(pr.shape[0]), _p(pr, C.c_float), C.c_int32(states_per_model)))
        return cls(h)

    def score_frames(self, frames, device: int = 0, mode: int = 0):
        """Companion GMM kernel on its own: [T, D] -> [T, n_gmm] log-likelihoods (mode: SCORE_EXACT, the default, or SCORE_FAST)."""
        x = _f32(frames)
        out = np.zeros((x.shape[0], self.n_gmms), np.float32)
        _check(lib().jd_am_score_frames_mode(self.h, C.c_int32(device), C.c_int32(mode), _p(x, C.c_float), C.c_int32(x.shape[0]),
                                             _p(out, C.c_float)))
        return out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jd_am_destroy(self.h)
            self.h = None


class Decoder:
    """WFSTDecoderLite counterpart over the C ABI (one jd_dec)."""

    def __init__(self, net: Network, models: Models, start_beam=0.0, main_beam=0.0, end_beam=0.0,
                 word_beam=0.0, max_hyps=0, block_size=5, device=0, max_streams=1,
                 max_slots=0, max_paths=0, max_items=0):
        L = lib()
        self.net, self.models = net, models       # keep alive: the decoder does not own them
        self.max_streams = int(max_streams)
        self.h = C.c_void_p()
        _check(L.jd_dec_create(C.byref(self.h), net.h, models.h, C.c_float(start_beam), C.c_float(main_beam),
                               C.c_float(end_beam), C.c_float(word_beam), C.c_int32(max_hyps),
                               C.c_int32(block_size), C.c_int32(device), C.c_int32(max_streams)))
        if max_slots or max_paths or max_items:
            _check(L.jd_dec_set_capacity(self.h, C.c_int64(max_slots), C.c_int64(max_paths), C.c_int64(max_items)))

    # -- IDecoder protocol on one stream
    def stream_init(self, s: int = 0):
        _check(lib().jd_stream_init(self.h, C.c_int32(s)))

    def stream_push(self, s: int, frames):
        x = _f32(frames)
        _check(lib().jd_stream_push(self.h, C.c_int32(s), _p(x, C.c_float), C.c_int32(x.shape[0])))

    def stream_finish(self, s: int = 0) -> Hyp:
        h = CHyp()
        _check(lib().jd_stream_finish(self.h, C.c_int32(s), C.byref(h)))
        return _hyp_from_c(h)

    def streams_push(self, streams, frames):
        """jd_stream_push for several streams at once: one scoring launch + one search launch (jd_streams_push)."""
        xs = [_f32(f) for f in frames]
        n = len(xs)
        ss = _i32(list(streams))
        ptrs = (C.POINTER(C.c_float) * n)(*[_p(x, C.c_float) for x in xs])
        nfr = _i32([x.shape[0] for x in xs])
        _check(lib().jd_streams_push(self.h, C.c_int32(n), _p(ss, C.c_int32), ptrs, _p(nfr, C.c_int32)))

    def set_max_alloc_models(self, v: int):
        """WFSTDecoderLite::setMaxAllocModels (:807-820): percentage / MB / count, see juicer_amd.h."""
        _check(lib().jd_dec_set_max_alloc_models(self.h, C.c_int32(v)))

    # -- PARTIAL_DECODING (WFSTDecoderLite.cpp:822-896)
    def set_partial_interval(self, interval: int):
        _check(lib().jd_dec_set_partial_interval(self.h, C.c_int32(interval)))

    def stream_collect_info(self, s: int = 0):
        """(collectPaths runs of the stream's utterance so far, lastPathCollectFrame) - counted while PARTIAL_DECODING is on."""
        n, last = C.c_int32(0), C.c_int32(-1)
        _check(lib().jd_stream_collect_info(self.h, C.c_int32(s), C.byref(n), C.byref(last)))
        return n.value, last.value

    def quiesce(self):
        """The decoder's resident kernel (FLOW_RESIDENT) leaves the device - call before a device-wide synchronisation (jd_dec_quiesce)."""
        _check(lib().jd_dec_quiesce(self.h))

    def set_pipeline(self, mode: int, depth: int = 0, slots: int = 0):
        """How batches that follow each other share the chip (jd_dec_set_pipeline): FLOW_SERIAL, FLOW_TWO_IN_FLIGHT (default) or
        FLOW_RESIDENT (depth = batches announced and not handed back at most, slots = one-workgroup slots of the resident kernel)."""
        _check(lib().jd_dec_set_pipeline(self.h, C.c_int32(mode), C.c_int32(depth), C.c_int32(slots)))

    def set_scoring(self, mode: int):
        """How the likelihood tables are scored (jd_dec_set_scoring): SCORE_EXACT (default: the reference's roundings, bit-identical
        log-likelihoods) or SCORE_FAST (fused multiply-add distance, fp32 logAdd: scores within 1e-4, the table at 0.4 of the cost)."""
        _check(lib().jd_dec_set_scoring(self.h, C.c_int32(mode)))

    def pipeline_stats(self) -> dict:
        """What FLOW_RESIDENT has done so far, cumulative (jd_dec_pipeline_stats)."""
        t = PipeStats()
        _check(lib().jd_dec_pipeline_stats(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in PipeStats._fields_ if not f.startswith("pad")}

    def stream_path_counts(self, s: int = 0):
        """(nPath, nPathNew, exact): collectPaths' trigger counts; exact = they are the reference's (jd_stream_path_counts)."""
        a, b, e = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(lib().jd_stream_path_counts(self.h, C.c_int32(s), C.byref(a), C.byref(b), C.byref(e)))
        return a.value, b.value, bool(e.value)

    def stream_partial(self, s: int = 0, trace_now: bool = False):
        """(found, partialPaths as [(label, frame)], oldest first); trace_now runs tracePartialPath first."""
        n, found = C.c_int32(0), C.c_int32(0)
        cap, was_found = 256, False
        while True:
            lab, tim = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            _check(lib().jd_stream_partial(self.h, C.c_int32(s), C.c_int32(1 if trace_now else 0), C.c_int32(cap), C.byref(n),
                                           _p(lab, C.c_int32), _p(tim, C.c_int32), C.byref(found)))
            if trace_now:
                was_found = bool(found.value)
            if n.value <= cap:
                break
            cap, trace_now = n.value, False
        return was_found, [(int(lab[i]), int(tim[i])) for i in range(n.value)]

    # -- DecoderBatchTest inner loop
    def decode_batch(self, feats: Sequence[np.ndarray]) -> List[Hyp]:
        n = len(feats)
        xs = [_f32(f) for f in feats]
        ptrs = (C.POINTER(C.c_float) * n)(*[_p(x, C.c_float) for x in xs])
        nfr = _i32([x.shape[0] for x in xs])
        hyps = (CHyp * n)()
        rc = lib().jd_decode_batch(self.h, C.c_int32(n), ptrs, _p(nfr, C.c_int32), hyps)
        _check(rc)
        return [_hyp_from_c(hyps[i]) for i in range(n)]

    def decode_batch_device(self, d_feats_ptr: int, offs, hip_stream: int = 0, raw: bool = False):
        """Features already in HBM: d_feats_ptr = device address of [total_frames, D] floats."""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        n = offs.shape[0] - 1
        hyps = (CHyp * n)()
        rc = lib().jd_decode_batch_device(self.h, C.c_int32(n), C.c_void_p(d_feats_ptr), _p(offs, C.c_int64),
                                          C.c_void_p(hip_stream), hyps)
        _check(rc)
        if raw:
            return hyps
        return [_hyp_from_c(hyps[i]) for i in range(n)]

    def prefetch_scores(self, d_feats_ptr: int, offs, hip_stream: int = 0):
        """Announce the batch the NEXT decode_batch_device call will decode (same arguments): its likelihood table is
        scored on the CUs the current batch's search leaves idle (jd_dec_prefetch_scores).  The device buffer and the
        offsets must stay as they are until that call."""
        offs = np.ascontiguousarray(offs if offs is not None else [0], dtype=np.int64)
        n = offs.shape[0] - 1                                          # (one entry = no utterance: drops what was scored ahead)
        _check(lib().jd_dec_prefetch_scores(self.h, C.c_int32(n), C.c_void_p(d_feats_ptr), _p(offs, C.c_int64),
                                            C.c_void_p(hip_stream)))

    def last_timing(self) -> dict:
        t = Timing()
        _check(lib().jd_dec_last_timing(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timing._fields_}

    def debug_cells(self, enable: bool):
        """Mark (enable) / count (not enable -> (cells read, cells of the last decode's table)) the likelihood cells the search reads."""
        a, b = C.c_int64(0), C.c_int64(0)
        _check(lib().jd_dec_debug_cells(self.h, C.c_int32(1 if enable else 0), C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_trace(self, enable: int = 0, fetch: bool = False):
        """In-kernel cycle accounting of k_search: enable, or fetch [1024, 16] int64 sums per workgroup
        {lists A, phase A, wg wait, barrier 1, lists X, phase X, wg wait, barriers X, frames} in 100 MHz ticks."""
        buf = np.zeros((1024, 16), np.int64) if fetch else None
        _check(lib().jd_dec_debug_trace(self.h, C.c_int32(enable), None if buf is None else _p(buf, C.c_int64)))
        return buf

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jd_dec_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class BrokerStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("ticks", "frames", "stream_ticks", "us_idle", "us_coalesce", "us_init", "us_push", "us_finish",
                                         "us_search", "resident")]


class Broker:
    """Many serial IDecoder callers (threads) on the streams of one Decoder (jd_broker_*): each caller opens a client
    and drives it with init / push / finish; a worker thread of the library coalesces what the clients have pushed into
    one scoring launch and one search launch per tick.  ctypes releases the GIL during the calls, so Python threads
    do run side by side.  The Decoder must not be used directly while the broker lives."""

    def __init__(self, dec: Decoder, n_clients: int = 0):
        self.dec = dec
        self.h = C.c_void_p()
        _check(lib().jd_broker_create(C.byref(self.h), dec.h, C.c_int32(n_clients or dec.max_streams)))

    def open(self) -> int:
        c = C.c_int32(-1)
        _check(lib().jd_broker_open(self.h, C.byref(c)))
        return c.value

    def close_client(self, client: int):
        _check(lib().jd_broker_close(self.h, C.c_int32(client)))

    def init(self, client: int):
        _check(lib().jd_broker_init(self.h, C.c_int32(client)))

    def push(self, client: int, frames):
        x = _f32(frames)
        _check(lib().jd_broker_push(self.h, C.c_int32(client), _p(x, C.c_float), C.c_int32(x.shape[0])))

    def finish(self, client: int) -> Hyp:
        h = CHyp()
        _check(lib().jd_broker_finish(self.h, C.c_int32(client), C.byref(h)))
        return _hyp_from_c(h)

    def stats(self) -> dict:
        st = BrokerStats()
        _check(lib().jd_broker_get_stats(self.h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in BrokerStats._fields_}

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.jd_broker_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

