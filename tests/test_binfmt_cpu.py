"""Juicer's binary caches (JWNT network, JMBI models) and HTK feature files: the C loaders
against oracle/binfmt.py (an independent numpy restatement of the reference's readers) on files
written by independent Python writers, save -> load round trips, and the malformed-file errors.
Host-side only: no kernel is launched."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def small():
    from juicer_amd import synth
    return synth.config_small()


def symbol_lists(net):
    inl = ["<eps>"] + ["ph%d" % i for i in range(1, int(net.ilab.max()) + 1)]
    outl = ["<eps>"] + ["w%d" % i for i in range(1, int(net.olab.max()) + 1)]
    outl[3] = None                                   # an unused id: label length 0 in the file
    return inl, outl


@pytest.mark.parametrize("scale,pen", [(1.0, 0.0), (13.0, -2.5), (0.7, 4.0)])
def test_jwnt_loader_matches_restated_reader(built, tmp_path, scale, pen):
    from juicer_amd import capi, io as jio
    from oracle import binfmt
    am, net, _, _ = small()
    p = str(tmp_path / "clg.fsm.bin")
    inl, outl = symbol_lists(net)
    jio.write_jwnt(p, net, inl, outl)
    g = capi.Network.from_jwnt_file(p, scale, pen)
    c, r = g.csr(), binfmt.read_jwnt(p, scale, pen)
    order = np.concatenate(r["trans"])               # state-major = the library's CSR order
    assert (g.n_arcs, g.n_states, g.init_state) == (net.n_arcs, r["label"].shape[0], r["init"])
    assert np.array_equal(np.diff(c["row_ptr"]), [len(t) for t in r["trans"]])
    for k in ("to", "ilab", "olab"):
        assert np.array_equal(c[k], r[k][order])
    assert np.array_equal(bits(c["w"]), bits(r["w"][order]))
    assert np.array_equal(bits(c["fin_w"]), bits(r["fin_w"]))
    assert r["alphabets"][0][0][1] == "ph1" and r["alphabets"][1][0][3] is None
    # arithmetic of the binary path: stored * scale (+ penalty on labelled arcs), float32
    labelled = r["olab"] > 0
    want = (r["w_stored"] * np.float32(scale)).astype(np.float32) if scale != 1.0 else r["w_stored"]
    want = np.where(labelled, (want + np.float32(pen)).astype(np.float32), want) if pen != 0.0 else want
    assert np.array_equal(bits(r["w"]), bits(want))


def test_jwnt_text_and_binary_paths_agree_at_unit_scale(built, tmp_path):
    """With lmScale 1 / insPenalty 0 the cache holds the text loader's weights (as values: the text
    path adds the zero penalty to labelled arcs, -0.0 + 0.0 = +0.0, WFSTNetwork.cpp:485-486, while
    readBinary skips a zero penalty, :1351, and keeps -0.0)."""
    from juicer_amd import capi, io as jio
    am, net, _, _ = small()
    p = str(tmp_path / "a.bin")
    jio.write_jwnt(p, net)
    a, b = capi.Network.from_synth(net).csr(), capi.Network.from_jwnt_file(p).csr()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    unl = a["olab"] == 0
    assert np.array_equal(bits(a["w"][unl]), bits(b["w"][unl]))


@pytest.mark.parametrize("scale,pen", [(1.0, 0.0), (13.0, -2.5)])
def test_jwnt_save_load_round_trip(built, tmp_path, scale, pen):
    from juicer_amd import capi
    from oracle import binfmt
    am, net, _, _ = small()
    g0 = capi.Network.from_synth(net, scale, pen)
    p = str(tmp_path / "rt.bin")
    g0.save_jwnt(p)
    raw = binfmt.read_jwnt(p)                        # scale 1, penalty 0: the stored values
    c0 = g0.csr()
    x = c0["w"].copy()
    if pen != 0.0:                                   # WFSTNetwork.cpp:1108-1126: penalty off, then scale off
        x = np.where(c0["olab"] > 0, (x - np.float32(pen)).astype(np.float32), x)
    if scale != 1.0:
        x = (x / np.float32(scale)).astype(np.float32)
    assert np.array_equal(bits(raw["w_stored"]), bits(x))
    assert raw["alphabets"] == [None, None] and raw["init"] == g0.init_state
    assert raw["max_out"] == int(np.diff(c0["row_ptr"]).max())
    c1 = capi.Network.from_jwnt_file(p, scale, pen).csr()
    for k in ("row_ptr", "to", "ilab", "olab"):
        assert np.array_equal(c0[k], c1[k])
    assert np.array_equal(bits(c1["fin_w"]), bits(c0["fin_w"]))   # final weights are never rescaled
    y = (x * np.float32(scale)).astype(np.float32) if scale != 1.0 else x
    y = np.where(c0["olab"] > 0, (y + np.float32(pen)).astype(np.float32), y) if pen != 0.0 else y
    assert np.array_equal(bits(c1["w"]), bits(y))
    assert np.allclose(c1["w"], c0["w"], rtol=1e-6, atol=1e-6)


def test_jwnt_malformed_files(built, tmp_path):
    from juicer_amd import capi, io as jio
    am, net, _, _ = small()
    good = str(tmp_path / "g.bin")
    jio.write_jwnt(good, net)
    data = open(good, "rb").read()
    cases = {"magic": b"XXXX" + data[4:], "truncated": data[:len(data) // 2], "tail": data[:-4] + b"JWNX", "empty": b""}
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".bin"))
        open(p, "wb").write(blob)
        with pytest.raises(capi.JuicerAmdError) as e:
            capi.Network.from_jwnt_file(p)
        assert "readBinary" in str(e.value), name
    with pytest.raises(capi.JuicerAmdError):
        capi.Network.from_jwnt_file(str(tmp_path / "missing.bin"))
    # auxiliary symbols in an embedded alphabet: rejected like the text loader does
    inl, outl = symbol_lists(net)
    inl.append("#1")
    p = str(tmp_path / "aux.bin")
    jio.write_jwnt(p, net, inl, outl, aux_in=(len(inl) - 1,))
    with pytest.raises(capi.JuicerAmdError) as e:
        capi.Network.from_jwnt_file(p)
    assert "auxiliary" in str(e.value)
    # a state whose transition list is not one contiguous run (the Lite core takes first + count,
    # WFSTNetwork.cpp:709-721, and would silently walk other arcs): patch trans[1] of the first
    # state that has two transitions
    import struct
    blob = bytearray(data)
    off = 32                                                        # ID + 7 header ints
    while True:
        _, _, nt = struct.unpack_from("<3i", blob, off)
        if nt >= 2:
            t1 = struct.unpack_from("<i", blob, off + 16)[0]
            struct.pack_into("<i", blob, off + 16, t1 + 1)
            break
        off += 12 + 4 * nt
    p = str(tmp_path / "noncontig.bin")
    open(p, "wb").write(bytes(blob))
    with pytest.raises(capi.JuicerAmdError) as e:
        capi.Network.from_jwnt_file(p)
    assert "not contiguous" in str(e.value)


def test_jmbi_round_trip_and_restated_reader(built, tmp_path):
    from juicer_amd import capi, synth
    from oracle import binfmt
    for am in (small()[0], synth.make_models(3, n_gmm=40, n_hmm=30, n_mix=5, D=13, n_tm=7, with_tee=True)):
        m0 = capi.Models.from_htk(am)
        p = str(tmp_path / "models.mmf.bin")
        m0.save_jmbi(p)
        m1 = capi.Models.from_jmbi_file(p)
        r = binfmt.read_jmbi(p)
        assert (m1.n_hmms, m1.n_gmms, m1.vec_size, m1.max_states, m1.max_mix) == \
               (am.n_hmm, am.n_gmm, am.D, am.max_n, am.max_mix)
        for a, b, c in zip(m0.flat(), m1.flat(), (r["det"], r["mean"], r["ivar"])):
            assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(c))
        t0, t1 = m0.trans(), m1.trans()
        assert np.array_equal(bits(t0[0]), bits(t1[0])) and np.array_equal(bits(t0[0]), bits(r["trP"]))
        assert np.array_equal(t0[1], t1[1]) and np.array_equal(t0[1], r["se"])
        assert np.array_equal(bits(t0[2]), bits(t1[2])) and np.array_equal(bits(t0[2]), bits(r["tee"]))
        for a, b in zip(m0.topology(), m1.topology()):
            assert np.array_equal(a, b)
        assert np.array_equal(bits(r["minus_half_over_vars"]), bits(np.float32(-0.5) / r["vars"]))
        assert not r["hybrid"]
        m1.save_jmbi(str(tmp_path / "again.bin"))                  # a loaded cache can be written again
        assert open(p, "rb").read() == open(str(tmp_path / "again.bin"), "rb").read()


def test_jmbi_foreign_file_with_names_and_shared_variances(built, tmp_path):
    """A cache written by someone else: named records, a pooled (shared) variance vector table,
    and derived values that are NOT what this library would compute - they must be used as
    stored (HTKFlatModels.cpp:163,174; HTKModels.cpp:2357-2361)."""
    from juicer_amd import capi, io as jio, synth
    from oracle import binfmt
    am = synth.make_models(5, n_gmm=12, n_hmm=9, n_mix=3, D=7, n_tm=4, with_tee=True)
    am.var[:, 1] = am.var[:, 0]                                    # identical vectors -> one pooled entry
    m0 = capi.Models.from_htk(am)
    det, _, _ = m0.flat()
    trP, _, _ = m0.trans()
    rng = np.random.default_rng(1)
    slv = rng.normal(-40.0, 3.0, size=(am.n_gmm, am.max_mix)).astype(np.float32)
    slv[:, 1] = slv[:, 0]                                          # consistent with the pooling
    lw = rng.normal(-1.5, 0.3, size=slv.shape).astype(np.float32)
    trP2 = np.where(trP > -1e30, trP + np.float32(0.125), trP).astype(np.float32)
    p = str(tmp_path / "foreign.bin")
    jio.write_jmbi(p, am, dict(sum_log_var=slv, log_weight=lw, trP=trP2), share_vars=True)
    m1 = capi.Models.from_jmbi_file(p)
    r = binfmt.read_jmbi(p)
    assert r["vars"].shape[0] < am.n_gmm * am.max_mix               # the pool really is shared
    assert r["hmm_names"][2] == "hmm2"
    d1, mu1, iv1 = m1.flat()
    assert np.array_equal(bits(d1), bits((slv + lw).astype(np.float32))) and np.array_equal(bits(d1), bits(r["det"]))
    assert np.array_equal(bits(mu1), bits(am.mean)) and np.array_equal(bits(iv1), bits(m0.flat()[2]))
    t1 = m1.trans()
    assert np.array_equal(bits(t1[0]), bits(trP2)) and np.array_equal(t1[1], m0.trans()[1])
    assert np.array_equal(bits(t1[2]), bits(r["tee"]))
    assert np.isclose(float(t1[2][am.sp_hmm]), float(m0.trans()[2][am.sp_hmm]) + 0.125)


def test_jmbi_malformed_files(built, tmp_path):
    from juicer_amd import capi
    am = small()[0]
    good = str(tmp_path / "g.bin")
    capi.Models.from_htk(am).save_jmbi(good)
    data = open(good, "rb").read()
    cases = {"magic": b"JMBX" + data[4:], "truncated": data[:len(data) // 3], "hybrid": data[:-1] + b"\1",
             "record": data.replace(b"JMGM", b"JMGX", 1)}
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".bin"))
        open(p, "wb").write(blob)
        with pytest.raises(capi.JuicerAmdError):
            capi.Models.from_jmbi_file(p)
    # mixtureInd != gmmInd (shared mixture pools): outside HTKFlatModels' assumption, refused
    i = data.index(b"JMGM") + 4 + 4                                 # first GMM: ID, name length 0
    blob = data[:i] + np.asarray([1], "<i4").tobytes() + data[i + 4:]
    p = str(tmp_path / "mixind.bin")
    open(p, "wb").write(blob)
    with pytest.raises(capi.JuicerAmdError) as e:
        capi.Models.from_jmbi_file(p)
    assert "mixtureInd" in str(e.value)


def test_batch_test_cli_accepts_htk_feature_files_without_gpu_work(built, tmp_path):
    """jd_batch_test parses HTK parameter files (big-endian) and the .bin caches; without a GPU
    it must stop at decoder creation with the no-device error, not before."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    from juicer_amd import build as jbuild, capi, io as jio
    am, net, feats, _ = small()
    jio.write_fsm(str(tmp_path / "clg.fsm"), net)
    capi.Network.from_synth(net).save_jwnt(str(tmp_path / "clg.fsm.bin"))
    jio.write_mmf(str(tmp_path / "m.mmf"), am)
    capi.Models.from_htk(am).save_jmbi(str(tmp_path / "m.mmf.bin"))
    jio.write_htk(str(tmp_path / "u0.htk"), feats[0])
    open(str(tmp_path / "list"), "w").write(str(tmp_path / "u0.htk") + "\n")
    r = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "clg.fsm"), "-htkModelsFName", str(tmp_path / "m.mmf"),
                        "-inputFName", str(tmp_path / "list"), "-mainBeam", "120"], capture_output=True, text=True)
    assert "pre-existing binary file" in r.stderr and r.stderr.count("pre-existing") == 2
    assert r.returncode != 0 and ("device" in r.stderr.lower() or "gpu" in r.stderr.lower() or "hip" in r.stderr.lower())
    # a file that is neither HTK nor .jdf is refused by the feature reader itself
    open(str(tmp_path / "bad.htk"), "wb").write(b"\0" * 40)
    open(str(tmp_path / "list2"), "w").write(str(tmp_path / "bad.htk") + "\n")
    r = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", str(tmp_path / "clg.fsm"), "-htkModelsFName", str(tmp_path / "m.mmf"),
                        "-inputFName", str(tmp_path / "list2")], capture_output=True, text=True)
    assert r.returncode != 0 and "neither" in r.stderr
