"""Exhaustive check of the kernels' own expf (csrc/jd_gmm.h: jd_expf, a replica of glibc's
algorithm) against the host libm for EVERY float32 in [-18.5, -0.001]: the argument range of
HTKFlatModels::logAdd (HTKFlatModels.cpp:266-293: diff = y - x <= 0, cut at -18.42), plus the
few values above it.  Bit-exact or the GPU log-likelihoods cannot be."""
import ctypes as C

import numpy as np
import pytest

LO, HI = np.float32(-18.5), np.float32(-0.001)


def _all_floats():
    # negative floats: the bit pattern grows with the magnitude
    a = int(np.float32(HI).view(np.uint32))
    b = int(np.float32(LO).view(np.uint32))
    return np.arange(a, b + 1, dtype=np.uint32).view(np.float32)


def _libm_expf(x):
    from oracle.oracle import lib
    out = np.empty_like(x)
    assert lib().jo_expf_array(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.shape[0]),
                               out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return out


def _jd_expf(x, device):
    from juicer_amd import capi
    out = np.empty_like(x)
    rc = capi.lib().jd_debug_expf(C.c_int32(device), x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.shape[0]),
                                  out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, capi.lib().jd_last_error()
    return out


def _extra():
    return np.asarray([0.0, -0.0, -1e-10, -1e-5, -5e-4, -18.42, -18.4200001, -30.0, -87.0], np.float32)


def test_expf_host_twin_equals_libm_everywhere(built):
    """The host twin is compiled from the same source as the device function."""
    x = _all_floats()
    assert x.shape[0] > 100_000_000
    for part in np.array_split(x, 8):
        assert np.array_equal(_jd_expf(part, -1).view(np.uint32), _libm_expf(part).view(np.uint32))
    e = _extra()
    assert np.array_equal(_jd_expf(e, -1).view(np.uint32), _libm_expf(e).view(np.uint32))


@pytest.mark.gpu
def test_expf_device_equals_libm_everywhere(built):
    x = _all_floats()
    bad = 0
    for part in np.array_split(x, 8):
        bad += int((_jd_expf(part, 0).view(np.uint32) != _libm_expf(part).view(np.uint32)).sum())
    assert bad == 0
    e = _extra()
    assert np.array_equal(_jd_expf(e, 0).view(np.uint32), _libm_expf(e).view(np.uint32))
