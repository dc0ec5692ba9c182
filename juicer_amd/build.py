"""In-tree build of libjuicer_amd.so (HIP kernels + C ABI) for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the GMM
kernel must round (x-mu)^2*ivar and the running sum separately, exactly like the
reference's x86-64 build (HTKFlatModels.cpp:249-250), and the host-side
parameter preparation must not be contracted either.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libjuicer_amd.so")
BATCH_TEST = os.path.join(HERE, "jd_batch_test")
BATCH_SRC = os.path.join(CSRC, "jd_batch_test.cpp")
SOURCES = [os.path.join(CSRC, "jd_host.cpp"), os.path.join(CSRC, "jd_device.hip"), os.path.join(CSRC, "jd_multi.cpp"),
           os.path.join(CSRC, "jd_compose.hip"), os.path.join(CSRC, "jd_broker.cpp")]
HEADERS_EXTRA = [os.path.join(CSRC, f) for f in ("jd_search.h", "jd_lazy.h", "jd_gmm.h", "jd_gc.h", "jd_resident.h", "jd_slot.h", "jd_host_resident.h", "jd_host_scoring.h", "jd_host_launch.h", "jd_host_stream.h")]
HEADERS = [os.path.join(CSRC, "jd_internal.h"), os.path.join(ROOT, "include", "juicer_amd.h"),
           os.path.join(ROOT, "include", "juicer_amd_decoder.hpp"), BATCH_SRC]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: juicer_amd needs ROCm to build its HIP extension")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(BATCH_TEST):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + HEADERS_EXTRA + [__file__])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + os.environ.get("JD_EXTRA_FLAGS", "").split() + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    # the DecoderBatchTest counterpart (C++ host over the C ABI + the IDecoder adapter)
    cmd2 = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", BATCH_TEST, BATCH_SRC,
            "-L", HERE, "-ljuicer_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd2), file=sys.stderr)
    subprocess.check_call(cmd2)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
