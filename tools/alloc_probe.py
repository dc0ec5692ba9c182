#!/usr/bin/env python
"""How long HIP takes to hand out and to clear large buffers (GPU box): the decoder's arena set-up is this."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize()


def run(n, size):
    ps = []
    t0 = time.perf_counter()
    for _ in range(n):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), size) == 0
        ps.append(p)
    hip.hipDeviceSynchronize()
    t1 = time.perf_counter()
    for p in ps:
        hip.hipMemset(p, 0, size)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    for p in ps:
        hip.hipFree(p)
    hip.hipDeviceSynchronize()
    t3 = time.perf_counter()
    gb = n * size / 1e9
    print("%4d x %7.1f MB = %6.1f GB: malloc %.3f s (%.1f GB/s), memset %.3f s (%.1f GB/s), free %.3f s" %
          (n, size / 1e6, gb, t1 - t0, gb / (t1 - t0), t2 - t1, gb / (t2 - t1), t3 - t2))


for n, size in ((1, 1 << 30), (1024, 64 << 20), (256, 256 << 20), (128, 512 << 20), (64, 1 << 30), (32, 2 << 30), (16, 4 << 30), (8, 8 << 30), (64, 3 << 30), (192, 1 << 30), (1024, 64 << 20)):
    run(n, size)
