#!/usr/bin/env python
"""Which environment keeps RCCL's version banner off stdout?  (GPU box; development)
python tools/rccl_banner_probe.py -> one line per variant: bytes RCCL wrote to stdout / stderr during ncclCommInitAll"""
import os
import subprocess
import sys

CHILD = r'''
import ctypes, os, sys
L = ctypes.CDLL("librccl.so.1")
comm = (ctypes.c_void_p * 1)()
dev = (ctypes.c_int * 1)(0)
rc = L.ncclCommInitAll(comm, 1, dev)
sys.stdout.flush()
os.write(2, ("rc=%d\n" % rc).encode())
'''
variants = {
    "as is": {},
    "NCCL_DEBUG=WARN + NCCL_DEBUG_FILE=/dev/stderr": {"NCCL_DEBUG": "WARN", "NCCL_DEBUG_FILE": "/dev/stderr"},
    "NCCL_DEBUG=NONE": {"NCCL_DEBUG": "NONE"},
    "RCCL_LOG_LEVEL=0": {"RCCL_LOG_LEVEL": "0"},
    "NCCL_DEBUG_FILE=/dev/stderr": {"NCCL_DEBUG_FILE": "/dev/stderr"},
    "NCCL_DEBUG=INFO + NCCL_DEBUG_FILE=/dev/stderr": {"NCCL_DEBUG": "INFO", "NCCL_DEBUG_FILE": "/dev/stderr"},
}
print("environment of this box:", {k: v for k, v in os.environ.items() if "NCCL" in k or "RCCL" in k})
for name, env in variants.items():
    e = {k: v for k, v in os.environ.items() if k not in ("NCCL_DEBUG", "NCCL_DEBUG_FILE", "RCCL_LOG_LEVEL")} if env else dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=120)
    print("%-50s stdout %4d bytes %r | stderr %5d bytes" % (name, len(r.stdout), r.stdout[:60], len(r.stderr)))
