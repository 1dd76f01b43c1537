#!/usr/bin/env python3
"""FETCH_SIZE calibration on a known byte count (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").

Launches eftb_stream_read_probe (1 GiB, past the 256 MiB Infinity Cache) at 8 and 16 bytes per lane; run it under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and divide the bytes by FETCH_SIZE*1024 per kernel (tools/profile_collect.py does)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import _lib as L

BYTES = 1 << 30
lib = L.load()
for w in (8, 16):
    v = C.c_double()
    L.check(lib.eftb_stream_read_probe(0, BYTES, w, C.byref(v)))
    print(f"stream read {BYTES} bytes at {w} B/lane: {v.value:.0f} GB/s", flush=True)
