"""Build libeftbird.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False, verbose=False):
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode:
        raise RuntimeError("hipcc build of libeftbird.so failed")
    return os.path.join(_HERE, "libeftbird.so")
