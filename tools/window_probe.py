"""Window precompute timing on the GPU box: host tables, device kernels, whole call, and the host builder beside it.
Usage: python tools/window_probe.py [Nk ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth, tables as TB  # noqa: E402
from eftpipe_amd.window import window_matrix_device  # noqa: E402

tab = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "win_NGC_LRG_sQ024.npy"))
for Nk in [int(a) for a in sys.argv[1:]] or [50, 512]:
    k = synth.survey_kgrid(Nk) if Nk >= 128 else np.linspace(0.005, 0.3, Nk)
    window_matrix_device(k[:8], tab[:, 0], tab[:, 1:].T, 3, 3)   # context + code objects
    best = None
    for _ in range(3):
        tm = {}
        t = time.perf_counter()
        Wal, p, Waldk, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, timing=tm)
        tm["total_s"] = time.perf_counter() - t
        best = tm if best is None or tm["total_s"] < best["total_s"] else best
    t = time.perf_counter()
    Wal_h, p_h = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    Wf_h, _ = TB.window_fold(k, Wal_h, p_h)
    th = time.perf_counter() - t
    nx = TB.window_tables(k, tab[:, 0], tab[:, 1:].T, 3, 3)[0].size
    flops = 2.0 * 9 * Nk * nx * p.size + 2.0 * 9 * Nk * p.size * Nk
    print(f"Nk={Nk} nx={nx} Np={p.size}: device kernels {best['device_kernels_ms']:.3f} ms ({flops / best['device_kernels_ms'] / 1e9:.1f} TFLOP/s), "
          f"host tables {best['host_tables_s']:.3f} s, call incl. copies {best['call_s']:.3f} s, total {best['total_s']:.3f} s; "
          f"host builder {th:.3f} s; max rel diff {np.abs(Wfold - Wf_h).max() / np.abs(Wf_h).max():.2e}", flush=True)
