"""ctypes binding of libeftbird.so (include/eftbird.h).  Thin on purpose: no numerics live here.

The library is built in-tree (``eftpipe_amd/libeftbird.so``) by ``eftpipe_amd.build.build()``
(``hipcc --offload-arch=gfx950``).  There is no CPU fallback: if the shared object is missing, or no
HIP device is visible when an engine is created, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EFTB_LIB") or os.path.join(_HERE, "libeftbird.so")  # EFTB_LIB: another build of the same library (kernel A/B runs)


class EftbError(RuntimeError):
    pass


class Config(C.Structure):
    """struct eftb_config (include/eftbird.h)"""

    _fields_ = [(n, C.c_int32) for n in (
        "device", "Nl", "Nk", "Nkin", "max_batch", "with_resum", "with_ap", "ap_stochastic", "nmu",
        "ntail", "nxtail", "nbasis", "nbasis13", "NIR", "Na", "Nklow", "with_nnlo", "optiresum", "dual_coef", "step_batch")]


# enum eftb_table / eftb_buffer / eftb_stage (same order as the header)
TABLES = ("K S LNKIN SKT GCT ECT LNXTAIL AD EXP22 EXPC MLJ LINVEC SYNK SYNS LINK LINS L11 LCT L22 L13 GRP "
          "BXT SPCBAND SPLOCAL TYT LNXXTAIL WQLAST2 QPOLY H RSBASIS RSBASISS RSROWS MU WMU LEGMU SPBAND APFID LCTN BAO GCT2 GCT2T").split()
T = {n: i for i, n in enumerate(TABLES)}
BUFFERS = "PIN F DA H P11 P22 P13 C11 CCT CC CLOOPL TEMPL XY Q BIAS PLK COEF GROWS LOGP CCTN TEMPLN BIASN GROWSN".split()
B = {n: i for i, n in enumerate(BUFFERS)}
S_PREP, S_LOOPS, S_CF, S_REGROUP, S_RESUM, S_AP, S_PROJECT, S_REDUCE, K_P22, K_C22, K_RESUM, S_LOGP, K_IRFILTER = (1 << i for i in range(13))

EXPORTS = ("eftb_create eftb_set_table eftb_finalize eftb_set_option eftb_dominant_time eftb_kernel_time eftb_kernel_time_ex eftb_set_likelihood eftb_destroy eftb_add_operator eftb_apply_operator "
           "eftb_set_operator_stochastic eftb_set_tracers eftb_set_pipeline_operator_tracer eftb_set_pipeline_operator eftb_set_template_dims eftb_put eftb_get eftb_buffer_size eftb_run "
           "eftb_sync eftb_run_timed eftb_stage_inputs eftb_run_staged eftb_fetch_previous eftb_fetch_back eftb_fetch_view eftb_step eftb_flush eftb_set_step_output eftb_step_trace eftb_submit_stats eftb_eval_batch eftb_eval_logp_batch eftb_host_alloc eftb_host_free eftb_comm_unique_id eftb_comm_init eftb_gather_plk eftb_fetch_gathered eftb_gathered_view "
           "eftb_window_precompute eftb_mfma_f64_peak eftb_stream_read_probe eftb_last_error eftb_version eftb_source_hash").split()

_lib = None


def load():
    """dlopen the library and declare the prototypes; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EftbError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950); eftpipe_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    dp, vp, sz, i32 = C.POINTER(C.c_double), C.c_void_p, C.c_size_t, C.c_int
    lib.eftb_create.argtypes, lib.eftb_create.restype = [C.POINTER(Config), C.POINTER(vp)], i32
    lib.eftb_set_table.argtypes, lib.eftb_set_table.restype = [vp, i32, vp, sz], i32
    lib.eftb_finalize.argtypes, lib.eftb_finalize.restype = [vp], i32
    lib.eftb_set_option.argtypes, lib.eftb_set_option.restype = [vp, i32, i32], i32
    lib.eftb_dominant_time.argtypes, lib.eftb_dominant_time.restype = [vp, dp, C.POINTER(C.c_longlong), i32], i32
    lib.eftb_kernel_time.argtypes, lib.eftb_kernel_time.restype = [vp, i32, dp, C.POINTER(C.c_longlong), i32], i32
    lib.eftb_kernel_time_ex.argtypes, lib.eftb_kernel_time_ex.restype = [vp, i32, dp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), i32], i32
    lib.eftb_set_likelihood.argtypes, lib.eftb_set_likelihood.restype = [vp, i32, C.POINTER(C.c_int32), dp, dp, i32, dp, dp], i32
    lib.eftb_add_operator.argtypes, lib.eftb_add_operator.restype = [vp, i32, i32, i32, i32, dp, C.POINTER(C.c_int)], i32
    lib.eftb_set_operator_stochastic.argtypes, lib.eftb_set_operator_stochastic.restype = [vp, i32, i32], i32
    lib.eftb_set_tracers.argtypes, lib.eftb_set_tracers.restype = [vp, i32], i32
    lib.eftb_set_pipeline_operator_tracer.argtypes, lib.eftb_set_pipeline_operator_tracer.restype = [vp, i32, i32], i32
    lib.eftb_apply_operator.argtypes, lib.eftb_apply_operator.restype = [vp, i32, i32], i32
    lib.eftb_set_pipeline_operator.argtypes, lib.eftb_set_pipeline_operator.restype = [vp, i32], i32
    lib.eftb_set_template_dims.argtypes, lib.eftb_set_template_dims.restype = [vp, i32, i32], i32
    lib.eftb_destroy.argtypes, lib.eftb_destroy.restype = [vp], None
    lib.eftb_put.argtypes, lib.eftb_put.restype = [vp, i32, sz, dp, sz], i32
    lib.eftb_get.argtypes, lib.eftb_get.restype = [vp, i32, sz, dp, sz], i32
    lib.eftb_buffer_size.argtypes, lib.eftb_buffer_size.restype = [vp, i32], sz
    lib.eftb_run.argtypes, lib.eftb_run.restype = [vp, i32, i32], i32
    lib.eftb_sync.argtypes, lib.eftb_sync.restype = [vp], i32
    lib.eftb_run_timed.argtypes, lib.eftb_run_timed.restype = [vp, i32, i32, i32, C.POINTER(C.c_float)], i32
    lib.eftb_eval_batch.argtypes, lib.eftb_eval_batch.restype = [vp, i32, dp, dp, dp, dp, dp, dp, dp], i32
    lib.eftb_eval_logp_batch.argtypes, lib.eftb_eval_logp_batch.restype = [vp, i32, dp, dp, dp, dp, dp, dp, dp, dp], i32
    lib.eftb_stage_inputs.argtypes, lib.eftb_stage_inputs.restype = [vp, i32, vp, vp, vp, vp, vp, vp], i32  # (const double* parameters declared void*: engine.stage_inputs passes plain addresses)
    lib.eftb_run_staged.argtypes, lib.eftb_run_staged.restype = [vp, i32, i32], i32
    lib.eftb_fetch_previous.argtypes, lib.eftb_fetch_previous.restype = [vp, i32, dp, sz], i32
    lib.eftb_fetch_back.argtypes, lib.eftb_fetch_back.restype = [vp, i32, i32, dp, sz], i32
    lib.eftb_fetch_view.argtypes, lib.eftb_fetch_view.restype = [vp, i32, i32, C.POINTER(dp), C.POINTER(sz)], i32
    lib.eftb_step.argtypes, lib.eftb_step.restype = [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(vp), C.POINTER(sz)], i32
    lib.eftb_flush.argtypes, lib.eftb_flush.restype = [vp], i32
    lib.eftb_set_step_output.argtypes, lib.eftb_set_step_output.restype = [vp, vp, sz], i32
    lib.eftb_step_trace.argtypes, lib.eftb_step_trace.restype = [vp, dp, i32, C.POINTER(i32)], i32
    lib.eftb_submit_stats.argtypes, lib.eftb_submit_stats.restype = [vp, i32, i32, dp], i32
    lib.eftb_host_alloc.argtypes, lib.eftb_host_alloc.restype = [sz], vp
    lib.eftb_host_free.argtypes, lib.eftb_host_free.restype = [vp], None
    lib.eftb_comm_unique_id.argtypes, lib.eftb_comm_unique_id.restype = [C.c_char_p], i32
    lib.eftb_comm_init.argtypes, lib.eftb_comm_init.restype = [vp, i32, i32, C.c_char_p], i32
    lib.eftb_gather_plk.argtypes, lib.eftb_gather_plk.restype = [vp, i32, i32, dp], i32
    lib.eftb_fetch_gathered.argtypes, lib.eftb_fetch_gathered.restype = [vp, i32, dp, sz], i32
    lib.eftb_gathered_view.argtypes, lib.eftb_gathered_view.restype = [vp, i32, C.POINTER(dp), C.POINTER(sz)], i32
    lib.eftb_window_precompute.argtypes, lib.eftb_window_precompute.restype = [i32] * 6 + [dp] * 5 + [i32, C.c_double] + [dp] * 5, i32
    lib.eftb_mfma_f64_peak.argtypes, lib.eftb_mfma_f64_peak.restype = [i32, dp], i32
    lib.eftb_stream_read_probe.argtypes, lib.eftb_stream_read_probe.restype = [i32, C.c_size_t, i32, dp], i32
    lib.eftb_last_error.argtypes, lib.eftb_last_error.restype = [], C.c_char_p
    lib.eftb_version.argtypes, lib.eftb_version.restype = [], C.c_char_p
    lib.eftb_source_hash.argtypes, lib.eftb_source_hash.restype = [], C.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise EftbError(load().eftb_last_error().decode())


def pinned_empty(shape):
    """float64 ndarray in page-locked host memory (eftb_host_alloc); freed when the array and its views are collected."""
    import weakref

    n = int(np.prod(shape))
    lib = load()
    p = lib.eftb_host_alloc(max(n, 1) * 8)
    if not p:
        raise EftbError(lib.eftb_last_error().decode())
    raw = (C.c_double * max(n, 1)).from_address(p)
    weakref.finalize(raw, lib.eftb_host_free, p)
    return np.frombuffer(raw, dtype=np.float64, count=n).reshape(shape)  # the ndarray keeps `raw` alive as its base


def vptr(a):
    """float64 C-contiguous ndarray -> its address as an integer for a `const double*` parameter declared void* (None -> NULL): half the cost of
    dptr's typed pointer object, for the calls a sampler makes every step"""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def dptr(a):
    """float64 C-contiguous ndarray -> double* (None -> NULL)"""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))
