"""Batched device engine: one object = one (process, GPU) handle of libeftbird.so.

``Engine`` owns the constant tables (built on the host by ``tables.build_tables``, uploaded once)
and the device-resident state for up to ``max_batch`` cosmologies.  ``eval_batch`` is the batched
counterpart of ``EFTLeafKernel.calculate_power_spectrum`` (reference eftpipe/theory.py:557-585);
the stage-by-stage methods back the PyBird-compatible classes in ``eftpipe_amd.pybird``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .tables import EngineConfig, build_tables

NROW = 24
ROWS = dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24))


class Engine:
    def __init__(self, cfg: EngineConfig, max_batch: int = 1, device: int = 0, loop_cache=None, coalesce: int = 1):
        """max_batch: cosmologies per call / per staged step.  coalesce > 1: the device state is sized for coalesce x max_batch, and staged steps
        that are still queued when the library's submission thread reaches them leave as one launch of up to that many cosmologies (every step is
        still fetched by itself; same bits per cosmology)."""
        self.lib = L.load()
        self.cfg = cfg
        t = self.tables = build_tables(cfg, loop_cache)
        self.Nl, self.Nk, self.Nkin = cfg.Nl, t["k"].size, t["kin"].size
        self.k, self.kin, self.s = t["k"], t["kin"], t["s"]
        self.max_batch = int(max_batch)
        self.dims = (cfg.Nl, self.Nk)      # shape (nl, nx) of the template block after the last stage run
        self._op_shapes = []               # (nl_out, nx_out, nl_in, nx_in) per registered operator
        self._pipeline_op = -1
        self.ntracers = 1
        self._tracer_ops = []
        self._latency_mode = os.environ.get("EFTB_LATENCY_MODE", "1") != "0"  # the library's default (eftb_set_option(EFTB_O_LATENCY_MODE))
        self._plk_direct = False
        self._step_ptr, self._step_n, self._views = C.c_void_p(), C.c_size_t(), {}
        c = L.Config()
        c.device, c.Nl, c.Nk, c.Nkin, c.max_batch = device, cfg.Nl, self.Nk, self.Nkin, self.max_batch * max(1, int(coalesce))
        c.step_batch = self.max_batch if coalesce > 1 else 0
        c.with_resum, c.with_ap, c.ap_stochastic = int(cfg.with_resum), int(cfg.with_ap), int(cfg.APst)
        c.nmu = cfg.nbinsmu
        c.ntail = t["lnx_tail"].size
        c.nbasis = t["comb22"].shape[1]
        c.nbasis13 = t["comb13"].shape[1] if cfg.with_resum else 0
        if cfg.with_resum:
            c.nxtail = t["lnx_xtail"].size
            c.NIR, c.Na, c.Nklow = (int(x) for x in t["resum_dims"])
        c.with_nnlo = int(cfg.with_NNLO)
        c.dual_coef = int("Gc2" in t)
        c.optiresum = int(cfg.optiresum and cfg.with_resum)
        self._cconf = c
        h = C.c_void_p()
        L.check(self.lib.eftb_create(C.byref(c), C.byref(h)))
        self._h = h
        self._upload(t)
        L.check(self.lib.eftb_finalize(self._h))

    # ------------------------------------------------------------------ tables
    def _set(self, name, arr, dtype=np.float64):
        a = np.ascontiguousarray(arr, dtype=dtype)
        L.check(self.lib.eftb_set_table(self._h, L.T[name], a.ctypes.data_as(C.c_void_p), a.nbytes))

    def _upload(self, t):
        cfg = self.cfg
        self._set("K", t["k"])
        self._set("S", t["s"])
        self._set("LNKIN", np.log(t["kin"]))
        # first-stage operators as GEMM operands: K-major, zero padded to a multiple of 48 rows (include/eftbird.h)
        kpad = lambda n: (n + 47) // 48 * 48
        Nkin, ntail = t["kin"].size, t["lnx_tail"].size

        def stack_gc(G):  # [2, 129, Nkin] (+ the tail operator Ec [2, 129, ntail]) -> [KP, 258]
            out = np.zeros((kpad(Nkin + ntail), 2 * 129))
            for c_ in range(2):
                out[:Nkin, c_ * 129 : (c_ + 1) * 129] = G[c_].T
                out[Nkin : Nkin + ntail, c_ * 129 : (c_ + 1) * 129] = t["Ec"][c_].T
            return out

        skt = np.zeros((kpad(Nkin), self.Nk))
        skt[:Nkin] = t["Sk"].T
        self._set("SKT", skt)
        gct = stack_gc(t["Gc"])
        self._set("GCT", gct)
        self._set("ECT", gct.T)
        self._set("LNXTAIL", t["lnx_tail"])
        cplx = lambda z: np.stack([z.real, z.imag], axis=-1)  # complex -> (re, im) pairs (device double2)
        self._set("AD", cplx(t["ad"].transpose(1, 2, 0)))                      # [j'][t][matrix]: wave-uniform 144-byte records
        exp22 = np.zeros((28, 8))
        exp22[:, : t["comb22"].shape[1]] = t["comb22"]
        self._set("EXP22", exp22)
        self._set("LINVEC", cplx(t["linvec"]))
        self._set("SYNK", t["syn_k"])
        self._set("LINK", t["lin_k"])
        for n in ("L11", "LCT", "L22", "L13"):
            self._set(n, t[n.lower()])
        self._set("GRP", np.concatenate([t["grp22"], t["grp13"]]), np.int32)
        if cfg.with_NNLO:
            self._set("LCTN", t["lctn"])
        if "bao" in t:
            self._set("BAO", t["bao"])
        if "Gc2" in t:
            gct2 = stack_gc(t["Gc2"])
            self._set("GCT2", gct2)
            self._set("GCT2T", gct2.T)
        if cfg.with_resum:
            expc = np.zeros((cfg.Nl * 38, 32))
            expc[:, : t["expand_c"].shape[1]] = t["expand_c"]
            self._set("EXPC", expc)
            self._set("MLJ", cplx(t["mlj"]))
            self._set("SYNS", t["syn_s"])
            self._set("LINS", t["lin_s"])
            nxt = t["lnx_xtail"].size
            bxt = np.zeros((kpad(Nkin + nxt), 2 * 80))   # X | Y = [Pin | tail'] . (BX BY ; TX TY)
            bxt[:Nkin, :80], bxt[:Nkin, 80:] = t["BX"].T, t["BY"].T
            bxt[Nkin : Nkin + nxt, :80], bxt[Nkin : Nkin + nxt, 80:] = t["TX"].T, t["TY"].T
            self._set("BXT", bxt)
            self._set("LNXXTAIL", t["lnx_xtail"])
            self._set("WQLAST2", t["wq_last2"])
            self._set("QPOLY", t["Qpoly"])
            self._set("H", t["H"].transpose(0, 2, 1))
            if "rs_rows" in t:
                self._set("RSBASIS", t["rs_basis"])
                self._set("RSBASISS", t["rs_basis_scaled"])
                self._set("RSROWS", t["rs_rows"], np.int32)
        if cfg.with_ap:
            self._set("MU", t["mu"])
            self._set("WMU", t["wmu"])
            self._set("LEGMU", t["legmu"])
            self._set("SPBAND", t["sp_band"])
            self._set("SPCBAND", t["sp_cband"])
            self._set("SPLOCAL", t["sp_local"])
            self._set("APFID", t["ap_fid"])

    def set_ap_fiducial(self, DA, H):
        """Replace the fiducial (DA, H) of the AP stage (APeffect constructor, pybird.py:1522-1530)."""
        self._set("APFID", np.array([DA, H], dtype=np.float64))

    def set_graph_replay(self, flag):
        """Replay whole-pipeline runs from captured HIP graphs (one host call per step; for busy hosts)."""
        L.check(self.lib.eftb_set_option(self._h, 2, int(bool(flag))))

    def set_check_finite(self, flag):
        """The REDUCE stage flags non-finite P_l(k); the next synchronising call raises, naming the cosmology (off by default)."""
        L.check(self.lib.eftb_set_option(self._h, 3, int(bool(flag))))

    def time_dominant(self, every):
        """Bracket every `every`-th resummation launch with HIP events on its own stream (0 / False: off; True: every launch; measurement only,
        see dominant_time)."""
        L.check(self.lib.eftb_set_option(self._h, 4, int(every)))

    def time_kernels(self, which):
        """Which launches time_dominant brackets, a set of: 1 the resummation kernel (default), 2 the synthesis launch of the loop stages, 4 the AP
        knot weights; read each with kernel_time(kind = 0, 1, 2)."""
        L.check(self.lib.eftb_set_option(self._h, 7, int(which)))

    def kernel_time(self, kind, reset=True, cosmologies=False):
        """(sum of the bracketed durations [ms], launches) of kind 0 resummation / 1 synthesis / 2 AP kernel since the last reset; cosmologies=True:
        (ms, launches, cosmologies those launches carried) -- a launch of coalesced staged steps carries several steps' batches."""
        ms, n, nc = C.c_double(), C.c_longlong(), C.c_longlong()
        L.check(self.lib.eftb_kernel_time_ex(self._h, int(kind), C.byref(ms), C.byref(n), C.byref(nc), int(bool(reset))))
        return (ms.value, n.value, nc.value) if cosmologies else (ms.value, n.value)

    def dominant_time(self, reset=True):
        """(sum of the bracketed durations [ms], launches) since the last reset; waits for launches in flight."""
        ms, n = C.c_double(), C.c_longlong()
        L.check(self.lib.eftb_dominant_time(self._h, C.byref(ms), C.byref(n), int(bool(reset))))
        return ms.value, n.value

    def set_latency_mode(self, flag):
        """True (default): a step staged while the GPU is idle runs in latency mode (one queue, zero-copy P_lin, P_l written to mapped host memory) --
        for samplers whose next step depends on this one.  False: always the three-stream layout (loops that keep several steps queued)."""
        L.check(self.lib.eftb_set_option(self._h, 5, int(bool(flag))))
        self._latency_mode = bool(flag)

    def set_plk_direct(self, flag):
        """True: whole-pipeline runs that end in REDUCE (no PROJECT / LOGP; Nl = 3, fast AP path) take the bias contraction of reduce_Plk FIRST --
        it commutes with the resummation and with the AP stage -- so one row per multipole instead of 24 goes through them.  Same P_l(k)
        (summation order aside); the template block "TEMPL" does not hold the templates of such a run.  False (default): templates first."""
        L.check(self.lib.eftb_set_option(self._h, 6, int(bool(flag))))
        self._plk_direct = bool(flag)

    def set_submit_thread(self, flag):
        """True (default): staged steps handed in while earlier ones are still queued or running are issued by the library's submission thread
        (the sampler's thread only fills the staging block and queues the step).  False: the calling thread issues every launch itself.
        2: every step is queued, also one that finds the GPU idle (tests)."""
        L.check(self.lib.eftb_set_option(self._h, 8, int(flag)))

    def step_trace(self, flag=None):
        """Measurement: step_trace(True) starts recording GPU timestamps at nine points of every staged direct-P_l launch; step_trace() returns
        the last (up to 64) launches as an array [n, 14] (columns 11-13: upload end, first-stage products end, anti-diagonal sums end): launch number, cosmologies, then upload start, front end, synthesis start, operand build
        end, resummation start, resummation end, spline start, AP end, copy-out end in microseconds since the recording started."""
        if flag is not None:
            L.check(self.lib.eftb_set_option(self._h, 10, int(bool(flag))))
            return None
        out, n = np.zeros((64, 14)), C.c_int()
        L.check(self.lib.eftb_step_trace(self._h, L.dptr(out), 64, C.byref(n)))
        return out[: n.value]

    def hold_submissions(self, flag):
        """Tests: True makes the submission thread leave queued steps in the queue until hold_submissions(False) (so that they leave together)."""
        L.check(self.lib.eftb_set_option(self._h, 9, int(bool(flag))))

    def submit_stats(self, enable=True, reset=True):
        """Host-side cost of the staged steps since the last reset (``eftb_submit_stats``): dict with steps, steps issued by the caller's thread, and
        microseconds per step spent issuing (either thread), filling the staging block, and waiting for results in fetch calls."""
        out = np.zeros(6)
        L.check(self.lib.eftb_submit_stats(self._h, int(bool(enable)), int(bool(reset)), L.dptr(out)))
        n = max(out[0], 1.0)
        return {"steps": int(out[0]), "launches": int(out[5]), "issued_by_caller": int(out[1]), "issue_us_per_step": out[2] / n, "fill_us_per_step": out[3] / n,
                "wait_us_per_step": out[4] / n}

    def set_ap_stochastic(self, flag):
        L.check(self.lib.eftb_set_option(self._h, 0, int(bool(flag))))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.eftb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state transfer
    def put(self, name, arr, offset=0):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        L.check(self.lib.eftb_put(self._h, L.B[name], offset, L.dptr(a), a.size))

    def get(self, name, shape, offset=0):
        out = np.empty(shape, dtype=np.float64)
        L.check(self.lib.eftb_get(self._h, L.B[name], offset, L.dptr(out), out.size))
        return out

    def run(self, mask, B=1, sync=True):
        L.check(self.lib.eftb_run(self._h, mask, B))
        if mask & L.S_REGROUP:
            self.dims = (self.Nl, self.Nk)
        if mask & L.S_PROJECT:
            self.dims = self.out_dims()
        if sync:
            L.check(self.lib.eftb_sync(self._h))

    def sync(self):
        L.check(self.lib.eftb_sync(self._h))

    def run_timed(self, mask, B, repeats=1):
        """Device milliseconds for `repeats` back-to-back launches of the stage set (HIP events)."""
        ms = C.c_float()
        L.check(self.lib.eftb_run_timed(self._h, mask, B, repeats, C.byref(ms)))
        return ms.value / repeats

    # ------------------------------------------------------------------ linear operators (window / binning / chained)
    def add_operator(self, op, stochastic=None):
        """op [nl_out, nl_in, nx_out, nx_in]: out[a, row, x] = sum_{l,k} op[a,l,x,k] in[l, row, k]  -> operator id.
        stochastic: a second matrix of the same shape for the Pstl rows (window_st=False / fiberst=False semantics)."""
        op = np.ascontiguousarray(op, dtype=np.float64)
        nl_out, nl_in, nx_out, nx_in = op.shape
        oid = C.c_int()
        L.check(self.lib.eftb_add_operator(self._h, nl_out, nx_out, nl_in, nx_in, L.dptr(op), C.byref(oid)))
        self._op_shapes.append((nl_out, nx_out, nl_in, nx_in))
        if stochastic is not None:
            L.check(self.lib.eftb_set_operator_stochastic(self._h, oid.value, self.add_operator(stochastic)))
        return oid.value

    def apply_operator(self, op_id, B=1, sync=True):
        L.check(self.lib.eftb_apply_operator(self._h, op_id, B))
        self.dims = self._op_shapes[op_id][:2]
        if sync:
            self.sync()

    def set_pipeline_operator(self, op_id):
        """Operator applied by the PROJECT stage of run()/eval_batch (-1 = none)."""
        L.check(self.lib.eftb_set_pipeline_operator(self._h, op_id))
        self._pipeline_op = op_id

    def set_template_dims(self, nl, nx):
        L.check(self.lib.eftb_set_template_dims(self._h, nl, nx))
        self.dims = (nl, nx)

    def set_tracers(self, ntr, operators=None):
        """ntr tracers per likelihood point: batch entry = walker * ntr + tracer (EFTLike(tracers=[...]), reference
        likelihood.py:483-549).  operators: one registered operator id per tracer for the PROJECT stage (same shape; pad a
        chained output with zero multipoles).  Resets any likelihood set before."""
        L.check(self.lib.eftb_set_tracers(self._h, int(ntr)))
        self.ntracers, self._tracer_ops = int(ntr), []
        if operators is not None:
            if len(operators) != ntr:
                raise ValueError("one operator per tracer")
            for t, op in enumerate(operators):
                L.check(self.lib.eftb_set_pipeline_operator_tracer(self._h, t, int(op)))
            self._tracer_ops = [int(op) for op in operators]

    def out_dims(self):
        """(nl, nx) of the templates / P_l that a full pipeline run produces."""
        if self._tracer_ops:
            return self._op_shapes[self._tracer_ops[0]][:2]
        return self._op_shapes[self._pipeline_op][:2] if self._pipeline_op >= 0 else (self.Nl, self.Nk)

    def full_mask(self, reduce=False):
        m = L.S_PREP | L.S_LOOPS | L.S_REGROUP
        if self.cfg.with_resum:
            m |= L.S_CF | L.S_RESUM
        if self.cfg.with_ap:
            m |= L.S_AP
        if self._pipeline_op >= 0 or self._tracer_ops:
            m |= L.S_PROJECT
        if reduce:
            m |= L.S_REDUCE
        return m

    # ------------------------------------------------------------------ whole path
    def _inputs(self, Pin, f, DA, H):
        Pin = np.ascontiguousarray(np.atleast_2d(Pin), dtype=np.float64)
        B = Pin.shape[0]

        def as1(x):  # (arrays that already are float64 [B] pass through: this runs once per step of a sampler loop, and the general form costs 3 us each)
            if x is None or (type(x) is np.ndarray and x.dtype == np.float64 and x.shape == (B,) and x.flags["C_CONTIGUOUS"]):
                return x
            return np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (B,)))

        return B, Pin, as1(f), as1(DA), as1(H)

    def eval_batch(self, Pin, f, DA=None, H=None, bias=None, templates=True, out=None, bias_nnlo=None):
        """Pin [B, Nkin], f/DA/H [B] -> templates [B, nl, 24, nx] (+ P_l [B, nl, nx] if bias [B, 24]);
        (nl, nx) = out_dims(): (Nl, Nk) unless a pipeline operator (window / binning / chained) is set.
        templates=False (needs bias): only P_l crosses PCIe.  out: a template array to fill, e.g. ``pinned_empty`` memory
        re-used across calls (D2H at the PCIe rate instead of through the pageable staging copy).
        With cfg.with_NNLO the NNLO block stays on the device: ``get("TEMPLN", (B, nl, 24, nx))[:, :, 3:6]`` is PctNNLOl."""
        B, Pin, f, DA, H = self._inputs(Pin, f, DA, H)
        nl, nx = self.out_dims()
        if self.cfg.with_NNLO and bias is not None:  # bctNNLO [B, 3] of the NNLO block (parambasis.nnlo_vector); zeros if not given
            self.put("BIASN", np.zeros((B, 3)) if bias_nnlo is None else np.ascontiguousarray(bias_nnlo, dtype=np.float64).reshape(B, 3))
        if not templates and bias is None:
            raise ValueError("templates=False needs bias (nothing to return otherwise)")
        templ = None
        if templates:
            templ = np.empty((B, nl, NROW, nx)) if out is None else out
            if templ.shape != (B, nl, NROW, nx) or templ.dtype != np.float64 or not templ.flags["C_CONTIGUOUS"]:
                raise ValueError(f"out must be C-contiguous float64 {(B, nl, NROW, nx)}")
        plk = None
        if bias is not None:
            bias = np.ascontiguousarray(bias, dtype=np.float64).reshape(B, NROW)
            plk = np.empty((B, nl, nx))
        L.check(self.lib.eftb_eval_batch(self._h, B, L.dptr(Pin), L.dptr(f), L.dptr(DA), L.dptr(H), L.dptr(templ),
                                         L.dptr(bias), L.dptr(plk)))
        if not templates:
            return plk
        return (templ, plk) if bias is not None else templ

    # ------------------------------------------------------------------ pipelined steps (double-buffered inputs / outputs)
    def stage_inputs(self, Pin, f, DA=None, H=None, bias=None, rows=None):
        """Copy the inputs of the NEXT step into the idle input set while the current step runs (``eftb_stage_inputs``)."""
        B, Pin, f, DA, H = self._inputs(Pin, f, DA, H)
        as2 = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        bias, rows = as2(bias), as2(rows)
        L.check(self.lib.eftb_stage_inputs(self._h, B, L.vptr(Pin), L.vptr(f), L.vptr(DA), L.vptr(H), L.vptr(bias), L.vptr(rows)))
        return B

    def run_staged(self, mask, B):
        """Make the staged set current and launch (asynchronous; does not wait for the step in flight)."""
        L.check(self.lib.eftb_run_staged(self._h, mask, B))
        if mask & L.S_REGROUP:
            self.dims = (self.Nl, self.Nk)
        if mask & L.S_PROJECT:
            self.dims = self.out_dims()

    def step(self, mask, Pin, f, DA=None, H=None, bias=None, rows=None, back=-1, fetch="PLK", shape=None, out=None):
        """One sampler step in one library call (``eftb_step``): stage these inputs, launch the step, and -- with back >= 0 -- hand out a read-only
        view of the PLK / LOGP block of the step `back` steps before this one (None while fewer than back + 1 steps are behind it).  Arrays that
        already are C-contiguous float64 of the right shape are passed as they are.  ``out``: a writable C-contiguous float64 array in page-locked
        memory (``pinned_empty``, or a slice of one) that receives THIS step's P_l straight from the copy-out behind the step
        (``eftb_set_step_output``) -- it is complete once a later call has handed out this step (the view returned then IS that memory)."""
        if out is not None:
            if not (type(out) is np.ndarray and out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]):
                raise L.EftbError("step: out must be a writable C-contiguous float64 ndarray (in page-locked memory: Engine.pinned_empty)")
            L.check(self.lib.eftb_set_step_output(self._h, out.ctypes.data, out.size))
        B, Pin, f, DA, H = self._inputs(Pin, f, DA, H)
        if bias is not None and not (type(bias) is np.ndarray and bias.dtype == np.float64 and bias.flags["C_CONTIGUOUS"]):
            bias = np.ascontiguousarray(bias, dtype=np.float64)
        if rows is not None and not (type(rows) is np.ndarray and rows.dtype == np.float64 and rows.flags["C_CONTIGUOUS"]):
            rows = np.ascontiguousarray(rows, dtype=np.float64)
        ptr, n = self._step_ptr, self._step_n
        L.check(self.lib.eftb_step(self._h, mask, B, L.vptr(Pin), L.vptr(f), L.vptr(DA), L.vptr(H), L.vptr(bias), L.vptr(rows), int(back), L.B[fetch],
                                   C.byref(ptr), C.byref(n)))
        if mask & L.S_REGROUP:
            self.dims = (self.Nl, self.Nk)
        if mask & L.S_PROJECT:
            self.dims = self.out_dims()
        if back < 0 or not ptr.value:
            return None
        key = (ptr.value, shape)
        view = self._views.get(key)   # (the engine rotates a handful of page-locked blocks: the ndarray wrappers are made once)
        if view is None:
            size = int(np.prod(shape))
            if n.value < size:
                raise L.EftbError(f"step: the block holds {n.value} elements, asked {size}")
            view = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(size,)).reshape(shape)
            view.flags.writeable = False
            if len(self._views) >= 64:  # (steps with destinations of their own bring a new address every time: the cache is for the engine's few blocks)
                self._views.clear()
            self._views[key] = view
        return view

    def flush(self):
        """No more steps for now: what is queued leaves at once instead of waiting for a launch to fill (``eftb_flush``) -- before fetching the tail of a burst."""
        L.check(self.lib.eftb_flush(self._h))

    def fetch_previous(self, name, shape, out=None, back=1, copy=True):
        """PLK / LOGP of the step launched `back` (0: the last one; 1 ... 15) steps before the last one (``out``: a C-contiguous float64 array of ``shape``
        to fill).  back=3 keeps three steps queued while the host works (see ``pipeline``).  copy=False: a read-only view of the engine's page-locked
        host copy of the block (no 1.5 MB memcpy on the sampler's critical path), valid until fifteen more steps have been staged."""
        if not copy:
            ptr, n = C.POINTER(C.c_double)(), C.c_size_t()
            L.check(self.lib.eftb_fetch_view(self._h, int(back), L.B[name], C.byref(ptr), C.byref(n)))
            size = int(np.prod(shape))
            if n.value < size:
                raise L.EftbError(f"fetch_previous: the block holds {n.value} elements, asked {size}")
            view = np.ctypeslib.as_array(ptr, shape=(size,)).reshape(shape)
            view.flags.writeable = False
            return view
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        L.check(self.lib.eftb_fetch_back(self._h, int(back), L.B[name], L.dptr(out), out.size))
        return out

    def pipeline(self, steps, mask=None, fetch="PLK", depth=None):
        """Generator over a stream of steps (dicts with Pin, f, DA, H and bias and/or rows): stages step i + 1 and launches it
        before fetching the results of step i, so the GPU never waits for the host.  Yields one result per step, in order:
        P_l [B, nl, nx] (fetch="PLK", needs bias) or the raw LOGP block [walkers, 26] (fetch="LOGP", needs rows + a likelihood).
        depth: steps kept queued on the GPU while the oldest one is copied out (1 ... 15; default 3, 4 for direct-P_l runs, whose
        steps are four pipeline stages deep)."""
        if mask is None:
            mask = self.full_mask(reduce=(fetch == "PLK")) | (L.S_LOGP if fetch == "LOGP" else 0)
        if depth is None:
            depth = 4 if self._plk_direct else 3
        if not 1 <= int(depth) <= 15:
            raise ValueError("depth must be 1 ... 15 (the engine rotates sixteen sets of per-launch inputs / outputs)")
        depth = int(depth)
        nl, nx = self.out_dims()
        shape_of = lambda B: (B, nl, nx) if fetch == "PLK" else (B // self.ntracers, 2 + 24)
        queued = []  # batch sizes of the steps launched and not yet fetched (at most `depth`)
        latency_before = self._latency_mode
        self.set_latency_mode(False)  # several steps stay queued: the first one must not take the single-queue form
        try:
            for st in steps:
                B = self.stage_inputs(st["Pin"], st["f"], st.get("DA"), st.get("H"), bias=st.get("bias"), rows=st.get("rows"))
                self.run_staged(mask, B)
                queued.append(B)
                if len(queued) == depth + 1:  # `depth` steps stay queued on the GPU while the oldest one is copied out
                    yield self.fetch_previous(fetch, shape_of(queued.pop(0)), back=depth)
            while queued:  # drain: back = 0 is the step launched last
                yield self.fetch_previous(fetch, shape_of(queued.pop(0)), back=len(queued))
        finally:  # (also when the consumer abandons the generator or a step raises: the caller's setting comes back)
            self.set_latency_mode(latency_before)

    def pinned_empty(self, shape):
        """Page-locked float64 host array for eval_batch(out=...) / put / get."""
        return L.pinned_empty(shape)

    def load_inputs(self, Pin, f, DA=None, H=None, bias=None):
        """Make a batch resident in HBM (what a sampler would keep there between steps)."""
        Pin = np.atleast_2d(Pin)
        self.put("PIN", Pin)
        self.put("F", np.broadcast_to(f, (Pin.shape[0],)))
        if DA is not None:
            self.put("DA", np.broadcast_to(DA, (Pin.shape[0],)))
            self.put("H", np.broadcast_to(H, (Pin.shape[0],)))
        if bias is not None:
            self.put("BIAS", bias)
        return Pin.shape[0]


    # ------------------------------------------------------------------ multi-GPU gather (RCCL)
    def comm_init(self, nranks, rank, unique_id: bytes):
        assert len(unique_id) == 128
        L.check(self.lib.eftb_comm_init(self._h, nranks, rank, unique_id))
        self.nranks, self.rank = nranks, rank

    def gather_plk(self, B, root=0, to_host=False):
        """RCCL gather of P_l[0:B] from every rank to `root` (asynchronous unless to_host)."""
        out = None
        nranks = getattr(self, "nranks", 1)
        if to_host and getattr(self, "rank", 0) == root:
            nl, nx = self.out_dims()
            out = np.empty((nranks, B, nl, nx))
        L.check(self.lib.eftb_gather_plk(self._h, B, root, L.dptr(out)))
        return out


    def fetch_gathered(self, B, back=1, out=None, copy=True):
        """Root only: the gathered block [nranks, B, nl, nx] of the last exchange enqueued (back=0) or of the one `back` exchanges before it.
        copy=False returns a read-only view of the engine's page-locked copy of the block (no host copy; valid until three more exchanges
        have been enqueued) -- at 8 ranks the block is 12.6 MB per step, more than one host thread copies in a step's time."""
        nl, nx = self.out_dims()
        nr = getattr(self, "nranks", 1)
        if not copy:
            ptr, n = C.POINTER(C.c_double)(), C.c_size_t()
            L.check(self.lib.eftb_gathered_view(self._h, int(back), C.byref(ptr), C.byref(n)))
            if n.value < nr * B * nl * nx:
                raise L.EftbError(f"fetch_gathered: the exchange holds {n.value} elements, asked {nr * B * nl * nx}")
            view = np.ctypeslib.as_array(ptr, shape=(nr * B * nl * nx,)).reshape(nr, B, nl, nx)
            view.flags.writeable = False
            return view
        if out is None:
            out = np.empty((nr, B, nl, nx))
        L.check(self.lib.eftb_fetch_gathered(self._h, int(back), L.dptr(out), out.size))
        return out


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    L.check(L.load().eftb_comm_unique_id(buf))
    return buf.raw


def mfma_f64_peak(device=0):
    """Measured v_mfma_f64_16x16x4_f64 issue rate in TFLOP/s."""
    lib = L.load()
    v = C.c_double()
    L.check(lib.eftb_mfma_f64_peak(device, C.byref(v)))
    return v.value
