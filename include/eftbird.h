/*
 * eftbird.h -- C ABI of the MI355X-native EFTofLSS one-loop engine (libeftbird.so).
 *
 * Drop-in boundary for the PyBird theory-vector hot path of zhaoruiyang98/eftpipe.  The reference has
 * no FFI (it is pure Python); each entry point below names the reference interface it replaces
 * (paths relative to the reference root).  The host side (the eftpipe_amd Python package) binds these with ctypes and
 * re-exposes the reference's own class surface (Common, Bird, NonLinear.PsCf, Bird.setPsCfl, Resum.Ps,
 * APeffect.AP, Window.Window, Binning.transform, Chained.transform, reduce_Plk).
 *
 * Conventions: plain C types only; all arrays are C-contiguous float64 unless stated; the caller owns
 * every host buffer, the library owns the engine and all device memory; one engine per (process, GPU),
 * not re-entrant per engine.  Every function returns 0 on success, non-zero on error; the message is
 * available from eftb_last_error().  There is NO CPU fallback: without a HIP device eftb_create fails.
 */
#ifndef EFTBIRD_H
#define EFTBIRD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eftb_engine eftb_engine;

/* Dimensions + switches fixed at construction.  Replaces the constructor arguments of
 * pybird.Common (eftpipe/pybird/pybird.py:498-514), pybird.NonLinear (:907-915), pybird.Resum (:1230),
 * pybird.APeffect (:1503-1518) as used by EFTLeafKernel.initialize_with_provider (eftpipe/theory.py:423-495). */
typedef struct eftb_config {
    int32_t device;        /* HIP device ordinal */
    int32_t Nl;            /* multipoles computed: 2 (l=0,2) or 3 (l=0,2,4)            Common.Nl */
    int32_t Nk;            /* |co.k|                                                    Common.Nk */
    int32_t Nkin;          /* |kin| of the input linear spectrum (200)                  theory.py:562 */
    int32_t max_batch;     /* device state is sized for this many cosmologies per call */
    int32_t with_resum;    /* IR-resummation on (needs the C11/Cct/C22/C13 pieces)      theory.py:573 */
    int32_t with_ap;       /* Alcock-Paczynski on                                       theory.py:582 */
    int32_t ap_stochastic; /* APeffect(APst=True): distort Pstl too                     pybird.py:1618 */
    int32_t nmu;           /* AP mu nodes (nbinsmu*accboost)                            pybird.py:1538 */
    int32_t ntail;         /* high-k power-law tail length of the loop FFTLog           fftlog.py:146-151 */
    int32_t nxtail;        /* same for the 32-point IR-filter FFTLog                    pybird.py:1293 */
    int32_t nbasis;        /* dimension of the span of the 28 M22 loop matrices (7)     tables.py loop_basis */
    int32_t nbasis13;      /* dimension of the span of the 10 M13 vectors (2)           tables.py loop_basis */
    int32_t NIR, Na, Nklow;/* Resum.NIR, Resum.Na, Common.Nklow                         pybird.py:1247-1259, 560 */
    int32_t with_nnlo;     /* Common.with_NNLO: the k^4 P11 counter-terms PctNNLOl       pybird.py:741-748, 1447-1458, 1615 */
    int32_t optiresum;     /* Common.optiresum: only the BAO peak is resummed (EFTB_T_BAO); the s axis keeps 80 slots, the first 52
                              hold arange(70, 200, 2.5), the rest repeat the last value and carry no weight   pybird.py:553-556, 1382-1400 */
    int32_t dual_coef;     /* Common.IRcutoff = "loop" | "resum": the xi-space pieces use a second FFTLog operator (EFTB_T_GCT2)
                              pybird.py:1151-1160  ("all" needs no switch: both operators are the cut one) */
    int32_t step_batch;    /* largest batch of ONE staged step (0 = max_batch).  With step_batch < max_batch, staged steps that are still queued
                              when the submission thread reaches them are launched TOGETHER, as one batch of up to max_batch cosmologies (same
                              bits per cosmology; each step is still fetched by itself).  No reference counterpart. */
} eftb_config;

/* Constant tables (built on the host by eftpipe_amd/tables.py; shapes in that file).  The first-stage operators are stored as GEMM operands,
 * K-major and zero padded to KP(n) = n rounded up to a multiple of 48 (the K chunk of the matrix-core kernel):
 *   EFTB_T_SKT [KP(Nkin)][Nk]                   P11 = Pin . SKT                                        (pybird.py:694-695)
 *   EFTB_T_GCT [KP(Nkin + ntail)][2 * 129]      FFTLog coefficients (re | im halves) = [Pin | tail] . GCT   (fftlog.py:84-166)
 *   EFTB_T_ECT [2 * 129][KP(Nkin + ntail)]      = GCT transposed: the same product with the batch as the column dimension
 *   EFTB_T_BXT [KP(Nkin + nxtail)][2 * 80]      IR filters X | Y = [Pin | tail'] . BXT                 (pybird.py:1316-1353)
 *   EFTB_T_SPCBAND [65][Nk], EFTB_T_SPLOCAL [Nk][4][4]   the AP splines in B-spline form: coefficient operator and per-interval pieces (tables.bspline_tables)
 * (EFTB_T_TYT is an unused id kept for numbering.) */
enum eftb_table {
    EFTB_T_K = 0, EFTB_T_S, EFTB_T_LNKIN, EFTB_T_SKT, EFTB_T_GCT, EFTB_T_ECT, EFTB_T_LNXTAIL,
    /* one-loop pieces in anti-diagonal form (tables.py antidiagonal_tables, synthesis_table) */
    EFTB_T_AD, EFTB_T_EXP22, EFTB_T_EXPC, EFTB_T_MLJ, EFTB_T_LINVEC, EFTB_T_SYNK, EFTB_T_SYNS, EFTB_T_LINK, EFTB_T_LINS,
    EFTB_T_L11, EFTB_T_LCT, EFTB_T_L22, EFTB_T_L13, EFTB_T_GRP,
    EFTB_T_BXT, EFTB_T_SPCBAND, EFTB_T_SPLOCAL, EFTB_T_TYT, EFTB_T_LNXXTAIL, EFTB_T_WQLAST2, EFTB_T_QPOLY, EFTB_T_H,
    EFTB_T_RSBASIS, EFTB_T_RSBASISS, EFTB_T_RSROWS,   /* matrix-core IR-resummation (Nl = 3): tables.py resum_mfma_tables */
    EFTB_T_MU, EFTB_T_WMU, EFTB_T_LEGMU, EFTB_T_SPBAND, EFTB_T_APFID,
    EFTB_T_LCTN,      /* with_nnlo: Common.lctNNLO [Nl][3] zero padded to [Nl][6]     pybird.py:575 */
    EFTB_T_BAO,       /* optiresum: a(s)[80], b(s)[80], then i_lo, i_hi, i_first, i_end as doubles: inside [i_first, i_end)
                         bao(s) = C(s) - a(s) C(s[i_lo]) - b(s) C(s[i_hi]), 0 elsewhere     Resum.extractBAO pybird.py:1382-1400 */
    EFTB_T_GCT2,      /* dual_coef: FFTLog operator of the xi-space coefficients (layout of EFTB_T_GCT) */
    EFTB_T_GCT2T,     /* dual_coef: its transpose (layout of EFTB_T_ECT) */
    EFTB_T_COUNT
};


/* Device-resident state (per engine, [max_batch] leading axis unless noted). */
enum eftb_buffer {
    EFTB_B_PIN = 0,   /* [B][Nkin]            Bird.Pin                            pybird.py:693 */
    EFTB_B_F,         /* [B]                  Bird.f                                              */
    EFTB_B_DA,        /* [B]                  Bird.DA                                             */
    EFTB_B_H,         /* [B]                  Bird.H                                              */
    EFTB_B_P11,       /* [B][Nk]              Bird.P11                            pybird.py:695 */
    EFTB_B_P22,       /* [B][28][Nk]          Bird.P22                            pybird.py:1076 */
    EFTB_B_P13,       /* [B][10][Nk]          Bird.P13                            pybird.py:1082 */
    EFTB_B_C11,       /* [B][Nl][80]          Bird.C11                            pybird.py:1090 */
    EFTB_B_CCT,       /* [B][Nl][80]          Bird.Cct                            pybird.py:1094 */
    EFTB_B_CC,        /* [B][Nl*38][80]       Bird.C22 ([Nl][28][80]) then Bird.C13 ([Nl][10][80]) */
    EFTB_B_CLOOPL,    /* [B][Nl][12][80]      Bird.Cloopl                         pybird.py:805-846   (written by a REGROUP stage that is not
                                              followed by RESUM in the same run: whole-pipeline runs regroup straight into the resummation records) */
    EFTB_B_TEMPL,     /* [B][nl][24][nx]      rows 0-2 P11l, 3-8 Pctl, 9-20 Ploopl, 21-23 Pstl; (nl, nx) = (Nl, Nk) until an
                                              operator (window / binning / chained) is applied, then that operator's output shape */
    EFTB_B_XY,        /* [B][2][80]           IR filters X(s), Y(s)               pybird.py:1316-1353 */
    EFTB_B_Q,         /* [B][2][Nl][Nl][Nn]   Resum.Q                             pybird.py:1367-1380 */
    EFTB_B_BIAS,      /* [B][24]              b11(3), bct(6), bloop(12), bst(3)   parambasis.py:69-126 */
    EFTB_B_PLK,       /* [B][nl][nx]          reduce_Plk(...).sum() without Picc  parambasis.py:128-136 */
    EFTB_B_COEF,      /* [B][2][129]          FFTLog coefficients (independent half, re/im) */
    EFTB_B_GROWS,     /* [B][25][24]          coefficient rows of P_NG (row 0) and dP/d(gaussian parameter) (rows 1..nG)   parambasis.py:249-316 */
    EFTB_B_LOGP,      /* [B][26]              marginalised ln P, full chi2 at the best fit, best-fit gaussian parameters  marginal.py:79-140 */
    /* with_nnlo only.  The NNLO counter-terms travel as a second template block whose Pctl slots (rows 3-5) hold PctNNLOl and
     * whose other rows are zero: every linear stage (Resum with Q[1] and lctNNLO, AP, the operators) runs on it unchanged. */
    EFTB_B_CCTN,      /* [B][Nl][80]          Bird.CctNNLO                        pybird.py:1098-1101 */
    EFTB_B_TEMPLN,    /* [B][nl][24][nx]      rows 3-5 = Bird.PctNNLOl            pybird.py:741-748 */
    EFTB_B_BIASN,     /* [B][3]               bctNNLO                             parambasis.py:96-106 */
    EFTB_B_GROWSN,    /* [B][25][3]           NNLO part of the EFTB_B_GROWS rows: coefficients of PctNNLOl in P_NG (row 0) and in
                                              dP/d(cr4, cr6 | ctilde) (parambasis.py:303-307, 429-435); zero-initialised */
    EFTB_B_COUNT
};

/* Stages, in the order EFTLeafKernel.calculate_power_spectrum runs them (theory.py:557-609). */
enum eftb_stage {
    EFTB_S_PREP    = 1 << 0,  /* Bird.__init__ spline + NonLinear.Coef        pybird.py:694-695, 1127-1141 */
    EFTB_S_LOOPS   = 1 << 1,  /* makeP22, makeP13                             pybird.py:1074-1086 */
    EFTB_S_CF      = 1 << 2,  /* makeC11, makeCct, makeC22, makeC13           pybird.py:1088-1125 */
    EFTB_S_REGROUP = 1 << 3,  /* Bird.setPsCfl                                pybird.py:737-866 */
    EFTB_S_RESUM   = 1 << 4,  /* Resum.Ps                                     pybird.py:1413-1464 */
    EFTB_S_AP      = 1 << 5,  /* APeffect.AP                                  pybird.py:1598-1621 */
    EFTB_S_PROJECT = 1 << 6,  /* the pipeline operator (eftb_set_pipeline_operator): Window.Window / Binning / Chained, folded
                                 window.py:371-415, binning.py:131-162, chained.py:56-68 */
    EFTB_S_REDUCE  = 1 << 7,  /* reduce_Plk                                   parambasis.py:42-136 */
    EFTB_S_ALL     = 0xff,
    /* single-kernel selectors (profiling / roofline measurement only; need the stage's inputs in place) */
    EFTB_K_P22     = 1 << 8,  /* makeP22 alone: anti-diagonal sums + rows + synthesis */
    EFTB_K_C22     = 1 << 9,  /* makeC22 + makeC13 alone */
    EFTB_K_RESUM   = 1 << 10, /* the main kernel of Resum.Ps alone (operands of an earlier EFTB_S_RESUM run) */
    EFTB_S_LOGP    = 1 << 11, /* Marginalizable.marginalized_logp on the current template block   marginal.py:79-140 */
    EFTB_K_IRFILTER = 1 << 12 /* Resum.IRFilters + Resum.makeQ alone: X(s), Y(s), Q(f) from the inputs (EFTB_B_XY, EFTB_B_Q)   pybird.py:1316-1380 */
};

int  eftb_create(const eftb_config* cfg, eftb_engine** out);
int  eftb_set_table(eftb_engine* e, int table_id, const void* host, size_t nbytes);
int  eftb_finalize(eftb_engine* e);
void eftb_destroy(eftb_engine* e);

/* Run-time switches that the reference keeps on plugin objects rather than on Common. */
enum eftb_option {
    EFTB_O_AP_STOCHASTIC = 0, /* APeffect.APst                                pybird.py:1514, 1618 */
    EFTB_O_JEFFREYS = 1,      /* marginalized_logp(jeffreys=True): drop ln det(F2 / 2 pi)   marginal.py:118-121 */
    EFTB_O_GRAPH = 2,         /* replay whole-pipeline runs (stage masks starting at PREP) from captured HIP graphs: one host call per
                                 step; for busy hosts (no reference counterpart; off by default, also EFTB_GRAPH=1) */
    EFTB_O_CHECK_FINITE = 3,  /* the REDUCE stage flags non-finite P_l(k): the next synchronising call (eftb_sync, eftb_get, eftb_eval_*,
                                 eftb_fetch_*) then returns non-zero naming the cosmology (off by default) */
    EFTB_O_TIME_DOMINANT = 4  /* value n > 0: bracket every n-th launch of the resummation kernel with HIP events on the stream it runs on
                                 (measurement only: bench.py's roofline; read with eftb_dominant_time; inactive while EFTB_O_GRAPH replays
                                 captured runs).  The two event packets cost the pipelined loop about 1.5 % when every launch carries them. */,
    EFTB_O_LATENCY_MODE = 5,  /* 1 (default): a step staged (eftb_stage_inputs) while the GPU is idle runs in latency mode -- one queue, P_lin read from
                                 the staging block, P_l written to mapped host memory by the kernel that forms it: what a sampler wants whose next
                                 step depends on this step's result.  0: always the three-stream layout (a caller that knows more steps follow
                                 at once: the first step of a pipelined loop then does not hold the main queue with its AP stage) */
    EFTB_O_PLK_DIRECT = 6,    /* 1: whole-pipeline runs that end in REDUCE (PREP .. AP | REDUCE, no PROJECT / LOGP; Nl = 3, fast AP path) take the
                                 bias contraction of reduce_Plk (parambasis.py:42-136) FIRST -- it commutes with Resum.Ps and APeffect.AP, linear maps
                                 that act on every template row alike (pybird.py:1413-1464, 1581-1621) -- so one row per multipole instead of 24
                                 goes through them: same P_l(k) (summation order aside), but EFTB_B_TEMPL does not hold the templates of such a
                                 run, and the stage taps EFTB_B_P13 / C11 / CCT (and the synthesised basis rows behind P22 / CC) hold rows already
                                 CONTRACTED with the bias.  Only a run whose mask holds PREP, LOOPS, CF, REGROUP, RESUM, AP and REDUCE together is
                                 taken this way: a sequence split into several eftb_run calls always takes the template path.
                                 0 (default): templates first, as the reference computes them (BirdSnapshot semantics) */
    EFTB_O_TIME_KERNEL = 7,   /* which launches EFTB_O_TIME_DOMINANT brackets, a set of: 1 (default) the resummation kernel, 2 the synthesis launch of
                                 the loop stages (synth_kernel), 4 the heaviest AP kernel (ap_weights_kernel; direct-P_l runs: ap_plk_kernel) -- measurement only */
    EFTB_O_STEP_TRACE = 10,   /* measurement: timing events at nine points of every staged direct-P_l launch (upload start, front end, synthesis start,
                                 operand build end, resummation start / end, spline start, AP end, copy-out end), each on its own stream; read with
                                 eftb_step_trace.  Switching it on (again) restarts the clock. */
    EFTB_O_SUBMIT_HOLD = 9,   /* tests: 1 makes the submission thread leave queued steps where they are until the option is cleared (or any entry point
                                 that needs the queue empty is called) -- the steps queued meanwhile then leave together, as eftb_config.step_batch allows */
    EFTB_O_SUBMIT_THREAD = 8  /* 1 (default; also EFTB_SUBMIT_THREAD=0/1): staged steps handed in while earlier ones are still queued or running are
                                 issued by a submission thread inside the library -- eftb_stage_inputs fills the page-locked block, eftb_run_staged
                                 queues the step and returns; the thread uploads the block and launches the kernels (50-95 us of HIP calls per step
                                 that the sampler's thread no longer pays).  A step staged while the engine is quiescent and the GPU idle is issued by
                                 the caller as before.  Same bits either way.  0: the caller's thread issues every step; 2: every step is queued, whatever the
                                 GPU is doing (tests).  No reference counterpart. */
};
int  eftb_set_option(eftb_engine* e, int option, int value);
/* Sum of the event-bracketed durations [ms] and number of resummation launches since the last reset (EFTB_O_TIME_DOMINANT); waits for
 * the launches still in flight. */
int  eftb_dominant_time(eftb_engine* e, double* ms_sum, long long* launches, int reset);
/* The same for one of the launches EFTB_O_TIME_KERNEL selects: kind 0 the resummation kernel (= eftb_dominant_time), 1 the synthesis launch, 2 the AP
 * kernel. */
int  eftb_kernel_time(eftb_engine* e, int kind, double* ms_sum, long long* launches, int reset);
/* ... and the cosmologies those launches carried: a launch of coalesced staged steps (eftb_config.step_batch) carries several steps' batches. */
int  eftb_kernel_time_ex(eftb_engine* e, int kind, double* ms_sum, long long* launches, long long* cosmologies, int reset);

/* Likelihood of the EFTB_S_LOGP stage.  Replaces, for a batch of walkers on the device,
 * EFTLike.PNG / PG (likelihood.py:483-549: flatten the multipoles over the masked k bins) and
 * Marginalizable.marginalized_logp (marginal.py:79-140).  index[a] = l * nx + x selects data point a from the current
 * template block [nl][24][nx] (after the pipeline operator, if any); data[ndata]; invcov[ndata][ndata] symmetric;
 * 0 <= nG <= 24 marginalised parameters (nG = 0: the plain Gaussian likelihood -chi2 / 2 of row 0) with Gaussian prior N(mu[i], sigma_i^2), sigma_inv[i] = 1 / sigma_i^2 (all zero: flat).
 * Per walker the caller puts the coefficient rows EFTB_B_GROWS (eftpipe_amd.parambasis.gaussian_rows) and reads
 * EFTB_B_LOGP after eftb_run(..., EFTB_S_LOGP, B). */
int  eftb_set_likelihood(eftb_engine* e, int ndata, const int32_t* index, const double* data, const double* invcov,
                         int nG, const double* mu, const double* sigma_inv);

/* Linear post-AP projections.  Window.Window (window.py:371-415), Binning.transform (binning.py:131-162) and
 * Chained.transform (chained.py:56-68) are linear maps of the template block on the k axis and the multipole
 * axis; each is registered once as a dense operator  out[a][row][x] = sum_{l,k} op[a][l][x][k] * in[l][row][k]
 * (the cubic splines onto the window p grid / the bin quadrature points are folded in on the host) and applied
 * on the FP64 matrix cores.  `op` is [nl_out][nl_in][nx_out][nx_in]; requires nl_out*nx_out <= Nl*Nk. */
int  eftb_add_operator(eftb_engine* e, int nl_out, int nx_out, int nl_in, int nx_in, const double* op, int* op_id);
/* The stochastic rows (Pstl, rows 21-23) of operator `op_id` go through the matrix of `st_op_id` (same shape) instead:
 * Window(window_st=False) (window.py:412-415) and FiberCollision(fiberst=False) (pybird.py:1788-1797) leave Pstl alone while
 * binning / chained still act on it.  st_op_id = -1 restores one matrix for all rows. */
int  eftb_set_operator_stochastic(eftb_engine* e, int op_id, int st_op_id);
/* Several tracers per likelihood point (EFTLike(tracers=[LRG, ELG, X]): reference likelihood.py:483-549, cfg 3 of BASELINE).
 * The batch then holds ntr consecutive entries per walker (entry = walker * ntr + tracer), each with its own P_lin, f, DA, H
 * (for a per-tracer AP fiducial pass DA * DA_fid_engine / DA_fid_tracer and H * H_fid_engine / H_fid_tracer) and bias rows.
 *  - eftb_set_pipeline_operator_tracer: the PROJECT stage uses operator op_id for the entries of `tracer` (window / fibre /
 *    binning / chained differ per tracer); all operators must share one shape -- pad a chained output with zero multipoles;
 *  - the LOGP stage treats the ntr entries of a walker as ONE block of ntr * nl multipoles: index[a] = (tracer * nl + l) * nx + x,
 *    rows (EFTB_B_GROWS) per entry with zero rows for the parameters that do not act on that tracer, results per walker
 *    (EFTB_B_LOGP [B / ntr][26], eftb_eval_logp_batch outputs [B / ntr]).
 * eftb_set_tracers resets the per-tracer operators and the likelihood; call it first. */
int  eftb_set_tracers(eftb_engine* e, int ntr);
int  eftb_set_pipeline_operator_tracer(eftb_engine* e, int tracer, int op_id);
/* Apply to the current template block of cosmologies [0, B); the block takes the operator's output shape. */
int  eftb_apply_operator(eftb_engine* e, int op_id, int B);
/* Operator run by the EFTB_S_PROJECT stage of eftb_run / eftb_eval_batch (-1 = none). */
int  eftb_set_pipeline_operator(eftb_engine* e, int op_id);
/* Declare the shape of the template block before eftb_put(EFTB_B_TEMPL, ...) of already-projected templates. */
int  eftb_set_template_dims(eftb_engine* e, int nl, int nx);

/* Input guards.  P_lin must be finite, and positive at its last two samples: the FFTLog extrapolates it beyond kin[-1] with the power
 * law through them (reference fftlog.py:146-151 takes their logarithm); f must be finite, DA and H finite and positive.  The entry
 * points that take host inputs (eftb_eval_batch, eftb_eval_logp_batch, eftb_stage_inputs) check this before anything is copied and
 * return non-zero; for inputs placed with eftb_put the first kernel raises a flag instead and the next synchronising call (eftb_sync,
 * eftb_get, eftb_fetch_*) returns non-zero, naming the cosmology.  The reference has no such check: NaNs propagate silently there. */

/* Host <-> device state.  `offset`/`count` are in elements (doubles). */
int  eftb_put(eftb_engine* e, int buffer_id, size_t offset, const double* host, size_t count);
int  eftb_get(eftb_engine* e, int buffer_id, size_t offset, double* host, size_t count);
size_t eftb_buffer_size(const eftb_engine* e, int buffer_id); /* elements */

/* Launch the selected stages for cosmologies [0, B) (asynchronous; every other entry point orders itself behind it).
 * A full pipeline run spreads over three streams so that consecutive runs overlap: the front half (first stage .. expansions)
 * of run i+1 and the back half (spline, AP, projection, likelihood, reduce) of run i execute beside the resummation.
 * Environment switches, read by eftb_create: EFTB_PREP_OVERLAP=0 / EFTB_AP_OVERLAP=0 put the front / back half back on
 * the main stream (results are bit-identical either way). */
int  eftb_run(eftb_engine* e, int stage_mask, int B);
int  eftb_sync(eftb_engine* e);
/* Same, bracketed by HIP events on the engine stream; *ms receives the elapsed device time. */
int  eftb_run_timed(eftb_engine* e, int stage_mask, int B, int repeats, float* ms);

/* One call = reference theory.py:557-585 for a batch: host inputs in, templates out.
 * templ is [B][nl][24][nx] (rows as EFTB_B_TEMPL; (nl, nx) = the pipeline operator's output shape, else (Nl, Nk));
 * plk [B][nl][nx] may be NULL, else bias must be [B][24]; templ may be NULL when plk is requested (P_l only: 12 KB instead
 * of 0.3 MB per cosmology over PCIe at Nl=3, Nk=512). */
int  eftb_eval_batch(eftb_engine* e, int B, const double* Pin, const double* f, const double* DA,
                     const double* H, double* templ, const double* bias, double* plk);

/* One call = theory + likelihood for a batch of walkers: reference theory.py:557-609, then EFTLike.PNG/PG
 * (likelihood.py:483-549) and Marginalizable.marginalized_logp (marginal.py:79-140).  Host inputs in, one marginalised
 * log-posterior per walker out; nothing else crosses PCIe.  Needs eftb_set_likelihood (and the pipeline operator that
 * brings the templates to the data's shape).  rows is PACKED [B][nG+1][24] (eftpipe_amd.parambasis.gaussian_rows);
 * logp [B / ntr] (NaN where det F2 <= 0: the reference raises there); fullchi2 [B / ntr] and best [B / ntr][nG] may be NULL.
 * With with_nnlo the NNLO part of the rows is whatever EFTB_B_GROWSN holds (eftb_put it beforehand; zero-initialised). */
int  eftb_eval_logp_batch(eftb_engine* e, int B, const double* Pin, const double* f, const double* DA, const double* H,
                          const double* rows, double* logp, double* fullchi2, double* best);

/* Pipelined sampler steps.  The per-step inputs (Pin, f, DA, H, bias rows, likelihood rows) and outputs (EFTB_B_PLK, EFTB_B_LOGP)
 * exist three times: one set is being evaluated, the next is already queued behind it, the third is being fetched from / refilled --
 *     eftb_stage_inputs(step i+1);  eftb_run_staged(step i+1);  eftb_fetch_previous(step i);   ...
 * nothing in this loop waits for the step in flight, so consecutive steps overlap on the GPU (the front half of step i+1 runs beside
 * the resummation / AP of step i) and the PCIe traffic hides behind the kernels.  The host arrays may be reused as soon as
 * eftb_stage_inputs returns (they are copied into page-locked staging memory).  bias ([B][24]) and rows (packed [B][nG+1][24])
 * may be NULL.  No reference counterpart (cobaya evaluates one point at a time). */
int  eftb_stage_inputs(eftb_engine* e, int B, const double* Pin, const double* f, const double* DA, const double* H,
                       const double* bias, const double* rows);
int  eftb_run_staged(eftb_engine* e, int stage_mask, int B);
/* (Round 4.)  A staged step handed in while earlier ones are still queued or running is only QUEUED by eftb_run_staged: the library's submission
 * thread issues it (EFTB_O_SUBMIT_THREAD), together with whatever else is queued by then if eftb_config.step_batch allows.  A launch error of such
 * a step cannot be returned by eftb_run_staged any more: the step still counts (`back` indices are positions in the sequence of steps handed in) and
 * the eftb_fetch_* call of THAT step returns the error; the same holds for a step the calling thread issued itself. */
/* EFTB_B_PLK or EFTB_B_LOGP of the step before the one in flight (waits only for that step). */
int  eftb_fetch_previous(eftb_engine* e, int buffer_id, double* host, size_t count);
/* The same for the step launched `back` (0 = the last one itself, 1 ... 15) steps before the last one.  The engine keeps sixteen sets of per-launch inputs / outputs, so the
 * loop  stage(i); run_staged(i); fetch_back(3) [= step i - 3]  keeps THREE steps queued on the GPU while the host copies results out and
 * prepares the next inputs: the inputs of step i + 1 are then staged before the back half of step i - 1 has finished, and the look-ahead
 * of consecutive steps never runs dry (with back = 2 the look-ahead stream idled ~0.1 ms per step waiting for the host).  Direct-P_l runs
 * (EFTB_O_PLK_DIRECT) are four pipeline stages deep -- front, syntheses + contraction, resummation, AP -- and want back = 4 or more; the set
 * read with back = 15 is the one the sixteenth launch from there refills. */
int  eftb_fetch_back(eftb_engine* e, int back, int buffer_id, double* host, size_t count);
/* The same results without the host copy: *block points at the engine's page-locked host copy of the step's EFTB_B_PLK / EFTB_B_LOGP block
 * (*count = its capacity in elements), valid until NSETS - 1 = 15 more steps have been staged.  For samplers that consume P_l in place (the
 * dependent loop of reference likelihood.py:570-594: chi^2 from P_l, then the next proposal). */
int  eftb_fetch_view(eftb_engine* e, int back, int buffer_id, const double** block, size_t* count);
/* The end of a burst: the steps queued so far leave now, in as few launches as eftb_config.step_batch allows, instead of waiting for more steps to
 * fill a launch (the submission thread lets the queue grow while two launches are in flight).  Returns at once; a sampler calls it before it
 * fetches the last steps of a batch of proposals (reference: the end of one Cobaya `Model.logposterior` sweep over the walkers). */
int  eftb_flush(eftb_engine* e);
/* Where the P_l of the staged step launched NEXT goes: the caller's own page-locked array (eftb_host_alloc / hipHostMalloc / hipHostRegister;
 * `count` doubles, at least B x Nl x Nx of the step's output) instead of the engine's host block -- the copy-out behind the step writes it with
 * the DMA engine, so a sampler that keeps every step's P_l (emulator training sets, the chains' derived output of cobaya's `output_params`,
 * reference theory.py:557-609 per point) pays no second host copy.  One step only; dst = NULL withdraws it.  eftb_fetch_view of that step
 * hands out dst once the step has finished (eftb_fetch_back copies from it); the array must stay allocated and untouched until then.  Refused:
 * pageable memory, engines with a communicator, runs that form no P_l (LOGP); an array that turns out too small for the step's rows fails the
 * launch, reported by the step's fetch.  May be called while earlier steps are queued (it does not wait for them). */
int  eftb_set_step_output(eftb_engine* e, double* dst, size_t count);
/* One sampler step in one call: eftb_stage_inputs + eftb_run_staged + (back >= 0 and that many steps already behind this one)
 * eftb_fetch_view(back, buffer_id).  *block stays NULL while the pipeline is still filling.  Replaces, per step of a batched sampler, what
 * EFTLeafKernel.calculate_power_spectrum + reduce_Plk do per point (theory.py:557-609, parambasis.py:42-136). */
int  eftb_step(eftb_engine* e, int stage_mask, int B, const double* Pin, const double* f, const double* DA, const double* H,
               const double* bias, const double* rows, int back, int buffer_id, const double** block, size_t* count);

/* Host-side cost of the staged steps, for bench.py's accounting (measurement only): enable != 0 switches the clocks on; out (may be NULL) receives
 * {steps issued, of which by the caller's thread, us spent issuing (upload kernel + launches + events, whichever thread), us spent copying inputs
 * into the staging blocks, us the caller spent waiting for results in eftb_fetch_*, launches that carried those steps}; reset != 0 clears the sums. */
int  eftb_submit_stats(eftb_engine* e, int enable, int reset, double out[6]);

/* The last (up to 64) traced launches (EFTB_O_STEP_TRACE), oldest first: out[i][0] = launch number, out[i][1] = cosmologies it carried, out[i][2..13] = the
 * twelve timestamps (the nine of EFTB_O_STEP_TRACE, then upload end, first-stage products end, anti-diagonal sums end) in microseconds since the option was switched on (-1000: point not recorded).  Waits for everything in flight. */
int  eftb_step_trace(eftb_engine* e, double* out, int max_launches, int* n_out);

/* Page-locked host memory for the I/O buffers of eftb_eval_batch / eftb_put / eftb_get: D2H of the template block runs at
 * the PCIe rate instead of through the driver's pageable staging copy.  NULL on failure (eftb_last_error). */
void* eftb_host_alloc(size_t bytes);
void  eftb_host_free(void* p);

/* Multi-GPU: cosmologies are sharded over ranks (one process per GPU); the only exchange is the gather of
 * the per-cosmology P_l(k) to `root` over RCCL (xGMI).  No reference counterpart: cobaya chains are
 * independent MPI processes (reference README.md:29-34).  The 128-byte id comes from rank 0
 * (eftb_comm_unique_id) and is handed to every rank by the host (any side channel). */
int  eftb_comm_unique_id(char id[128]);
int  eftb_comm_init(eftb_engine* e, int nranks, int rank, const char id[128]);
/* Gather EFTB_B_PLK rows [0, B) of every rank into root's device buffer (rank-major), in line behind the kernel that
 * wrote them (on the back-half stream of an overlapped run, else on the compute stream; EFTB_GATHER_ASYNC=1 moves it to a
 * communication stream instead);
 * if host_out != NULL (root only) the gathered block [nranks][B][Nl][Nx] is copied out after the gather. */
int  eftb_gather_plk(eftb_engine* e, int B, int root, double* host_out);
/* Pipelined multi-GPU steps (eftb_stage_inputs / eftb_run_staged / eftb_gather_plk per step, nothing waits for the step in flight): on the
 * root the gathered block rotates through four device buffers, and each is copied on to page-locked host memory by a copy kernel on its own
 * stream as soon as its exchange has finished -- 12.6 MB per step at 8 ranks, which a blocking device-to-host copy into pageable memory would
 * turn into the bottleneck of the whole node.  eftb_fetch_gathered copies the block ([nranks][B][nl][nx], count elements) of the last exchange
 * enqueued (back = 0) or of the one `back` (1 ... 3) exchanges before it into `host`, waiting only for that exchange; eftb_gathered_view hands
 * out the page-locked block itself instead (no host copy; valid until three more exchanges have been enqueued).  With a communicator
 * (eftb_comm_init, to be called before the first eftb_stage_inputs) the staged sets keep P_l in device memory, where RCCL reads it. */
int  eftb_fetch_gathered(eftb_engine* e, int back, double* host, size_t count);
int  eftb_gathered_view(eftb_engine* e, int back, const double** block, size_t* count);

/* Window precompute (reference Window._compute_Wal / _compute_Waldk, window.py:262-359) on the device.  The caller passes
 * the k-independent tables of eftpipe_amd.tables.window_tables: x [nx] (FFTLog samples inside the tabulated window), Qt
 * [Na][Nl][nx] (Q_al(x) with the FFTLog tilt and (-1)^a), T [Nl][nx][Np] (FFT + power-law sum collapsed per l) and the p grid;
 *   Wal [Na][Nl][Nk][Np]  = sum_i Qt_al[i] j_{2a}(k x_i) T_l[i][p]                (the reference's *.npy cache content)
 *   Waldk (optional)      = Wal * [|p - k| < windowk, if withmask] * dp            (window.py:348-359)
 *   Wfold (optional, needs S [Np][Nk], the k -> p cubic-spline matrix) = Waldk S  [Na][Nl][Nk][Nk], what Window.Window applies.
 * All pointers are host buffers, C-contiguous float64; device_ms (optional) receives the kernel time without the copies. */
int  eftb_window_precompute(int device, int Na, int Nl, int Nk, int nx, int Np, const double* k, const double* x,
                            const double* Qt, const double* T, const double* p, int withmask, double windowk,
                            const double* S, double* Wal, double* Waldk, double* Wfold, double* device_ms);

/* Measured FP64 MFMA issue rate (v_mfma_f64_16x16x4_f64), TFLOP/s, for the roofline denominator. */
int  eftb_mfma_f64_peak(int device, double* tflops);

/* Counter calibration: one kernel that streams `bytes` of device memory once with 8 or 16 bytes per lane (two launches; the second is
 * timed).  Run under `rocprofv3 --pmc FETCH_SIZE` it tells how the counter tallies each access width (tools/fetch_calib.py). */
int  eftb_stream_read_probe(int device, size_t bytes, int bytes_per_lane, double* gbps);

const char* eftb_last_error(void);
const char* eftb_version(void);
/* first 16 hex digits of sha256(eftbird.hip + eftb_kernels.hpp) of the build (csrc/Makefile); the key of csrc/isa_counts.json */
const char* eftb_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* EFTBIRD_H */
