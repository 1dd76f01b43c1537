// eftbird.hip -- C ABI of libeftbird.so (see include/eftbird.h) and the stage scheduler.
// gfx950 only.  No CPU fallback: every entry point needs a HIP device.
#include "../../include/eftbird.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "eftb_kernels.hpp"

using namespace eftb;

constexpr size_t GEMM_LDS = (size_t)64 * 258 * sizeof(double);  // A tile of gemm_rows_kernel

// What-if builds only (make whatif -> libeftbird_whatif.so, tools/whatif_probe.py): EFTB_WHATIF_SKIP is a set of direct-P_l kernels that are NOT
// launched, to see what each one costs the pipelined step (the results of such a run are garbage).  The product library has no such switch.
#ifdef EFTB_WHATIF
static const int g_whatif = getenv("EFTB_WHATIF_SKIP") ? atoi(getenv("EFTB_WHATIF_SKIP")) : 0;
#define WHATIF_SKIP(bit) ((g_whatif & (bit)) != 0)
#else
#define WHATIF_SKIP(bit) false
#endif

static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct eftb_engine {
    eftb_config c;
    hipStream_t stream = nullptr, side = nullptr;   // side: the small input-only kernels (IR filters, AP prefix sums) run beside the loop path
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evFork = nullptr, evJoin = nullptr, evJoinAP = nullptr;  // evJoin: X, Y, Q(f) ready; evJoinAP: AP prefix sums + knot weights ready
    bool finalized = false;
    void* tab[EFTB_T_COUNT] = {nullptr};
    size_t tab_bytes[EFTB_T_COUNT] = {0};
    double* buf[EFTB_B_COUNT] = {nullptr};
    size_t buf_elems[EFTB_B_COUNT] = {0};
    // scratch
    double *SD = nullptr, *Talt = nullptr, *part = nullptr;
    double *RSA = nullptr, *RSC = nullptr;  // matrix-core resum: A = Q V8^T [B][80][8] (Nl = 2), per-s records [B][NS][48]
    double *RSAS = nullptr, *RSAS2 = nullptr;  // Nl = 3: the A operand per s [B][NS][4][64][2] (resum_as_kernel: inputs only), two sets like RSA / RSC
    hipEvent_t evXY = nullptr, evAS = nullptr;  // X, Y of this run are in place (look-ahead stream); the per-s operand built beside the chain is ready
    double *RSA2 = nullptr, *RSC2 = nullptr;  // second operand set: the look-ahead builds run i+1's operands while run i's resummation reads its own
    hipEvent_t evRsDone[2] = {nullptr, nullptr};  // the resummation that read operand set [slot] has finished
    unsigned rs_step = 0;
    // flow control of asynchronous runs: at most RUN_DEPTH runs are in flight (a caller that never synchronises cannot grow the runtime's
    // command queues without bound; four runs ahead is more than the three-stream layout can use)
    static constexpr int RUN_DEPTH = 8;
    hipEvent_t evRun[RUN_DEPTH] = {};
    unsigned long long run_seq = 0;
    bool nnlo_inline = true;  // EFTB_NNLO_INLINE=0: with_NNLO steps never take the three-stream layout
    bool prep_ahead = true;  // EFTB_PREP_AHEAD=0: regrouping and operand build stay on the main stream in front of the resummation
    // duration of the dominant kernel inside pipelined steps (EFTB_O_TIME_DOMINANT): HIP events on the stream it is launched on
    static constexpr int NTIMER = 8;
    hipEvent_t evT0[NTIMER] = {}, evT1[NTIMER] = {};
    bool timer_busy[NTIMER] = {};
    unsigned timer_next = 0;
    int time_dominant = 0;          // 0 off; n: every n-th launch of the timed kernel is bracketed
    int time_kernel = 1;            // EFTB_O_TIME_KERNEL, bits: 1 the resummation kernel, 2 the synthesis launch of the loop stages, 4 the AP kernel (knot weights; direct-P_l runs: ap_plk_kernel)
    unsigned long long time_seqk[3] = {};
    int timer_kind[NTIMER] = {};    // which kernel the pair of a slot brackets (see time_kernel)
    double timer_ms[3] = {};
    long long timer_n[3] = {};
    long long timer_cosmo[3] = {};   // cosmologies the timed launches carried (a launch of coalesced steps carries several steps' worth)
    int timer_B[8] = {};             // batch of the launch slot t brackets (NTIMER = 8)
    int launch_B = 0;                // batch of the launch being issued (launch_stages_impl)
    double *APP = nullptr, *APR = nullptr, *APP2 = nullptr, *APR2 = nullptr;  // AP: prefix sums over mu [B][nmu+1][Nl*Nl*4], roots [B][nmu]
    // AP fast path (ap_weights_kernel / ap_rows_kernel): knot weights [B][tiles][APW_DCAP][Nl][Nl][2][64], lowest knot per k [B][tiles * 64],
    // window per tile [B][tiles]; second set for the look-ahead of overlapped runs (swapped together with APP / APR)
    double *APW = nullptr, *APW2 = nullptr;
    int *API = nullptr, *API2 = nullptr;
    int4 *APM = nullptr, *APM2 = nullptr;
    bool ap_fast = true;  // EFTB_AP_FAST=0: every tile through ap_direct_kernel (the reference's quadrature, otherwise the fallback)
    // EFTB_AP_MODE: 0 = knot weights + banded product (ap_weights / ap_rows; tiles it cannot hold fall back to ap_direct), 1 = interval moments
    // from the mu prefix sums (ap_moments_kernel: cost grows with the intervals crossed, not with a table size), 2 = the reference's quadrature
    // everywhere; default: 1 for k grids so fine that a 2 % distortion at the last k crosses more than half the knots the weight tables hold
    int ap_mode = 0;
    int ad_waves = 2;     // waves per workgroup of antidiag_kernel (EFTB_AD_WAVES=2|4): same-box A/B 2 against 4: +1.3 % resident, +1.8 % staged loop (two free wave slots on a CU are found sooner than four)
    int gd_waves = 4;     // waves per workgroup (K split) of gemm_direct_kernel (EFTB_GD_WAVES=2|4)
    int ap_ring = 2;      // knots whose weights ap_rows_kernel keeps in flight (EFTB_AP_RING=2|4)
    bool fuse_cf = true;  // EFTB_FUSE_CF=0: always through regroup_cf_kernel (A/B switch)
    double *PA1 = nullptr, *PA2 = nullptr, *PA2T = nullptr, *PA3 = nullptr;  // operand rows of the first-stage GEMMs (prep_rows_kernel)
    double* coefT = nullptr;                 // FFTLog coefficients, cosmology-contiguous [2][129][B]
    double2* SAD = nullptr;                  // anti-diagonal partial sums S[AD_CH][B][nbasis + nbasis13][257]
    double *A22 = nullptr, *A13 = nullptr;   // synthesis rows of the P22 basis [B][BAS22][KSYN] and of P13 [B][10][KLIN]
    double *ACF = nullptr, *ALC = nullptr;   // ... of the weighted xi basis [B][BASC][KSYN] and of C11 / Cct [B][2 Nl][KLIN]
    double *Y22 = nullptr, *YCF = nullptr;   // synthesised basis rows [B][BAS22][Nk], [B][BASC][NS]
    // linear post-AP operators (window / binning / chained), stored K-major for gemm_rows_kernel
    struct Op { int nl_out, nx_out, nl_in, nx_in, ld; double* dev; int st_op; };  // st_op >= 0: matrix used for the Pstl rows instead
    std::vector<Op> ops;
    int pipeline_op = -1;
    int ntr = 1;                    // tracers per likelihood point: batch entry = walker * ntr + tracer (eftb_set_tracers)
    std::vector<int> tracer_ops;    // per-tracer pipeline operators (eftb_set_pipeline_operator_tracer), empty = pipeline_op for all
    // likelihood of the LOGP stage (eftb_set_likelihood)
    int like_ndata = 0, like_nG = 0, jeffreys = 0;
    int like_nl = 0, like_nx = 0;   // template-block shape the data index was validated against (checked again when the LOGP stage launches)
    // input / output guards: kernels raise flags in mapped page-locked memory (status[0]: 1 + index of a cosmology whose P_lin is
    // non-finite or non-positive at its last two samples; status[1]: 1 + index of a cosmology with a non-finite P_l, EFTB_O_CHECK_FINITE)
    // One pair per rotating set of the staged API (status[2 q], status[2 q + 1]: raised and cleared with step q's own results, whatever runs
    // beside it) and one pair (slot NSETS) for runs on the engine's own buffers (eftb_put / eftb_run / eftb_eval_batch).
    int* status = nullptr;
    int status_slot = 16;           // = NSETS: pair the kernels of the next launch raise (eftb_run_staged: the step's set)
    bool check_finite = false;
    std::vector<double> like_host;  // eftb_eval_logp_batch: D2H landing block [B][MARG_OUT]
    int* like_index = nullptr;
    double *like_data = nullptr, *like_invcov = nullptr, *like_mu = nullptr, *like_sinv = nullptr;
    double *like_V = nullptr, *like_U = nullptr;  // LOGP scratch: V and U = V C^-1, packed [walkers][nG + 1][ndata]
    // EFTB_O_GRAPH / EFTB_GRAPH=1: whole-pipeline runs (masks that start at PREP) are captured once into a HIP graph per launch
    // state and replayed -- one host call per step instead of ~30, for hosts whose cores are busy or throttled.  Off by default: on
    // ROCm 7.2 the replay is 2-3 % slower than the plain launches when the host keeps up (0.566 vs 0.553 ms per 128, 0.169 vs 0.144 ms at B = 1)
    struct GraphEntry {
        int mask, B, cur_nl, cur_nx;
        unsigned long long epoch;
        const double *templ, *talt, *templn;
        hipGraph_t graph;
        hipGraphExec_t exec;
        const double *post_templ, *post_talt, *post_templn;  // state after the run
        int post_nl, post_nx;
    };
    std::vector<GraphEntry> graphs;
    unsigned long long epoch = 0;  // bumped by every setter that can change what launch_stages launches
    bool use_graphs = false;
    // Back-to-back asynchronous runs: the first stage of run i+1 (P11 spline + FFTLog, 20 us, inputs only) goes to its own stream and
    // overlaps the tail of run i (resum / AP) -- it waits for evInFree, recorded once run i no longer reads P11 / the coefficients
    hipStream_t pre = nullptr;
    hipEvent_t evPrep = nullptr, evInFree = nullptr;
    bool prep_overlap = true, inputs_settled = false;  // EFTB_PREP_OVERLAP=0 disables; inputs_settled: set by eftb_run only
    // back half (spline, AP, reduce) of run i on its own stream beside the resummation of run i+1 (EFTB_AP_OVERLAP=1): three template
    // blocks rotate (work block of this run, work block of the previous run = AP output of this one, AP output of the previous run)
    hipStream_t back = nullptr;
    hipEvent_t evResum = nullptr, evBack[2] = {nullptr, nullptr};
    double* T3 = nullptr;
    double *TaltN = nullptr, *T3N = nullptr;  // the same rotation for the NNLO template block (with_NNLO steps on the three-stream layout)
    bool ap_overlap = true, back_pending = false, allow_back = false;  // EFTB_AP_OVERLAP=0 disables
    unsigned back_step = 0;
    hipStream_t opstream = nullptr;  // where the operator launchers put their kernels (null: the main stream)
    // Pipelined sampler steps (eftb_stage_inputs / eftb_run_staged / eftb_fetch_back): four sets of the per-step inputs (PIN, F, DA,
    // H, BIAS, GROWS) and outputs (PLK, LOGP) -- up to three launched and not yet fetched, one whose results are being fetched / refilled
    static constexpr int NSETS = 16; // rotating sets of per-launch inputs / outputs: the host may run fifteen steps ahead of the step it fetches (round 3: 8; a step's way from
                                     // the staging block to the fetched P_l is four pipeline stages long since the direct-P_l runs: four sets left bubbles)
    double* setbuf[NSETS][EFTB_B_COUNT] = {{nullptr}};
    double* setblock[NSETS] = {};   // one contiguous input block per set (PIN, F, DA, H, BIAS, GROWS at stage_off[])
    double* orig[EFTB_B_COUNT] = {nullptr};              // the engine's own buffers, current until the first staged run
    size_t stage_off[EFTB_B_COUNT] = {0};
    // page-locked staging, one block per STEP (a launch may carry several queued steps -- see SubCmd -- so staging blocks and device sets
    // rotate separately: slot = step % NSLOT, set = launch % NSETS)
    static constexpr int NSLOT = 32;
    double* stage_host[NSLOT] = {};
    size_t stage_elems = 0;
    hipStream_t cpy = nullptr;
    static constexpr int NLRING = 64;       // ring of the upload events, by launch number (a staging block is refilled 16 steps <= 16 launches later)
    hipEvent_t evStagedR[NLRING] = {}, evStagedAllR[NLRING] = {};
    hipEvent_t evSetDone[NSETS] = {};
    // Latency mode (a sampler whose next step depends on this step's P_l: nothing is queued behind the step being staged).  eftb_stage_inputs
    // finds the GPU idle -> this set's step runs on ONE queue (every cross-queue hand-over costs 15-18 us of dispatch latency, and there is no
    // neighbouring step to overlap with), the first kernel reads P_lin straight from the page-locked staging block (the 0.2 MB device copy
    // follows off the critical path; only the few KB of f / DA / H / bias are waited for, copied on the compute queue itself), and P_l is
    // written to mapped host memory by the kernel that forms it (no DMA phase).  (Measured and dropped: the AP tables beside the resummation
    // instead of beside the loop chain -- the chain gained 20 us, the resummation lost 39.)
    // direct-P_l runs in the three-stream layout keep their FRONT (operand rows, first-stage products, anti-diagonal sums, synthesis rows, Q(f):
    // functions of the inputs alone) on the side stream, one run ahead of the rest of the look-ahead chain (syntheses, contraction, operand
    // build): two sets of everything the front writes, swapped per run
    struct FrontSet {
        double *PA1 = nullptr, *PA2 = nullptr, *PA2T = nullptr, *PA3 = nullptr, *coefT = nullptr, *A22 = nullptr, *A13 = nullptr, *ACF = nullptr, *ALC = nullptr;
        double *P11 = nullptr, *COEF = nullptr, *XY = nullptr, *Q = nullptr;
        double2* SAD = nullptr;
    } alt;
    hipEvent_t evFront = nullptr, evFrontFree[2] = {nullptr, nullptr};  // this run's front is done (side stream); the readers of front set [slot] are done
    unsigned front_step = 0;
    bool prev_front_side = false;
    bool upload_on_side = true;         // EFTB_UPLOAD_ON_SIDE=0: staged uploads always on the copy stream
    bool ap_plk_fused = false;          // the AP stage of direct-P_l runs as ap_plk_fused_kernel (k grids whose tables fit the LDS; EFTB_AP_PLK_FUSED=0: ap_prefix + ap_plk_mom)
    bool ap_plk_nodes = false;          // EFTB_AP_PLK_NODES=1: the AP stage of direct-P_l runs as the node quadrature (ap_plk_kernel, round 3) instead of the moment form
    double* PLK0 = nullptr;             // direct-P_l runs with a PROJECT stage: P_l [B][Nl][Nk] as the AP stage leaves it (the operator then takes ONE row per cosmology)
    bool plk_host_written = false;      // the kernel that formed the final P_l of the last launch also stored it to plk_host_out
    bool plk_direct = false;            // EFTB_O_PLK_DIRECT: whole-pipeline runs that end in REDUCE contract with the bias first (regroup_plk_kernel)
    bool latency_auto = true;           // EFTB_LATENCY_MODE=0 disables
    bool set_latency[NSETS] = {};        // the launch on this set is a latency-mode step (evStagedAllR: the whole staging block has been uploaded; evStagedR: the part its first kernels wait for)
    bool lat_run = false;                // the run being launched is a latency-mode step
    double* plk_host_out = nullptr;      // latency mode: where P_l goes besides the device buffer
    int cur_set = 0;
    // staged sets keep P_l in device memory: with a communicator RCCL sends from it; without one the step's last stream copies it to page-locked
    // host memory with the DMA engine (plk_host) -- measured 0.447 ms per step against 0.455 ms with REDUCE writing mapped host memory over PCIe
    // from its waves (EFTB_STAGED_PLK_MAPPED=1 keeps that form)
    bool staged_plk_device = false;
    double* plk_host[NSETS] = {};
    bool generic_resum = false;  // EFTB_GENERIC_RESUM=1: Nl = 2 on resum_kernel<2> (the pre-matrix-core kernel, kept for A/B checks)
    int cur_nl = 0, cur_nx = 0;  // shape of the template block
    int resum_splits = 1;
    int Nn = 0;
    double* sm2 = nullptr;  // s^-2 row scale of Cct
    double* sm4 = nullptr;  // s^-4 row scale of CctNNLO
    double* XB = nullptr;  // optiresum: BAO-extracted copies of C11 [B][Nl][80], Cct [B][Nl][80], Cloopl [B][Nl][12][80], in this order
    double *coef2 = nullptr, *coefT2 = nullptr;  // dual_coef: FFTLog coefficients of the xi-space pieces (layouts of EFTB_B_COEF / coefT)
    double *ZC = nullptr, *ZC2 = nullptr;  // with_nnlo: zeros standing in for C11 [B][Nl][80] / Cloopl [B][Nl][12][80] in the NNLO pass of Resum.Ps
    // RCCL gather (multi-GPU batches)
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    double* gathered = nullptr;
    double* gathered2[NSETS] = {};  // the gathered block rotates through three buffers: the root reads step i while steps i + 1, i + 2 are exchanged
    hipEvent_t evGath2[NSETS] = {};
    // the root's gathered block is copied on to page-locked host memory by the DMA engine, in line behind the exchange (no host call blocks on
    // the transfer): eftb_fetch_gathered / eftb_gathered_view read it there
    double* gath_host[NSETS] = {};
    hipEvent_t evGathHost[NSETS] = {};
    size_t gath_elems[NSETS] = {};
    int gath_set[NSETS];  // (all NSETS after eftb_create) set whose P_l exchange `slot` carries (its flags are checked when the gathered block is handed out; NSETS: the engine's own buffers)
    static_assert(NSETS == 16, "status_slot's default names NSETS");
    int gather_slot = 0;
    // the gather runs on its own stream from a snapshot of P_l, so that it overlaps the next step's kernels
    hipStream_t comm_stream = nullptr;
    hipEvent_t evSnap = nullptr, evGathered = nullptr;
    double* plk_snap = nullptr;
    // Submission thread (EFTB_O_SUBMIT_THREAD, default on).  A staged step costs the host 50-95 us of HIP calls (nine kernel launches, ~20 event
    // records / waits) on top of ~30 us of validation and copies into the staging block -- more than the 0.12 ms the GPU needs for a direct-P_l
    // step of 128.  So while earlier steps are still in flight, eftb_stage_inputs only fills the staging block and eftb_run_staged only QUEUES the
    // step: this thread issues the upload kernel and every launch of the step (exactly the code the caller's thread runs otherwise), and the
    // caller goes on to the next step's inputs.  A step that finds the engine quiescent and the GPU idle (a dependent sampler's step) is issued by the
    // caller itself, as before: no hand-over latency.  Every other entry point waits until the queue is empty (sub_drain), so the engine's state
    // is only ever touched by one thread at a time; the queue and the per-step completion records are the only shared data.
    // Coalescing: steps that are still queued when the submission thread gets to them leave as ONE launch chain (their staging blocks are gathered
    // into one device set, row after row; each step's results are its rows of the set's output blocks).  At 128 cosmologies every kernel of a
    // step is a single ragged round of waves; two or four steps together cost far less than two or four launches (measured: 0.108 ms per 128 at 256
    // against 0.126).  Needs eftb_config.step_batch < max_batch (the device state is sized for max_batch; a step brings at most step_batch).
    struct SubCmd { int slot, mask, B, has_rows; unsigned long long step; double* out; size_t out_count; };  // out: the caller's page-locked destination of this step's P_l (eftb_set_step_output), or null
    static constexpr int SUBQ = 64;          // ring of queued steps (at most NSLOT - 1 can be pending: every step owns a staging block)
    SubCmd sub_ring[SUBQ];
    std::atomic<unsigned long long> sub_tail{0};   // commands pushed (caller's thread)
    std::atomic<unsigned long long> sub_head{0};   // commands completed (submission thread)
    std::atomic<bool> sub_flush{false};            // eftb_flush: what is queued leaves now (the flow control does not wait for a full group); cleared once the queue is empty
    std::atomic<bool> sub_sleeping{false}, sub_stop{false}, sub_hold{false};   // sub_hold (EFTB_O_SUBMIT_HOLD, tests): queued steps are not taken until it is cleared
    std::mutex sub_mx;
    std::condition_variable sub_cv;
    std::thread sub_thread;
    int sub_mode = 1;   // 0: the caller issues every step; 1: queued unless the engine is quiescent and the GPU idle; 2: always queued (tests)
    bool sub_started = false;
    static constexpr int SUBREC = 64;        // records of the last steps: where the step's rows are, rc + message of its launch (surface when the step is fetched)
    struct StepRec { int set, row, B, rc, nl, nx; unsigned long long launch; double* out; };
    StepRec rec[SUBREC] = {};
    char sub_err[SUBREC][256] = {};
    int coalesce_max = 1;                    // steps per launch at most (EFTB_COALESCE; 1 unless step_batch < max_batch)
    int sub_low = 2;                         // ... and below which it launches whatever is queued at once (EFTB_SUB_LOW); between the two it waits for a full group
    int sub_inflight = 4;                    // launches the submission thread keeps in flight before it lets the queue grow (EFTB_SUB_INFLIGHT)
    unsigned long long launch_seq = 0;       // staged launches issued so far (issuing thread)
    unsigned long long launch_done = 0;      // ... of which known to have finished (issuing thread's view)
    int lring_set[NLRING] = {};              // set of launch (seq % NLRING)
    std::atomic<unsigned long long> steps_launched{0};  // staged steps whose launch has been issued (by either thread), in order
    unsigned long long steps_submitted = 0;   // staged steps handed in so far (caller's thread) = what eftb_fetch_back counts `back` from
    int stg_slot = -1, stg_B = 0, stg_rows = 0;  // inputs staged into block stg_slot, waiting for eftb_run_staged
    double* stg_out = nullptr;                   // eftb_set_step_output: where the P_l of the step launched NEXT goes (the caller's page-locked array), ...
    size_t stg_out_count = 0;                    // ... which holds this many doubles
    bool stg_inline = false, stg_lat = false;    // ... to be issued by the caller's thread (engine quiescent, GPU idle), as a latency-mode step
    unsigned long long slot_step[NSLOT] = {}; // 1 + the step that used the staging block last (0: never used): its launch must be over before the block is refilled
    bool slot_latency[NSLOT] = {};            // ... which was a latency-mode step (its first kernel read P_lin from the block itself)
    size_t slot_off[EFTB_B_COUNT] = {0};      // layout of a staging block (one step: step_batch rows per array)
    size_t slot_elems = 0;
    unsigned long long launch_n = 0;          // (statistics: launches that carried the issue_n steps)
    // completion words of the staged sets in mapped page-locked memory: the step's last stream writes 1 + its step number behind everything else
    // (hipStreamWriteValue64), so the fetch of a step polls plain memory -- a hipEventQuery loop on the caller's thread takes the runtime's locks
    // thousands of times per step and slows the submission thread's launches down (measured: no gain from the thread at all with the event spin)
    volatile unsigned long long* set_done = nullptr;
    unsigned long long set_word[NSETS] = {};  // 1 + the launch whose completion write was enqueued for the set (else the set's event is what to wait for)
    unsigned long long set_cword[NSETS] = {}; // ... whose compute-done write was (set_done[NSETS + q])
    bool plk_dma = true;                      // EFTB_PLK_DMA=0: P_l of a pipelined launch through copy16_kernel instead of the DMA engine
    bool done_words = true;                   // EFTB_DONE_WORDS=0: event queries (A/B); also the fall-back when the write command is refused
    // EFTB_O_STEP_TRACE: GPU timestamps (timing events) at nine points of every staged direct-P_l launch -- upload start, front end, synthesis
    // start, operand build end, resummation start / end, spline start, AP end, copy-out end -- on their own streams: an undistorted timeline of
    // the pipelined loop (rocprofv3 doubles the host's launch cost, and under it the host, not the GPU, sets the pace); tools/step_trace.py
    static constexpr int NTP = 12, NTRACE = 64;   // (points 9-11: inside the front -- upload end, first-stage products end, anti-diagonal sums end)
    hipEvent_t evTrace[NTRACE][NTP] = {};
    hipEvent_t evTraceBase = nullptr;
    int trace_B[NTRACE] = {};
    unsigned long long trace_launch[NTRACE] = {};
    bool trace_on = false;
    int trace_slot = -1;      // ring slot of the launch being issued (-1: not traced)
    unsigned long long trace_n = 0;
    // EFTB_SUB_STATS=1: host time the issuing thread spends per step (printed by eftb_destroy)
    bool sub_stats = false;
    double issue_ns = 0.0, fill_ns = 0.0, wait_ns = 0.0;
    unsigned long long issue_n = 0, inline_n = 0;
};

static void sub_stop_thread(eftb_engine* e);
static inline void sub_drain(eftb_engine* e);   // every entry point but the staged-step calls first lets the submission thread finish its queue

// RCCL is resolved lazily so that single-GPU users never load it
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
    if (g_rccl.handle) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("cannot load librccl.so: %s", dlerror());
#define RSYM(field, name)                                                 \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if (!g_rccl.field) return fail("librccl.so lacks %s", name);
    RSYM(GetUniqueId, "ncclGetUniqueId")
    RSYM(CommInitRank, "ncclCommInitRank")
    RSYM(CommDestroy, "ncclCommDestroy")
    RSYM(GroupStart, "ncclGroupStart")
    RSYM(GroupEnd, "ncclGroupEnd")
    RSYM(Send, "ncclSend")
    RSYM(Recv, "ncclRecv")
    RSYM(GetErrorString, "ncclGetErrorString")
#undef RSYM
    g_rccl.handle = h;
    return 0;
}

#define NCCLCHK(expr)                                                                            \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess) return fail("%s failed: %s", #expr, g_rccl.GetErrorString(_r));   \
    } while (0)

static inline void trace_point(eftb_engine* e, int p, hipStream_t st) {
    if (e->trace_slot >= 0) (void)hipEventRecord(e->evTrace[e->trace_slot][p], st);
}

static inline size_t kpad(int n) { return (size_t)(n + SYN_KPAD - 1) / SYN_KPAD * SYN_KPAD; }  // K of a first-stage GEMM: padded like the synthesis tables

static size_t need_table_bytes(const eftb_config& c, int id) {
    const size_t D = sizeof(double);
    const int Nn = 2 * c.NIR * c.Na;
    switch (id) {
        case EFTB_T_K: return D * c.Nk;
        case EFTB_T_S: return D * NS;
        case EFTB_T_LNKIN: return D * c.Nkin;
        case EFTB_T_SKT: return D * kpad(c.Nkin) * c.Nk;
        case EFTB_T_GCT: case EFTB_T_ECT: return D * kpad(c.Nkin + c.ntail) * 2 * NCH;
        case EFTB_T_LNXTAIL: return D * c.ntail;
        case EFTB_T_AD: return 2 * D * (size_t)(c.nbasis + (c.with_resum ? c.nbasis13 : 0)) * NPOW * AD_T;
        case EFTB_T_EXP22: return D * 28 * BAS22;
        case EFTB_T_EXPC: return c.with_resum ? D * (size_t)c.Nl * 38 * BASC : 0;
        case EFTB_T_MLJ: return c.with_resum ? 2 * D * c.Nl * NPOW : 0;
        case EFTB_T_LINVEC: return 2 * D * (size_t)(10 + (c.with_resum ? (c.with_nnlo ? 3 : 2) * c.Nl : 0)) * NCH;  // M13, Mcf11, Mcfct (, McfctNNLO)
        case EFTB_T_SYNK: return D * (size_t)KSYN * c.Nk;
        case EFTB_T_LINK: return D * (size_t)KLIN * c.Nk;
        case EFTB_T_SYNS: return c.with_resum ? D * (size_t)KSYN * NS : 0;
        case EFTB_T_LINS: return c.with_resum ? D * (size_t)KLIN * NS : 0;
        case EFTB_T_L11: return D * c.Nl * 3;
        case EFTB_T_LCT: return D * c.Nl * 6;
        case EFTB_T_L22: return D * c.Nl * 28;
        case EFTB_T_L13: return D * c.Nl * 10;
        case EFTB_T_GRP: return sizeof(int32_t) * 38 * 2;
        case EFTB_T_BXT: return c.with_resum ? D * kpad(c.Nkin + c.nxtail) * 2 * NS : 0;
        case EFTB_T_TYT: return 0;  // (unused id)
        case EFTB_T_LNXXTAIL: return c.with_resum ? D * c.nxtail : 0;
        case EFTB_T_WQLAST2: return c.with_resum ? D * 2 : 0;
        case EFTB_T_QPOLY: return c.with_resum ? D * 2 * c.Nl * c.Nl * Nn * 15 : 0;
        case EFTB_T_H: return c.with_resum ? D * c.Na * NS * c.Nk : 0;
        case EFTB_T_RSBASIS: case EFTB_T_RSBASISS: return c.with_resum ? D * RS_NB * 16 : 0;
        case EFTB_T_RSROWS: return c.with_resum ? sizeof(int32_t) * 2 * RS3_ROWS : 0;  // Nl = 3: X part, Y part per row; Nl = 2: the first 80
        case EFTB_T_MU: case EFTB_T_WMU: return c.with_ap ? D * c.nmu : 0;
        case EFTB_T_LEGMU: return c.with_ap ? D * c.Nl * c.nmu : 0;
        case EFTB_T_SPBAND: return c.with_ap ? D * (size_t)(2 * SPL_HB + 1) * c.Nk : 0;
        case EFTB_T_SPCBAND: return c.with_ap ? D * (size_t)(2 * SPL_HB + 1) * c.Nk : 0;
        case EFTB_T_SPLOCAL: return c.with_ap ? D * (size_t)16 * c.Nk : 0;
        case EFTB_T_APFID: return c.with_ap ? D * 2 : 0;
        case EFTB_T_LCTN: return c.with_nnlo ? D * c.Nl * 6 : 0;
        case EFTB_T_BAO: return c.optiresum && c.with_resum ? D * (2 * NS + 4) : 0;
        case EFTB_T_GCT2: case EFTB_T_GCT2T: return c.dual_coef ? D * kpad(c.Nkin + c.ntail) * 2 * NCH : 0;
    }
    return 0;
}

static size_t need_buffer_elems(const eftb_config& c, int id) {
    const size_t B = c.max_batch;
    const int Nn = 2 * c.NIR * c.Na;
    switch (id) {
        case EFTB_B_PIN: return B * c.Nkin;
        case EFTB_B_F: case EFTB_B_DA: case EFTB_B_H: return B;
        case EFTB_B_P11: return B * c.Nk;
        case EFTB_B_P22: return B * 28 * c.Nk;
        case EFTB_B_P13: return B * 10 * c.Nk;
        case EFTB_B_C11: case EFTB_B_CCT: return c.with_resum ? B * c.Nl * NS : 0;
        case EFTB_B_CC: return c.with_resum ? B * c.Nl * 38 * NS : 0;
        case EFTB_B_CLOOPL: return c.with_resum ? B * c.Nl * 12 * NS : 0;
        case EFTB_B_TEMPL: return B * c.Nl * NROW * c.Nk;
        case EFTB_B_XY: return c.with_resum ? B * 2 * NS : 0;
        case EFTB_B_Q: return c.with_resum ? B * 2 * c.Nl * c.Nl * Nn : 0;
        case EFTB_B_BIAS: return B * NROW;
        case EFTB_B_PLK: return B * c.Nl * c.Nk;
        case EFTB_B_COEF: return B * 2 * NCH;
        case EFTB_B_GROWS: return B * MARG_NG1 * NROW;
        case EFTB_B_LOGP: return B * MARG_OUT;
        case EFTB_B_CCTN: return c.with_nnlo && c.with_resum ? B * c.Nl * NS : 0;
        case EFTB_B_TEMPLN: return c.with_nnlo ? B * c.Nl * NROW * c.Nk : 0;
        case EFTB_B_BIASN: return c.with_nnlo ? B * 3 : 0;
        case EFTB_B_GROWSN: return c.with_nnlo ? B * MARG_NG1 * 3 : 0;
    }
    return 0;
}

template <typename T>
static inline const T* tb(const eftb_engine* e, int id) { return static_cast<const T*>(e->tab[id]); }

// out[row][x] = sum_q A[row][q] Tab[q][x] on the FP64 matrix cores (synth_kernel); rows = (group, member).  The requested
// syntheses of a stage set are queued into one SynthBatch and launched together.
static void queue_synth(SynthBatch& sb, const double* A, long long a_group, int groups, int rpg, int K, const double* Tab, int X, double* out,
                        long long o_group, const double* gscale, const double* xscale) {
    SynthDesc& d = sb.p[sb.n];
    d.A = A; d.Tab = Tab; d.out = out; d.gscale = gscale; d.xscale = xscale;
    d.a_group = a_group; d.o_group = o_group; d.M = groups * rpg; d.rpg = rpg; d.K = K; d.X = X;
    d.wgx = (X + 63) / 64;
    d.wg_end = (sb.n ? sb.p[sb.n - 1].wg_end : 0) + d.wgx * ((d.M + 31) / 32);
    ++sb.n;
}

static void launch_synth(hipStream_t st, const SynthBatch& sb) {
    if (sb.n) hipLaunchKernelGGL(synth_kernel, dim3(sb.p[sb.n - 1].wg_end), dim3(256), 0, st, sb);
}

// the same batch on gemm_direct_kernel (one wave per 16 x 32 tile, no LDS): for problems with few rows -- the first-stage products
static int launch_gemm_direct(hipStream_t st, SynthBatch sb, int waves = 4) {
    int end = 0;
    for (int i = 0; i < sb.n; ++i) {
        SynthDesc& d = sb.p[i];
        if (d.K % 16) return fail("eftb_run: K = %d of a first-stage product is not a multiple of 16", d.K);
        d.wgx = (d.X + 31) / 32;
        end += d.wgx * ((d.M + 15) / 16);
        d.wg_end = end;
    }
    if (sb.n && waves == 2) hipLaunchKernelGGL((gemm_direct_kernel<2>), dim3(end), dim3(128), 0, st, sb);
    else if (sb.n) hipLaunchKernelGGL((gemm_direct_kernel<4>), dim3(end), dim3(256), 0, st, sb);
    return 0;
}

// anti-diagonal sums of every loop matrix for the batch (shared by the k-space and the xi-space pieces), then the
// synthesis rows selected by `sets` (build_rows_kernel)
static int launch_antidiag_rows(eftb_engine* e, hipStream_t st, int B, int sets, const double* coef, const double* coefT, bool contracted = false) {
    const eftb_config& c = e->c;
    const int nc = c.nbasis + (c.with_resum ? c.nbasis13 : 0);
    const dim3 grid(NPOW, (B + 63) / 64), rgrid(B, 2, 5);  // antidiag: one workgroup of four waves per (j', 64 cosmologies); build_rows: 5 x 64 lanes per (cosmology, half)
#define AD_ARGS B, c.max_batch, coefT, tb<double2>(e, EFTB_T_AD), e->SAD
#define ROW_ARGS sets, c.max_batch, c.Nl, c.with_nnlo ? 3 : 2, c.nbasis, coef, e->SAD, tb<double2>(e, EFTB_T_MLJ), tb<double2>(e, EFTB_T_LINVEC), e->A22, e->A13, \
                 e->ACF, e->ALC
    if (nc == 9) {
        if (contracted && WHATIF_SKIP(4)) {
        } else if ((sets & 0x10) && e->ad_waves == 2) hipLaunchKernelGGL((antidiag_kernel<9, 2>), grid, dim3(128), 0, st, AD_ARGS);
        else if (sets & 0x10) hipLaunchKernelGGL((antidiag_kernel<9, 4>), grid, dim3(256), 0, st, AD_ARGS);
        if (contracted) trace_point(e, 11, st);
        if (contracted && WHATIF_SKIP(8)) {
        } else if (contracted)  // direct-P_l runs: the rows contracted with the bias before the synthesis (3 per cosmology and space)
            hipLaunchKernelGGL((build_rows_plk_kernel<9>), rgrid, dim3(64), 0, st, c.max_batch, c.nbasis, coef, e->SAD, tb<double2>(e, EFTB_T_MLJ),
                               tb<double2>(e, EFTB_T_LINVEC), e->buf[EFTB_B_BIAS], e->buf[EFTB_B_F], tb<double>(e, EFTB_T_L11), tb<double>(e, EFTB_T_LCT),
                               tb<double>(e, EFTB_T_L22), tb<double>(e, EFTB_T_L13), tb<int>(e, EFTB_T_GRP), tb<double>(e, EFTB_T_EXP22),
                               tb<double>(e, EFTB_T_EXPC), e->A22, e->A13, e->ACF, e->ALC);
        else
        hipLaunchKernelGGL((build_rows_kernel<9>), rgrid, dim3(64), 0, st, ROW_ARGS);
    } else if (nc == 7) {
        if ((sets & 0x10) && e->ad_waves == 2) hipLaunchKernelGGL((antidiag_kernel<7, 2>), grid, dim3(128), 0, st, AD_ARGS);
        else if (sets & 0x10) hipLaunchKernelGGL((antidiag_kernel<7, 4>), grid, dim3(256), 0, st, AD_ARGS);
        hipLaunchKernelGGL((build_rows_kernel<7>), rgrid, dim3(64), 0, st, ROW_ARGS);
    } else {
        return fail("loop-matrix basis of dimension %d + %d is not instantiated (expected 7 + 2)", c.nbasis, c.with_resum ? c.nbasis13 : 0);
    }
#undef AD_ARGS
#undef ROW_ARGS
    return 0;
}

// out[w][a][r][x] = sum_{l,k} T[w][l][r][k] * opT[(l,k)][(a,x)] on the FP64 matrix cores; the block changes shape
// Entries t0, t0 + tstride, ... of the batch go through operator `id`; with commit the block pointers are swapped and the
// block takes the operator's output shape (the last of a set of per-tracer launches commits).
static int launch_operator(eftb_engine* e, int id, int B, int t0 = 0, int tstride = 1, bool commit = true) {
    if (id < 0 || id >= (int)e->ops.size()) return fail("operator id %d out of range", id);
    const eftb_engine::Op& o = e->ops[id];
    if (o.nl_in != e->cur_nl || o.nx_in != e->cur_nx)
        return fail("operator %d expects templates [%d][24][%d], the block is [%d][24][%d]", id, o.nl_in, o.nx_in, e->cur_nl, e->cur_nx);
    // rows [r0, r0 + nr) of every (cosmology, multipole) through matrix `m`
    auto launch = [&](const eftb_engine::Op& m, int r0, int nr) {
        GemmDesc g{};
        const long long bin = (long long)o.nl_in * NROW * o.nx_in, bout = (long long)o.nl_out * NROW * o.nx_out;
        g.A = e->buf[EFTB_B_TEMPL] + (size_t)t0 * bin + (size_t)r0 * o.nx_in; g.a_group = bin * tstride; g.a_row = o.nx_in;
        g.a_seg = (long long)NROW * o.nx_in;
        g.rows = ((B - t0 + tstride - 1) / tstride) * nr; g.rows_per_group = nr; g.nseg = o.nl_in; g.kseg = o.nx_in;
        g.B = m.dev; g.ldb = m.ld; g.ncols = o.nl_out * o.nx_out;
        g.C = e->Talt + (size_t)t0 * bout + (size_t)r0 * o.nx_out; g.c_group = bout * tstride; g.c_row = o.nx_out;
        g.c_colgroup = (long long)NROW * o.nx_out;
        g.cols_per_group = o.nx_out;
        if (g.ncols <= 16 * GN_MAXT)  // few output columns: K split over the waves, 16-row workgroups
            hipLaunchKernelGGL(gemm_narrow_kernel, dim3((g.rows + 15) / 16, 1), dim3(256), 0, (e->opstream ? e->opstream : e->stream), g, GemmZ{});
        else
            hipLaunchKernelGGL(gemm_rows_kernel, dim3((g.rows + 63) / 64, (g.ncols + 255) / 256), dim3(256), GEMM_LDS, (e->opstream ? e->opstream : e->stream), g);
    };
    if (o.st_op < 0) launch(o, 0, NROW);
    else {
        launch(o, 0, NROW - 3);                 // P11l, Pctl, Ploopl
        launch(e->ops[o.st_op], NROW - 3, 3);   // Pstl (window_st / fiberst semantics of the reference)
    }
    if (!commit) return 0;
    std::swap(e->buf[EFTB_B_TEMPL], e->Talt);
    e->cur_nl = o.nl_out;
    e->cur_nx = o.nx_out;
    return 0;
}

// PROJECT stage: one operator for every entry, or one per tracer on the strided subsets
static int launch_pipeline_operator(eftb_engine* e, int B) {
    if (e->tracer_ops.empty()) {
        if (e->pipeline_op < 0) return fail("eftb_run: stage PROJECT needs eftb_set_pipeline_operator");
        return launch_operator(e, e->pipeline_op, B);
    }
    if (B % e->ntr) return fail("eftb_run: batch %d is not a multiple of the %d tracers per likelihood point", B, e->ntr);
    const eftb_engine::Op& o = e->ops[e->tracer_ops[0]];
    bool one_launch = e->ntr <= GN_MAXZ && o.nl_out * o.nx_out <= 16 * GN_MAXT;
    for (int t = 0; t < e->ntr; ++t) one_launch = one_launch && e->ops[e->tracer_ops[t]].st_op < 0;
    if (!one_launch) {
        for (int t = 0; t < e->ntr; ++t)
            if (int rc = launch_operator(e, e->tracer_ops[t], B, t, e->ntr, t == e->ntr - 1)) return rc;
        return 0;
    }
    // narrow outputs, one matrix per tracer: the tracers are the z dimension of a single launch
    if (o.nl_in != e->cur_nl || o.nx_in != e->cur_nx)
        return fail("the tracers' operators expect templates [%d][24][%d], the block is [%d][24][%d]", o.nl_in, o.nx_in, e->cur_nl, e->cur_nx);
    GemmDesc g{};
    GemmZ z{};
    const long long bin = (long long)o.nl_in * NROW * o.nx_in, bout = (long long)o.nl_out * NROW * o.nx_out;
    g.A = e->buf[EFTB_B_TEMPL]; g.a_group = bin * e->ntr; g.a_row = o.nx_in; g.a_seg = (long long)NROW * o.nx_in;
    g.rows = (B / e->ntr) * NROW; g.rows_per_group = NROW; g.nseg = o.nl_in; g.kseg = o.nx_in;
    g.B = o.dev; g.ldb = o.ld; g.ncols = o.nl_out * o.nx_out;
    g.C = e->Talt; g.c_group = bout * e->ntr; g.c_row = o.nx_out; g.c_colgroup = (long long)NROW * o.nx_out; g.cols_per_group = o.nx_out;
    for (int t = 0; t < e->ntr; ++t) z.B[t] = e->ops[e->tracer_ops[t]].dev;
    z.a_off = bin;
    z.c_off = bout;
    hipLaunchKernelGGL(gemm_narrow_kernel, dim3((g.rows + 15) / 16, 1, e->ntr), dim3(256), 0, (e->opstream ? e->opstream : e->stream), g, z);
    std::swap(e->buf[EFTB_B_TEMPL], e->Talt);
    e->cur_nl = o.nl_out;
    e->cur_nx = o.nx_out;
    return 0;
}

// PROJECT stage of a direct-P_l run: the same operators on ONE row per cosmology, P_l [B][Nl][Nk] (PLK0) -> EFTB_B_PLK [B][nl_out][nx_out]
static int launch_pipeline_operator_plk(eftb_engine* e, int B, hipStream_t st) {
    const bool per_tracer = !e->tracer_ops.empty();
    if (per_tracer && (B % e->ntr)) return fail("eftb_run: batch %d is not a multiple of the %d tracers per likelihood point", B, e->ntr);
    const int nz = per_tracer ? e->ntr : 1;
    const eftb_engine::Op& o = e->ops[per_tracer ? e->tracer_ops[0] : e->pipeline_op];
    GemmDesc g{};
    GemmZ z{};
    const long long bin = (long long)o.nl_in * o.nx_in, bout = (long long)o.nl_out * o.nx_out;
    g.A = e->PLK0; g.a_group = bin * nz; g.a_row = 0; g.a_seg = o.nx_in;
    g.rows = B / nz; g.rows_per_group = 1; g.nseg = o.nl_in; g.kseg = o.nx_in;
    g.B = o.dev; g.ldb = o.ld; g.ncols = o.nl_out * o.nx_out;
    g.C = e->buf[EFTB_B_PLK]; g.c_group = bout * nz; g.c_row = 0; g.c_colgroup = o.nx_out; g.cols_per_group = o.nx_out;
    if (g.ncols <= 16 * GN_MAXT && nz <= GN_MAXZ) {  // few output columns: K split over the waves, 16-row workgroups; the tracers are the z dimension
        for (int t = 0; t < nz; ++t) z.B[t] = e->ops[per_tracer ? e->tracer_ops[t] : e->pipeline_op].dev;
        z.a_off = bin;
        z.c_off = bout;
        hipLaunchKernelGGL(gemm_narrow_kernel, dim3((g.rows + 15) / 16, 1, nz), dim3(256), 0, st, g, z);
    } else {
        for (int t = 0; t < nz; ++t) {
            GemmDesc gt = g;
            gt.A = g.A + (size_t)t * bin;
            gt.C = g.C + (size_t)t * bout;
            gt.B = e->ops[per_tracer ? e->tracer_ops[t] : e->pipeline_op].dev;
            if (gt.ncols <= 16 * GN_MAXT) hipLaunchKernelGGL(gemm_narrow_kernel, dim3((gt.rows + 15) / 16, 1), dim3(256), 0, st, gt, GemmZ{});
            else hipLaunchKernelGGL(gemm_rows_kernel, dim3((gt.rows + 63) / 64, (gt.ncols + 255) / 256), dim3(256), GEMM_LDS, st, gt);
        }
    }
    e->cur_nl = o.nl_out;
    e->cur_nx = o.nx_out;
    return 0;
}

static void launch_prep_rows(eftb_engine* e, hipStream_t st, int B, bool first, bool ir) {
    const eftb_config& c = e->c;
    hipLaunchKernelGGL(prep_rows_kernel, dim3(B), dim3(256), (size_t)c.Nkin * sizeof(double), st, c.Nkin, c.ntail, c.nxtail, (int)kpad(c.Nkin),
                       (int)kpad(c.Nkin + c.ntail), (int)kpad(c.Nkin + c.nxtail), c.max_batch, e->buf[EFTB_B_PIN], tb<double>(e, EFTB_T_LNKIN),
                       tb<double>(e, EFTB_T_LNXTAIL), tb<double>(e, EFTB_T_LNXXTAIL), tb<double>(e, EFTB_T_WQLAST2), first ? e->PA1 : nullptr, e->PA2, e->PA2T,
                       ir ? e->PA3 : nullptr, e->status + 2 * e->status_slot);
}

static void queue_xy(eftb_engine* e, SynthBatch& sb, int B) {  // X(s), Y(s) [B][2][80] = [Pin | tail'] (BX BY ; TX TY)
    const eftb_config& c = e->c;
    queue_synth(sb, e->PA3, 0, 1, B, (int)kpad(c.Nkin + c.nxtail), tb<double>(e, EFTB_T_BXT), 2 * NS, e->buf[EFTB_B_XY], 0, nullptr, nullptr);
}

// Resum.IRFilters + Resum.makeQ: X, Y (operand rows + one GEMM, unless the first stage of this run already produced them) and Q(f)
static void launch_irfilter(eftb_engine* e, hipStream_t st, int B, bool xy = true) {
    const eftb_config& c = e->c;
    if (xy) {
        launch_prep_rows(e, st, B, false, true);
        SynthBatch sb{};
        queue_xy(e, sb, B);
        (void)launch_gemm_direct(st, sb, e->gd_waves);
    }
    hipLaunchKernelGGL(qf_kernel, dim3(B), dim3(256), 0, st, c.Nl * c.Nl * e->Nn, e->buf[EFTB_B_F], tb<double>(e, EFTB_T_QPOLY), e->buf[EFTB_B_Q]);
}

static int timer_begin(eftb_engine* e, hipStream_t st, int kind);
static void timer_end(eftb_engine* e, hipStream_t st, int tslot);

// per-s A operand of the Nl = 3 resummation from Q(f), X(s), Y(s) (resum_as_kernel)
static void launch_resum_as(eftb_engine* e, hipStream_t st, int B) {
    const eftb_config& c = e->c;
    hipLaunchKernelGGL(resum_as_kernel, dim3(B, 4), dim3(256), 0, st, e->Nn, c.NIR, c.Na, e->buf[EFTB_B_Q], tb<double>(e, EFTB_T_RSBASISS),
                       tb<int>(e, EFTB_T_RSROWS), e->buf[EFTB_B_XY], e->RSAS);
}

static void launch_ap_prefix(eftb_engine* e, hipStream_t st, int B, bool weights = true) {
    const eftb_config& c = e->c;
    double** b = e->buf;
    const size_t pflds = ((size_t)(1 + 2 * c.Nl) * c.nmu + (size_t)c.Nl * c.Nl * 4 * 8) * sizeof(double);
#define PF_ARGS c.nmu, b[EFTB_B_DA], b[EFTB_B_H], tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), tb<double>(e, EFTB_T_WMU), \
                tb<double>(e, EFTB_T_LEGMU), e->APP, e->APR
    if (c.Nl == 3) hipLaunchKernelGGL((ap_prefix_kernel<3>), dim3(B), dim3(320), pflds, st, PF_ARGS);
    else hipLaunchKernelGGL((ap_prefix_kernel<2>), dim3(B), dim3(128), pflds, st, PF_ARGS);
#undef PF_ARGS
    if (!e->ap_fast || !weights) return;
    // knot weights of the fast path: inputs only as well, so they ride with the prefix sums (look-ahead stream in overlapped runs)
    const dim3 wgrid(((c.Nk + 63) / 64) * B);  // flat: (k tile, cosmology) decoded XCD-aware in the kernel
    const size_t wlds = ((size_t)c.Nk + c.nmu + (size_t)c.Nl * c.Nl * 4 * 64) * sizeof(double);  // knots, roots, the waves' coefficient windows
#define APW_ARGS c.Nk, c.nmu, tb<double>(e, EFTB_T_K), b[EFTB_B_DA], b[EFTB_B_H], tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), e->APP, e->APR, \
                 tb<double>(e, EFTB_T_SPLOCAL), e->APW, e->API, e->APM
    const int tslot = timer_begin(e, st, 2);
    if (c.Nl == 3) hipLaunchKernelGGL((ap_weights_kernel<3>), wgrid, dim3(192), wlds, st, APW_ARGS);
    else hipLaunchKernelGGL((ap_weights_kernel<2>), wgrid, dim3(128), wlds, st, APW_ARGS);
    timer_end(e, st, tslot);
#undef APW_ARGS
}

// nnlo_pass: the linear stages (RESUM / AP / PROJECT) once more, on the NNLO block (pointers swapped in by launch_stages)
// the main stream picks up behind a back half that is still in flight on its own stream (see engine.back)
static inline void join_back(eftb_engine* e) {
    if (!e->back_pending) return;
    (void)hipStreamWaitEvent(e->stream, e->evBack[(e->back_step + 1) & 1], 0);  // the last one recorded
    e->back_pending = false;
}

// every stream of the engine drained (setters that replace resident tables / likelihood data)
static hipError_t sync_all(eftb_engine* e) {
    join_back(e);
    for (hipStream_t q : {e->stream, e->side, e->pre, e->back, e->cpy, e->comm_stream})
        if (q) {
            const hipError_t rc = hipStreamSynchronize(q);
            if (rc != hipSuccess) return rc;
        }
    return hipSuccess;
}

// flags raised by the kernels of finished runs (call after the runs in question have finished); reported once, then cleared.
// slot >= 0: only that pair (a staged step's own flags: the steps queued behind it keep theirs); slot < 0: every pair, after a synchronisation
static int check_status(eftb_engine* e, const char* who, int slot = -1) {
    if (!e->status) return 0;
    volatile int* s = e->status;
    const int lo = slot < 0 ? 0 : slot, hi = slot < 0 ? eftb_engine::NSETS : slot;
    int bad_in = 0, bad_out = 0;
    for (int q = lo; q <= hi; ++q) {
        if (!bad_in && !bad_out) {
            bad_in = s[2 * q];
            bad_out = s[2 * q + 1];
        }
        s[2 * q] = s[2 * q + 1] = 0;
    }
    if (!bad_in && !bad_out) return 0;
    if (bad_in)
        return fail("%s: P_lin of cosmology %d is non-finite, or not positive at its last two samples (the FFTLog power-law extrapolation needs "
                    "them positive, reference fftlog.py:146-151); the outputs of that run are invalid", who, bad_in - 1);
    return fail("%s: non-finite P_l(k) for cosmology %d (EFTB_O_CHECK_FINITE)", who, bad_out - 1);
}

// bounded spin on an event (the sampler thread is about to enqueue the next step: the wake-up latency of a blocking wait would be paid once per step)
static int spin_event(hipEvent_t ev, const char* who, const char* what) {
    static const double limit_s = getenv("EFTB_FETCH_TIMEOUT_S") ? atof(getenv("EFTB_FETCH_TIMEOUT_S")) : 60.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail("%s: %s", who, hipGetErrorString(q));
        if ((spins & 0xfff) == 0xfff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
            return fail("%s: %s did not finish within %.0f s (EFTB_FETCH_TIMEOUT_S)", who, what, limit_s);
    }
}

// host inputs of one batch, before anything is copied: finite everywhere, P_lin positive where its logarithm is taken, distances positive
static int validate_inputs(const eftb_config& c, const char* who, int B, const double* Pin, const double* f, const double* DA, const double* H) {
    for (int w = 0; w < B; ++w) {
        const double* p = Pin + (size_t)w * c.Nkin;
        // exponent field all ones <=> Inf or NaN: an integer pass per row (the floating-point form, a sum of 0 x p[j], is one serial chain of adds per
        // row: 15 us per step of 128 on the sampler's thread against 7), the element-wise search only when it trips
        unsigned long long bad = 0;
        for (int j = 0; j < c.Nkin; ++j) {
            unsigned long long u;
            memcpy(&u, p + j, sizeof u);
            bad |= (unsigned long long)((u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
        }
        if (bad)
            for (int j = 0; j < c.Nkin; ++j)
                if (!std::isfinite(p[j])) return fail("%s: Pin[%d][%d] is not finite", who, w, j);
        if (!(p[c.Nkin - 1] > 0.0) || !(p[c.Nkin - 2] > 0.0))
            return fail("%s: Pin[%d] must be positive at its last two samples (power-law extrapolation of the FFTLog, reference fftlog.py:146-151)", who, w);
        if (!std::isfinite(f[w])) return fail("%s: f[%d] is not finite", who, w);
        if (c.with_ap && (!std::isfinite(DA[w]) || !std::isfinite(H[w]) || !(DA[w] > 0.0) || !(H[w] > 0.0)))
            return fail("%s: DA[%d], H[%d] must be finite and positive", who, w, w);
    }
    return 0;
}

// adds the finished event pairs of the dominant-kernel timer to the running sum (wait = true: after a synchronisation, all of them)
static void collect_timer(eftb_engine* e, int slot, bool wait) {
    for (int t = 0; t < eftb_engine::NTIMER; ++t) {
        if ((slot >= 0 && t != slot) || !e->timer_busy[t]) continue;
        if (wait) (void)hipEventSynchronize(e->evT1[t]);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->evT0[t], e->evT1[t]) == hipSuccess) {
            e->timer_ms[e->timer_kind[t]] += ms;
            ++e->timer_n[e->timer_kind[t]];
            e->timer_cosmo[e->timer_kind[t]] += e->timer_B[t];
            e->timer_busy[t] = false;
        }
    }
}

// EFTB_O_TIME_DOMINANT: the launch that follows on `st` is bracketed with an event pair if it is this sample's turn (-> slot, or -1)
static int timer_begin(eftb_engine* e, hipStream_t st, int kind) {
    if (!e->time_dominant || !(e->time_kernel & (1 << kind)) || !e->evT0[0] || e->use_graphs || (e->time_seqk[kind]++ % (unsigned)e->time_dominant) != 0)
        return -1;  // (not under graph replay: the event records would be captured)
    const int tslot = (int)(e->timer_next++ % eftb_engine::NTIMER);
    collect_timer(e, tslot, false);
    if (e->timer_busy[tslot]) return -1;  // its previous pair has not finished: skip this sample
    e->timer_kind[tslot] = kind;
    e->timer_B[tslot] = e->launch_B;
    return hipEventRecord(e->evT0[tslot], st) == hipSuccess ? tslot : -1;
}
static void timer_end(eftb_engine* e, hipStream_t st, int tslot) {
    if (tslot >= 0 && hipEventRecord(e->evT1[tslot], st) == hipSuccess) e->timer_busy[tslot] = true;
}

static int launch_stages_impl(eftb_engine* e, int mask, int B, bool nnlo_pass, bool nnlo_inline = false) {
    const eftb_config& c = e->c;
    e->launch_B = B;
    hipStream_t st = e->stream;
    const int Nk = c.Nk, Nl = c.Nl;
    double** b = e->buf;
    // every configuration error is raised before the first launch: no half-issued run is left behind on the look-ahead stream
    if ((mask & (EFTB_S_CF | EFTB_K_C22)) && !c.with_resum) return fail("eftb_run: stage CF needs with_resum=1");
    if ((mask & (EFTB_S_RESUM | EFTB_K_RESUM)) && !c.with_resum) return fail("eftb_run: stage RESUM needs with_resum=1");
    if ((mask & EFTB_S_AP) && !c.with_ap) return fail("eftb_run: stage AP needs with_ap=1");
    if ((mask & EFTB_S_PROJECT) && e->tracer_ops.empty() && e->pipeline_op < 0) return fail("eftb_run: stage PROJECT needs eftb_set_pipeline_operator");
    if ((mask & EFTB_S_LOGP) && !e->like_ndata) return fail("eftb_run: stage LOGP needs eftb_set_likelihood");
    if ((mask & (EFTB_S_PROJECT | EFTB_S_LOGP)) && (B % e->ntr)) return fail("eftb_run: batch %d is not a multiple of the %d tracers per likelihood point", B, e->ntr);
    // The IR filters / Q(f) and the AP prefix sums depend on the inputs only: when they are part of a longer stage set they
    // run on a side stream beside the (latency-bound) loop path and are joined right before their consumers.
    const bool side_ir = !nnlo_pass && (mask & EFTB_S_RESUM) && c.with_resum && (mask & (EFTB_S_PREP | EFTB_S_LOOPS | EFTB_S_CF | EFTB_S_REGROUP));
    const bool side_ap0 = !nnlo_pass && (mask & EFTB_S_AP) && c.with_ap && (mask & (EFTB_S_PREP | EFTB_S_LOOPS | EFTB_S_CF | EFTB_S_REGROUP | EFTB_S_RESUM));
    // cross-run overlap of the front half (see engine.pre): only for asynchronous runs whose inputs are already in place
    // with_NNLO: CctNNLO rides in the resummation records of the first pass when the batch is large enough for unsplit s sums
    const bool nnlo_fused = c.with_nnlo && c.with_resum && Nl == 3 && e->resum_splits == 1 && !c.optiresum && !e->generic_resum;
    // (with_NNLO steps stay on one stream: with only the front half overlapped they measured 0.66 ms per 128 against 0.64 ms in line)
    // nnlo_inline (launch_stages): this call carries the NNLO block through regrouping, resummation (fused accumulator), AP and REDUCE itself
    const bool pre_side = (mask & EFTB_S_PREP) && (mask & EFTB_S_REGROUP) && e->prep_overlap && e->inputs_settled && !e->use_graphs && !nnlo_pass &&
                          (!c.with_nnlo || nnlo_inline);
    const bool ap_side = pre_side && e->ap_overlap && e->allow_back && (mask & EFTB_S_RESUM) && (mask & EFTB_S_AP) && (Nl == 3 || !e->generic_resum);
    if (!ap_side) join_back(e);
    // three-stream runs keep the regrouping and the operand build of the resummation on the look-ahead stream as well: the main stream
    // then carries nothing but the resummation kernels, back to back
    const bool ahead = ap_side && e->prep_ahead && e->RSA2;
    const int bslot = e->back_step & 1;  // evBack[bslot] was recorded two runs ago
    const bool xy_in_prep = (mask & EFTB_S_PREP) && (mask & EFTB_S_RESUM) && c.with_resum && !nnlo_pass;  // X, Y ride with the first-stage GEMMs
    // whole-pipeline runs regroup C22 / C13 into the resummation records directly (resum_prep_kernel): no regroup_cf_kernel, no Cloopl
    // buffer on the way (EFTB_B_CLOOPL then keeps what the last stand-alone REGROUP stage left there)
    const bool fuse_cf = (mask & EFTB_S_REGROUP) && (mask & EFTB_S_RESUM) && c.with_resum && (Nl == 3 || !e->generic_resum) && !c.optiresum && !c.with_nnlo &&
                         !nnlo_pass && e->fuse_cf;
    // P_l = sum_row b_row T[l][row]: two FMA chains split at msplit_cfg (the row split of ap_rows_kernel's half waves), everywhere
    const int msplit_cfg = c.with_ap ? ((c.ap_stochastic ? NROW : 21) + 1) / 2 : NROW / 2;
    const bool fuse_reduce = (mask & EFTB_S_AP) && (mask & EFTB_S_REDUCE) && !(mask & (EFTB_S_PROJECT | EFTB_S_LOGP)) && c.with_ap && e->ap_mode == 0 &&
                             !c.with_nnlo && !nnlo_pass;
    // direct-P_l runs (EFTB_O_PLK_DIRECT): the bias contraction first, then resummation and AP on ONE row per multipole (regroup_plk_kernel,
    // resum_prep_plk_kernel, resum_plk_kernel, spline / ap_rows on row 0); the template block is not produced
    // (only runs that also hold PREP, LOOPS and CF: the front of a direct run leaves CONTRACTED rows in Y22 / P13 / YCF / C11 / CCT, which is what
    // back_prep_plk_kernel reads -- a run split in front of REGROUP finds un-contracted rows there and takes the template path)
    // Round 4: also with a PROJECT stage behind AP (window / binning / chained are linear maps of the k and multipole axes that act on every row
    // alike: reference window.py:371-415, binning.py:131-162, chained.py:56-68 -- the operator takes ONE row per cosmology instead of 24) unless an
    // operator keeps a second matrix for the stochastic rows; and on every k grid (the AP stage of a direct run is the moment form on B-spline
    // pieces, which needs neither the knot-weight tables nor their grid limit).  LOGP runs stay templates-first: they need 1 + n_G contracted rows.
    bool proj_plk = true;  // every operator of the PROJECT stage can act on the contracted row
    if (mask & EFTB_S_PROJECT) {
        auto one = [&](int id) { return id >= 0 && id < (int)e->ops.size() && e->ops[id].st_op < 0 && e->ops[id].nl_in == Nl && e->ops[id].nx_in == Nk; };
        if (e->tracer_ops.empty()) proj_plk = one(e->pipeline_op);
        else
            for (int t = 0; t < e->ntr; ++t) proj_plk = proj_plk && one(e->tracer_ops[t]) && e->ops[e->tracer_ops[t]].nl_out == e->ops[e->tracer_ops[0]].nl_out &&
                                                        e->ops[e->tracer_ops[t]].nx_out == e->ops[e->tracer_ops[0]].nx_out;
    }
    const bool direct_tail = (mask & EFTB_S_AP) && (mask & EFTB_S_REDUCE) && !(mask & EFTB_S_LOGP) && c.with_ap && !c.with_nnlo && !nnlo_pass && proj_plk && e->PLK0;
    const bool direct = e->plk_direct && direct_tail && fuse_cf && Nl == 3 && !e->generic_resum && !c.dual_coef && !e->use_graphs &&
                        (mask & EFTB_S_PREP) && (mask & EFTB_S_LOOPS) && (mask & EFTB_S_CF) && (mask & EFTB_S_REGROUP) && e->RSAS;
    const bool direct_proj = direct && (mask & EFTB_S_PROJECT);
    e->plk_host_written = true;
    // ... and the per-s A operand of the Nl = 3 resummation (inputs only: Q(f), X, Y) is built on the side stream, off the chain
    const bool as_side = ahead && Nl == 3 && !e->generic_resum && e->RSAS2 && !direct;
    // ... whose front runs a step ahead on the side stream (see FrontSet)
    const bool front_side = direct && ahead && e->alt.A22 && (mask & EFTB_S_PREP) && (mask & (EFTB_S_LOOPS | EFTB_S_CF)) && xy_in_prep;
    hipStream_t fst = st;  // where the front's kernels go
    const bool side_ap = side_ap0 && !direct;  // (direct-P_l runs take the AP stage as the node quadrature on one row: no prefix sums, no knot weights)
    if ((side_ir || side_ap) && !pre_side) {
        if (hipEventRecord(e->evFork, st) != hipSuccess || hipStreamWaitEvent(e->side, e->evFork, 0) != hipSuccess)
            return fail("eftb_run: stream fork failed");
        if (side_ir) launch_irfilter(e, e->side, B, !xy_in_prep);
        if (hipEventRecord(e->evJoin, e->side) != hipSuccess) return fail("eftb_run: stream join failed");
        if (side_ap) launch_ap_prefix(e, e->side, B);
        if (hipEventRecord(e->evJoinAP, e->side) != hipSuccess) return fail("eftb_run: stream join failed");
    }
    bool joined = !side_ir, joined_ap = !side_ap;
    if (mask & EFTB_S_PREP) {
        hipStream_t st0 = st;
        if (front_side) {
            // the front of this run goes to the side stream, into the set the run before the previous one used (its readers -- syntheses,
            // contraction, operand build -- have signalled evFrontFree); the rest of the chain (pre stream) picks up at the syntheses
            st = e->pre;
            eftb_engine::FrontSet& a = e->alt;
            std::swap(e->PA1, a.PA1); std::swap(e->PA2, a.PA2); std::swap(e->PA2T, a.PA2T); std::swap(e->PA3, a.PA3); std::swap(e->coefT, a.coefT);
            std::swap(e->A22, a.A22); std::swap(e->A13, a.A13); std::swap(e->ACF, a.ACF); std::swap(e->ALC, a.ALC); std::swap(e->SAD, a.SAD);
            std::swap(b[EFTB_B_P11], a.P11); std::swap(b[EFTB_B_COEF], a.COEF); std::swap(b[EFTB_B_XY], a.XY); std::swap(b[EFTB_B_Q], a.Q);
            ++e->epoch;  // (graphs captured earlier hold the other set's pointers)
            fst = e->side;
            if (hipStreamWaitEvent(fst, e->evFrontFree[e->front_step & 1], 0) != hipSuccess) return fail("eftb_run: stream wait failed");
            if (!e->prev_front_side && hipStreamWaitEvent(fst, e->evInFree, 0) != hipSuccess) return fail("eftb_run: stream wait failed");
            joined = joined_ap = true;  // (X, Y, Q(f) reach the rest of the chain with evFront; no AP tables in a direct run)
        } else if (pre_side) {
            st = e->pre;
            if (hipStreamWaitEvent(st, e->evInFree, 0) != hipSuccess) return fail("eftb_run: stream wait failed");
            // (X, Y of the previous run are also read by its operand build on the side stream, which evInFree does not cover)
            if (as_side && hipStreamWaitEvent(st, e->evAS, 0) != hipSuccess) return fail("eftb_run: stream wait failed");
            // the input-only kernels (IR filters / Q(f); AP prefix sums and knot weights) get their own low-priority stream beside the front
            // half: X, Y, Q are free since the previous run built its resummation operands (evInFree); the AP tables alternate between two
            // sets because the previous run's AP reads its own late (the set written here was last read two runs ago: evBack)
            if (side_ir || side_ap) {
                if (hipStreamWaitEvent(e->side, e->evInFree, 0) != hipSuccess) return fail("eftb_run: stream wait failed");
                if (side_ir) launch_irfilter(e, e->side, B, !xy_in_prep);
                if (hipEventRecord(e->evJoin, e->side) != hipSuccess) return fail("eftb_run: stream join failed");
                if (side_ap) {
                    std::swap(e->APP, e->APP2);
                    std::swap(e->APR, e->APR2);
                    std::swap(e->APW, e->APW2);
                    std::swap(e->API, e->API2);
                    std::swap(e->APM, e->APM2);
                    if (ap_side && hipStreamWaitEvent(e->side, e->evBack[bslot], 0) != hipSuccess) return fail("eftb_run: stream wait failed");  // its reader
                    launch_ap_prefix(e, e->side, B);
                }
                if (hipEventRecord(e->evJoinAP, e->side) != hipSuccess) return fail("eftb_run: stream join failed");
            }
        }
        // operand rows, then every first-stage product in one launch on the matrix cores: P11, the FFTLog coefficients (and their
        // cosmology-contiguous transpose for the anti-diagonal pass), the second coefficient set of IRcutoff "loop" / "resum" (reference
        // pybird.py:1151-1160), and X(s), Y(s) when a resummation follows in this run
        if (!front_side) fst = st;
        if (front_side && WHATIF_SKIP(1)) {
        } else if (front_side)  // operand rows and Q(f) in one launch
            hipLaunchKernelGGL(prep_rows_qf_kernel, dim3(2 * B), dim3(256), (size_t)c.Nkin * sizeof(double), fst, B, c.Nkin, c.ntail, c.nxtail, (int)kpad(c.Nkin),
                               (int)kpad(c.Nkin + c.ntail), (int)kpad(c.Nkin + c.nxtail), c.max_batch, e->buf[EFTB_B_PIN], tb<double>(e, EFTB_T_LNKIN),
                               tb<double>(e, EFTB_T_LNXTAIL), tb<double>(e, EFTB_T_LNXXTAIL), tb<double>(e, EFTB_T_WQLAST2), e->PA1, e->PA2, e->PA2T, e->PA3,
                               e->status + 2 * e->status_slot, c.Nl * c.Nl * e->Nn, e->buf[EFTB_B_F], tb<double>(e, EFTB_T_QPOLY), e->buf[EFTB_B_Q]);
        else
        launch_prep_rows(e, fst, B, true, xy_in_prep);
        {
            SynthBatch sb{};
            const int KP1 = (int)kpad(c.Nkin), KP2 = (int)kpad(c.Nkin + c.ntail);
            queue_synth(sb, e->PA1, 0, 1, B, KP1, tb<double>(e, EFTB_T_SKT), Nk, b[EFTB_B_P11], 0, nullptr, nullptr);
            queue_synth(sb, e->PA2, 0, 1, B, KP2, tb<double>(e, EFTB_T_GCT), 2 * NCH, b[EFTB_B_COEF], 0, nullptr, nullptr);
            queue_synth(sb, tb<double>(e, EFTB_T_ECT), KP2, 2 * NCH, 1, KP2, e->PA2T, c.max_batch, e->coefT, c.max_batch, nullptr, nullptr);
            if (c.dual_coef) {
                queue_synth(sb, e->PA2, 0, 1, B, KP2, tb<double>(e, EFTB_T_GCT2), 2 * NCH, e->coef2, 0, nullptr, nullptr);
                queue_synth(sb, tb<double>(e, EFTB_T_GCT2T), KP2, 2 * NCH, 1, KP2, e->PA2T, c.max_batch, e->coefT2, c.max_batch, nullptr, nullptr);
            }
            if (xy_in_prep) queue_xy(e, sb, B);
            if (!(front_side && WHATIF_SKIP(2)))
                if (int rc = launch_gemm_direct(fst, sb, e->gd_waves)) return rc;
            if (front_side) trace_point(e, 10, fst);
        }
        if (as_side) {  // behind the AP tables on the side stream; the set written here was last read by the resummation two runs ago
            if (xy_in_prep && (hipEventRecord(e->evXY, st) != hipSuccess || hipStreamWaitEvent(e->side, e->evXY, 0) != hipSuccess))
                return fail("eftb_run: stream join failed");
            std::swap(e->RSAS, e->RSAS2);
            if (hipStreamWaitEvent(e->side, e->evRsDone[e->rs_step & 1], 0) != hipSuccess) return fail("eftb_run: stream wait failed");
            launch_resum_as(e, e->side, B);
            if (hipEventRecord(e->evAS, e->side) != hipSuccess) return fail("eftb_run: event record failed");
        }
        if (!pre_side) st = st0;
    }
    // with pre_side the whole front half (first stage, anti-diagonal sums, rows, syntheses, expansions: inputs -> P22, P13, C11, Cct, CC)
    // stays on the `pre` stream; the main stream picks up at the regrouping
    hipStream_t st_main = e->stream;
    if ((mask & (EFTB_S_CF | EFTB_K_C22)) && !c.with_resum) return fail("eftb_run: stage CF needs with_resum=1");
    // the anti-diagonal sums serve both the k-space and the xi-space pieces: one pass, then the synthesis rows of
    // whatever is requested (bit 0/1: quadratic rows of k / xi space, bit 2/3: single-sum rows, bit 4: the sums themselves)
    if (mask & (EFTB_S_LOOPS | EFTB_S_CF | EFTB_K_P22 | EFTB_K_C22)) {
        int sets = 0x10;
        if (mask & (EFTB_S_LOOPS | EFTB_K_P22)) sets |= 0x1;
        if (mask & EFTB_S_LOOPS) sets |= 0x4;
        if (mask & (EFTB_S_CF | EFTB_K_C22)) sets |= 0x2;
        if (mask & EFTB_S_CF) sets |= 0x8;
        if (!c.dual_coef) {
            if (int rc = launch_antidiag_rows(e, front_side ? fst : st, B, sets, b[EFTB_B_COEF], e->coefT, direct)) return rc;
            if (front_side) trace_point(e, 1, fst);
            if (front_side && (hipEventRecord(e->evFront, fst) != hipSuccess || hipStreamWaitEvent(st, e->evFront, 0) != hipSuccess))
                return fail("eftb_run: stream join failed");
            if (front_side) trace_point(e, 2, st);
        } else {  // k-space rows from the first coefficient set, xi-space rows from the second (the sums are recomputed in between)
            if (sets & 0x5)
                if (int rc = launch_antidiag_rows(e, st, B, (sets & 0x5) | 0x10, b[EFTB_B_COEF], e->coefT)) return rc;
            if (sets & 0xa)
                if (int rc = launch_antidiag_rows(e, st, B, (sets & 0xa) | 0x10, e->coef2, e->coefT2)) return rc;
        }
    }
    {
        // every synthesis of the requested pieces in one launch, then both expansions in one launch
        SynthBatch sb{};
        const bool k22 = mask & (EFTB_S_LOOPS | EFTB_K_P22), c22 = mask & (EFTB_S_CF | EFTB_K_C22);
        // (only the rows in use: 7 of the 8 padded basis rows per cosmology, Nl (7 + 2) = 27 of the 32 weighted ones -- 10 % of the launch's
        // matrix-core work was padding; the padded rows of Y22 / YCF stay at their initial zeros and meet zero columns in expand_kernel)
        const int ncf = c.nbasis + (c.with_resum ? c.nbasis13 : 0);
        // (direct-P_l runs: 3 contracted rows per cosmology in each of these three -- same strides, fewer rows)
        if (k22) queue_synth(sb, e->A22, (long long)BAS22 * KSYN, B, direct ? 3 : c.nbasis, KSYN, tb<double>(e, EFTB_T_SYNK), Nk, e->Y22, (long long)BAS22 * Nk, nullptr, nullptr);
        if (c22) queue_synth(sb, e->ACF, (long long)BASC * KSYN, B, direct ? 3 : Nl * ncf, KSYN, tb<double>(e, EFTB_T_SYNS), NS, e->YCF, (long long)BASC * NS, nullptr, nullptr);
        if (mask & EFTB_S_LOOPS)
            queue_synth(sb, e->A13, 10LL * KLIN, B, direct ? 3 : 10, KLIN, tb<double>(e, EFTB_T_LINK), Nk, b[EFTB_B_P13], 10LL * Nk, b[EFTB_B_P11], nullptr);
        if (mask & EFTB_S_CF) {
            const long long ag = (c.with_nnlo ? 3LL : 2LL) * Nl * KLIN;
            queue_synth(sb, e->ALC, ag, B, Nl, KLIN, tb<double>(e, EFTB_T_LINS), NS, b[EFTB_B_C11], (long long)Nl * NS, nullptr, nullptr);
            queue_synth(sb, e->ALC + (size_t)Nl * KLIN, ag, B, Nl, KLIN, tb<double>(e, EFTB_T_LINS), NS, b[EFTB_B_CCT], (long long)Nl * NS,
                        nullptr, e->sm2);
            if (c.with_nnlo)  // makeCctNNLO (reference pybird.py:1098-1101)
                queue_synth(sb, e->ALC + (size_t)2 * Nl * KLIN, ag, B, Nl, KLIN, tb<double>(e, EFTB_T_LINS), NS, b[EFTB_B_CCTN], (long long)Nl * NS,
                            nullptr, e->sm4);
        }
        {
            const int tslot = (mask & EFTB_S_REGROUP) ? timer_begin(e, st, 1) : -1;
            if (direct && WHATIF_SKIP(16)) {
            } else launch_synth(st, sb);
            timer_end(e, st, tslot);
        }
        if ((k22 || c22) && !direct) {  // (direct-P_l runs contract the basis rows themselves: regroup_plk_kernel, resum_prep_plk_kernel)
            const int n22 = k22 ? ((Nk + 63) / 64) * ((28 + EXP_RPB - 1) / EXP_RPB) * B : 0;
            const int ncf = c22 ? ((NS + 63) / 64) * ((Nl * 38 + EXP_RPB - 1) / EXP_RPB) * B : 0;
            hipLaunchKernelGGL(expand_kernel, dim3(n22 + ncf), dim3(64), 0, st, n22, Nk, B, e->Y22, tb<double>(e, EFTB_T_EXP22), b[EFTB_B_P22], Nl * 38,
                               e->YCF, c22 ? tb<double>(e, EFTB_T_EXPC) : nullptr, b[EFTB_B_CC]);
        }
    }
    if (pre_side && !ahead) {
        if (hipEventRecord(e->evPrep, st) != hipSuccess || hipStreamWaitEvent(st_main, e->evPrep, 0) != hipSuccess) return fail("eftb_run: stream join failed");
        st = st_main;
    }
    if (ap_side) {
        // regroup into the block the run before the previous one left its AP output in (its readers are done: evBack)
        if (hipStreamWaitEvent(st, e->evBack[bslot], 0) != hipSuccess) return fail("eftb_run: stream wait failed");
        std::swap(b[EFTB_B_TEMPL], e->T3);
        if (nnlo_inline) std::swap(b[EFTB_B_TEMPLN], e->T3N);
    }
    if (mask & EFTB_S_REGROUP) {
        if (direct) {  // (regroup_plk rides in the launch of the operand build below: back_prep_plk_kernel)
        } else
        hipLaunchKernelGGL(regroup_kernel, dim3((Nk + 255) / 256, B, Nl), dim3(256), 0, st, Nk, Nl, tb<double>(e, EFTB_T_K), b[EFTB_B_F],
                           b[EFTB_B_P11], b[EFTB_B_P22], b[EFTB_B_P13], tb<double>(e, EFTB_T_L11), tb<double>(e, EFTB_T_LCT),
                           tb<double>(e, EFTB_T_L22), tb<double>(e, EFTB_T_L13), tb<int>(e, EFTB_T_GRP), b[EFTB_B_TEMPL]);
        if (c.with_nnlo)
            hipLaunchKernelGGL(nnlo_rows_kernel, dim3((Nk + 255) / 256, B, Nl), dim3(256), 0, st, Nk, Nl, tb<double>(e, EFTB_T_K), b[EFTB_B_P11],
                               tb<double>(e, EFTB_T_LCTN), b[EFTB_B_TEMPLN]);
        if (c.with_resum && !fuse_cf)
            hipLaunchKernelGGL(regroup_cf_kernel, dim3(12, Nl, B), dim3(128), 0, st, Nl, b[EFTB_B_F], b[EFTB_B_CC], tb<double>(e, EFTB_T_L22),
                               tb<double>(e, EFTB_T_L13), tb<int>(e, EFTB_T_GRP), b[EFTB_B_CLOOPL]);
        // nothing later in this run reads the front half's outputs (P11, coefficients, P22, P13, CC; C11 / Cct only if a resummation
        // follows): the next run's front half may overwrite them from here on
        if (!(mask & EFTB_S_RESUM) && !e->use_graphs && !nnlo_pass && hipEventRecord(e->evInFree, st) != hipSuccess)
            return fail("eftb_run: event record failed");
    }
    if ((mask & EFTB_K_IRFILTER) && !(mask & EFTB_S_RESUM)) {
        if (!c.with_resum) return fail("eftb_run: EFTB_K_IRFILTER needs with_resum=1");
        launch_irfilter(e, st, B);
    }
    if (mask & (EFTB_S_RESUM | EFTB_K_RESUM)) {
        if (!c.with_resum) return fail("eftb_run: stage RESUM needs with_resum=1");
        const bool full = mask & EFTB_S_RESUM;  // EFTB_K_RESUM alone: only the main kernel, on the operands of an earlier full run
        if (full && !side_ir && !nnlo_pass) launch_irfilter(e, st, B);  // X, Y, Q(f) of the first pass stay valid
        if (!joined) {
            if (hipStreamWaitEvent(st, e->evJoin, 0) != hipSuccess) return fail("eftb_run: stream join failed");
            joined = true;
        }
        const double *c11 = b[EFTB_B_C11], *cct = b[EFTB_B_CCT], *cloopl = b[EFTB_B_CLOOPL];
        if (c.optiresum && full) {  // Resum.extractBAO (reference pybird.py:1382-1400) into scratch: the stage inputs stay as the CF stage left them
            double *x11 = e->XB, *xct = x11 + (size_t)c.max_batch * Nl * NS, *xl = xct + (size_t)c.max_batch * Nl * NS;
            hipLaunchKernelGGL(extract_bao_kernel, dim3(B * Nl), dim3(128), 0, st, tb<double>(e, EFTB_T_BAO), c11, x11);
            hipLaunchKernelGGL(extract_bao_kernel, dim3(B * Nl), dim3(128), 0, st, tb<double>(e, EFTB_T_BAO), cct, xct);
            hipLaunchKernelGGL(extract_bao_kernel, dim3(B * Nl * 12), dim3(128), 0, st, tb<double>(e, EFTB_T_BAO), cloopl, xl);
            c11 = x11; cct = xct; cloopl = xl;
        }
        if (Nl == 3 || !e->generic_resum) {
            // matrix-core form: polynomials as [80 | 32 x 8] x [8 x 16 points] MFMAs, one wave = 16 k x one slice of the s sum
            // with_nnlo, large batches: CctNNLO rides in the records and the main kernel accumulates PctNNLOl beside Pctl (no second pass)
            const bool fused_nnlo = nnlo_fused && !nnlo_pass;
#define RP_ARGS e->Nn, c.NIR, c.Na, b[EFTB_B_Q], tb<double>(e, EFTB_T_RSBASISS), tb<int>(e, EFTB_T_RSROWS), b[EFTB_B_XY], c11, cct, cloopl, e->RSA, e->RSC, \
                fused_nnlo ? b[EFTB_B_CCTN] : nullptr, fuse_cf ? b[EFTB_B_CC] : nullptr, b[EFTB_B_F], tb<double>(e, EFTB_T_L22), tb<double>(e, EFTB_T_L13), \
                tb<int>(e, EFTB_T_GRP)
            const dim3 rpgrid(B, fuse_cf ? 5 : 1);  // the fused regrouping is 38 conditional terms per record entry: spread over five workgroups
            const int rslot = e->rs_step & 1;
            if (ahead) {  // the operand set written here was last read by the resummation two runs ago
                std::swap(e->RSA, e->RSA2);
                std::swap(e->RSC, e->RSC2);
                if (direct) std::swap(e->RSAS, e->RSAS2);  // (here the coefficient table of resum_plk_kernel)
                if (hipStreamWaitEvent(st, e->evRsDone[rslot], 0) != hipSuccess) return fail("eftb_run: stream wait failed");
            }
            if (full && direct) {
                constexpr int nparts = 5;  // slices of the s range per cosmology (2 ... 10 measured the same since the coefficient part keeps Q(f) in registers: 20-22 us alone at 512 per launch, 8 at 128)
                const int nsl = (NS + nparts - 1) / nparts;
                const size_t plds = ((size_t)2 * 3 * nsl + 2 * nsl) * sizeof(double);
                const int nkx = (Nk + 255) / 256, nreg = nkx * B;   // (regroup: one workgroup per (256 k, cosmology), all three l)
                if (!WHATIF_SKIP(32))
                hipLaunchKernelGGL(back_prep_plk_kernel, dim3(nreg + nparts * B), dim3(BPP_THREADS), plds, st, nreg, nkx, B, nparts, Nk, Nl, tb<double>(e, EFTB_T_K), b[EFTB_B_P11], e->Y22,
                                   b[EFTB_B_P13], tb<double>(e, EFTB_T_L11), tb<double>(e, EFTB_T_LCT), b[EFTB_B_BIAS], b[EFTB_B_TEMPL], c.ap_stochastic ? 1 : 0,
                                   e->Nn, c.NIR, c.Na, b[EFTB_B_Q], b[EFTB_B_XY], c11, cct, e->YCF, e->RSAS);
                trace_point(e, 3, st);
                if (front_side) {  // the last reader of this run's front set
                    if (hipEventRecord(e->evFrontFree[e->front_step & 1], st) != hipSuccess) return fail("eftb_run: event record failed");
                    ++e->front_step;
                }
            }
            else if (full && Nl == 3 && !as_side) launch_resum_as(e, st, B);  // (in line: X, Y, Q(f) are in place behind evJoin)
            if (full && direct) {
            } else
            if (full && Nl == 3) hipLaunchKernelGGL((resum_prep_kernel<3>), rpgrid, dim3(256), 0, st, RP_ARGS);
            else if (full) hipLaunchKernelGGL((resum_prep_kernel<2>), rpgrid, dim3(256), 0, st, RP_ARGS);
#undef RP_ARGS
            // C11 / Cct / Cloopl now live in the per-s records: the next run's front half may overwrite its outputs (see the regrouping)
            if (full && (mask & EFTB_S_REGROUP) && !e->use_graphs && !nnlo_pass && hipEventRecord(e->evInFree, st) != hipSuccess)
                return fail("eftb_run: event record failed");
            if (ahead) {
                if (hipEventRecord(e->evPrep, st) != hipSuccess || hipStreamWaitEvent(st_main, e->evPrep, 0) != hipSuccess) return fail("eftb_run: stream join failed");
                if (as_side && full && hipStreamWaitEvent(st_main, e->evAS, 0) != hipSuccess) return fail("eftb_run: stream join failed");
                st = st_main;
            }
            if (direct) trace_point(e, 4, st);
            const int tslot = full ? timer_begin(e, st, 0) : -1;
            const int kblocks = (Nk - c.Nklow + 63) / 64;
            int nsplit = 1;
            while (!direct && nsplit < e->resum_splits && (size_t)kblocks * 4 * B * nsplit < 2048) nsplit *= 2;
            const int schunk = (NS + nsplit - 1) / nsplit;
#define RM_ARGS Nk, c.Nklow, schunk, tb<double>(e, EFTB_T_K), tb<double>(e, EFTB_T_H), tb<double>(e, EFTB_T_RSBASIS), Nl == 3 ? e->RSAS : e->RSA, e->RSC, tb<double>(e, EFTB_T_L11), \
                tb<double>(e, nnlo_pass ? EFTB_T_LCTN : EFTB_T_LCT), b[EFTB_B_TEMPL], e->part, nsplit
            const int nkb = (Nk - (c.Nklow & ~15) + 63) / 64;  // Nl = 3: k tiles aligned to 16, (k block, cosmology) decoded from a flat index
            if (direct) {
                // four k per lane, four slices of the s range (round 4, same-box sweep of six shapes: 34.9 us alone at B = 128 / 105 at 384; 4 x 2, the
                // round-3 shape: 39.0 / 120; 2 x 4: 38.6 / 117; 8 x 1: 57 / 117; 8 x 2: 58 / 153; 8 x 4: 49 / 134)
                const int nkd = (Nk + 64 * 4 - 1) / (64 * 4);
                if (kblocks > 0 && !WHATIF_SKIP(64))
                    hipLaunchKernelGGL((resum_plk_kernel<4, 4>), dim3(nkd * 3 * B), dim3(192 * 4), 0, st, Nk, c.Nklow, tb<double>(e, EFTB_T_K), tb<double>(e, EFTB_T_H),
                                       e->RSAS, b[EFTB_B_TEMPL], nkd);
            } else if (kblocks > 0 && Nl == 3 && fused_nnlo)
                hipLaunchKernelGGL((resum_mfma_kernel<true>), dim3(nkb * B, 1, nsplit), dim3(256), 0, st, RM_ARGS, tb<double>(e, EFTB_T_LCTN), b[EFTB_B_TEMPLN], nkb);
            else if (kblocks > 0 && Nl == 3)
                hipLaunchKernelGGL((resum_mfma_kernel<false>), dim3(nkb * B, 1, nsplit), dim3(256), 0, st, RM_ARGS, nullptr, nullptr, nkb);
            else if (kblocks > 0) hipLaunchKernelGGL(resum_mfma2_kernel, dim3(kblocks, B, nsplit), dim3(256), 0, st, RM_ARGS);
#undef RM_ARGS
            if (nsplit > 1)
                hipLaunchKernelGGL(resum_sum_kernel, dim3((Nk + 255) / 256, 21 * Nl, B), dim3(256), 0, st, Nk, Nl, nsplit, e->part, b[EFTB_B_TEMPL]);
            timer_end(e, st, tslot);
            if (direct) trace_point(e, 5, st);
            if (ahead) {
                if (hipEventRecord(e->evRsDone[rslot], st) != hipSuccess) return fail("eftb_run: event record failed");
                ++e->rs_step;
            }
        } else {
            const int kblocks = (Nk + 255) / 256;
            // waves = kblocks*4 x (2*Nl) x (B*Nl*nsplit): split the s sum further only for small batches
            int nsplit = 1;
            while (nsplit < e->resum_splits && (size_t)kblocks * 4 * 4 * Nl * Nl * B * nsplit < 8192) nsplit *= 2;
            const int schunk = (NS + nsplit - 1) / nsplit;
            hipLaunchKernelGGL((resum_kernel<2>), dim3(kblocks, 4 * Nl, B * Nl * nsplit), dim3(256), 0, st, Nk, c.Nklow, schunk, tb<double>(e, EFTB_T_K),
                               b[EFTB_B_XY], b[EFTB_B_Q], tb<double>(e, EFTB_T_H), c11, cct, cloopl,
                               tb<double>(e, EFTB_T_L11), tb<double>(e, nnlo_pass ? EFTB_T_LCTN : EFTB_T_LCT), e->part, nsplit);
            hipLaunchKernelGGL(resum_sum_kernel, dim3((Nk + 255) / 256, 21 * Nl, B), dim3(256), 0, st, Nk, Nl, 2 * Nl * nsplit, e->part, b[EFTB_B_TEMPL]);
            if (full && (mask & EFTB_S_REGROUP) && !e->use_graphs && !nnlo_pass && hipEventRecord(e->evInFree, st) != hipSuccess)
                return fail("eftb_run: event record failed");
        }
    }
    if (ap_side) {
        // (with the operands built ahead the resummation's own event -- recorded right behind it on this stream -- is the fork point: one
        // event record less on the host's path)
        if (ahead && (mask & EFTB_S_RESUM)) {
            if (hipStreamWaitEvent(e->back, e->evRsDone[(e->rs_step + 1) & 1], 0) != hipSuccess) return fail("eftb_run: stream fork failed");
        } else if (hipEventRecord(e->evResum, st) != hipSuccess || hipStreamWaitEvent(e->back, e->evResum, 0) != hipSuccess) return fail("eftb_run: stream fork failed");
        st = e->back;
    }
    if (mask & EFTB_S_AP) {
        if (!c.with_ap) return fail("eftb_run: stage AP needs with_ap=1");
        // one AP pass over a template block: *pin -> *palt, then the two trade places.  nn: the NNLO block carries only the three
        // counter-term rows 3-5 of every multipole: spline and AP touch those alone
        auto ap_pass = [&](bool nn, double** pin, double** palt) -> int {
            const bool moments = e->ap_mode == 1;  // (works on whole blocks: the NNLO block's zero rows stay zero)
            const bool dir = direct && !nn;  // direct-P_l runs: row 0 of every (cosmology, l) block is the only one that goes through the stage
            const int nr = dir ? 1 : nn ? (moments ? 21 : 6) : (c.ap_stochastic ? NROW : 21);
            // rows the spline data is needed for: [rlo, rlo + rsel) of every (cosmology, l) -- the counter-term rows of the NNLO block; on the fast
            // path only the rows the stage distorts (Pstl passes through unless APst: 21 of 24 rows)
            const int rlo = nn && !moments ? 3 : 0, rsel = dir ? 1 : nn && !moments ? 3 : (e->ap_mode == 0 ? nr : NROW);
            const int nseries = B * Nl * rsel;
            if (dir) trace_point(e, 6, st);
            {
                const int kt = (Nk + 63) / 64;
                int ysplit = std::max(1, std::min((nseries + 15) / 16, 1024 / kt));  // ~4 workgroups per CU, each sweeping its share of the series
                if (ysplit >= 8) ysplit &= ~7;  // shares in multiples of 8: the k tiles of a share then sit on one XCD (xcd_decode)
                // (the fast path keeps the splines as B-spline coefficients -- one number per knot; the moment / quadrature forms as knot slopes)
                if (!(dir && WHATIF_SKIP(128)))
                hipLaunchKernelGGL(spline_kernel, dim3(kt * ysplit), dim3(256), 0, st, Nk, nseries, rlo, rsel, *pin, tb<double>(e, e->ap_mode == 0 || dir ? EFTB_T_SPCBAND : EFTB_T_SPBAND), e->SD);
            }
            // (a PROJECT stage follows: the stage leaves P_l in PLK0 and the operator writes the final block)
            double* const plk_dst = direct_proj ? e->PLK0 : b[EFTB_B_PLK];
            double* const plk_hst = direct_proj ? nullptr : e->plk_host_out;
            if (dir && e->ap_plk_nodes) {  // the node quadrature on the contracted row (ap_plk_kernel); nothing else of the stage runs
                const size_t lds = ((size_t)Nk + (size_t)c.nmu * 8 + 3 * APD_WMAX * 4 + 3 * 3 * 64) * sizeof(double);
                const int tslot = timer_begin(e, st, 2);
                hipLaunchKernelGGL((ap_plk_kernel<3>), dim3(((Nk + 63) / 64) * B), dim3(256), lds, st, Nk, c.nmu, tb<double>(e, EFTB_T_K), b[EFTB_B_DA], b[EFTB_B_H],
                                   tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), tb<double>(e, EFTB_T_WMU), tb<double>(e, EFTB_T_LEGMU), e->SD,
                                   tb<double>(e, EFTB_T_SPLOCAL), *pin, b[EFTB_B_BIAS], plk_dst, plk_hst,
                                   e->check_finite ? e->status + 2 * e->status_slot + 1 : nullptr, c.ap_stochastic ? NROW : 21);
                timer_end(e, st, tslot);
                std::swap(*pin, *palt);
                return 0;
            }
            if (dir && e->ap_plk_fused) {
                // moment form with everything in LDS (ap_plk_fused_kernel): prefix sums, pieces and the walk in one launch, two workgroups per cosmology
                const size_t lds = ((size_t)((Nk + 1) & ~1) + (size_t)c.nmu * (2 + 2 * 3) + 36 * 16 + (size_t)(c.nmu + 1) * 36 + (size_t)(Nk - 1) * 3 * 4) * sizeof(double);
                const int tslot = timer_begin(e, st, 2);
#define APF_ARGS Nk, c.nmu, tb<double>(e, EFTB_T_K), b[EFTB_B_DA], b[EFTB_B_H], tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), tb<double>(e, EFTB_T_WMU),   \
                 tb<double>(e, EFTB_T_LEGMU), e->SD, tb<double>(e, EFTB_T_SPLOCAL), *pin, b[EFTB_B_BIAS], plk_dst, plk_hst,                                      \
                 e->check_finite ? e->status + 2 * e->status_slot + 1 : nullptr, c.ap_stochastic ? NROW : 21
                // one workgroup of eight waves per cosmology (22.3 us alone at B = 128, 40 at 384; two workgroups of four waves measured 22.7 / 61 -- the
                // LDS tables allow one workgroup per CU either way)
                if (!WHATIF_SKIP(256)) hipLaunchKernelGGL((ap_plk_fused_kernel<3, 8>), dim3(B), dim3(512), lds, st, APF_ARGS);
#undef APF_ARGS
                trace_point(e, 7, st);
                timer_end(e, st, tslot);
                std::swap(*pin, *palt);
                return 0;
            }
            if (dir) {
                // moment form on the contracted row (ap_plk_mom_kernel): the mu prefix sums of this cosmology batch (a function of DA, H alone) in
                // line in front of it -- the stage is ~20 us of latency-bound launches either way, and in line it needs no second set of sums
                launch_ap_prefix(e, st, B, false);
                const size_t lds = ((size_t)Nk + c.nmu + 3 * 3 * 64) * sizeof(double);
                const int tslot = timer_begin(e, st, 2);
                hipLaunchKernelGGL((ap_plk_mom_kernel<3>), dim3(((Nk + 63) / 64) * B), dim3(256), lds, st, Nk, c.nmu, tb<double>(e, EFTB_T_K), b[EFTB_B_DA], b[EFTB_B_H],
                                   tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), e->APP, e->APR, e->SD, tb<double>(e, EFTB_T_SPLOCAL), *pin, b[EFTB_B_BIAS],
                                   plk_dst, plk_hst, e->check_finite ? e->status + 2 * e->status_slot + 1 : nullptr, c.ap_stochastic ? NROW : 21);
                timer_end(e, st, tslot);
                std::swap(*pin, *palt);
                return 0;
            }
            // prefix sums over mu per cosmology (side stream when possible), then interval moments by differences x cubic coefficients
            if (!side_ap && !nn) launch_ap_prefix(e, st, B);
            if (!joined_ap) {
                if (hipStreamWaitEvent(st, e->evJoinAP, 0) != hipSuccess) return fail("eftb_run: stream join failed");
                joined_ap = true;
            }
            if (moments) {
                const dim3 apgrid((Nk + 63) / 64, B, 3);
                const size_t aplds = ((size_t)Nk + c.nmu + (size_t)4 * Nl * ((nr + 2) / 3) * 64) * sizeof(double);
#define APM_ARGS Nk, c.nmu, tb<double>(e, EFTB_T_K), b[EFTB_B_DA], b[EFTB_B_H], tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), e->APP, e->APR, *pin, e->SD, *palt
                if (Nl == 3 && nr == 21) hipLaunchKernelGGL((ap_moments_kernel<3, 21, 3>), apgrid, dim3(256), aplds, st, APM_ARGS);
                else if (Nl == 3) hipLaunchKernelGGL((ap_moments_kernel<3, NROW, 3>), apgrid, dim3(256), aplds, st, APM_ARGS);
                else if (nr == 21) hipLaunchKernelGGL((ap_moments_kernel<2, 21, 3>), apgrid, dim3(256), aplds, st, APM_ARGS);
                else hipLaunchKernelGGL((ap_moments_kernel<2, NROW, 3>), apgrid, dim3(256), aplds, st, APM_ARGS);
#undef APM_ARGS
                std::swap(*pin, *palt);
                return 0;
            }
            // REDUCE directly behind the AP stage: the bias contraction rides in the epilogue of ap_rows_kernel (and of the fallback tiles'
            // ap_direct_kernel), in the summation order of reduce_kernel(msplit) -- no separate pass over the 33 MB of AP output
            const bool red = fuse_reduce && !nn;
            const int msplit = red ? msplit_cfg : (rlo + nr + 1) / 2;  // rows [rlo, msplit) / [msplit, nr) to the two half waves
            const double* rb = red ? b[EFTB_B_BIAS] : nullptr;
            double* rp = red ? b[EFTB_B_PLK] : nullptr;
            double* rph = red ? e->plk_host_out : nullptr;
            int* rflag = red && e->check_finite ? e->status + 2 * e->status_slot + 1 : nullptr;
            if (e->ap_fast) {
                // banded product of the knot weights with the spline data; rows outside [rlo, nr) are copied through
                const int kt2 = 2 * ((Nk + 63) / 64), nh = (nr - rlo + 1) / 2, nre = 2 * nh;
                const size_t lds = 0;  // (the window is a static array: 37 KB at most)
                if (rlo + nre > NROW || msplit - rlo > nh || nr - msplit > nh || (nh != 2 && nh != 11 && nh != 12))
                    return fail("eftb_run: AP rows [%d, %d) split at %d do not fit the window layouts built into ap_rows_kernel", rlo, nr, msplit);
#define APR_ARGS Nk, rlo, nr, msplit, b[EFTB_B_DA], b[EFTB_B_H], tb<double>(e, EFTB_T_APFID), e->APW, e->API, e->APM, *pin, e->SD, *palt, rb, rp, rph, rflag
#define APR_LAUNCH(NLV, NHV) do { if (e->ap_ring == 2) hipLaunchKernelGGL((ap_rows_kernel<NLV, NHV, 2>), dim3(kt2 * B), dim3(64 * NLV), lds, st, APR_ARGS); \
                                  else hipLaunchKernelGGL((ap_rows_kernel<NLV, NHV, 4>), dim3(kt2 * B), dim3(64 * NLV), lds, st, APR_ARGS); } while (0)
                if (Nl == 3 && nh == 11) APR_LAUNCH(3, 11);
                else if (Nl == 3 && nh == 12) APR_LAUNCH(3, 12);
                else if (Nl == 3) APR_LAUNCH(3, 2);
                else if (nh == 11) APR_LAUNCH(2, 11);
                else if (nh == 12) APR_LAUNCH(2, 12);
                else APR_LAUNCH(2, 2);
#undef APR_LAUNCH
#undef APR_ARGS
            }
            // the reference's own quadrature: every tile (EFTB_AP_FAST=0), or only the tiles the fast path flagged (strong distortions)
            {
                const int4* gate = e->ap_fast ? e->APM : nullptr;
                const dim3 dgrid(((Nk + 63) / 64) * B);
#define APD_ARGS Nk, c.nmu, rlo, nr, tb<double>(e, EFTB_T_K), b[EFTB_B_DA], b[EFTB_B_H], tb<double>(e, EFTB_T_APFID), tb<double>(e, EFTB_T_MU), tb<double>(e, EFTB_T_WMU), \
                 tb<double>(e, EFTB_T_LEGMU), e->APR, *pin, e->SD, *palt, gate, rb, rp, rph, msplit, rflag, tb<double>(e, EFTB_T_SPLOCAL)
                if (e->ap_fast && Nl == 3) hipLaunchKernelGGL((ap_direct_kernel<3, true>), dgrid, dim3(64), 0, st, APD_ARGS);
                else if (e->ap_fast) hipLaunchKernelGGL((ap_direct_kernel<2, true>), dgrid, dim3(64), 0, st, APD_ARGS);
                else if (Nl == 3) hipLaunchKernelGGL((ap_direct_kernel<3, false>), dgrid, dim3(64), 0, st, APD_ARGS);
                else hipLaunchKernelGGL((ap_direct_kernel<2, false>), dgrid, dim3(64), 0, st, APD_ARGS);
#undef APD_ARGS
            }
            std::swap(*pin, *palt);
            return 0;
        };
        if (int rc = ap_pass(nnlo_pass, &e->buf[EFTB_B_TEMPL], &e->Talt)) return rc;
        if (nnlo_inline)
            if (int rc = ap_pass(true, &e->buf[EFTB_B_TEMPLN], &e->TaltN)) return rc;
    }
    if (mask & EFTB_S_REGROUP) {
        e->cur_nl = Nl;
        e->cur_nx = Nk;
    }
    if (direct_proj) {
        if (int rc = launch_pipeline_operator_plk(e, B, st)) return rc;
        e->plk_host_written = false;  // (the operator writes device memory only: whoever wants P_l in host memory copies it behind the launch)
    } else if (mask & EFTB_S_PROJECT) {
        e->opstream = st;
        const int rc = launch_pipeline_operator(e, B);
        e->opstream = nullptr;
        if (rc) return rc;
    }
    if (mask & EFTB_S_LOGP) {
        if (!e->like_ndata) return fail("eftb_run: stage LOGP needs eftb_set_likelihood");
        if (B % e->ntr) return fail("eftb_run: batch %d is not a multiple of the %d tracers per likelihood point", B, e->ntr);
        if (e->cur_nl != e->like_nl || e->cur_nx != e->like_nx)
            return fail("eftb_run: stage LOGP: the likelihood's data index addresses templates [%d][24][%d], the block is [%d][24][%d]", e->like_nl,
                        e->like_nx, e->cur_nl, e->cur_nx);
        const int nw = B / e->ntr, ng1 = e->like_nG + 1, nd = e->like_ndata;
        const size_t lds = (size_t)e->ntr * ng1 * NROW * sizeof(double);
        if (lds > 64 * 1024) return fail("eftb_run: stage LOGP: %d tracers x %d rows do not fit the coefficient block in LDS", e->ntr, ng1);
        // V (residual and derivatives on the data vector) per walker, U = V C^-1 for all walkers in one matrix-core GEMM, then the
        // (nG + 1)^2 products and the small dense solve per walker
        hipLaunchKernelGGL(marg_build_kernel, dim3(nw), dim3(256), lds, st, e->cur_nl, e->cur_nx, e->ntr, nd, e->like_nG, e->like_index, e->like_data,
                           b[EFTB_B_GROWS], b[EFTB_B_TEMPL], c.with_nnlo ? b[EFTB_B_GROWSN] : nullptr, c.with_nnlo ? b[EFTB_B_TEMPLN] : nullptr,
                           e->like_V);
        GemmDesc gd{};
        gd.A = e->like_V; gd.a_group = 0; gd.a_row = nd; gd.a_seg = 0; gd.rows = nw * ng1; gd.rows_per_group = nw * ng1; gd.nseg = 1; gd.kseg = nd;
        gd.B = e->like_invcov; gd.ldb = nd; gd.ncols = nd;
        gd.C = e->like_U; gd.c_group = 0; gd.c_row = nd; gd.c_colgroup = 0; gd.cols_per_group = nd;
        hipLaunchKernelGGL(gemm_narrow_kernel, dim3((gd.rows + 15) / 16, (gd.ncols + 16 * GN_MAXT - 1) / (16 * GN_MAXT)), dim3(256), 0, st, gd, GemmZ{});
        hipLaunchKernelGGL(marg_solve_kernel, dim3(nw), dim3(256), 0, st, nd, e->like_nG, e->jeffreys, e->like_mu, e->like_sinv, e->like_V, e->like_U,
                           b[EFTB_B_LOGP]);
    }
    if ((mask & EFTB_S_REDUCE) && !fuse_reduce && !direct)
        hipLaunchKernelGGL(reduce_kernel, dim3((e->cur_nx + 255) / 256, e->cur_nl, B), dim3(256), 0, st, e->cur_nx, e->cur_nl, msplit_cfg, b[EFTB_B_BIAS],
                           b[EFTB_B_TEMPL], b[EFTB_B_PLK], c.with_nnlo ? nullptr : e->plk_host_out, e->check_finite && !c.with_nnlo ? e->status + 2 * e->status_slot + 1 : nullptr);
    if ((mask & EFTB_S_REDUCE) && c.with_nnlo)
        hipLaunchKernelGGL(reduce_nnlo_kernel, dim3((e->cur_nx + 255) / 256, e->cur_nl, B), dim3(256), 0, st, e->cur_nx, e->cur_nl, b[EFTB_B_BIASN],
                           b[EFTB_B_TEMPLN], b[EFTB_B_PLK], e->check_finite ? e->status + 2 * e->status_slot + 1 : nullptr);
    if (!(mask & EFTB_S_REGROUP) && !e->use_graphs && !nnlo_pass && hipEventRecord(e->evInFree, st) != hipSuccess) return fail("eftb_run: event record failed");
    if (ap_side) {
        if (hipEventRecord(e->evBack[bslot], st) != hipSuccess) return fail("eftb_run: event record failed");
        ++e->back_step;
        e->back_pending = true;
    }
    if (mask & EFTB_S_PREP) e->prev_front_side = front_side;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail("kernel launch failed: %s", hipGetErrorString(le));
    return 0;
}

static int launch_stages(eftb_engine* e, int mask, int B) {
    const eftb_config& c = e->c;
    const int lin = mask & (EFTB_S_RESUM | EFTB_S_AP | EFTB_S_PROJECT);
    if (!c.with_nnlo || !lin) return launch_stages_impl(e, mask, B, false);
    // with_NNLO: the counter-terms k^4 P11 lctNNLO go through the same linear stages as a second template block whose Pctl
    // slots carry them (reference pybird.py:1447-1458 Resum with Q[1], CctNNLO, lctNNLO; :1615 AP; window.py:399, 410):
    // everything but the tail, then the linear stages again with the NNLO operands swapped in, then LOGP / REDUCE.
    // whole-pipeline steps without PROJECT / LOGP take the three-stream layout: the NNLO block has its own three rotating blocks and
    // follows the main block through every stage inside one call
    const bool fused0 = c.with_resum && c.Nl == 3 && e->resum_splits == 1 && !c.optiresum && !e->generic_resum;
    const int whole = EFTB_S_PREP | EFTB_S_REGROUP | EFTB_S_RESUM | EFTB_S_AP;
    if (fused0 && c.with_ap && (mask & whole) == whole && !(mask & (EFTB_S_PROJECT | EFTB_S_LOGP)) && e->prep_overlap && e->ap_overlap && e->inputs_settled &&
        e->allow_back && !e->use_graphs && e->T3N && e->nnlo_inline)
        return launch_stages_impl(e, mask, B, false, true);
    const int tail = mask & (EFTB_S_LOGP | EFTB_S_REDUCE);
    const int in_nl = (mask & EFTB_S_REGROUP) ? c.Nl : e->cur_nl, in_nx = (mask & EFTB_S_REGROUP) ? c.Nk : e->cur_nx;
    if (int rc = launch_stages_impl(e, mask & ~tail, B, false)) return rc;
    const int out_nl = e->cur_nl, out_nx = e->cur_nx;
    double** b = e->buf;
    auto swap_in = [&]() {
        std::swap(b[EFTB_B_TEMPL], b[EFTB_B_TEMPLN]);
        if (c.with_resum) {
            std::swap(b[EFTB_B_CCT], b[EFTB_B_CCTN]);
            std::swap(b[EFTB_B_C11], e->ZC);
            std::swap(b[EFTB_B_CLOOPL], e->ZC2);
        }
    };
    swap_in();
    e->cur_nl = in_nl;
    e->cur_nx = in_nx;
    // large batches at Nl = 3 accumulate PctNNLOl inside the resummation kernel of the first pass: only AP / PROJECT are left for the block
    const bool fused = c.with_resum && c.Nl == 3 && e->resum_splits == 1 && !c.optiresum && !e->generic_resum;
    const int lin2 = fused ? (lin & ~EFTB_S_RESUM) : lin;
    const int rc = lin2 ? launch_stages_impl(e, lin2, B, true) : 0;
    swap_in();  // the same swaps undo themselves
    e->cur_nl = out_nl;
    e->cur_nx = out_nx;
    if (rc) return rc;
    return tail ? launch_stages_impl(e, tail, B, false) : 0;
}

static void drop_graphs(eftb_engine* e) {
    for (auto& g : e->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    e->graphs.clear();
}

// launch_stages, replayed from a captured graph when the same launch state comes back (the usual case: a sampler calling the
// same pipeline step after step)
static int run_stages(eftb_engine* e, int mask, int B) {
    if (!e->use_graphs || !(mask & EFTB_S_PREP)) return launch_stages(e, mask, B);
    double** b = e->buf;
    for (auto& g : e->graphs)
        if (g.mask == mask && g.B == B && g.epoch == e->epoch && g.templ == b[EFTB_B_TEMPL] && g.talt == e->Talt && g.templn == b[EFTB_B_TEMPLN] &&
            g.cur_nl == e->cur_nl && g.cur_nx == e->cur_nx) {
            if (hipGraphLaunch(g.exec, e->stream) != hipSuccess) return fail("eftb_run: hipGraphLaunch failed");
            (void)hipEventRecord(e->evInFree, e->stream);
            b[EFTB_B_TEMPL] = const_cast<double*>(g.post_templ);
            e->Talt = const_cast<double*>(g.post_talt);
            b[EFTB_B_TEMPLN] = const_cast<double*>(g.post_templn);
            e->cur_nl = g.post_nl;
            e->cur_nx = g.post_nx;
            return 0;
        }
    eftb_engine::GraphEntry g{};
    g.mask = mask; g.B = B; g.epoch = e->epoch; g.cur_nl = e->cur_nl; g.cur_nx = e->cur_nx;
    g.templ = b[EFTB_B_TEMPL]; g.talt = e->Talt; g.templn = b[EFTB_B_TEMPLN];
    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        e->use_graphs = false;  // this runtime cannot capture: plain launches from now on
        return launch_stages(e, mask, B);
    }
    const int rc = launch_stages(e, mask, B);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
    if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    if (ce != hipSuccess || !graph) {
        (void)hipGetLastError();
        e->use_graphs = false;
        return fail("eftb_run: stream capture failed (%s); graphs disabled, call again", hipGetErrorString(ce));
    }
    if (hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        e->use_graphs = false;
        return fail("eftb_run: hipGraphInstantiate failed; graphs disabled, call again");
    }
    g.graph = graph;
    g.post_templ = b[EFTB_B_TEMPL]; g.post_talt = e->Talt; g.post_templn = b[EFTB_B_TEMPLN]; g.post_nl = e->cur_nl; g.post_nx = e->cur_nx;
    if (e->graphs.size() >= 8) {
        (void)hipGraphExecDestroy(e->graphs.front().exec);
        (void)hipGraphDestroy(e->graphs.front().graph);
        e->graphs.erase(e->graphs.begin());
    }
    e->graphs.push_back(g);
    if (hipGraphLaunch(g.exec, e->stream) != hipSuccess) return fail("eftb_run: hipGraphLaunch failed");
    (void)hipEventRecord(e->evInFree, e->stream);
    return 0;
}

// Measurement: EFTB_STREAM_PAD="a,b,c,d,e" creates (and keeps) that many extra streams of the same priority in front of the main / side / look-ahead /
// back / copy stream, each touched once so that its hardware queue exists.  Which hardware queue a stream lands on decides which other streams'
// barrier packets its kernels wait behind (round 4: the copy-out of P_l on a stream of its own ran at 0.16 or 0.103 ms per step depending on how
// many streams were created before it) -- a sweep of the placement of the engine's own five streams, tools/pad_sweep.sh.
static int pad_streams(int which, int prio) {
    const char* f = getenv("EFTB_STREAM_PAD");
    if (!f) return 0;
    int v[5] = {0, 0, 0, 0, 0};
    sscanf(f, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    for (int i = 0; i < v[which]; ++i) {
        hipStream_t dummy = nullptr;
        HIPCHK(hipStreamCreateWithPriority(&dummy, hipStreamNonBlocking, prio));
        hipLaunchKernelGGL(wake_kernel, dim3(1), dim3(64), 0, dummy);
    }
    return 0;
}

extern "C" {

const char* eftb_last_error(void) { return g_err.c_str(); }
#ifndef EFTB_SRC_HASH
#define EFTB_SRC_HASH "unknown"
#endif
const char* eftb_source_hash(void) { return EFTB_SRC_HASH; }
const char* eftb_version(void) { return "eftbird 0.3 (gfx950, fp64: anti-diagonal loops, mfma synthesis / resummation, overlapped steps)"; }

int eftb_create(const eftb_config* cfg, eftb_engine** out) {
    if (!cfg || !out) return fail("eftb_create: null argument");
    const eftb_config& c = *cfg;
    if (c.Nl != 2 && c.Nl != 3) return fail("eftb_create: Nl must be 2 or 3 (got %d)", c.Nl);
    if (c.Nk < 8 || c.Nkin < 4 || c.max_batch < 1) return fail("eftb_create: bad dimensions Nk=%d Nkin=%d max_batch=%d", c.Nk, c.Nkin, c.max_batch);
    if (c.step_batch < 0 || c.step_batch > c.max_batch) return fail("eftb_create: step_batch=%d outside [0, max_batch=%d]", c.step_batch, c.max_batch);
    if (c.nbasis < 1 || c.nbasis > 16) return fail("eftb_create: nbasis=%d outside [1, 16]", c.nbasis);
    if (c.nbasis > BAS22 || (c.with_resum && c.Nl * (c.nbasis + c.nbasis13) > BASC)) return fail("eftb_create: loop basis %d + %d too large", c.nbasis, c.nbasis13);
    if (c.with_resum && !((c.Nl == 3 && c.NIR == 16 && c.Na == 3) || (c.Nl == 2 && c.NIR == 8 && c.Na == 2)))
        return fail("eftb_create: (Nl, NIR, Na) = (%d, %d, %d) unsupported", c.Nl, c.NIR, c.Na);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("eftb_create: no HIP device visible -- libeftbird has no CPU fallback");
    if (c.device < 0 || c.device >= ndev) return fail("eftb_create: device %d out of range (%d visible)", c.device, ndev);
    HIPCHK(hipSetDevice(c.device));
    eftb_engine* e = new eftb_engine();
    e->c = c;
    for (int& g : e->gath_set) g = eftb_engine::NSETS;
    e->Nn = 2 * c.NIR * c.Na;
    e->cur_nl = c.Nl;
    e->cur_nx = c.Nk;
    if (const char* f = getenv("EFTB_GRAPH")) e->use_graphs = atoi(f) != 0;
    if (const char* f = getenv("EFTB_GENERIC_RESUM")) e->generic_resum = atoi(f) != 0;
    if (const char* f = getenv("EFTB_PREP_OVERLAP")) e->prep_overlap = atoi(f) != 0;
    if (const char* f = getenv("EFTB_PREP_AHEAD")) e->prep_ahead = atoi(f) != 0;
    if (const char* f = getenv("EFTB_NNLO_INLINE")) e->nnlo_inline = atoi(f) != 0;
    // Stream priorities.  With the regrouping and the operand build on the look-ahead stream, the look-ahead chain (a dozen latency-bound
    // kernels in series) is the critical path of a pipelined step and the resummation kernel the throughput work beside it: the chain gets
    // the high priority, the main stream the low one (measured at batch 128, evaluations/s: 266-269 k; all normal 265 k; main high and
    // look-ahead low, the round-1 assignment, 240 k; raising the wave priority of the small kernels with s_setprio lost 10 %).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    auto prio_of = [&](const char* name, int dflt) {  // EFTB_*_PRIO = 1 low, 0 normal, -1 high
        const char* f = getenv(name);
        return !f ? dflt : (atoi(f) > 0 ? prio_lo : (atoi(f) < 0 ? prio_hi : 0));
    };
    if (int rc = pad_streams(0, prio_of("EFTB_MAIN_PRIO", prio_lo))) return rc;
    HIPCHK(hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_of("EFTB_MAIN_PRIO", prio_lo)));
    if (int rc = pad_streams(1, prio_of("EFTB_SIDE_PRIO", 0))) return rc;
    HIPCHK(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, prio_of("EFTB_SIDE_PRIO", 0)));
    if (int rc = pad_streams(2, prio_of("EFTB_PRE_PRIO", prio_hi))) return rc;
    HIPCHK(hipStreamCreateWithPriority(&e->pre, hipStreamNonBlocking, prio_of("EFTB_PRE_PRIO", prio_hi)));
    HIPCHK(hipEventCreateWithFlags(&e->evPrep, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evFront, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evFrontFree[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evFrontFree[1], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evXY, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evAS, hipEventDisableTiming));
    if (const char* f = getenv("EFTB_AP_OVERLAP")) e->ap_overlap = atoi(f) != 0;
    {
        int bp = prio_lo;  // measured: 290 k evaluations/s with the back half at low priority, 286 k at high, 278 k without the stream
        if (const char* f = getenv("EFTB_BACK_PRIO")) bp = atoi(f) > 0 ? prio_lo : (atoi(f) < 0 ? prio_hi : 0);
        if (int rc = pad_streams(3, bp)) return rc;
        HIPCHK(hipStreamCreateWithPriority(&e->back, hipStreamNonBlocking, bp));
    }
    HIPCHK(hipEventCreateWithFlags(&e->evResum, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evBack[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evBack[1], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evInFree, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->evInFree, e->pre));
    HIPCHK(hipEventCreateWithFlags(&e->evFork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evJoin, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evJoinAP, hipEventDisableTiming));
    for (hipEvent_t& ev : e->evRsDone) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (hipEvent_t& ev : e->evRun) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int t = 0; t < eftb_engine::NTIMER; ++t) {
        HIPCHK(hipEventCreate(&e->evT0[t]));
        HIPCHK(hipEventCreate(&e->evT1[t]));
    }
    HIPCHK(hipEventCreate(&e->ev0));
    HIPCHK(hipEventCreate(&e->ev1));
    for (int id = 0; id < EFTB_B_COUNT; ++id) {
        const size_t n = need_buffer_elems(c, id);
        e->buf_elems[id] = n;
        if (n) {
            HIPCHK(hipMalloc(&e->buf[id], (n + 2) * sizeof(double)));  // (+2: ap_rows_kernel's 16-byte window loads may end one element past a row)
            HIPCHK(hipMemset(e->buf[id], 0, (n + 2) * sizeof(double)));
        }
    }
    const size_t B = c.max_batch;
    HIPCHK(hipMalloc(&e->sm2, NS * sizeof(double)));
    HIPCHK(hipMalloc(&e->sm4, NS * sizeof(double)));
    // scratch of the anti-diagonal pipeline; padded rows / coefficients of the synthesis rows stay zero
    {
        const size_t nc = c.nbasis + (c.with_resum ? c.nbasis13 : 0);
        auto zalloc = [](double** p, size_t n) {
            if (hipMalloc(p, n * sizeof(double)) != hipSuccess) return 1;
            return hipMemset(*p, 0, n * sizeof(double)) != hipSuccess ? 1 : 0;
        };
        int bad = zalloc(&e->coefT, 2 * NCH * B);
        bad |= zalloc(&e->PA1, B * kpad(c.Nkin)) | zalloc(&e->PA2, B * kpad(c.Nkin + c.ntail)) | zalloc(&e->PA2T, B * kpad(c.Nkin + c.ntail));
        if (c.with_resum) bad |= zalloc(&e->PA3, B * kpad(c.Nkin + c.nxtail));
        if (c.dual_coef) bad |= zalloc(&e->coef2, 2 * NCH * B) | zalloc(&e->coefT2, 2 * NCH * B);
        bad |= zalloc(reinterpret_cast<double**>(&e->SAD), 2 * (size_t)AD_CH * B * nc * NPOW);
        bad |= zalloc(&e->A22, B * BAS22 * KSYN) | zalloc(&e->A13, B * 10 * KLIN) | zalloc(&e->Y22, B * BAS22 * c.Nk);
        if (c.with_resum) bad |= zalloc(&e->ACF, B * BASC * KSYN) | zalloc(&e->ALC, B * (c.with_nnlo ? 3 : 2) * c.Nl * KLIN) | zalloc(&e->YCF, B * BASC * NS);
        if (bad) return fail("eftb_create: out of device memory for the loop scratch");
        if (c.with_resum && c.Nl == 3 && !c.dual_coef) {  // second set of the front's outputs (direct-P_l runs, see FrontSet)
            eftb_engine::FrontSet& a = e->alt;
            bad = zalloc(&a.coefT, 2 * NCH * B) | zalloc(&a.PA1, B * kpad(c.Nkin)) | zalloc(&a.PA2, B * kpad(c.Nkin + c.ntail)) | zalloc(&a.PA2T, B * kpad(c.Nkin + c.ntail)) |
                  zalloc(&a.PA3, B * kpad(c.Nkin + c.nxtail)) | zalloc(reinterpret_cast<double**>(&a.SAD), 2 * (size_t)AD_CH * B * nc * NPOW) |
                  zalloc(&a.A22, B * BAS22 * KSYN) | zalloc(&a.A13, B * 10 * KLIN) | zalloc(&a.ACF, B * BASC * KSYN) | zalloc(&a.ALC, B * (c.with_nnlo ? 3 : 2) * c.Nl * KLIN) |
                  zalloc(&a.P11, e->buf_elems[EFTB_B_P11]) | zalloc(&a.COEF, e->buf_elems[EFTB_B_COEF]) | zalloc(&a.XY, e->buf_elems[EFTB_B_XY]) |
                  zalloc(&a.Q, e->buf_elems[EFTB_B_Q]);
            if (bad) return fail("eftb_create: out of device memory for the second front set");
        }
    }
    HIPCHK(hipMalloc(&e->Talt, (e->buf_elems[EFTB_B_TEMPL] + 2) * sizeof(double)));
    if (c.with_ap) {
        HIPCHK(hipMalloc(&e->SD, (e->buf_elems[EFTB_B_TEMPL] + 2) * sizeof(double)));  // knot slopes of the splines, laid out like the template block
        HIPCHK(hipMemset(e->SD, 0, (e->buf_elems[EFTB_B_TEMPL] + 2) * sizeof(double)));
        HIPCHK(hipMalloc(&e->APP, (size_t)c.max_batch * (c.nmu + 1) * c.Nl * c.Nl * 4 * sizeof(double)));
        HIPCHK(hipMalloc(&e->APR, (size_t)c.max_batch * c.nmu * sizeof(double)));
        HIPCHK(hipMalloc(&e->APP2, (size_t)c.max_batch * (c.nmu + 1) * c.Nl * c.Nl * 4 * sizeof(double)));
        HIPCHK(hipMalloc(&e->APR2, (size_t)c.max_batch * c.nmu * sizeof(double)));
        const size_t kt = (c.Nk + 63) / 64;
        for (int q = 0; q < 2; ++q) {
            HIPCHK(hipMalloc(q ? &e->APW2 : &e->APW, (size_t)c.max_batch * kt * APW_DCAP * c.Nl * c.Nl * 64 * sizeof(double)));
            HIPCHK(hipMalloc(q ? &e->API2 : &e->API, (size_t)c.max_batch * kt * 64 * sizeof(int)));
            HIPCHK(hipMalloc(q ? &e->APM2 : &e->APM, (size_t)c.max_batch * 2 * kt * sizeof(int4)));  // one window record per tile of 32 k
        }
    }
    if (const char* f = getenv("EFTB_AP_FAST")) e->ap_fast = atoi(f) != 0;
    if (const char* f = getenv("EFTB_AP_RING")) e->ap_ring = atoi(f);
    if (const char* f = getenv("EFTB_AD_WAVES")) e->ad_waves = atoi(f);
    if (const char* f = getenv("EFTB_GD_WAVES")) e->gd_waves = atoi(f);
    if (const char* f = getenv("EFTB_FUSE_CF")) e->fuse_cf = atoi(f) != 0;
    if (const char* f = getenv("EFTB_AP_PLK_NODES")) e->ap_plk_nodes = atoi(f) != 0;
    if (const char* f = getenv("EFTB_UPLOAD_ON_SIDE")) e->upload_on_side = atoi(f) != 0;
    if (const char* f = getenv("EFTB_SUBMIT_THREAD")) e->sub_mode = std::max(0, std::min(2, atoi(f)));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->status), 2 * (eftb_engine::NSETS + 1) * sizeof(int), hipHostMallocMapped));
    memset(e->status, 0, 2 * (eftb_engine::NSETS + 1) * sizeof(int));
    HIPCHK(hipDeviceSynchronize());  // the zero fills above ran on the null stream, which the engine's non-blocking streams do not wait for
    *out = e;
    return 0;
}

int eftb_set_table(eftb_engine* e, int id, const void* host, size_t nbytes) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e || !host) return fail("eftb_set_table: null argument");
    if (id < 0 || id >= EFTB_T_COUNT) return fail("eftb_set_table: bad table id %d", id);
    const size_t need = need_table_bytes(e->c, id);
    if (need == 0) return fail("eftb_set_table: table %d is not used by this configuration", id);
    if (need != nbytes) return fail("eftb_set_table: table %d expects %zu bytes, got %zu", id, need, nbytes);
    HIPCHK(hipSetDevice(e->c.device));
    if (!e->tab[id]) HIPCHK(hipMalloc(&e->tab[id], nbytes));
    else if (e->finalized) HIPCHK(sync_all(e));  // replacing a resident table (e.g. the AP fiducial): no run may still be reading it
    HIPCHK(hipMemcpy(e->tab[id], host, nbytes, hipMemcpyHostToDevice));
    e->tab_bytes[id] = nbytes;
    if (id == EFTB_T_S) {
        std::vector<double> v(NS);
        const double* sv = static_cast<const double*>(host);
        for (int i = 0; i < NS; ++i) v[i] = 1.0 / (sv[i] * sv[i]);
        HIPCHK(hipMemcpy(e->sm2, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < NS; ++i) v[i] = 1.0 / (sv[i] * sv[i] * sv[i] * sv[i]);
        HIPCHK(hipMemcpy(e->sm4, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return 0;
}

int eftb_finalize(eftb_engine* e) {
    if (e) sub_drain(e);
    if (!e) return fail("eftb_finalize: null engine");
    for (int id = 0; id < EFTB_T_COUNT; ++id)
        if (need_table_bytes(e->c, id) && !e->tab[id]) return fail("eftb_finalize: table %d was never set", id);
    HIPCHK(hipSetDevice(e->c.device));
    const eftb_config& c = e->c;
    if (c.with_resum) {
        // split the s sum when the batch alone cannot fill the chip (deterministic two-pass reduction)
        e->resum_splits = c.max_batch >= 16 ? 1 : 8;
        {
            const size_t pbytes = (size_t)c.max_batch * e->resum_splits * 2 * c.Nl * c.Nl * 21 * c.Nk * sizeof(double);
            HIPCHK(hipMalloc(&e->part, pbytes));
            HIPCHK(hipMemset(e->part, 0, pbytes));  // the matrix-core kernel never touches k < Nklow
        }
        if (c.optiresum) HIPCHK(hipMalloc(&e->XB, (size_t)c.max_batch * c.Nl * 14 * NS * sizeof(double)));
        if (c.with_nnlo) {
            HIPCHK(hipMalloc(&e->ZC, e->buf_elems[EFTB_B_C11] * sizeof(double)));
            HIPCHK(hipMalloc(&e->ZC2, e->buf_elems[EFTB_B_CLOOPL] * sizeof(double)));
            HIPCHK(hipMemset(e->ZC, 0, e->buf_elems[EFTB_B_C11] * sizeof(double)));
            HIPCHK(hipMemset(e->ZC2, 0, e->buf_elems[EFTB_B_CLOOPL] * sizeof(double)));
        }
        {
            const size_t abytes = (size_t)c.max_batch * RS_ROWS * RS_NB * sizeof(double);
            HIPCHK(hipMalloc(&e->RSA, abytes));
            HIPCHK(hipMalloc(&e->RSC, ((size_t)c.max_batch * NS + RS_PF) * RS_REC * sizeof(double)));
            HIPCHK(hipMalloc(&e->RSA2, abytes));
            if (c.Nl == 3) {  // one operand per s
                HIPCHK(hipMalloc(&e->RSAS, ((size_t)c.max_batch * NS + RS_PF) * RS3_AS * sizeof(double)));
                HIPCHK(hipMalloc(&e->RSAS2, ((size_t)c.max_batch * NS + RS_PF) * RS3_AS * sizeof(double)));
            }
            HIPCHK(hipMalloc(&e->RSC2, ((size_t)c.max_batch * NS + RS_PF) * RS_REC * sizeof(double)));
        }
    }
    // opt in to the large dynamic LDS tiles
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (c.with_ap) {
        // (the kernel also holds a few static words: the dynamic part must leave room for them below the 160 KB of a CU)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ap_weights_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ap_weights_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        {   // which form of the AP stage (see engine.ap_mode): needs the k grid, so it is decided here
            std::vector<double> kh(c.Nk);
            HIPCHK(hipMemcpy(kh.data(), tb<double>(e, EFTB_T_K), c.Nk * sizeof(double), hipMemcpyDeviceToHost));
            const double dk = kh[c.Nk - 1] - kh[c.Nk - 2];
            e->ap_mode = dk > 0.0 && 0.02 * kh[c.Nk - 1] / dk > APW_DCAP / 2 ? 1 : 0;
            if (const char* f = getenv("EFTB_AP_MODE")) e->ap_mode = atoi(f);
            if (!e->ap_fast) e->ap_mode = 2;
            e->ap_fast = e->ap_mode == 0;
            {   // the all-in-LDS form of the direct-P_l AP stage, where its tables fit
                const size_t lds = ((size_t)((c.Nk + 1) & ~1) + (size_t)c.nmu * 8 + 36 * 16 + (size_t)(c.nmu + 1) * 36 + (size_t)(c.Nk - 1) * 12) * sizeof(double);
                e->ap_plk_fused = c.Nl == 3 && lds <= 150 * 1024 && c.nmu >= 2 && c.nmu <= 7 * 32 && !(getenv("EFTB_AP_PLK_FUSED") && !atoi(getenv("EFTB_AP_PLK_FUSED")));
                if (e->ap_plk_fused)
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ap_plk_fused_kernel<3, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            }
#define APM_LDS(NLV, NRV) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ap_moments_kernel<NLV, NRV, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256))
            APM_LDS(3, 21); APM_LDS(3, NROW); APM_LDS(2, 21); APM_LDS(2, NROW);
#undef APM_LDS
        }
    }
    if (c.with_ap && c.Nl == 3 && !e->PLK0) {
        HIPCHK(hipMalloc(&e->PLK0, e->buf_elems[EFTB_B_PLK] * sizeof(double)));
        HIPCHK(hipMemset(e->PLK0, 0, e->buf_elems[EFTB_B_PLK] * sizeof(double)));
    }
    if (e->ap_overlap && !e->T3) HIPCHK(hipMalloc(&e->T3, (e->buf_elems[EFTB_B_TEMPL] + 2) * sizeof(double)));  // third template block (engine.back)
    if (e->ap_overlap && e->c.with_nnlo && e->c.with_ap && !e->T3N) {
        HIPCHK(hipMalloc(&e->T3N, (e->buf_elems[EFTB_B_TEMPLN] + 2) * sizeof(double)));
        HIPCHK(hipMalloc(&e->TaltN, (e->buf_elems[EFTB_B_TEMPLN] + 2) * sizeof(double)));
        HIPCHK(hipMemset(e->T3N, 0, e->buf_elems[EFTB_B_TEMPLN] * sizeof(double)));
        HIPCHK(hipMemset(e->TaltN, 0, e->buf_elems[EFTB_B_TEMPLN] * sizeof(double)));
    }
    HIPCHK(hipDeviceSynchronize());  // null-stream zero fills (part, ZC, ZC2) are not ordered against the engine's non-blocking streams
    e->finalized = true;
    return 0;
}

int eftb_add_operator(eftb_engine* e, int nl_out, int nx_out, int nl_in, int nx_in, const double* op, int* op_id) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e || !op || !op_id) return fail("eftb_add_operator: null argument");
    if (nl_out < 1 || nx_out < 1 || nl_in < 1 || nx_in < 1) return fail("eftb_add_operator: bad shape");
    if ((size_t)nl_out * nx_out > (size_t)e->c.Nl * e->c.Nk || (size_t)nl_in * nx_in > (size_t)e->c.Nl * e->c.Nk)
        return fail("eftb_add_operator: operator shapes must not exceed the engine's [Nl=%d][Nk=%d] block", e->c.Nl, e->c.Nk);
    HIPCHK(hipSetDevice(e->c.device));
    eftb_engine::Op o{nl_out, nx_out, nl_in, nx_in, (nl_out * nx_out + 15) / 16 * 16, nullptr, -1};
    // K-major copy: opT[(l,k)][(a,x)]
    std::vector<double> t((size_t)nl_in * nx_in * o.ld, 0.0);
    for (int a = 0; a < nl_out; ++a)
        for (int l = 0; l < nl_in; ++l)
            for (int x = 0; x < nx_out; ++x) {
                const double* src = op + (((size_t)a * nl_in + l) * nx_out + x) * nx_in;
                for (int k = 0; k < nx_in; ++k) t[((size_t)l * nx_in + k) * o.ld + (size_t)a * nx_out + x] = src[k];
            }
    HIPCHK(hipMalloc(&o.dev, t.size() * sizeof(double)));
    HIPCHK(hipMemcpy(o.dev, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    e->ops.push_back(o);
    *op_id = (int)e->ops.size() - 1;
    return 0;
}

int eftb_set_operator_stochastic(eftb_engine* e, int op_id, int st_op_id) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e) return fail("eftb_set_operator_stochastic: null engine");
    const int n = (int)e->ops.size();
    if (op_id < 0 || op_id >= n || st_op_id < -1 || st_op_id >= n) return fail("eftb_set_operator_stochastic: operator id out of range");
    if (st_op_id >= 0) {
        const eftb_engine::Op &a = e->ops[op_id], &b = e->ops[st_op_id];
        if (a.nl_out != b.nl_out || a.nx_out != b.nx_out || a.nl_in != b.nl_in || a.nx_in != b.nx_in)
            return fail("eftb_set_operator_stochastic: operators %d and %d differ in shape", op_id, st_op_id);
    }
    e->ops[op_id].st_op = st_op_id;
    return 0;
}

int eftb_set_tracers(eftb_engine* e, int ntr) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e) return fail("eftb_set_tracers: null engine");
    if (ntr < 1 || ntr > e->c.max_batch) return fail("eftb_set_tracers: %d tracers outside [1, max_batch]", ntr);
    e->ntr = ntr;
    e->tracer_ops.clear();
    e->like_ndata = 0;  // a likelihood set for another grouping does not survive
    return 0;
}

int eftb_set_pipeline_operator_tracer(eftb_engine* e, int tracer, int op_id) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e) return fail("eftb_set_pipeline_operator_tracer: null engine");
    const int n = (int)e->ops.size();
    if (tracer < 0 || tracer >= e->ntr || op_id < 0 || op_id >= n) return fail("eftb_set_pipeline_operator_tracer: tracer %d / operator %d out of range", tracer, op_id);
    if (e->tracer_ops.empty()) e->tracer_ops.assign(e->ntr, op_id);  // until told otherwise every tracer uses the first one given
    const eftb_engine::Op& b = e->ops[op_id];
    for (int t = 0; t < e->ntr; ++t) {
        const eftb_engine::Op& a = e->ops[e->tracer_ops[t]];
        if (t != tracer && (a.nl_out != b.nl_out || a.nx_out != b.nx_out || a.nl_in != b.nl_in || a.nx_in != b.nx_in))
            return fail("eftb_set_pipeline_operator_tracer: the tracers' operators must share one shape (pad chained outputs with zero rows)");
    }
    e->tracer_ops[tracer] = op_id;
    return 0;
}

int eftb_apply_operator(eftb_engine* e, int op_id, int B) {
    if (e) sub_drain(e);
    if (!e) return fail("eftb_apply_operator: null engine");
    if (!e->finalized) return fail("eftb_apply_operator: engine not finalized");
    if (B < 1 || B > e->c.max_batch) return fail("eftb_apply_operator: batch %d outside [1, %d]", B, e->c.max_batch);
    HIPCHK(hipSetDevice(e->c.device));
    join_back(e);
    if (int rc = launch_operator(e, op_id, B)) return rc;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail("kernel launch failed: %s", hipGetErrorString(le));
    return 0;
}

int eftb_set_pipeline_operator(eftb_engine* e, int op_id) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e) return fail("eftb_set_pipeline_operator: null engine");
    if (op_id >= (int)e->ops.size()) return fail("eftb_set_pipeline_operator: operator id %d out of range", op_id);
    e->pipeline_op = op_id < 0 ? -1 : op_id;
    return 0;
}

int eftb_set_template_dims(eftb_engine* e, int nl, int nx) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e) return fail("eftb_set_template_dims: null engine");
    if (nl < 1 || nx < 1 || (size_t)nl * nx > (size_t)e->c.Nl * e->c.Nk) return fail("eftb_set_template_dims: bad shape [%d][24][%d]", nl, nx);
    e->cur_nl = nl;
    e->cur_nx = nx;
    return 0;
}

int eftb_set_option(eftb_engine* e, int option, int value) {
    if (e && option == EFTB_O_SUBMIT_HOLD) { e->sub_hold.store(value != 0, std::memory_order_release); return 0; }  // (no drain: the queue is meant to stand)
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e) return fail("eftb_set_option: null engine");
    switch (option) {
        case EFTB_O_AP_STOCHASTIC: e->c.ap_stochastic = value ? 1 : 0; return 0;
        case EFTB_O_JEFFREYS: e->jeffreys = value ? 1 : 0; return 0;
        case EFTB_O_GRAPH: e->use_graphs = value != 0; return 0;
        case EFTB_O_CHECK_FINITE: e->check_finite = value != 0; return 0;
        case EFTB_O_TIME_DOMINANT: e->time_dominant = value < 0 ? 0 : value; e->time_seqk[0] = e->time_seqk[1] = e->time_seqk[2] = 0; return 0;
        case EFTB_O_LATENCY_MODE: e->latency_auto = value != 0; return 0;
        case EFTB_O_PLK_DIRECT: e->plk_direct = value != 0; return 0;
        case EFTB_O_STEP_TRACE:
            if (value && !e->evTraceBase) {
                HIPCHK(hipSetDevice(e->c.device));
                HIPCHK(hipEventCreate(&e->evTraceBase));
                for (auto& row : e->evTrace)
                    for (hipEvent_t& ev : row) HIPCHK(hipEventCreate(&ev));
            }
            if (value) {
                HIPCHK(hipEventRecord(e->evTraceBase, e->stream));
                e->trace_n = 0;
            }
            e->trace_on = value != 0;
            return 0;
        case EFTB_O_SUBMIT_HOLD: e->sub_hold.store(value != 0, std::memory_order_release); return 0;
        case EFTB_O_SUBMIT_THREAD:
            if (value < 0 || value > 2) return fail("eftb_set_option: EFTB_O_SUBMIT_THREAD takes 0, 1 or 2");
            e->sub_mode = value; return 0;
        case EFTB_O_TIME_KERNEL:
            if (value < 1 || value > 7) return fail("eftb_set_option: EFTB_O_TIME_KERNEL takes a set of 1 (resummation) | 2 (synthesis) | 4 (AP knot weights)");
            e->time_kernel = value; return 0;
    }
    return fail("eftb_set_option: unknown option %d", option);
}

int eftb_dominant_time(eftb_engine* e, double* ms_sum, long long* launches, int reset) {
    if (e) sub_drain(e);
    if (!e || !ms_sum || !launches) return fail("eftb_dominant_time: null argument");
    HIPCHK(hipSetDevice(e->c.device));
    return eftb_kernel_time(e, 0, ms_sum, launches, reset);
}

int eftb_kernel_time(eftb_engine* e, int kind, double* ms_sum, long long* launches, int reset) {
    if (e) sub_drain(e);
    if (!e || !ms_sum || !launches) return fail("eftb_kernel_time: null argument");
    if (kind < 0 || kind > 2) return fail("eftb_kernel_time: kind %d (0 resummation, 1 synthesis, 2 AP knot weights)", kind);
    HIPCHK(hipSetDevice(e->c.device));
    collect_timer(e, -1, true);
    *ms_sum = e->timer_ms[kind];
    *launches = e->timer_n[kind];
    if (reset) {
        e->timer_ms[kind] = 0.0;
        e->timer_n[kind] = 0;
        e->timer_cosmo[kind] = 0;
    }
    return 0;
}

int eftb_kernel_time_ex(eftb_engine* e, int kind, double* ms_sum, long long* launches, long long* cosmologies, int reset) {
    if (!cosmologies) return fail("eftb_kernel_time_ex: null argument");
    if (e && kind >= 0 && kind <= 2) {
        sub_drain(e);
        (void)hipSetDevice(e->c.device);
        collect_timer(e, -1, true);
        *cosmologies = e->timer_cosmo[kind];
    }
    return eftb_kernel_time(e, kind, ms_sum, launches, reset);
}

int eftb_set_likelihood(eftb_engine* e, int ndata, const int32_t* index, const double* data, const double* invcov, int nG, const double* mu,
                        const double* sigma_inv) {
    if (e) sub_drain(e);
    if (e) ++e->epoch;  // invalidates the captured graphs
    if (!e || !index || !data || !invcov || (nG > 0 && (!mu || !sigma_inv))) return fail("eftb_set_likelihood: null argument");
    if (nG < 0 || nG > MARG_MAXG) return fail("eftb_set_likelihood: nG=%d outside [0, %d]", nG, MARG_MAXG);  // nG = 0: plain -chi2 / 2
    if (ndata < 1) return fail("eftb_set_likelihood: ndata=%d", ndata);
    // the shape the index refers to: what the PROJECT stage will produce if an operator is registered for it and the block has not been
    // projected yet, else the block as it is; the LOGP stage checks it again at launch
    int lnl = e->cur_nl, lnx = e->cur_nx;
    {
        const int op = !e->tracer_ops.empty() ? e->tracer_ops[0] : e->pipeline_op;
        if (op >= 0 && e->ops[op].nl_in == e->cur_nl && e->ops[op].nx_in == e->cur_nx) {
            lnl = e->ops[op].nl_out;
            lnx = e->ops[op].nx_out;
        }
    }
    const int npts = e->ntr * lnl * lnx;  // the walker's ntr entries are addressed as one block of ntr * nl multipoles
    for (int a = 0; a < ndata; ++a)
        if (index[a] < 0 || index[a] >= npts)
            return fail("eftb_set_likelihood: index[%d]=%d outside the template block [%d x %d][24][%d]", a, index[a], e->ntr, lnl, lnx);
    for (int a = 0; a < ndata; ++a)
        for (int b2 = 0; b2 < a; ++b2) {
            // symmetric up to the rounding of a matrix inversion (the reference passes np.linalg.inv(cov) as it comes out, likelihood.py:372):
            // measured against the diagonal, not against the element itself -- small off-diagonal entries carry the inversion's absolute error
            const double x = invcov[(size_t)a * ndata + b2], y = invcov[(size_t)b2 * ndata + a];
            const double sc = sqrt(fabs(invcov[(size_t)a * ndata + a] * invcov[(size_t)b2 * ndata + b2]));
            if (fabs(x - y) > 1e-8 * sc + 1e-300) return fail("eftb_set_likelihood: invcov is not symmetric at (%d, %d)", a, b2);
        }
    HIPCHK(hipSetDevice(e->c.device));
    HIPCHK(sync_all(e));  // a likelihood stage of an overlapped run may still be reading the old tables on the back-half stream
    for (void* p : {(void*)e->like_index, (void*)e->like_data, (void*)e->like_invcov, (void*)e->like_mu, (void*)e->like_sinv, (void*)e->like_V, (void*)e->like_U}) if (p) (void)hipFree(p);
    e->like_index = nullptr; e->like_data = e->like_invcov = e->like_mu = e->like_sinv = e->like_V = e->like_U = nullptr;
    {
        const size_t vb = (size_t)(e->c.max_batch / e->ntr + 1) * (nG + 1) * ndata * sizeof(double);
        HIPCHK(hipMalloc(&e->like_V, vb));
        HIPCHK(hipMalloc(&e->like_U, vb));
    }
    HIPCHK(hipMalloc(&e->like_index, ndata * sizeof(int)));
    HIPCHK(hipMalloc(&e->like_data, ndata * sizeof(double)));
    HIPCHK(hipMalloc(&e->like_invcov, (size_t)ndata * ndata * sizeof(double)));
    HIPCHK(hipMalloc(&e->like_mu, MARG_MAXG * sizeof(double)));
    HIPCHK(hipMalloc(&e->like_sinv, MARG_MAXG * sizeof(double)));
    HIPCHK(hipMemcpy(e->like_index, index, ndata * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->like_data, data, ndata * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->like_invcov, invcov, (size_t)ndata * ndata * sizeof(double), hipMemcpyHostToDevice));
    if (nG > 0) {
        HIPCHK(hipMemcpy(e->like_mu, mu, nG * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->like_sinv, sigma_inv, nG * sizeof(double), hipMemcpyHostToDevice));
    }
    e->like_ndata = ndata;
    e->like_nG = nG;
    e->like_nl = lnl;
    e->like_nx = lnx;
    return 0;
}

void eftb_destroy(eftb_engine* e) {
    if (!e) return;
    sub_stop_thread(e);
    (void)hipSetDevice(e->c.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
    for (hipStream_t q : {e->pre, e->side, e->cpy, e->back}) if (q) (void)hipStreamSynchronize(q);  // look-ahead work and staged uploads still in flight
    drop_graphs(e);
    if (e->cpy)
        for (int id = 0; id < EFTB_B_COUNT; ++id)
            if (e->orig[id]) e->buf[id] = e->orig[id];  // the staged sets are freed below, the engine's own buffers with e->buf
    for (auto& p : e->tab) if (p) (void)hipFree(p);
    for (auto& p : e->buf) if (p) (void)hipFree(p);
    for (double* p : {e->alt.PA1, e->alt.PA2, e->alt.PA2T, e->alt.PA3, e->alt.coefT, e->alt.A22, e->alt.A13, e->alt.ACF, e->alt.ALC, e->alt.P11, e->alt.COEF, e->alt.XY, e->alt.Q, reinterpret_cast<double*>(e->alt.SAD), e->PLK0, e->RSA, e->RSC, e->RSA2, e->RSC2, e->RSAS, e->RSAS2, e->APP, e->APR, e->APP2, e->APR2, e->SD, e->Talt, e->T3, e->TaltN, e->T3N, e->part, e->plk_snap, e->PA1, e->PA2, e->PA2T, e->PA3, e->A22, e->A13, e->ACF, e->ALC, e->Y22, e->YCF, e->coefT, e->sm2, e->sm4, e->ZC, e->ZC2, e->coef2, e->coefT2, e->XB, reinterpret_cast<double*>(e->SAD)}) if (p) (void)hipFree(p);
    for (int q = 0; q < eftb_engine::NSETS; ++q) {
        if (e->gathered2[q]) (void)hipFree(e->gathered2[q]);
        if (e->evGath2[q]) (void)hipEventDestroy(e->evGath2[q]);
        if (e->gath_host[q]) (void)hipHostFree(e->gath_host[q]);
        if (e->evGathHost[q]) (void)hipEventDestroy(e->evGathHost[q]);
    }
    e->gathered = nullptr;  // (one of gathered2)
    for (void* p : {(void*)e->APW, (void*)e->APW2, (void*)e->API, (void*)e->API2, (void*)e->APM, (void*)e->APM2}) if (p) (void)hipFree(p);
    for (auto& o : e->ops) if (o.dev) (void)hipFree(o.dev);
    for (void* p : {(void*)e->like_index, (void*)e->like_data, (void*)e->like_invcov, (void*)e->like_mu, (void*)e->like_sinv, (void*)e->like_V, (void*)e->like_U}) if (p) (void)hipFree(p);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    for (hipEvent_t ev : {e->ev0, e->ev1, e->evFork, e->evJoin, e->evJoinAP, e->evXY, e->evAS, e->evFront, e->evFrontFree[0], e->evFrontFree[1], e->evSnap, e->evGathered, e->evPrep, e->evInFree, e->evResum, e->evBack[0], e->evBack[1], e->evRsDone[0], e->evRsDone[1]}) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->evRun) if (ev) (void)hipEventDestroy(ev);
    for (int t = 0; t < eftb_engine::NTIMER; ++t) {
        if (e->evT0[t]) (void)hipEventDestroy(e->evT0[t]);
        if (e->evT1[t]) (void)hipEventDestroy(e->evT1[t]);
    }
    if (e->pre) (void)hipStreamDestroy(e->pre);
    if (e->back) (void)hipStreamDestroy(e->back);
    if (e->cpy) (void)hipStreamDestroy(e->cpy);
    for (int h = 0; h < eftb_engine::NSLOT; ++h)
        if (e->stage_host[h]) (void)hipHostFree(e->stage_host[h]);
    for (int r = 0; r < eftb_engine::NLRING; ++r) {
        if (e->evStagedR[r]) (void)hipEventDestroy(e->evStagedR[r]);
        if (e->evStagedAllR[r]) (void)hipEventDestroy(e->evStagedAllR[r]);
    }
    for (int q = 0; q < eftb_engine::NSETS; ++q) {
        if (e->evSetDone[q]) (void)hipEventDestroy(e->evSetDone[q]);
        if (e->setblock[q]) (void)hipFree(e->setblock[q]);
        for (int id : {EFTB_B_PLK, EFTB_B_LOGP})
            if (e->setbuf[q][id]) {
                if (id == EFTB_B_PLK && e->plk_host[q]) (void)hipHostFree(e->plk_host[q]);
                if (id == EFTB_B_PLK && e->staged_plk_device) (void)hipFree(e->setbuf[q][id]);
                else (void)hipHostFree(e->setbuf[q][id]);
            }
    }
    if (e->evTraceBase) {
        (void)hipEventDestroy(e->evTraceBase);
        for (auto& row : e->evTrace)
            for (hipEvent_t ev : row) if (ev) (void)hipEventDestroy(ev);
    }
    if (e->set_done) (void)hipHostFree(const_cast<unsigned long long*>(e->set_done));
    if (e->sub_stats && e->issue_n)
        fprintf(stderr, "[eftbird] staged steps: %llu issued (%llu by the caller's thread), %.1f us per step in the issuing thread, %.1f us filling the staging block, "
                        "%.1f us waiting for results\n", e->issue_n, e->inline_n, e->issue_ns / e->issue_n * 1e-3, e->fill_ns / e->issue_n * 1e-3, e->wait_ns / e->issue_n * 1e-3);
    if (e->status) (void)hipHostFree(e->status);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->comm_stream) (void)hipStreamDestroy(e->comm_stream);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

size_t eftb_buffer_size(const eftb_engine* e, int id) {
    if (!e || id < 0 || id >= EFTB_B_COUNT) return 0;
    return e->buf_elems[id];
}

int eftb_put(eftb_engine* e, int id, size_t offset, const double* host, size_t count) {
    if (e) sub_drain(e);
    if (!e || !host) return fail("eftb_put: null argument");
    if (id < 0 || id >= EFTB_B_COUNT || !e->buf[id]) return fail("eftb_put: buffer %d not available in this configuration", id);
    if (offset + count > e->buf_elems[id]) return fail("eftb_put: buffer %d holds %zu elements, asked [%zu, %zu)", id, e->buf_elems[id], offset, offset + count);
    HIPCHK(hipSetDevice(e->c.device));
    join_back(e);
    HIPCHK(hipMemcpyAsync(e->buf[id] + offset, host, count * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int eftb_get(eftb_engine* e, int id, size_t offset, double* host, size_t count) {
    if (e) sub_drain(e);
    if (!e || !host) return fail("eftb_get: null argument");
    if (id < 0 || id >= EFTB_B_COUNT || !e->buf[id]) return fail("eftb_get: buffer %d not available in this configuration", id);
    if (offset + count > e->buf_elems[id]) return fail("eftb_get: buffer %d holds %zu elements, asked [%zu, %zu)", id, e->buf_elems[id], offset, offset + count);
    HIPCHK(hipSetDevice(e->c.device));
    join_back(e);
    HIPCHK(hipMemcpyAsync(host, e->buf[id] + offset, count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return check_status(e, "eftb_get");
}

int eftb_run(eftb_engine* e, int mask, int B) {
    if (e) sub_drain(e);
    if (!e) return fail("eftb_run: null engine");
    if (!e->finalized) return fail("eftb_run: engine not finalized");
    if (B < 1 || B > e->c.max_batch) return fail("eftb_run: batch %d outside [1, %d]", B, e->c.max_batch);
    HIPCHK(hipSetDevice(e->c.device));
    const int slot = (int)(e->run_seq % eftb_engine::RUN_DEPTH);
    if (e->run_seq >= eftb_engine::RUN_DEPTH)  // the run launched RUN_DEPTH calls ago must have finished (spin: it almost always has)
        if (int rc = spin_event(e->evRun[slot], "eftb_run", "the run launched RUN_DEPTH calls ago")) return rc;
    e->status_slot = eftb_engine::NSETS;
    e->inputs_settled = true;  // eftb_put is synchronous: the inputs of this run are in place, its first stage may start early
    e->allow_back = true;
    const int rc = run_stages(e, mask, B);
    e->inputs_settled = e->allow_back = false;
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->evRun[slot], e->back_pending ? e->back : e->stream));  // the run ends where its back half ran
    ++e->run_seq;
    return 0;
}

int eftb_sync(eftb_engine* e) {
    if (e) sub_drain(e);
    if (!e) return fail("eftb_sync: null engine");
    join_back(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->comm_stream) HIPCHK(hipStreamSynchronize(e->comm_stream));  // an asynchronous gather may still be in flight
    return check_status(e, "eftb_sync");
}

int eftb_run_timed(eftb_engine* e, int mask, int B, int repeats, float* ms) {
    if (e) sub_drain(e);
    if (!e || !ms) return fail("eftb_run_timed: null argument");
    if (!e->finalized) return fail("eftb_run_timed: engine not finalized");
    if (B < 1 || B > e->c.max_batch) return fail("eftb_run_timed: batch %d outside [1, %d]", B, e->c.max_batch);
    if (repeats < 1) repeats = 1;
    HIPCHK(hipSetDevice(e->c.device));
    join_back(e);
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int r = 0; r < repeats; ++r)
        if (int rc = launch_stages(e, mask, B)) return rc;
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipEventSynchronize(e->ev1));
    HIPCHK(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return 0;
}

// inputs of one batch -> device (engine stream), then the theory stages of reference theory.py:557-585
static int upload_and_launch(eftb_engine* e, const char* who, int B, const double* Pin, const double* f, const double* DA, const double* H,
                             int extra_mask) {
    if (!e->finalized) return fail("%s: engine not finalized", who);
    const eftb_config& c = e->c;
    if (B < 1 || B > c.max_batch) return fail("%s: batch %d outside [1, %d]", who, B, c.max_batch);
    if (c.with_ap && (!DA || !H)) return fail("%s: DA and H are required when with_ap=1", who);
    if (int rc = validate_inputs(c, who, B, Pin, f, DA, H)) return rc;
    HIPCHK(hipSetDevice(c.device));
    join_back(e);
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(e->buf[EFTB_B_PIN], Pin, (size_t)B * c.Nkin * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->buf[EFTB_B_F], f, (size_t)B * sizeof(double), hipMemcpyHostToDevice, st));
    if (c.with_ap) {
        HIPCHK(hipMemcpyAsync(e->buf[EFTB_B_DA], DA, (size_t)B * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->buf[EFTB_B_H], H, (size_t)B * sizeof(double), hipMemcpyHostToDevice, st));
    }
    int mask = EFTB_S_PREP | EFTB_S_LOOPS | EFTB_S_REGROUP | extra_mask;
    if (c.with_resum) mask |= EFTB_S_CF | EFTB_S_RESUM;
    if (c.with_ap) mask |= EFTB_S_AP;
    if (e->pipeline_op >= 0 || !e->tracer_ops.empty()) mask |= EFTB_S_PROJECT;
    return run_stages(e, mask, B);
}

int eftb_eval_batch(eftb_engine* e, int B, const double* Pin, const double* f, const double* DA, const double* H, double* templ,
                    const double* bias, double* plk) {
    if (e) sub_drain(e);
    if (!e || !Pin || !f || (!templ && !plk)) return fail("eftb_eval_batch: null argument");
    if (plk && !bias) return fail("eftb_eval_batch: bias is required when plk is requested");
    if (plk && B >= 1 && B <= e->c.max_batch) {
        HIPCHK(hipSetDevice(e->c.device));
        join_back(e);  // a back half still in flight (reduce / likelihood of an asynchronous eftb_run) reads BIAS
        HIPCHK(hipMemcpyAsync(e->buf[EFTB_B_BIAS], bias, (size_t)B * NROW * sizeof(double), hipMemcpyHostToDevice, e->stream));
    }
    if (int rc = upload_and_launch(e, "eftb_eval_batch", B, Pin, f, DA, H, plk ? EFTB_S_REDUCE : 0)) return rc;
    hipStream_t st = e->stream;
    if (templ)
        HIPCHK(hipMemcpyAsync(templ, e->buf[EFTB_B_TEMPL], (size_t)B * e->cur_nl * NROW * e->cur_nx * sizeof(double), hipMemcpyDeviceToHost, st));
    if (plk) HIPCHK(hipMemcpyAsync(plk, e->buf[EFTB_B_PLK], (size_t)B * e->cur_nl * e->cur_nx * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_status(e, "eftb_eval_batch");
}

int eftb_eval_logp_batch(eftb_engine* e, int B, const double* Pin, const double* f, const double* DA, const double* H, const double* rows,
                         double* logp, double* fullchi2, double* best) {
    if (e) sub_drain(e);
    if (!e || !Pin || !f || !rows || !logp) return fail("eftb_eval_logp_batch: null argument");
    if (!e->finalized) return fail("eftb_eval_logp_batch: engine not finalized");
    if (!e->like_ndata) return fail("eftb_eval_logp_batch: needs eftb_set_likelihood");
    if (B < 1 || B > e->c.max_batch) return fail("eftb_eval_logp_batch: batch %d outside [1, %d]", B, e->c.max_batch);
    HIPCHK(hipSetDevice(e->c.device));
    hipStream_t st = e->stream;
    const int ng1 = e->like_nG + 1;
    join_back(e);  // a back half still in flight reads GROWS
    // rows arrive packed [B][nG+1][24]; the device block is [B][MARG_NG1][24]
    HIPCHK(hipMemcpy2DAsync(e->buf[EFTB_B_GROWS], (size_t)MARG_NG1 * NROW * sizeof(double), rows, (size_t)ng1 * NROW * sizeof(double),
                            (size_t)ng1 * NROW * sizeof(double), B, hipMemcpyHostToDevice, st));
    if (int rc = upload_and_launch(e, "eftb_eval_logp_batch", B, Pin, f, DA, H, EFTB_S_LOGP)) return rc;
    const int nw = B / e->ntr;  // one result per likelihood point
    e->like_host.resize((size_t)nw * MARG_OUT);
    HIPCHK(hipMemcpyAsync(e->like_host.data(), e->buf[EFTB_B_LOGP], (size_t)nw * MARG_OUT * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (int rc = check_status(e, "eftb_eval_logp_batch")) return rc;
    for (int w = 0; w < nw; ++w) {
        const double* o = e->like_host.data() + (size_t)w * MARG_OUT;
        logp[w] = o[0];
        if (fullchi2) fullchi2[w] = o[1];
        if (best)
            for (int i = 0; i < e->like_nG; ++i) best[(size_t)w * e->like_nG + i] = o[2 + i];
    }
    return 0;
}

void* eftb_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail("eftb_host_alloc: hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void eftb_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

static const int kStagedIds[] = {EFTB_B_PIN, EFTB_B_F, EFTB_B_DA, EFTB_B_H, EFTB_B_BIAS, EFTB_B_GROWS, EFTB_B_PLK, EFTB_B_LOGP};

static const int kStagedIn[] = {EFTB_B_PIN, EFTB_B_F, EFTB_B_DA, EFTB_B_H, EFTB_B_BIAS, EFTB_B_GROWS};  // order inside a set's input block
static const int kStagedOut[] = {EFTB_B_PLK, EFTB_B_LOGP};

// Every set keeps its inputs in ONE device block with the layout of the page-locked staging blocks, so that staging is one upload
// kernel (small separate copies are carried out by the host thread once the stream's dependencies have resolved,
// which would stall the sampler loop for a whole step).
static int staged_setup(eftb_engine* e) {
    if (e->cpy) return 0;
    HIPCHK(hipSetDevice(e->c.device));
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (int rc = pad_streams(4, prio_hi)) return rc;
    HIPCHK(hipStreamCreateWithPriority(&e->cpy, hipStreamNonBlocking, prio_hi));
    size_t off = 0;
    for (int id : kStagedIn) {
        e->stage_off[id] = off;
        off += e->buf_elems[id];
    }
    e->stage_elems = (off + 1) & ~(size_t)1;  // whole double2s for the copy kernel
    for (int id : kStagedIds) e->orig[id] = e->buf[id];
    e->staged_plk_device = e->comm != nullptr || !(getenv("EFTB_STAGED_PLK_MAPPED") && atoi(getenv("EFTB_STAGED_PLK_MAPPED")));
    for (int q = 0; q < eftb_engine::NSETS; ++q) {
        HIPCHK(hipMalloc(&e->setblock[q], e->stage_elems * sizeof(double)));
        HIPCHK(hipMemset(e->setblock[q], 0, e->stage_elems * sizeof(double)));
        for (int id : kStagedIn) e->setbuf[q][id] = e->buf_elems[id] ? e->setblock[q] + e->stage_off[id] : nullptr;
        // per-step outputs: ln P (a few KB) lives in page-locked host memory mapped into the device and is written by the kernel itself;
        // P_l stays in device memory and is copied to page-locked host memory by the DMA engine on the stream the step ends on (the back-half
        // stream in pipelined steps: round 1 measured ~70 us of stalled compute queue per step for such a transfer on the COMPUTE stream)
        for (int id : kStagedOut) {
            if (id == EFTB_B_PLK && e->staged_plk_device) {
                HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->setbuf[q][id]), e->buf_elems[id] * sizeof(double)));
                HIPCHK(hipMemset(e->setbuf[q][id], 0, e->buf_elems[id] * sizeof(double)));
                if (!e->comm) {
                    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->plk_host[q]), e->buf_elems[id] * sizeof(double), hipHostMallocMapped));  // (mapped: latency-mode steps write it from the kernel)
                    memset(e->plk_host[q], 0, e->buf_elems[id] * sizeof(double));
                }
                continue;
            }
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->setbuf[q][id]), e->buf_elems[id] * sizeof(double), hipHostMallocMapped));
            memset(e->setbuf[q][id], 0, e->buf_elems[id] * sizeof(double));
        }
        HIPCHK(hipEventCreateWithFlags(&e->evSetDone[q], hipEventDisableTiming));
        HIPCHK(hipEventRecord(e->evSetDone[q], e->cpy));
    }
    // a staging block is sized for one step (step_batch cosmologies), laid out like the device block of a set that holds only that step
    const size_t sb = e->c.step_batch > 0 ? (size_t)e->c.step_batch : (size_t)e->c.max_batch;
    size_t hoff = 0;
    for (int id : kStagedIn) {
        e->slot_off[id] = hoff;
        hoff += e->buf_elems[id] / e->c.max_batch * sb;
    }
    e->slot_elems = (hoff + 1) & ~(size_t)1;
    for (int h = 0; h < eftb_engine::NSLOT; ++h) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->stage_host[h]), e->slot_elems * sizeof(double), hipHostMallocMapped));
        memset(e->stage_host[h], 0, e->slot_elems * sizeof(double));
    }
    for (int r = 0; r < eftb_engine::NLRING; ++r) {
        HIPCHK(hipEventCreateWithFlags(&e->evStagedR[r], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->evStagedAllR[r], hipEventDisableTiming));
        HIPCHK(hipEventRecord(e->evStagedR[r], e->cpy));
        HIPCHK(hipEventRecord(e->evStagedAllR[r], e->cpy));
    }
    if (const char* f = getenv("EFTB_LATENCY_MODE")) e->latency_auto = atoi(f) != 0;
    if (const char* f = getenv("EFTB_DONE_WORDS")) e->done_words = atoi(f) != 0;
    if (const char* f = getenv("EFTB_PLK_DMA")) e->plk_dma = atoi(f) != 0;
    if (const char* f = getenv("EFTB_SUB_STATS")) e->sub_stats = atoi(f) != 0;
    if (const char* f = getenv("EFTB_SUB_INFLIGHT")) e->sub_inflight = std::max(1, std::min(eftb_engine::NSETS - 1, atoi(f)));
    if (const char* f = getenv("EFTB_SUB_LOW")) e->sub_low = std::max(1, atoi(f));
    e->sub_low = std::min(e->sub_low, e->sub_inflight);
    e->coalesce_max = e->comm ? 1 : (int)std::min<size_t>(8, (size_t)e->c.max_batch / sb);   // (with a communicator every step is exchanged by itself)
    if (const char* f = getenv("EFTB_COALESCE")) e->coalesce_max = std::max(1, std::min(e->coalesce_max, atoi(f)));
    {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, 2 * eftb_engine::NSETS * sizeof(unsigned long long), hipHostMallocMapped));
        memset(p, 0, 2 * eftb_engine::NSETS * sizeof(unsigned long long));
        e->set_done = static_cast<volatile unsigned long long*>(p);
    }
    e->cur_set = eftb_engine::NSETS - 1;  // the engine's own buffers are current until the first staged launch; sets 0, 1, 2, 3 follow in turn
    HIPCHK(hipDeviceSynchronize());  // the zero fills ran on the null stream; the copy stream is about to write into these blocks
    return 0;
}

// ---- staged steps: the host part (inputs -> page-locked block `slot`) ...
static void stage_fill(eftb_engine* e, int slot, int B, const double* Pin, const double* f, const double* DA, const double* H, const double* bias,
                       const double* rows) {
    const eftb_config& c = e->c;
    double* h = e->stage_host[slot];
    memcpy(h + e->slot_off[EFTB_B_PIN], Pin, (size_t)B * c.Nkin * sizeof(double));
    memcpy(h + e->slot_off[EFTB_B_F], f, (size_t)B * sizeof(double));
    if (c.with_ap) {
        memcpy(h + e->slot_off[EFTB_B_DA], DA, (size_t)B * sizeof(double));
        memcpy(h + e->slot_off[EFTB_B_H], H, (size_t)B * sizeof(double));
    }
    if (bias) memcpy(h + e->slot_off[EFTB_B_BIAS], bias, (size_t)B * NROW * sizeof(double));
    if (rows) {  // packed [B][nG+1][24] -> device rows of MARG_NG1
        const int ng1 = e->like_nG + 1;
        for (int w = 0; w < B; ++w)
            memcpy(h + e->slot_off[EFTB_B_GROWS] + (size_t)w * MARG_NG1 * NROW, rows + (size_t)w * ng1 * NROW, (size_t)ng1 * NROW * sizeof(double));
    }
}

// ... and the device part (issued by the caller's thread or by the submission thread): n queued steps become ONE launch on the next device set --
// their staging blocks are gathered into the set's input block row after row by one upload kernel, then every stage runs on the joint batch
static int issue_group(eftb_engine* e, const eftb_engine::SubCmd* cmds, int n, bool lat) {
    const eftb_config& c = e->c;
    const int q = (e->cur_set + 1) % eftb_engine::NSETS;
    const unsigned long long L = e->launch_seq;
    const int lr = (int)(L % eftb_engine::NLRING);
    const int mask = cmds[0].mask;
    const bool has_rows = cmds[0].has_rows != 0;
    int Bt = 0;
    for (int j = 0; j < n; ++j) Bt += cmds[j].B;
    // the upload goes where the launch's first kernels go: direct-P_l launches start on the side stream (their front), so the upload in front of
    // them on the same queue needs no cross-queue hand-over (20-35 us each on this runtime, and the chain of a launch has four more)
    hipStream_t cs = (e->plk_direct && !lat && e->upload_on_side) ? e->side : e->cpy;
    HIPCHK(hipStreamWaitEvent(cs, e->evSetDone[q], 0));  // the last launch on this set (and the fetch of its results) is over
    // (an upload kernel reading the mapped staging blocks, not a DMA transfer: 0.4 MB is latency, and the DMA form measured the same or worse)
    // only what was staged travels: P_lin, f, DA, H, the bias rows -- and the likelihood rows (0.6 MB at 128 walkers) when there are any.  Reads
    // of host memory from a kernel run at ~16 GB/s: the whole 0.85 MB block took 48-53 us, on the critical path of a dependent sampler
    const size_t width[6] = {(size_t)c.Nkin, 1, 1, 1, (size_t)NROW, (size_t)MARG_NG1 * NROW};
    auto gather = [&](int id_lo, int id_hi, hipStream_t st) {  // arrays kStagedIn[id_lo .. id_hi) of every step of the group
        StageSegs sg{};
        unsigned total = 0;
        int row = 0;
        for (int j = 0; j < n; ++j) {
            for (int a = id_lo; a < id_hi; ++a) {
                const int id = kStagedIn[a];
                if (!e->buf_elems[id] || (id == EFTB_B_GROWS && !has_rows)) continue;
                StageSeg& g = sg.s[sg.n++];
                g.src = e->stage_host[cmds[j].slot] + e->slot_off[id];
                g.dst = e->setblock[q] + e->stage_off[id] + (size_t)row * width[a];
                g.n = (unsigned)((size_t)cmds[j].B * width[a]);
                g.first = total;
                total += (g.n + 63u) & ~63u;
            }
            row += cmds[j].B;
        }
        if (sg.n && !WHATIF_SKIP(1024)) hipLaunchKernelGGL(stage_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, st, sg);
    };
    e->trace_slot = -1;
    if (e->trace_on && !lat && e->evTraceBase) {
        e->trace_slot = (int)(e->trace_n++ % eftb_engine::NTRACE);
        e->trace_B[e->trace_slot] = Bt;
        e->trace_launch[e->trace_slot] = L;
        trace_point(e, 0, cs);
    }
    e->set_latency[q] = lat;
    if (lat) {
        // the small arrays go first and on the compute queue itself (in line in front of the step's kernels: no cross-queue hand-over; the
        // GPU is idle, nothing of an earlier step can still be reading the set): they are all the step's first kernels wait for -- its first
        // kernel reads P_lin from the staging block itself, whose device copy follows on the copy queue
        gather(1, 6, e->stream);
        HIPCHK(hipEventRecord(e->evStagedR[lr], e->stream));
        gather(0, 1, cs);
        HIPCHK(hipEventRecord(e->evStagedAllR[lr], cs));
    } else {
        gather(0, 6, cs);
        trace_point(e, 9, cs);
        HIPCHK(hipEventRecord(e->evStagedR[lr], cs));
    }
    // the set becomes current and the launch goes out
    e->cur_set = q;
    e->lring_set[lr] = q;
    ++e->launch_seq;
    int row = 0;
    for (int j = 0; j < n; ++j) {  // where each step's rows will be (the fetch needs it even when the launch fails below: to say so)
        eftb_engine::StepRec& r = e->rec[cmds[j].step % eftb_engine::SUBREC];
        r.set = q; r.row = row; r.B = cmds[j].B; r.launch = L; r.rc = 0; r.out = cmds[j].out;
        row += cmds[j].B;
    }
    for (int id : kStagedIds)
        if (e->setbuf[q][id]) e->buf[id] = e->setbuf[q][id];
    e->status_slot = q;  // this launch's kernels raise this set's flags (cleared here: whatever an abandoned step left is void)
    e->status[2 * q] = e->status[2 * q + 1] = 0;
    ++e->epoch;  // (captured graphs hold the other set's pointers)
    const hipStream_t staged_on = lat ? e->stream : cs;   // (where evStagedR was recorded: that stream is behind it already -- a call saved per launch)
    if (staged_on != e->stream) HIPCHK(hipStreamWaitEvent(e->stream, e->evStagedR[lr], 0));  // the side stream forks from here, so it inherits the wait
    if (staged_on != e->pre) HIPCHK(hipStreamWaitEvent(e->pre, e->evStagedR[lr], 0));
    if (staged_on != e->side) HIPCHK(hipStreamWaitEvent(e->side, e->evStagedR[lr], 0));
    // latency mode: one queue for the whole step; P_lin is read from the page-locked staging block (its device copy arrives behind evStagedAllR);
    // P_l goes to mapped host memory from the kernel that forms it (REDUCE, or the AP epilogue) unless the NNLO pass adds to it afterwards
    const bool plk_direct = (lat || (e->plk_direct && !e->comm)) && e->plk_host[q] && (mask & EFTB_S_REDUCE) && !c.with_nnlo;
    // ... from the kernel that forms it in latency mode (one kernel less on the dependent sampler's chain); in pipelined steps by a small copy
    // kernel behind the step (round 3 stored from ap_plk_kernel: 1.6 MB of 8-byte PCIe stores kept every SIMD's waves resident for ~25 us)
    const bool plk_tail = plk_direct && !lat && !e->ap_plk_nodes;
    double* pin_dev = e->buf[EFTB_B_PIN];
    if (lat) e->buf[EFTB_B_PIN] = e->stage_host[cmds[0].slot] + e->slot_off[EFTB_B_PIN];
    e->lat_run = lat;
    // steps with a destination of their own (eftb_set_step_output): P_l always leaves through the copy-out below, one transfer per destination
    bool any_out = false;
    for (int j = 0; j < n; ++j) any_out = any_out || cmds[j].out != nullptr;
    e->plk_host_out = plk_direct && !plk_tail && !any_out ? e->plk_host[q] : nullptr;
    e->inputs_settled = e->allow_back = !lat;
    const int rc = run_stages(e, mask, Bt);
    e->inputs_settled = e->allow_back = false;
    e->lat_run = false;
    e->plk_host_out = nullptr;
    e->buf[EFTB_B_PIN] = pin_dev;
    e->status_slot = eftb_engine::NSETS;
    if (rc) { e->trace_slot = -1; return rc; }
    for (int j = 0; j < n; ++j) {  // shape of the output rows (after the pipeline operator, if the mask has one)
        eftb_engine::StepRec& r = e->rec[cmds[j].step % eftb_engine::SUBREC];
        r.nl = e->cur_nl; r.nx = e->cur_nx;
    }
    hipStream_t last = e->back_pending ? e->back : e->stream;  // the launch ends where its back half ran
    if (lat) HIPCHK(hipStreamWaitEvent(last, e->evStagedAllR[lr], 0));  // (the set is not "done" before its own upload is)
    constexpr int copy_wgs = 48;   // (8 ... 512 workgroups measured the same: the kernel's time is the PCIe transfer)
    hipStream_t done_on = last;  // where the launch's LAST work runs: what the set's event and completion word follow
    if (e->done_words && !lat) {   // compute finished: what the submission thread's flow control counts (the copy-out below is not part of it)
        if (hipStreamWriteValue64(last, const_cast<unsigned long long*>(e->set_done) + eftb_engine::NSETS + q, L + 1, 0) == hipSuccess) e->set_cword[q] = L + 1;
        else (void)hipGetLastError();
    }
    // P_l [rows of the launch][nl][nx] to host memory: the set's own page-locked block, or -- per step -- the caller's page-locked array; steps next to
    // each other that go to the set's block travel as one transfer
    auto copy_out = [&](hipStream_t st) -> int {
        const size_t per = (size_t)e->cur_nl * e->cur_nx;
        // the DMA engine, in line behind the launch's last kernel (same-box A/B over 200 steps: 0.091-0.096 ms per step against 0.102-0.104 with
        // copy16_kernel -- the what-if runs priced the kernel's PCIe stores at a fifth of the step; EFTB_PLK_DMA=0 brings the kernel back)
        const bool dma = e->plk_dma || !plk_direct;
        if (WHATIF_SKIP(512)) return 0;   // (what-if: P_l stays on the device)
        if (WHATIF_SKIP(2048) && e->PLK0) {   // (what-if: the copy kernel with a device destination -- the stream holds it, PCIe does not)
            hipLaunchKernelGGL(copy16_kernel, dim3(copy_wgs), dim3(256), 0, st, e->buf[EFTB_B_PLK], e->PLK0, (size_t)Bt * per);
            return 0;
        }
        for (int j = 0; j < n; ++j)
            if (cmds[j].out && cmds[j].out_count < (size_t)cmds[j].B * per)
                return fail("eftb_run_staged: the step's P_l has %zu elements, the destination of eftb_set_step_output holds %zu", (size_t)cmds[j].B * per, cmds[j].out_count);
        int row = 0;
        for (int j = 0; j < n;) {
            int j1 = j + 1, rows = cmds[j].B;
            if (!cmds[j].out)
                for (; j1 < n && !cmds[j1].out; ++j1) rows += cmds[j1].B;
            else   // destinations that follow each other in memory (the slices of one results array) travel as one transfer too
                for (; j1 < n && cmds[j1].out == cmds[j].out + (size_t)rows * per; ++j1) rows += cmds[j1].B;
            const double* src = e->buf[EFTB_B_PLK] + (size_t)row * per;
            double* dst = cmds[j].out ? cmds[j].out : e->plk_host[q] + (size_t)row * per;
            if (dma) HIPCHK(hipMemcpyAsync(dst, src, (size_t)rows * per * sizeof(double), hipMemcpyDeviceToHost, st));
            else hipLaunchKernelGGL(copy16_kernel, dim3(copy_wgs), dim3(256), 0, st, src, dst, (size_t)rows * per);
            row += rows;
            j = j1;
        }
        return 0;
    };
    const bool kernel_stored = plk_direct && !plk_tail && !any_out && e->plk_host_written;  // the kernel that formed P_l wrote it to the set's host block too
    if (plk_direct && !kernel_stored) {
        if (int crc = copy_out(done_on)) return crc;
    }
    if (e->plk_host[q] && (mask & EFTB_S_REDUCE) && !plk_direct) {
        if (int crc = copy_out(last)) return crc;
    }
    trace_point(e, 8, done_on);
    e->trace_slot = -1;
    HIPCHK(hipEventRecord(e->evSetDone[q], done_on));
    if (e->back_pending && e->plk_host[q] && (mask & EFTB_S_REDUCE) && !plk_direct)
        HIPCHK(hipEventRecord(e->evBack[(e->back_step + 1) & 1], last));  // whoever joins the back half also waits for the copy
    if (e->done_words) {
        if (hipStreamWriteValue64(done_on, const_cast<unsigned long long*>(e->set_done) + q, L + 1, 0) == hipSuccess) e->set_word[q] = L + 1;
        else {
            (void)hipGetLastError();
            e->done_words = false;  // this runtime refuses the command: event queries from here on (the words of the earlier launches stay valid)
        }
    }
    return 0;
}

// issue_group + bookkeeping shared by both issuing threads: the steps' records, the clocks, the publication of "issued"
static int issue_and_publish(eftb_engine* e, const eftb_engine::SubCmd* cmds, int n, bool lat, bool by_caller) {
    const auto ti0 = std::chrono::steady_clock::now();
    const int q_before = e->cur_set;
    const unsigned long long L_before = e->launch_seq;
    const int rc = issue_group(e, cmds, n, lat);
    if (rc && e->launch_seq == L_before) {  // failed before the set was taken: the steps still need records that say so
        for (int j = 0; j < n; ++j) {
            eftb_engine::StepRec& r = e->rec[cmds[j].step % eftb_engine::SUBREC];
            r.set = q_before; r.row = 0; r.B = cmds[j].B; r.launch = ~0ull; r.out = cmds[j].out;
        }
    }
    for (int j = 0; j < n; ++j) {
        const int ri = (int)(cmds[j].step % eftb_engine::SUBREC);
        e->rec[ri].rc = rc;
        if (rc) snprintf(e->sub_err[ri], sizeof e->sub_err[ri], "%s", g_err.c_str());
    }
    if (e->sub_stats) {
        e->issue_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - ti0).count();
        e->issue_n += n;
        ++e->launch_n;
        if (by_caller) e->inline_n += n;
    }
    e->steps_launched.store(cmds[n - 1].step + 1, std::memory_order_release);
    return rc;
}

static inline void cpu_pause() { __builtin_ia32_pause(); }

// has staged launch L finished?  (issuing thread; its completion word if one was enqueued, else its set's event)
static bool launch_finished(eftb_engine* e, unsigned long long L) {
    const int q = e->lring_set[L % eftb_engine::NLRING];
    if (e->set_cword[q] >= L + 1) return e->set_done[eftb_engine::NSETS + q] >= L + 1;   // its compute has finished (the copy-out of P_l may still run)
    if (e->set_word[q] >= L + 1) return e->set_done[q] >= L + 1;   // (words grow with the launch number: a later launch on the set says it too)
    return hipEventQuery(e->evSetDone[q]) == hipSuccess;
}

// ---- submission thread (see eftb_engine::SubCmd)
static void sub_main(eftb_engine* e) {
    (void)hipSetDevice(e->c.device);
    unsigned long long head = e->sub_head.load(std::memory_order_relaxed);
    for (;;) {
        // wait for a command: spin for a while (a pipelined loop delivers one every ~0.1 ms), then sleep
        unsigned spins = 0;
        while (e->sub_tail.load(std::memory_order_acquire) == head) {
            if (e->sub_stop.load(std::memory_order_acquire)) return;
            if (++spins < 200000) { cpu_pause(); continue; }   // ~ a few ms
            std::unique_lock<std::mutex> lk(e->sub_mx);
            e->sub_sleeping.store(true, std::memory_order_seq_cst);
            e->sub_cv.wait(lk, [&] { return e->sub_tail.load(std::memory_order_acquire) != head || e->sub_stop.load(std::memory_order_acquire); });
            e->sub_sleeping.store(false, std::memory_order_seq_cst);
            spins = 0;
        }
        while (e->sub_hold.load(std::memory_order_acquire) && !e->sub_stop.load(std::memory_order_acquire)) cpu_pause();
        // flow control: with sub_inflight launches still running the queue is left to grow -- what has piled up by the time one of them finishes
        // leaves as one launch (coalesce_max steps at most)
        while (e->launch_done < e->launch_seq && launch_finished(e, e->launch_done)) ++e->launch_done;
        {
            static const double limit_s = getenv("EFTB_FETCH_TIMEOUT_S") ? atof(getenv("EFTB_FETCH_TIMEOUT_S")) : 60.0;
            const auto tw0 = std::chrono::steady_clock::now();
            for (unsigned wspins = 0;;) {
                const long long inflight = (long long)(e->launch_seq - e->launch_done);
                // go: below the in-flight limit AND (a full group is waiting, or the GPU is about to run dry: fewer than sub_low launches left)
                if (inflight < e->sub_inflight &&
                    (inflight < e->sub_low || (long long)(e->sub_tail.load(std::memory_order_acquire) - head) >= e->coalesce_max ||
                     e->sub_flush.load(std::memory_order_acquire))) break;
                if (inflight > 0 && launch_finished(e, e->launch_done)) { ++e->launch_done; continue; }
                for (int i = 0; i < 16; ++i) cpu_pause();
                // (a launch that never finishes must not hold the queue -- and with it every entry point that drains it -- for ever: the steps go
                // out anyway and their fetches report the time-out)
                if ((++wspins & 0xffff) == 0 && (e->sub_stop.load(std::memory_order_acquire) ||
                                                 std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count() > limit_s)) break;
            }
        }
        eftb_engine::SubCmd grp[8];
        int n = 0, Bt = 0;
        const unsigned long long tail = e->sub_tail.load(std::memory_order_acquire);
        while (head + n < tail && n < e->coalesce_max) {
            const eftb_engine::SubCmd& c = e->sub_ring[(head + n) % eftb_engine::SUBQ];
            if (n && (c.mask != grp[0].mask || c.has_rows != grp[0].has_rows || Bt + c.B > e->c.max_batch || e->check_finite)) break;
            grp[n++] = c;
            Bt += c.B;
        }
        (void)issue_and_publish(e, grp, n, false, false);
        head += n;
        if (head == e->sub_tail.load(std::memory_order_acquire)) e->sub_flush.store(false, std::memory_order_release);  // (a step queued from here on waits for its group again)
        e->sub_head.store(head, std::memory_order_release);
    }
}

// nothing queued and the submission thread between commands: the engine's state belongs to the caller's thread
static inline bool sub_quiescent(const eftb_engine* e) {
    return e->sub_head.load(std::memory_order_acquire) == e->sub_tail.load(std::memory_order_relaxed);
}

static inline void sub_drain(eftb_engine* e) {
    if (!e->sub_started) return;
    e->sub_hold.store(false, std::memory_order_release);   // (an entry point that needs the queue empty ends a test's hold)
    while (!sub_quiescent(e)) cpu_pause();
}

static void sub_stop_thread(eftb_engine* e) {
    if (!e->sub_started) return;
    sub_drain(e);
    {
        std::lock_guard<std::mutex> lk(e->sub_mx);
        e->sub_stop.store(true, std::memory_order_release);
    }
    e->sub_cv.notify_all();
    if (e->sub_thread.joinable()) e->sub_thread.join();
    e->sub_started = false;
}

// waits until the launch of staged step `step` has been issued, and reports its launch error (if any) as this call's
static int sub_wait_launched(eftb_engine* e, unsigned long long step, const char* who) {
    static const double limit_s = getenv("EFTB_FETCH_TIMEOUT_S") ? atof(getenv("EFTB_FETCH_TIMEOUT_S")) : 60.0;
    if (e->steps_launched.load(std::memory_order_acquire) <= step) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; e->steps_launched.load(std::memory_order_acquire) <= step; ++spins) {
            cpu_pause();
            if ((spins & 0xffff) == 0xffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
                return fail("%s: the submission thread did not issue the step within %.0f s (EFTB_FETCH_TIMEOUT_S)", who, limit_s);
        }
    }
    const int r = (int)(step % eftb_engine::SUBREC);
    if (e->rec[r].rc) return fail("%s: the launch of this step failed: %s", who, e->sub_err[r]);
    return 0;
}

int eftb_stage_inputs(eftb_engine* e, int B, const double* Pin, const double* f, const double* DA, const double* H, const double* bias,
                      const double* rows) {
    if (!e || !Pin || !f) {
        if (e) e->stg_out = nullptr;
        return fail("eftb_stage_inputs: null argument");
    }
    struct OutGuard {   // a step that is refused here takes the destination named for it (eftb_set_step_output) with it: it must not pass to the next step
        eftb_engine* e;
        bool staged = false;
        ~OutGuard() { if (!staged) e->stg_out = nullptr; }
    } guard{e};
    if (!e->finalized) return fail("eftb_stage_inputs: engine not finalized");
    const eftb_config& c = e->c;
    const int step_max = c.step_batch > 0 ? c.step_batch : c.max_batch;
    if (B < 1 || B > step_max) return fail("eftb_stage_inputs: batch %d outside [1, %d]", B, step_max);
    if (c.with_ap && (!DA || !H)) return fail("eftb_stage_inputs: DA and H are required when with_ap=1");
    if (rows && !e->like_ndata) return fail("eftb_stage_inputs: rows need eftb_set_likelihood");
    if (int rc = validate_inputs(c, "eftb_stage_inputs", B, Pin, f, DA, H)) return rc;
    if (!e->cpy) {
        sub_drain(e);
        if (int rc = staged_setup(e)) return rc;
    }
    HIPCHK(hipSetDevice(c.device));
    const int slot = (int)(e->steps_submitted % eftb_engine::NSLOT);  // the oldest staging block: its step was fetched (or abandoned) long ago
    // the step that used this block last has been issued (its events exist) ...
    if (e->slot_step[slot]) {
        (void)sub_wait_launched(e, e->slot_step[slot] - 1, "eftb_stage_inputs");  // (a failed launch left nothing to wait for; its fetch reports it)
        const eftb_engine::StepRec& r = e->rec[(e->slot_step[slot] - 1) % eftb_engine::SUBREC];
        // (a launch whose completion word has been written is over, its upload with it: no runtime call -- hipEventSynchronize costs this thread 6 us per step)
        const bool over = r.launch != ~0ull && e->set_word[r.set] >= r.launch + 1 && e->set_done[r.set] >= r.launch + 1;
        if (r.launch != ~0ull && !over) {
            const int lr = (int)(r.launch % eftb_engine::NLRING);
            // ... and the block is free again (the upload from it has finished; a latency-mode step's first kernel read P_lin from the block itself:
            // that step must be over too -- it almost always is)
            HIPCHK(hipEventSynchronize(e->slot_latency[slot] ? e->evStagedAllR[lr] : e->evStagedR[lr]));
            if (e->slot_latency[slot])
                if (int rc = spin_event(e->evSetDone[r.set], "eftb_stage_inputs", "the latency-mode step that read this staging block")) return rc;
        }
    }
    // Who issues the step?  With earlier steps still queued or in flight, the submission thread (this call only fills the staging block).  With
    // the engine quiescent and the GPU idle -- a sampler whose next step depends on this step's result -- this thread, at once, as a latency-mode
    // step if that is enabled (see latency_auto)
    const bool quiet = sub_quiescent(e);
    const bool gpu_idle = quiet && hipEventQuery(e->evSetDone[e->cur_set]) == hipSuccess;
    // (with latency mode off the caller has said that more steps follow at once: queued from the first step on -- the caller goes on staging
    // while the thread issues; 2-3 % of a 20-step loop)
    e->stg_inline = !(e->sub_mode == 2 || (e->sub_mode == 1 && (!gpu_idle || !e->latency_auto)));
    // nothing in flight (the step launched last has finished, or none was launched): the step staged here has the GPU to itself -- see latency_auto
    e->stg_lat = e->stg_inline && e->latency_auto && gpu_idle && e->slot_off[EFTB_B_PIN] == 0;
    e->slot_latency[slot] = e->stg_lat;
    if (e->stg_lat) {
        sub_drain(e);
        hipLaunchKernelGGL(wake_kernel, dim3(1), dim3(64), 0, e->stream);  // (see wake_kernel: its start-up runs under the host copies below)
    }
    const auto tf0 = std::chrono::steady_clock::now();
    stage_fill(e, slot, B, Pin, f, DA, H, bias, rows);
    if (e->sub_stats) e->fill_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tf0).count();
    e->stg_slot = slot; e->stg_B = B; e->stg_rows = rows ? 1 : 0;
    guard.staged = true;
    return 0;
}

int eftb_flush(eftb_engine* e) {
    if (!e) return fail("eftb_flush: null engine");
    if (e->sub_started && !sub_quiescent(e)) e->sub_flush.store(true, std::memory_order_release);
    return 0;
}

int eftb_set_step_output(eftb_engine* e, double* dst, size_t count) {
    if (!e) return fail("eftb_set_step_output: null engine");
    if (!dst) { e->stg_out = nullptr; e->stg_out_count = 0; return 0; }
    if (e->comm) return fail("eftb_set_step_output: P_l of an engine with a communicator stays in device memory for the exchange (eftb_gathered_view)");
    HIPCHK(hipSetDevice(e->c.device));
    // the DMA engine writes the array behind the step: it must be page-locked (eftb_host_alloc, hipHostMalloc, hipHostRegister) -- pageable memory would
    // turn every transfer into a synchronous staged copy on the issuing thread
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, dst) != hipSuccess || at.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail("eftb_set_step_output: the destination is not page-locked host memory (allocate it with eftb_host_alloc / Engine.pinned_empty)");
    }
    e->stg_out = dst;
    e->stg_out_count = count;
    return 0;
}

int eftb_run_staged(eftb_engine* e, int mask, int B) {
    if (!e) return fail("eftb_run_staged: null engine");
    if (!e->cpy || e->stg_slot < 0) return fail("eftb_run_staged: nothing staged (eftb_stage_inputs first)");
    if (B != e->stg_B) return fail("eftb_run_staged: batch %d, but %d cosmologies were staged", B, e->stg_B);
    HIPCHK(hipSetDevice(e->c.device));
    const unsigned long long step = e->steps_submitted;
    double* out = e->stg_out;
    e->stg_out = nullptr;   // (one step only, launched or not)
    if (out) {
        if (!(mask & EFTB_S_REDUCE) || (mask & EFTB_S_LOGP)) return fail("eftb_run_staged: eftb_set_step_output names a destination for P_l, but this run forms none (mask 0x%x)", mask);
        if (!e->plk_host[0]) return fail("eftb_run_staged: eftb_set_step_output needs an engine whose P_l leaves through the copy-out (no communicator, no EFTB_STAGED_PLK_MAPPED)");
    }
    const eftb_engine::SubCmd cmd{e->stg_slot, mask, B, e->stg_rows, step, out, e->stg_out_count};
    if (!e->stg_inline) {  // queued: the submission thread uploads the staging block and launches the step (with whatever else is queued by then)
        if (!e->sub_started) {
            e->sub_stop.store(false);
            e->sub_thread = std::thread(sub_main, e);
            e->sub_started = true;
        }
        // (never more than NSLOT - 1 pending: each owns a staging block, and eftb_stage_inputs has waited for the block's previous user)
        const unsigned long long tail = e->sub_tail.load(std::memory_order_relaxed);
        e->sub_ring[tail % eftb_engine::SUBQ] = cmd;
        e->sub_tail.store(tail + 1, std::memory_order_release);
        if (e->sub_sleeping.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(e->sub_mx);
            e->sub_cv.notify_one();
        }
        e->slot_step[e->stg_slot] = step + 1;
        e->stg_slot = -1;
        ++e->steps_submitted;
        return 0;
    }
    sub_drain(e);
    // (launch bookkeeping of the issuing thread: bring "finished" up to date, as the submission thread does before its launches)
    while (e->launch_done < e->launch_seq && launch_finished(e, e->launch_done)) ++e->launch_done;
    const int rc = issue_and_publish(e, &cmd, 1, e->stg_lat, true);
    e->slot_step[e->stg_slot] = step + 1;
    e->stg_slot = -1;
    ++e->steps_submitted;
    return rc;
}

// the launch that carried a step has finished (its completion word, or its set's event)
static int wait_step_done(eftb_engine* e, const eftb_engine::StepRec& r, const char* who) {
    const auto tw0 = std::chrono::steady_clock::now();
    int rc = 0;
    const int t = r.set;
    if (e->set_word[t] >= r.launch + 1) {   // (else: no completion write was enqueued for this launch -- EFTB_DONE_WORDS=0, or the runtime refused it)
        static const double limit_s = getenv("EFTB_FETCH_TIMEOUT_S") ? atof(getenv("EFTB_FETCH_TIMEOUT_S")) : 60.0;
        for (unsigned spins = 0; e->set_done[t] < r.launch + 1; ++spins) {
            cpu_pause();
            if ((spins & 0xfffff) == 0xfffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count() > limit_s) {
                rc = fail("%s: the step did not finish within %.0f s (EFTB_FETCH_TIMEOUT_S)", who, limit_s);
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        rc = spin_event(e->evSetDone[t], who, "the step");
    }
    if (e->sub_stats) e->wait_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tw0).count();
    return rc;
}

// record of the staged step `back` steps before the one submitted last, once its launch has been issued
static int staged_step(eftb_engine* e, const char* who, int back, const eftb_engine::StepRec** rec) {
    if (back < 0 || back >= eftb_engine::NSETS) return fail("%s: back must be 0 (the step launched last) ... %d (that many steps before it)", who, eftb_engine::NSETS - 1);
    if (!e->cpy) return fail("%s: no staged run yet", who);
    if ((unsigned long long)back >= e->steps_submitted)
        return fail("%s: back = %d, but only %llu staged step(s) have been launched", who, back, e->steps_submitted);
    const unsigned long long step = e->steps_submitted - 1 - back;
    if (int rc = sub_wait_launched(e, step, who)) return rc;
    *rec = &e->rec[step % eftb_engine::SUBREC];
    return 0;
}

// where a step's rows of a per-set output block start: P_l [row][nl][nx]; ln P per likelihood point [row / ntr][MARG_OUT]
static inline size_t step_offset(const eftb_engine* e, const eftb_engine::StepRec& r, int id) {
    return id == EFTB_B_PLK ? (size_t)r.row * r.nl * r.nx : (size_t)(r.row / e->ntr) * MARG_OUT;
}

int eftb_fetch_back(eftb_engine* e, int back, int id, double* host, size_t count) {
    if (!e || !host) return fail("eftb_fetch_back: null argument");
    if (id != EFTB_B_PLK && id != EFTB_B_LOGP) return fail("eftb_fetch_back: only EFTB_B_PLK and EFTB_B_LOGP are per-set outputs");
    if (count > e->buf_elems[id]) return fail("eftb_fetch_back: buffer %d holds %zu elements, asked %zu", id, e->buf_elems[id], count);
    const eftb_engine::StepRec* r = nullptr;
    if (int rc = staged_step(e, "eftb_fetch_back", back, &r)) return rc;
    HIPCHK(hipSetDevice(e->c.device));
    const int t = r->set;
    if (int rc = wait_step_done(e, *r, "eftb_fetch_back")) return rc;
    const size_t off = step_offset(e, *r, id);
    if (off + count > e->buf_elems[id]) return fail("eftb_fetch_back: the step's rows start at element %zu of %zu, asked %zu", off, e->buf_elems[id], count);
    if (id == EFTB_B_PLK && r->out) {
        if (count > (size_t)r->B * r->nl * r->nx) return fail("eftb_fetch_back: the step's P_l has %zu elements, asked %zu", (size_t)r->B * r->nl * r->nx, count);
        if (host != r->out) memcpy(host, r->out, count * sizeof(double));  // (it went to the caller's own array: eftb_set_step_output)
    } else if (id == EFTB_B_PLK && e->plk_host[t])
        memcpy(host, e->plk_host[t] + off, count * sizeof(double));  // copied out behind the step
    else if (e->staged_plk_device && id == EFTB_B_PLK)  // multi-GPU runs keep P_l on the device for the RCCL exchange
        HIPCHK(hipMemcpy(host, e->setbuf[t][id] + off, count * sizeof(double), hipMemcpyDeviceToHost));
    else
        memcpy(host, e->setbuf[t][id] + off, count * sizeof(double));  // the results are already in (mapped) host memory
    return check_status(e, "eftb_fetch_back", t);  // this launch's own flags only: the steps queued behind it report with their own fetch
}

int eftb_fetch_previous(eftb_engine* e, int id, double* host, size_t count) { return eftb_fetch_back(e, 1, id, host, count); }

int eftb_fetch_view(eftb_engine* e, int back, int id, const double** block, size_t* count) {
    if (!e || !block) return fail("eftb_fetch_view: null argument");
    if (id != EFTB_B_PLK && id != EFTB_B_LOGP) return fail("eftb_fetch_view: only EFTB_B_PLK and EFTB_B_LOGP are per-set outputs");
    const eftb_engine::StepRec* r = nullptr;
    if (int rc = staged_step(e, "eftb_fetch_view", back, &r)) return rc;
    HIPCHK(hipSetDevice(e->c.device));
    const int t = r->set;
    const double* p = id == EFTB_B_PLK ? (e->plk_host[t] ? e->plk_host[t] : (e->staged_plk_device ? nullptr : e->setbuf[t][id])) : e->setbuf[t][id];
    if (!p) return fail("eftb_fetch_view: P_l of this engine stays in device memory for the RCCL exchange (eftb_gathered_view hands out the gathered block)");
    if (int rc = wait_step_done(e, *r, "eftb_fetch_view")) return rc;
    const size_t off = step_offset(e, *r, id);
    if (id == EFTB_B_PLK && r->out) {   // the step's P_l went to the caller's own array (eftb_set_step_output): that is the block
        *block = r->out;
        if (count) *count = (size_t)r->B * r->nl * r->nx;
        return check_status(e, "eftb_fetch_view", t);
    }
    *block = p + off;
    if (count) *count = e->buf_elems[id] - off;
    return check_status(e, "eftb_fetch_view", t);
}

int eftb_submit_stats(eftb_engine* e, int enable, int reset, double out[6]) {
    if (!e) return fail("eftb_submit_stats: null engine");
    sub_drain(e);
    if (out) {
        out[0] = (double)e->issue_n; out[1] = (double)e->inline_n; out[2] = e->issue_ns * 1e-3; out[3] = e->fill_ns * 1e-3; out[4] = e->wait_ns * 1e-3;
        out[5] = (double)e->launch_n;
    }
    if (reset) { e->issue_n = e->inline_n = e->launch_n = 0; e->issue_ns = e->fill_ns = e->wait_ns = 0.0; }
    e->sub_stats = enable != 0;
    return 0;
}

int eftb_step_trace(eftb_engine* e, double* out, int max_launches, int* n_out) {
    if (!e || !out || !n_out) return fail("eftb_step_trace: null argument");
    sub_drain(e);
    HIPCHK(hipSetDevice(e->c.device));
    if (!e->evTraceBase) return fail("eftb_step_trace: EFTB_O_STEP_TRACE was never switched on");
    HIPCHK(sync_all(e));
    const unsigned long long total = e->trace_n, first = total > (unsigned long long)eftb_engine::NTRACE ? total - eftb_engine::NTRACE : 0;
    int n = 0;
    for (unsigned long long i = first; i < total && n < max_launches; ++i) {
        const int sl = (int)(i % eftb_engine::NTRACE);
        double* row = out + (size_t)n * (eftb_engine::NTP + 2);
        row[0] = (double)e->trace_launch[sl];
        row[1] = (double)e->trace_B[sl];
        bool ok = true;
        for (int p = 0; p < eftb_engine::NTP; ++p) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e->evTraceBase, e->evTrace[sl][p]) != hipSuccess) { (void)hipGetLastError(); ok = false; ms = -1.f; }
            row[2 + p] = ms * 1e3;
        }
        (void)ok;
        ++n;
    }
    *n_out = n;
    return 0;
}

int eftb_step(eftb_engine* e, int mask, int B, const double* Pin, const double* f, const double* DA, const double* H, const double* bias,
              const double* rows, int back, int id, const double** block, size_t* count) {
    if (int rc = eftb_stage_inputs(e, B, Pin, f, DA, H, bias, rows)) return rc;
    if (int rc = eftb_run_staged(e, mask, B)) return rc;
    if (block) *block = nullptr;
    if (back < 0 || !block || (unsigned long long)back >= e->steps_submitted) return 0;  // (the pipeline is still filling: nothing to hand out yet)
    return eftb_fetch_view(e, back, id, block, count);
}

int eftb_comm_unique_id(char id[128]) {
    if (!id) return fail("eftb_comm_unique_id: null argument");
    if (int rc = rccl_load()) return rc;
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return 0;
}

int eftb_comm_init(eftb_engine* e, int nranks, int rank, const char id[128]) {
    if (e) sub_drain(e);
    if (!e || !id) return fail("eftb_comm_init: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("eftb_comm_init: bad rank %d of %d", rank, nranks);
    if (e->cpy) return fail("eftb_comm_init: call it before the first eftb_stage_inputs (the staged sets place P_l where the exchange can read it)");
    if (int rc = rccl_load()) return rc;
    HIPCHK(hipSetDevice(e->c.device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    NCCLCHK(g_rccl.CommInitRank(&e->comm, nranks, u, rank));
    e->nranks = nranks;
    e->rank = rank;
    return 0;
}

int eftb_gather_plk(eftb_engine* e, int B, int root, double* host_out) {
    if (e) sub_drain(e);
    if (!e) return fail("eftb_gather_plk: null engine");
    if (B < 1 || B > e->c.max_batch) return fail("eftb_gather_plk: batch %d outside [1, %d]", B, e->c.max_batch);
    if (root < 0 || root >= e->nranks) return fail("eftb_gather_plk: bad root %d", root);
    HIPCHK(hipSetDevice(e->c.device));
    const size_t count = (size_t)B * e->cur_nl * e->cur_nx;
    if (e->rank == root) {
        e->gather_slot = (e->gather_slot + 1) % eftb_engine::NSETS;
        const int q = e->gather_slot;
        if (!e->gathered2[q]) {
            const size_t bytes = (size_t)e->nranks * e->c.max_batch * e->c.Nl * e->c.Nk * sizeof(double);
            HIPCHK(hipMalloc(&e->gathered2[q], bytes));
            HIPCHK(hipEventCreateWithFlags(&e->evGath2[q], hipEventDisableTiming));
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->gath_host[q]), bytes, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&e->evGathHost[q], hipEventDisableTiming));
        }
        e->gathered = e->gathered2[q];
    }
    if (!e->comm_stream) {
        int prio_lo = 0, prio_hi = 0;  // the exchange kernel is small and must not queue behind the next step's compute
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        HIPCHK(hipStreamCreateWithPriority(&e->comm_stream, hipStreamNonBlocking, prio_hi));
        HIPCHK(hipEventCreateWithFlags(&e->evSnap, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->evGathered, hipEventDisableTiming));
        HIPCHK(hipMalloc(&e->plk_snap, (size_t)e->c.max_batch * e->c.Nl * e->c.Nk * sizeof(double)));
        HIPCHK(hipEventRecord(e->evGathered, e->comm_stream));
    }
    // The exchange runs in line on the compute stream, straight from the P_l buffer: measured 0.018 ms per step, against 0.24 ms for the
    // variant on a communication stream behind a snapshot (EFTB_GATHER_ASYNC=1) -- the RCCL kernel holds up new dispatches on every queue while it
    // runs, so keeping the compute stream "free" beside it buys nothing, and the look-ahead stream carries the next step's front half anyway.
    static const bool inline_exchange = !(getenv("EFTB_GATHER_ASYNC") && atoi(getenv("EFTB_GATHER_ASYNC")));
    hipStream_t cs = e->comm_stream;
    const bool on_back = inline_exchange && e->back_pending;  // P_l is being written on the back-half stream: the exchange follows it there
    if (inline_exchange) {
        cs = on_back ? e->back : e->stream;
    } else {
        join_back(e);
        // snapshot on the compute stream (after the previous gather has let go of the snapshot buffer) ...
        HIPCHK(hipStreamWaitEvent(e->stream, e->evGathered, 0));
        hipLaunchKernelGGL(copy_kernel, dim3(64), dim3(256), 0, e->stream, e->buf[EFTB_B_PLK], e->plk_snap, count);
        HIPCHK(hipEventRecord(e->evSnap, e->stream));
        // ... exchange on the communication stream, concurrently with whatever the compute stream does next
        HIPCHK(hipStreamWaitEvent(cs, e->evSnap, 0));
    }
    const double* sendbuf = inline_exchange ? e->buf[EFTB_B_PLK] : e->plk_snap;
    if (e->nranks == 1 && !e->comm) {
        HIPCHK(hipMemcpyAsync(e->gathered, sendbuf, count * sizeof(double), hipMemcpyDeviceToDevice, cs));
    } else {  // with a communicator even a single rank goes through the RCCL send / recv group (self exchange)
        if (!e->comm) return fail("eftb_gather_plk: eftb_comm_init was not called");
        NCCLCHK(g_rccl.GroupStart());
        if (e->rank == root)
            for (int r = 0; r < e->nranks; ++r) NCCLCHK(g_rccl.Recv(e->gathered + (size_t)r * count, count, ncclDouble, r, e->comm, cs));
        NCCLCHK(g_rccl.Send(sendbuf, count, ncclDouble, root, e->comm, cs));
        NCCLCHK(g_rccl.GroupEnd());
    }
    HIPCHK(hipEventRecord(e->evGathered, cs));
    if (e->rank == root) {
        const int q = e->gather_slot;
        HIPCHK(hipEventRecord(e->evGath2[q], cs));
        e->gath_elems[q] = (size_t)e->nranks * count;
        e->gath_set[q] = e->buf[EFTB_B_PLK] == e->orig[EFTB_B_PLK] || !e->cpy ? eftb_engine::NSETS : e->cur_set;
        // copy-out with the DMA engine, in line behind the exchange (measured with the PCIe traffic of eight ranks, 12.6 MB per step: 0.441 ms
        // per step; on a stream of its own 0.52 ms -- a sixth stream shares a hardware queue with the look-ahead --; a copy kernel writing
        // mapped host memory 0.62 ms)
        HIPCHK(hipMemcpyAsync(e->gath_host[q], e->gathered2[q], e->gath_elems[q] * sizeof(double), hipMemcpyDeviceToHost, cs));
        HIPCHK(hipEventRecord(e->evGathHost[q], cs));
    }
    if (on_back) HIPCHK(hipEventRecord(e->evBack[(e->back_step + 1) & 1], cs));  // whoever joins the back half also waits for the exchange
    if (host_out && e->rank == root) {
        HIPCHK(hipEventSynchronize(e->evGathHost[e->gather_slot]));
        memcpy(host_out, e->gath_host[e->gather_slot], (size_t)e->nranks * count * sizeof(double));
    }
    return 0;
}

// waits for the copy-out of the exchange `which` exchanges back and returns where its block sits in host memory
static int gathered_ready(eftb_engine* e, const char* who, int which, int* slot) {
    if (which < 0 || which >= eftb_engine::NSETS) return fail("%s: `back` must be 0 (the last exchange enqueued) ... %d (that many exchanges before it)", who, eftb_engine::NSETS - 1);
    const int q = (e->gather_slot + eftb_engine::NSETS - which) % eftb_engine::NSETS;
    if (!e->gathered2[q] || !e->gath_elems[q]) return fail("%s: no such exchange yet", who);
    HIPCHK(hipSetDevice(e->c.device));
    if (int rc = spin_event(e->evGathHost[q], who, "the exchange")) return rc;
    *slot = q;
    return check_status(e, who, e->gath_set[q]);  // the flags of the step whose P_l this block carries (the root's own share)
}

int eftb_gathered_view(eftb_engine* e, int which, const double** block, size_t* count) {
    if (!e || !block) return fail("eftb_gathered_view: null argument");
    int q = 0;
    if (int rc = gathered_ready(e, "eftb_gathered_view", which, &q)) return rc;
    *block = e->gath_host[q];
    if (count) *count = e->gath_elems[q];
    return 0;
}

int eftb_fetch_gathered(eftb_engine* e, int which /* = back */, double* host, size_t count) {
    if (!e || !host) return fail("eftb_fetch_gathered: null argument");
    int q = 0;
    if (int rc = gathered_ready(e, "eftb_fetch_gathered", which, &q)) return rc;
    if (count > e->gath_elems[q]) return fail("eftb_fetch_gathered: the exchange holds %zu elements, asked %zu", e->gath_elems[q], count);
    memcpy(host, e->gath_host[q], count * sizeof(double));
    return 0;
}

int eftb_mfma_f64_peak(int device, double* tflops) {
    if (!tflops) return fail("eftb_mfma_f64_peak: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("eftb_mfma_f64_peak: no HIP device visible");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    double* sink = nullptr;
    HIPCHK(hipMalloc(&sink, sizeof(double)));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    const int iters = 40000, blocks = prop.multiProcessorCount;  // 8 waves per CU = 2 waves per SIMD
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(512), 0, 0, 100, sink);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(512), 0, 0, iters, sink);
    HIPCHK(hipEventRecord(b, 0));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 8 * iters * 4 * (2.0 * 16 * 16 * 4);
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(sink);
    return 0;
}

int eftb_stream_read_probe(int device, size_t bytes, int bytes_per_lane, double* gbps) {
    if (bytes_per_lane != 8 && bytes_per_lane != 16) return fail("eftb_stream_read_probe: bytes_per_lane must be 8 or 16");
    if (bytes < 4096 || bytes > ((size_t)64 << 30)) return fail("eftb_stream_read_probe: bytes out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("eftb_stream_read_probe: no HIP device visible");
    HIPCHK(hipSetDevice(device));
    const size_t n = bytes / sizeof(double) / 2 * 2;
    double *src = nullptr, *sink = nullptr;
    HIPCHK(hipMalloc(&src, n * sizeof(double)));
    HIPCHK(hipMalloc(&sink, sizeof(double)));
    HIPCHK(hipMemset(src, 0, n * sizeof(double)));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {  // the second launch is the timed one
        HIPCHK(hipEventRecord(a, 0));
        if (bytes_per_lane == 8) hipLaunchKernelGGL(stream_read_kernel<1>, dim3(4096), dim3(256), 0, 0, src, n, sink);
        else hipLaunchKernelGGL(stream_read_kernel<2>, dim3(4096), dim3(256), 0, 0, src, n, sink);
        HIPCHK(hipEventRecord(b, 0));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms, a, b));
    }
    if (gbps) *gbps = (double)n * sizeof(double) / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(src);
    (void)hipFree(sink);
    return 0;
}

/* Window precompute on the device (reference window.py:262-359); see window_bessel_kernel. */
int eftb_window_precompute(int device, int Na, int Nl, int Nk, int nx, int Np, const double* k, const double* x, const double* Qt,
                           const double* T, const double* p, int withmask, double windowk, const double* S, double* Wal,
                           double* Waldk, double* Wfold, double* device_ms) {
    if (!k || !x || !Qt || !T || !p || !Wal) return fail("eftb_window_precompute: null argument");
    if (Na < 1 || Na > 3 || Nl < 1 || Nl > 3 || Na > Nl) return fail("eftb_window_precompute: need 1 <= Na <= Nl <= 3 (got Na=%d, Nl=%d)", Na, Nl);
    if (Nk < 1 || nx < 1 || Np < 1 || Nk > 65535) return fail("eftb_window_precompute: bad sizes Nk=%d nx=%d Np=%d", Nk, nx, Np);
    if ((Waldk || Wfold) && !(windowk > 0.0)) return fail("eftb_window_precompute: windowk must be positive");
    if (Wfold && !S) return fail("eftb_window_precompute: Wfold needs the spline matrix S [Np][Nk]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("eftb_window_precompute: no HIP device visible");
    HIPCHK(hipSetDevice(device));
    const size_t R = (size_t)Na * Nl, nW = R * Nk * Np;
    struct Pool {
        std::vector<void*> v;
        ~Pool() { for (void* q : v) (void)hipFree(q); }
    } pool;
    auto dalloc = [&](size_t n, double** out) -> hipError_t {
        hipError_t st = hipMalloc((void**)out, n * sizeof(double));
        if (st == hipSuccess) pool.v.push_back(*out);
        return st;
    };
    double *dk, *dx, *dQ, *dT, *dp, *dA, *dW, *dWd = nullptr, *dS = nullptr, *dF = nullptr;
    HIPCHK(dalloc(Nk, &dk)); HIPCHK(dalloc(nx, &dx)); HIPCHK(dalloc(R * nx, &dQ)); HIPCHK(dalloc((size_t)Nl * nx * Np, &dT));
    HIPCHK(dalloc(Np, &dp)); HIPCHK(dalloc(R * Nk * nx, &dA)); HIPCHK(dalloc(nW, &dW));
    if (Waldk || Wfold) HIPCHK(dalloc(nW, &dWd));
    if (Wfold) { HIPCHK(dalloc((size_t)Np * Nk, &dS)); HIPCHK(dalloc(R * Nk * Nk, &dF)); }
    HIPCHK(hipMemcpy(dk, k, Nk * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dx, x, nx * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dQ, Qt, R * nx * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dT, T, (size_t)Nl * nx * Np * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dp, p, Np * sizeof(double), hipMemcpyHostToDevice));
    if (Wfold) HIPCHK(hipMemcpy(dS, S, (size_t)Np * Nk * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t ev0, ev1;
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    HIPCHK(hipEventRecord(ev0, 0));
    hipLaunchKernelGGL(window_bessel_kernel, dim3((nx + 255) / 256, Nk), dim3(256), 0, 0, Na, Nl, Nk, nx, dk, dx, dQ, dA);
    for (int l = 0; l < Nl; ++l) {  // rows = (a, k), A_al [Nk][nx] @ T_l [nx][Np]
        GemmDesc g{};
        g.A = dA + (size_t)l * Nk * nx; g.a_group = (long long)Nl * Nk * nx; g.a_row = nx; g.a_seg = 0;
        g.rows = Na * Nk; g.rows_per_group = Nk; g.nseg = 1; g.kseg = nx;
        g.B = dT + (size_t)l * nx * Np; g.ldb = Np; g.ncols = Np;
        g.C = dW + (size_t)l * Nk * Np; g.c_group = (long long)Nl * Nk * Np; g.c_row = Np; g.c_colgroup = 0; g.cols_per_group = Np;
        hipLaunchKernelGGL(gemm_rows_kernel, dim3((g.rows + 63) / 64, (g.ncols + 255) / 256), dim3(256), GEMM_LDS, 0, g);
    }
    if (dWd) hipLaunchKernelGGL(window_maskdp_kernel, dim3((unsigned)((nW + 255) / 256)), dim3(256), 0, 0, (int)R, Nk, Np, dk, dp, withmask, windowk, dW, dWd);
    if (Wfold) {  // rows = (a, l, k), Waldk [.][Np] @ S [Np][Nk]
        GemmDesc g{};
        g.A = dWd; g.a_group = 0; g.a_row = Np; g.a_seg = 0;
        g.rows = (int)(R * Nk); g.rows_per_group = g.rows; g.nseg = 1; g.kseg = Np;
        g.B = dS; g.ldb = Nk; g.ncols = Nk;
        g.C = dF; g.c_group = 0; g.c_row = Nk; g.c_colgroup = 0; g.cols_per_group = Nk;
        hipLaunchKernelGGL(gemm_rows_kernel, dim3((g.rows + 63) / 64, (g.ncols + 255) / 256), dim3(256), GEMM_LDS, 0, g);
    }
    HIPCHK(hipEventRecord(ev1, 0));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    if (device_ms) *device_ms = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    HIPCHK(hipMemcpy(Wal, dW, nW * sizeof(double), hipMemcpyDeviceToHost));
    if (Waldk) HIPCHK(hipMemcpy(Waldk, dWd, nW * sizeof(double), hipMemcpyDeviceToHost));
    if (Wfold) HIPCHK(hipMemcpy(Wfold, dF, R * Nk * Nk * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
