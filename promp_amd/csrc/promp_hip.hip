// promp_hip.hip -- host side of libpromp_hip.so: context, device memory, launch sequences, C ABI.
// See include/promp_hip.h for the contract.  gfx950 only; built by __graft_entry__.build():
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC promp_hip.hip -lrccl -o libpromp_hip.so
#include "promp_kernels_chain.h"
#include "promp_kernels_pass.h"
#include "promp_kernels_policy.h"
#include "promp_kernels_policy_wide.h"
#include "promp_kernels_wide_bf16.h"
#include "promp_kernels_sample.h"
#include "promp_kernels_rollout.h"
#include "promp_kernels_generic.h"
#include "promp_kernels_generic_bf16.h"
#include "../../include/promp_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define PROMP_ARCH_NAME(prop) (prop).gcnArchName

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHECK(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct StepData {
    int n_paths = 0, n_rows = 0;
    int n_work[2] = {0, 0};            // [0]: one workgroup per CU (wide passes, gram, fit), [1]: two per CU (k_normalize)
    int rollout_B = 0, rollout_T = 0;  // environments per task / horizon of a device-side rollout in progress (promp_begin_rollout)
    bool rollout_ragged = false;       // promp_begin_collection: rows go to the staging area, rollout_T = its capacity in vectorised steps
    int n_chain_wg = 0;                // workgroups of the register-chained kernels k_pass / k_chain_hvp (segment table)
    bool has_policy = false, processed = false, has_adv = false;
    unsigned long long data_version = 0;   // bumped by every entry point that may change what the policy passes read from this step
    unsigned long long cache_tag = 0;      // identity of the primal cache's contents (every storing pass draws a new one)
    int ls_per_row = 0;
    int feat_dim = 0;
    float *obs = nullptr, *act = nullptr, *rew = nullptr, *old_mean = nullptr, *old_ls = nullptr;
    unsigned* obs_absmax = nullptr;    // [tasks]: bits of the largest |observation| of each task's rows (k_obs_range; the FP16 split's scale)
    bool obs_range_valid = false;      // false from every entry point that writes S.obs until k_obs_range has been enqueued behind it
    double* rew64 = nullptr;           // promp_set_rewards_f64 (allocated on first use); valid while has_rew64
    // DiCE (promp_set_dice_rewards; allocated on first use): per-row adjusted rewards, row tangents of the R-operator pass,
    // coupling weights, scan scratch
    float *dice_rw = nullptr, *dice_c = nullptr, *dice_u = nullptr;
    double* dice_tmp = nullptr;
    bool has_dice = false;
    bool has_rew64 = false;
    float *ret32 = nullptr, *adv32 = nullptr;
    double *ret64 = nullptr, *adv64 = nullptr;
    float* hcache = nullptr;           // primal cache (promp_kernels_chain.h: chain_cache_row), allocated on first use
    int *path_row_offsets = nullptr, *path_task = nullptr, *row_t = nullptr;
    int *task_row_offsets = nullptr, *task_path_offsets = nullptr;
    int* task_wg_offsets[2] = {nullptr, nullptr};
    ChainSeg* chain_segs = nullptr;                          // segment table of k_pass / k_chain_hvp
    int* chain_wg_offsets = nullptr;                         // [workgroups+1]
    int* chain_slot_offsets = nullptr;                       // [tasks+1]: partial rows (= segments) of each task
    double *path_ret0 = nullptr, *path_undisc = nullptr, *path_rsq = nullptr, *path_mom = nullptr;
    double* coeffs = nullptr;
    WorkItem* work[2] = {nullptr, nullptr};
    // second-stream sample processing (promp_process_samples of steps >= 1, see stage_a_stream):
    hipEvent_t ev_use = nullptr;       // main stream: the last enqueued work that reads or writes this step's slabs
    hipEvent_t ev_done = nullptr;      // side stream: the step's processed outputs are complete
    bool use_set = false, side_pending = false, dirty = false;
    // staged uploads (promp_stage_step / promp_commit_step): the slab set is written by the copy stream
    hipEvent_t ev_ready = nullptr;     // copy stream: every array of this slab set has arrived
    bool ready_set = false, wait_ready_main = false, wait_ready_side = false, staged = false;
    std::shared_ptr<void> host_tables; // host-side sources of the asynchronous table copies, alive until the next staging
    std::vector<int> lay_tpo, lay_pro; // the offsets the set's device-side tables were built from (set_step_layout)
};

// waves per workgroup of k_pass / k_chain_hvp (one per SIMD: 512 registers per lane)
constexpr int CHAIN_NW_HVP = 4;
// layer-1 k-steps k_chain_hvp is instantiated for (4 observation entries per step, zero-padded)
int chain_ksteps(int obs_dim) { return obs_dim <= 8 ? 2 : obs_dim <= 20 ? 5 : 8; }
// k_pass instances: (hidden_0 / 16, hidden_1 / 16)
#define PROMP_PASS_ALL(X) X(2, 2) X(2, 4) X(4, 2) X(4, 4)
#define PROMP_WB_ALL(X) X(1, 4, 2) X(2, 7, 4) X(3, 8, 4)
#define PROMP_CHAIN_ALL(X) X(2, 2, 2) X(2, 2, 5) X(2, 2, 8) X(2, 4, 2) X(2, 4, 5) X(2, 4, 8) X(4, 2, 2) X(4, 2, 5) X(4, 2, 8) X(4, 4, 2) X(4, 4, 5) X(4, 4, 8)

struct ProfSlot {
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t used = 0;
    double total_ms = 0.0;
    long long launches = 0, rows = 0;
};

}  // namespace

struct promp_ctx {
    promp_dims d;
    int device = 0, n_cus = 256, clock_mhz = 0;
    char dev_name[256];
    int NP = 0, Dmax = 0, coeff_stride = 0, max_work = 0, partial_stride = 0, gram_stride = 0;
    promp_dims du;                       // the caller's dims (d holds the instantiated, possibly zero-padded hidden widths)
    int NPu = 0;                         // parameter count in the caller's layout
    bool padded = false;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;          // sample processing of steps >= 1 runs here, under the main stream's step-0 work
    hipStream_t copy = nullptr;          // promp_stage_step: host -> device copies of the NEXT batch, under the current one's compute
    std::vector<StepData> back;          // the slab sets being staged (swapped with `steps` entries by promp_commit_step)
    bool overlap = true;
    double *gram_partials_side = nullptr, *fit_scratch_side = nullptr;
    std::vector<StepData> steps;
    float *theta = nullptr, *step_sizes = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    long long adam_t = 0;
    float *theta_tasks = nullptr, *chain = nullptr, *lam = nullptr, *vbuf = nullptr;
    bool tasks_shared = false;           // switch_to_pre_update: every task's parameters ARE theta; theta_tasks is written when somebody reads it
    float* wbuf = nullptr;               // promp_constraint_hvp: [tasks][Theta], allocated on first use
    float *partials = nullptr, *scal_inner = nullptr, *scal_outer = nullptr, *scal_tmp = nullptr;
    float *red = nullptr, *grad_mean = nullptr, *stats = nullptr;
    float eta_last[PROMP_ETA_MAX] = {};
    double *gram_partials = nullptr, *red64 = nullptr;
    void* rollout_buf = nullptr;         // goals, start states and noise of a device rollout
    float* stage_rows = nullptr;         // promp_begin_collection: staging rows [steps][tasks * B] of observations | actions | means
    size_t stage_capacity = 0;           // floats
    size_t rollout_capacity = 0;
    double* fit_scratch = nullptr;       // k_fit_wide: [tasks][2][(D+1)^2] when the matrices do not fit in LDS
    size_t smem_fwd = 0, smem_hvp = 0;
    bool wide = false;                   // cooperative kernels for hidden 128 / obs_dim > 32
    int wbf = 0;                         // hidden 128, obs_dim <= 127: the BF16-pipe cooperative kernels (promp_kernels_wide_bf16.h);
                                         // 1..3 = the observation class (NKO, NXB) = (4,2) (7,4) (8,4): obs_dim <= 63 / 111 / 127
    size_t smem_wb_fwd = 0, smem_wb_bwd = 0, smem_wb_hvp = 0;
    unsigned *wb_planes = nullptr, *wb_vplanes = nullptr;   // [tasks][wb_planes_words]: k_wb_planes' output for theta / the direction
    // The planes of the META-parameters (theta itself, stride 0) have their own block: an epoch passes over step 0 at theta twice --
    // the inner gradient pass and, two passes later, the R-operator pass -- and the second finds the first one's planes (same
    // parameters, same slab, same observation scales): one k_wb_planes launch less per epoch.
    unsigned* wb_planes_meta = nullptr;
    float* cg_buf = nullptr;             // promp_cg_solve: x, r, d, (H + reg I) d, the gradient ahead, theta0 [+ the gradient at theta0]
    double* cg_scal = nullptr;           // [4] r.r, d.Hd, converged, x.Hx
    struct { bool valid = false; unsigned long long theta_version = 0, data_version = 0, sizes_version = 0; const void* step = nullptr; } wbp;
    unsigned* vdir_absmax = nullptr;     // [tasks]: k_vec_absmax's output for the direction (FP16 split)
    // layer-by-layer kernels (promp_kernels_generic.h) for every other shape: layer table, and one set of activation / tangent /
    // cotangent buffers for the whole context (the passes of a context run one after another on its stream)
    bool gramt_single = false;           // PROMP_GRAMT_SINGLE=1: k_gram_tiled with one feature tile (A/B runs against the double-buffered rounds)
    bool gram_untiled = false;           // PROMP_GRAM_UNTILED=1: k_gram_wide at every width (A/B runs against k_gram_tiled)
    int gramt_map_nblk = -1;             // the block count c->gramt_map was balanced for
    GramtMap gramt_map;                  // one-slice k_gram_tiled launches: wave -> rectangle
    bool fit_one_launch = false;         // PROMP_FIT_ONE_LAUNCH=1: k_fit_wide alone at every width (A/B runs against the per-phase launches)
    bool generic = false;
    bool gen_bf16 = true;                // their GEMMs on the BF16 matrix pipe (promp_kernels_generic_bf16.h); PROMP_GEN_FP32=1: the exact-FP32 kernels (A/B runs)
    unsigned short *gb_wplanes = nullptr, *gb_vplanes = nullptr;   // [tasks][gb_plane_stride]: k_gb_planes' output for theta / minus the direction
    long long gb_plane_stride = 0;
    int gb_pf_off[GEN_MAX_LIN] = {}, gb_pb_off[GEN_MAX_LIN] = {};
    int n_lin = 0, g_maxw = 0;
    GenLin lin[GEN_MAX_LIN];
    float *g_act[GEN_MAX_LIN] = {}, *g_ract[GEN_MAX_LIN] = {}, *g_mu = nullptr, *g_rmu = nullptr, *g_dz[2] = {}, *g_qz[2] = {};
    unsigned long long* dbg = nullptr;   // cycle stamps (developer tooling, tools/phase_timing.py)
    bool dbg_enabled = false;
    int* task_counters = nullptr;        // [tasks] arrival counters of the chain kernels' fused reductions (zero between launches)
    int* split_events = nullptr;         // [2] see PassArgs::split_events
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    float min_log_std = -13.815510558f;  // log(1e-6): GaussianMLPPolicy's default min_std
    bool learn_std = true;               // false: log_std is neither adapted (step size 0) nor trained (no Adam update)
    int fuse_min_tasks = 1 << 30;        // k_chain_hvp sums a task's partial rows in-launch from this many local tasks on (default: never)
    int stats_slot = 0;                  // promp_optimize parks the first epoch's statistics in slot 1 (loss_before)
    const float* pass_adv = nullptr;     // launch_pass: per-row weights instead of the step's advantages (DiCE coupling pass)
    float* pass_row_tan = nullptr;       // launch_pass (R-operator pass): where the rows' log-likelihood tangents go
    int pass_cache = 0;                  // launch_pass: 1 = the gradient pass fills the step's primal cache, 2 = the R-operator pass reads it
    int primal_cache = -1;               // promp_set_primal_cache: 1 on, 0 off, -1 default (on: primal_cache_worth)
    // promp_inner_adapt(step 0) from the meta-parameters evaluates exactly what the first epoch of the following optimisation
    // evaluates first (the inner pass at theta on step 0's slab): it leaves theta', the inner scalars and the primal cache
    // where that epoch expects them, and the epoch skips its pass while nothing it depends on has changed (reuse_adapt).
    unsigned long long version_counter = 0, theta_version = 0, sizes_version = 0;
    unsigned long long cache_counter = 0;   // tags of primal-cache fills: their own counter (whether a rank fills a cache depends on
                                            // its shard; promp_state_version must move alike on every rank)
    unsigned long long opt_theta_version = 0;   // theta_version when promp_optimize_begin returned (promp_optimize_end: ls_min)
    struct { bool valid = false; unsigned long long theta_version = 0, data_version = 0, sizes_version = 0; int inner_kind = 0; bool cached = false; float min_log_std = 0.f; bool learn_std = true; } adapt0;
    bool reuse_adapt = true;             // promp_set_reuse_adapt
    bool ls_known = false;               // ls_min is the smallest log_std entry of the current theta
    float ls_min = 0.f;
    // promp_constraint_hvp: what its primal caches (one per step) were filled at; the products of one conjugate-gradient solve run
    // at the same parameters on the same slabs, so the 2K + 1 R-operator passes of every product after the first read them back
    struct { bool valid = false; unsigned long long theta_version = 0, sizes_version = 0, data_version[PROMP_ETA_MAX + 1] = {},
             tag[PROMP_ETA_MAX + 1] = {}; int inner_kind = 0; float min_log_std = 0.f; } chvp;
    long long chvp_cached_passes = 0;          // R-operator passes of promp_constraint_hvp that read a primal cache (tests, tools)
    float* pass_next2 = nullptr;         // launch_pass: second destinations of the RED_STEP reduction (see adapt0)
    float* pass_scal2 = nullptr;
    long long adapt_passes_skipped = 0;
    bool force_split = false;            // take the multi-rank launch sequence (reduce / all-reduce / Adam) on one rank too
    bool fixed_order = false;            // exchange = ncclAllGather + sum in rank order (bitwise-identical replicas) instead of ncclAllReduce
    float* gather = nullptr;             // [nranks][Theta + K + 2]: every rank's sums side by side (fixed_order)
    size_t gather_len = 0;
    bool prof = false;
    ProfSlot prof_slots[PROMP_KERNEL_COUNT];
    float* stats_host = nullptr;         // pinned: promp_optimize_begin parks both statistics slots here (async copy)
    double* small_host = nullptr;        // pinned: promp_download_processed gathers the per-path sums and coefficients here
    size_t small_host_len = 0;
    unsigned* stats_seq_host = nullptr;  // pinned: sequence number the last launch of an optimisation writes behind the statistics
    unsigned stats_seq = 0;              // the number the pending optimisation will write
    bool publish_next = false;           // enqueue_meta: this launch is the one that publishes
    bool opt_pending = false;
    int opt_epochs = 0;
    float* fwd_buf = nullptr;            // staging for promp_policy_forward
    size_t fwd_capacity = 0;
};

namespace {

template <class T>
int dev_alloc(T** p, size_t n) {
    HIPCHECK(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
    HIPCHECK(hipMemset(*p, 0, (n ? n : 1) * sizeof(T)));
    return 0;
}

// ---- second stream for sample processing -------------------------------------------------------
// process_samples of a step >= 1 has no data dependence on the main stream's step-0 work (process_samples(0), _adapt):
// it is enqueued on c->side behind the last main-stream work that touched the step's slabs (ev_use), and the main stream
// picks the results up (ev_done) in front of the first launch that reads them.
//
// Event records are not free on the queue (a few microseconds of bubble each), so the uses are marked lazily: an entry
// point that touches a step only sets S.dirty on exit; the event is recorded when the next entry point that does NOT
// touch the step is about to enqueue (settle_others) -- in the training loop that is once per iteration, in front of
// process_samples(0).  A step still dirty when its processing goes to the side stream is marked on the spot, which
// orders it behind everything enqueued so far: always correct, merely no overlap.
int join_side(promp_ctx* c, StepData& S) {
    if (S.wait_ready_main) {           // slabs staged by the copy stream
        HIPCHECK(hipStreamWaitEvent(c->stream, S.ev_ready, 0));
        S.wait_ready_main = false;
    }
    if (!S.side_pending) return 0;
    HIPCHECK(hipStreamWaitEvent(c->stream, S.ev_done, 0));
    S.side_pending = false;
    return 0;
}
int mark_use(promp_ctx* c, StepData& S) {
    if (!S.dirty || !S.ev_use) return 0;
    HIPCHECK(hipEventRecord(S.ev_use, c->stream));
    S.use_set = true;
    S.dirty = false;
    return 0;
}
int settle_others(promp_ctx* c, const StepData* touched) {
    if (!c->overlap && c->back.empty()) return 0;
    for (auto& S : c->steps)
        if (&S != touched && mark_use(c, S)) return -2;
    for (auto& S : c->back)
        if (&S != touched && mark_use(c, S)) return -2;
    return 0;
}
struct StepScope {
    promp_ctx* c;
    StepData& S;
    int rc;
    // writes: the entry point may change what a policy pass reads from the step (slabs, advantages, layout); pure readers say so
    StepScope(promp_ctx* c_, StepData& S_, bool writes = true) : c(c_), S(S_), rc(join_side(c_, S_) | settle_others(c_, &S_)) {
        if (writes) S.data_version = ++c->version_counter;
    }
    ~StepScope() { S.dirty = true; }
};

// hidden_sizes of a context: promp_dims carries up to four widths (n_hidden == 0: the two-layer struct of ABI 2)
struct HiddenList {
    int n;
    int h[4];
};
static HiddenList hidden_list(const promp_dims* d) {
    HiddenList L;
    L.n = d->n_hidden > 0 ? d->n_hidden : 2;
    L.h[0] = d->hidden1; L.h[1] = d->hidden2; L.h[2] = d->hidden3; L.h[3] = d->hidden4;
    return L;
}
// which family of pass kernels serves a network shape (sample processing alone works for any obs_dim <= 128)
#define PROMP_LINFEAT_MAX_O 480     // LinearFeatureBaseline on the device: 2 obs_dim + 5 <= 965 columns (16 feature rows + their observations in LDS)
// what the layer-by-layer kernels read as GenArgs.act_kind: the hidden nonlinearity's code in the low byte, the output
// nonlinearity's (mlp.py:53-60, 114-117; none = identity) above it
static int gen_act_kinds(const promp_dims* d) {
    const int out = d->hidden_act >> PROMP_OUT_ACT_SHIFT;
    const int ok = out == PROMP_OUT_ACT_TANH ? GEN_ACT_TANH : out == PROMP_OUT_ACT_RELU ? GEN_ACT_RELU : GEN_ACT_IDENTITY;
    return (d->hidden_act & 0xff) | (ok << 8);
}
bool policy_shape_generic(const promp_dims* d) {   // layer-by-layer kernels (promp_kernels_generic.h): everything the fused ones do not cover
    const HiddenList L = hidden_list(d);
    return L.n != 2 || d->obs_dim > 128 || d->act_dim > 8 || d->hidden1 > 128 || d->hidden2 > 128 || d->hidden_act != PROMP_ACT_TANH;      // (an output nonlinearity sits in the upper bits: != too)
}
bool policy_shape_chain(const promp_dims* d) {     // register-chained kernels: hidden widths from {32, 64}, obs_dim <= 32
    return !policy_shape_generic(d) && d->obs_dim <= 32 && (d->hidden1 == 32 || d->hidden1 == 64) && (d->hidden2 == 32 || d->hidden2 == 64);
}
int wb_nko(int cls) { return cls == 1 ? 4 : cls == 2 ? 7 : 8; }      // K = 16 steps of the observation per class
bool policy_shape_coop(const promp_dims* d) {      // cooperative kernels: (128,128), or (64,64) with wide observations
    return !policy_shape_generic(d) && d->hidden1 == d->hidden2 && (d->hidden1 == 128 || (d->hidden1 == 64 && d->obs_dim > 32));
}

int check_dims(const promp_dims* d) {
    if (!d) return fail(-1, "dims is NULL");
    if (d->n_tasks < 1 || d->n_tasks_global < d->n_tasks) return fail(-1, "bad task counts (%d local, %d global)", d->n_tasks, d->n_tasks_global);
    if (d->obs_dim < 1 || d->obs_dim > 1024) return fail(-1, "obs_dim %d unsupported (1..1024)", d->obs_dim);
    if (d->act_dim < 1 || d->act_dim > GEN_MAX_A) return fail(-1, "act_dim %d unsupported (1..%d)", d->act_dim, GEN_MAX_A);
    const HiddenList L = hidden_list(d);
    if (d->n_hidden < 0 || L.n > 4) return fail(-1, "hidden_sizes of length %d unsupported (1..4 hidden layers)", L.n);
    for (int l = 0; l < L.n; ++l)
        if (L.h[l] < 1 || L.h[l] > GEN_MAX_N)
            return fail(-1, "hidden size %d (layer %d) unsupported: tanh layers of 1..%d units.  Two layers of up to 128 units run on the fused "
                        "kernels (narrower ones zero-padded on the instantiated widths: every combination of {32, 64} for obs_dim <= 32, "
                        "(64,64) / (128,128) otherwise); wider layers and other depths on the layer-by-layer kernels", L.h[l], l, GEN_MAX_N);
    if ((d->hidden_act & 0xff) > PROMP_ACT_IDENTITY || d->hidden_act < 0)
        return fail(-1, "hidden_act %d unknown (0 tanh, 1 relu, 2 identity)", d->hidden_act & 0xff);
    if ((d->hidden_act >> PROMP_OUT_ACT_SHIFT) > PROMP_OUT_ACT_RELU)
        return fail(-1, "output nonlinearity %d unknown (0 none, 1 tanh, 2 relu)", d->hidden_act >> PROMP_OUT_ACT_SHIFT);
    if (d->num_inner_steps < 1 || d->num_inner_steps > PROMP_ETA_MAX) return fail(-1, "num_inner_steps must be in [1, %d]", PROMP_ETA_MAX);
    if (d->max_rows < 1 || d->max_paths < 1) return fail(-1, "max_rows / max_paths must be positive");
    return 0;
}

int param_count(const promp_dims* d) {
    const HiddenList L = hidden_list(d);
    int n = 0, in = d->obs_dim;
    for (int l = 0; l < L.n; ++l) {
        n += in * L.h[l] + L.h[l];
        in = L.h[l];
    }
    return n + in * d->act_dim + d->act_dim + d->act_dim;
}

int feature_dim(const promp_dims* d, int kind) {
    if (kind == PROMP_BASELINE_LINEAR_FEATURE) return 2 * d->obs_dim + 4;
    if (kind == PROMP_BASELINE_LINEAR_TIME) return 4;
    return 0;
}

// ---- profiling helpers -------------------------------------------------------------------------
int prof_begin(promp_ctx* c, int id, long long rows) {
    if (!c->prof) return 0;
    ProfSlot& s = c->prof_slots[id];
    if (s.used + 2 > s.ev.size()) {
        hipEvent_t a, b;
        HIPCHECK(hipEventCreate(&a));
        HIPCHECK(hipEventCreate(&b));
        s.ev.push_back(a);
        s.ev.push_back(b);
    }
    HIPCHECK(hipEventRecord(s.ev[s.used], c->stream));
    s.rows += rows;
    return 0;
}
int prof_end(promp_ctx* c, int id) {
    if (!c->prof) return 0;
    ProfSlot& s = c->prof_slots[id];
    HIPCHECK(hipEventRecord(s.ev[s.used + 1], c->stream));
    s.used += 2;
    s.launches += 1;
    return 0;
}
int prof_collect(promp_ctx* c) {
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int id = 0; id < PROMP_KERNEL_COUNT; ++id) {
        ProfSlot& s = c->prof_slots[id];
        for (size_t i = 0; i + 1 < s.used; i += 2) {
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
            s.total_ms += ms;
        }
        s.used = 0;
    }
    return 0;
}

// ---- launches ----------------------------------------------------------------------------------
// A pass on the layer-by-layer kernels (promp_kernels_generic.h): forward chain, loss level, then per layer (output first) the
// weight gradient and the cotangent of the layer below.  One workgroup per entry of work table 0 in every launch; the partial
// rows are the cooperative kernels' (one per work item, summed by k_reduce_task).
#define PROMP_GEN_NBW(nbw, ...) \
    switch (nbw) { case 1: { constexpr int NBW = 1; __VA_ARGS__ } break; case 2: { constexpr int NBW = 2; __VA_ARGS__ } break; \
                   case 3: { constexpr int NBW = 3; __VA_ARGS__ } break; default: { constexpr int NBW = 4; __VA_ARGS__ } break; }
int launch_pass_generic(promp_ctx* c, StepData& S, const PassArgs& a, bool hvp, bool fwd_only) {
    GenArgs g;
    memset(&g, 0, sizeof g);
    g.work = a.work; g.task_row_offsets = a.task_row_offsets;
    g.n_lin = c->n_lin;
    for (int l = 0; l < c->n_lin; ++l) g.lin[l] = c->lin[l];
    g.O = a.O; g.A = a.A; g.NP = c->NP; g.act_kind = gen_act_kinds(&c->d);
    g.theta = a.theta; g.theta_task_stride = a.theta_task_stride; g.vdir = a.vdir;
    g.act[0] = a.obs;
    for (int l = 1; l < c->n_lin; ++l) { g.act[l] = c->g_act[l]; g.out_act[l] = c->g_act[l]; g.ract[l] = c->g_ract[l]; }
    g.mu = c->g_mu; g.rmu = c->g_rmu;
    g.dz[0] = c->g_dz[0]; g.dz[1] = c->g_dz[1]; g.qz[0] = c->g_qz[0]; g.qz[1] = c->g_qz[1];
    g.actions = a.act; g.adv = a.adv; g.old_mean = a.old_mean; g.old_log_std = a.old_log_std; g.ls_per_row = a.ls_per_row;
    g.partials = a.partials; g.partial_stride = a.partial_stride;
    g.loss_kind = a.loss_kind; g.clip_eps = a.clip_eps; g.clip_log_std = a.clip_log_std; g.min_log_std = a.min_log_std;
    g.kl_weight = a.kl_weight; g.row_tan = a.row_tan;
    const dim3 grid(S.n_work[0]);
    // k_gen_linear / k_gb_linear: the 64-row rounds of a work item (about one CU's share of the rows) dealt to GEN_SPLIT workgroups
    const dim3 lgrid(S.n_work[0], GEN_SPLIT);
    hipStream_t st = c->stream;
    const bool bf = c->gen_bf16;
    if (bf) {
        // the parameters' (and minus the direction's) kernels as BF16 planes, both orientations, in the GEMMs' chunk order
        GbPlaneArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.n_lin = c->n_lin;
        int most = 0;
        for (int l = 0; l < c->n_lin; ++l) {
            pa.lin[l] = c->lin[l]; pa.pf_off[l] = c->gb_pf_off[l]; pa.pb_off[l] = c->gb_pb_off[l];
            g.pf_off[l] = c->gb_pf_off[l]; g.pb_off[l] = c->gb_pb_off[l];
            const int oct = (int)(gb_f_elems(c->lin[l].K, c->lin[l].N) / 24);
            most = oct > most ? oct : most;
        }
        const int gx = (most + 255) / 256 < 16 ? (most + 255) / 256 : 16;
        pa.src = a.theta; pa.src_task_stride = a.theta_task_stride; pa.dst = c->gb_wplanes;
        pa.dst_task_stride = a.theta_task_stride ? c->gb_plane_stride : 0; pa.sign = 1.f;
        PROMP_LAUNCH(k_gb_planes, dim3(gx, 2 * c->n_lin, a.theta_task_stride ? c->d.n_tasks : 1), 256, 0, st, pa);
        g.wplanes = c->gb_wplanes; g.wplane_stride = pa.dst_task_stride;
        if (hvp) {
            pa.src = a.vdir; pa.src_task_stride = c->NP; pa.dst = c->gb_vplanes; pa.dst_task_stride = c->gb_plane_stride; pa.sign = -1.f;
            PROMP_LAUNCH(k_gb_planes, dim3(gx, 2 * c->n_lin, c->d.n_tasks), 256, 0, st, pa);
            g.vplanes = c->gb_vplanes; g.vplane_stride = c->gb_plane_stride;
        }
    }
    for (int li = 0; li < c->n_lin; ++li) {
        const int nbw = (c->lin[li].N + 63) / 64;
        PROMP_GEN_NBW(nbw,
            if (bf) {
                if (hvp) { auto k = k_gb_linear<GEN_FWD_T, NBW>; PROMP_LAUNCH(k, lgrid, 256, gb_smem(2, NBW), st, g, li, 0); }
                else { auto k = k_gb_linear<GEN_FWD, NBW>; PROMP_LAUNCH(k, lgrid, 256, gb_smem(1, NBW), st, g, li, 0); }
            } else if (hvp) { auto k = k_gen_linear<GEN_FWD_T, NBW>; PROMP_LAUNCH(k, lgrid, 256, gen_linear_smem(GEN_FWD_T, NBW), st, g, li, 0); }
            else { auto k = k_gen_linear<GEN_FWD, NBW>; PROMP_LAUNCH(k, lgrid, 256, gen_linear_smem(GEN_FWD, NBW), st, g, li, 0); })
    }
    if (hvp) { auto k = k_gen_loss<true, true>; PROMP_LAUNCH(k, grid, 256, gen_loss_smem(g.A), st, g, 0); }
    else if (fwd_only) { auto k = k_gen_loss<false, false>; PROMP_LAUNCH(k, grid, 256, gen_loss_smem(g.A), st, g, 0); }
    else { auto k = k_gen_loss<false, true>; PROMP_LAUNCH(k, grid, 256, gen_loss_smem(g.A), st, g, 0); }
    int pp = 0;
    for (int li = c->n_lin - 1; li >= 0 && !fwd_only; --li) {
        const int nbw = (c->lin[li].N + 63) / 64;
        const dim3 wgrid(S.n_work[0], (c->lin[li].K + GEN_KC - 1) / GEN_KC);     // one slab of 64 input units per workgroup
        PROMP_GEN_NBW(nbw,
            if (bf) {
                const dim3 bgrid(S.n_work[0] * Ly_K_slabs(c->lin[li].K));
                if (hvp) { auto k = k_gb_wgrad<2, NBW>; PROMP_LAUNCH(k, bgrid, 256, gb_smem(2, NBW), st, g, li, pp); }
                else { auto k = k_gb_wgrad<1, NBW>; PROMP_LAUNCH(k, bgrid, 256, gb_smem(1, NBW), st, g, li, pp); }
            } else if (hvp) { auto k = k_gen_wgrad<2, NBW>; PROMP_LAUNCH(k, wgrid, 256, gen_wgrad_smem(2, c->lin[li].N), st, g, li, pp); }
            else { auto k = k_gen_wgrad<1, NBW>; PROMP_LAUNCH(k, wgrid, 256, gen_wgrad_smem(1, c->lin[li].N), st, g, li, pp); })
        if (li == 0) break;
        const int nbk = (c->lin[li].K + 63) / 64;
        PROMP_GEN_NBW(nbk,
            if (bf) {
                if (hvp) { auto k = k_gb_linear<GEN_BWD_T, NBW>; PROMP_LAUNCH(k, lgrid, 256, gb_smem(2, NBW), st, g, li, pp); }
                else { auto k = k_gb_linear<GEN_BWD, NBW>; PROMP_LAUNCH(k, lgrid, 256, gb_smem(1, NBW), st, g, li, pp); }
            } else if (hvp) { auto k = k_gen_linear<GEN_BWD_T, NBW>; PROMP_LAUNCH(k, lgrid, 256, gen_linear_smem(GEN_BWD_T, NBW), st, g, li, pp); }
            else { auto k = k_gen_linear<GEN_BWD, NBW>; PROMP_LAUNCH(k, lgrid, 256, gen_linear_smem(GEN_BWD, NBW), st, g, li, pp); })
        pp ^= 1;
    }
    HIPCHECK(hipGetLastError());
    return 0;
}

// The per-task range of a step's observations (the FP16 split's scale, promp_kernels_pass.h: k_obs_range), enqueued on `st` behind
// whatever wrote the slab there.  The uploads call it on their own stream (its cost is the upload's); the device-side writers only
// mark the table stale and the first policy pass recomputes it.
int enqueue_obs_range(promp_ctx* c, StepData& S, hipStream_t st) {
    if (S.n_rows <= 0) return 0;
    HIPCHECK(hipMemsetAsync(S.obs_absmax, 0, sizeof(unsigned) * c->d.n_tasks, st));
    ObsRangeArgs r;
    r.obs = S.obs; r.task_row_offsets = S.task_row_offsets; r.absmax = S.obs_absmax; r.O = c->d.obs_dim;
    const int slices = (c->n_cus * 2 + c->d.n_tasks - 1) / c->d.n_tasks;
    PROMP_LAUNCH(k_obs_range, dim3(slices < 1 ? 1 : slices > 32 ? 32 : slices, c->d.n_tasks), 256, 16, st, r);
    HIPCHECK(hipGetLastError());
    S.obs_range_valid = true;
    c->wbp.valid = false;          // (the hidden_0 planes carry the scales this launch rewrites)
    return 0;
}

// Does the gradient pass fill the primal cache for the R-operator pass behind it?  promp_set_primal_cache: 1 / 0; -1 (default) = yes at
// every size.  Through round 5 the default was "from two rounds of tiles per compute unit" (a small shard's passes are all fixed cost
// and the stores cost what the R-operator pass got back: 0.750 vs 0.737 ms per step at 5 tasks).  Since the cache-reading instance
// runs every product on the FP16 pipe (round 6: 12.2 k cycles per tile against 26.3 k for the recomputing one) the cache pays on
// small shards too: 0.516 vs 0.545 ms at 5 tasks, 0.475 vs 0.485 at 3 (two A/B pairs, one box).
static bool primal_cache_worth(const promp_ctx* c, long long n_rows) {
    (void)n_rows;
    return c->primal_cache != 0;
}

// One policy pass over a step's slabs plus the per-task reduction that consumes it:
//   red_mode RED_STEP / RED_OUTER / RED_HVP / RED_PLAIN / RED_SCAL (promp_kernels_chain.h).
// k_chain_hvp can do both in one launch; k_pass and the cooperative kernels (hidden 128 / wide observations) are
// followed by k_reduce_task.
int launch_pass(promp_ctx* c, StepData& S, bool hvp, const float* theta, long long theta_stride, int loss_kind,
                float clip_eps, int clip_ls, float klw, bool fwd_only, int red_mode, const float* cur, long long cur_stride,
                float* next, float* scal) {
    if (!S.has_policy) return fail(-3, "step has no actions / agent_infos uploaded");
    if (!S.has_adv) return fail(-3, "step has no advantages: call promp_process_samples or promp_set_advantages first");
    if (!policy_shape_chain(&c->d) && !c->wide && !c->generic)
        return fail(-1, "internal: no pass kernel for this policy shape");
    if (!S.obs_range_valid && enqueue_obs_range(c, S, c->stream)) return -2;
    PassArgs a;
    memset(&a, 0, sizeof a);
    a.obs_absmax = (const float*)S.obs_absmax;
    a.obs = S.obs; a.act = S.act; a.adv = c->pass_adv ? c->pass_adv : S.adv32; a.old_mean = S.old_mean; a.old_log_std = S.old_ls;
    a.row_tan = hvp ? c->pass_row_tan : nullptr;
    const int cache = (c->wide || c->generic || fwd_only || !S.hcache) ? 0 : c->pass_cache;
    a.hcache = cache ? S.hcache : nullptr;
    if (cache == 1) S.cache_tag = ++c->cache_counter;
    a.ls_per_row = S.ls_per_row;
    a.task_row_offsets = S.task_row_offsets;
    a.work = S.work[0];
    a.segs = S.chain_segs; a.wg_seg_offsets = S.chain_wg_offsets;
    a.theta = theta; a.theta_task_stride = theta_stride;
    a.vdir = c->vbuf;
    a.partials = c->partials; a.partial_stride = c->partial_stride;
    a.O = c->d.obs_dim; a.A = c->d.act_dim;
    a.loss_kind = loss_kind; a.clip_eps = clip_eps; a.clip_log_std = clip_ls;
    a.min_log_std = c->min_log_std;   // GaussianMLPPolicy min_std (policies/gaussian_mlp_policy.py:31,35)
    a.kl_weight = klw;
    a.task_counters = c->task_counters; a.task_slot_offsets = S.chain_slot_offsets;
    // In-launch reduction (the last-arriving workgroup of a task sums its partial rows) against the grid-wide k_reduce_task
    // behind the launch: with few tasks every task finishes at once and one workgroup per task streaming ~50 partial rows
    // is exposed (66 us vs 46 + 5 us at 5 tasks); with 40 tasks the sums partly hide under other tasks' tiles, but since
    // k_reduce_task keeps eight rows per thread in flight the separate launch wins there too (123 + 5 us vs 137 us with the
    // primal cache, 147 + 5 vs 162 us without).  The in-launch form stays available (promp_set_schedule).
    a.fuse_reduce = (hvp && !c->wide && !c->generic && c->d.n_tasks >= c->fuse_min_tasks) ? 1 : 0;
    a.red_mode = red_mode; a.step_sizes = c->step_sizes; a.cur = cur; a.cur_task_stride = cur_stride; a.next = next;
    a.lam = c->lam; a.v = c->vbuf; a.scal = scal;
    a.dbg = c->dbg_enabled ? c->dbg : nullptr;
    a.split_events = c->split_events;
    const int id = hvp ? PROMP_KERNEL_HVP : fwd_only ? PROMP_KERNEL_FWD : PROMP_KERNEL_FWD_BWD;
    // (the timed slot covers the pass with the small launches that prepare its operands: k_wb_planes, k_vec_absmax)
    if (prof_begin(c, id, S.n_rows)) return -2;
    if (c->wbf) {
        // the parameters' (and the direction's) hidden kernels as BF16 planes in fragment order (one small launch each: 276 KB per task)
        const int nko = wb_nko(c->wbf);
        WbPlaneArgs pa;
        const bool at_meta = theta == c->theta && theta_stride == 0 && c->wb_planes_meta;
        const bool standing = at_meta && c->wbp.valid && c->wbp.theta_version == c->theta_version && c->wbp.data_version == S.data_version &&
                              c->wbp.sizes_version == c->sizes_version && c->wbp.step == (const void*)&S;
        pa.src = theta; pa.src_stride = theta_stride; pa.dst = at_meta ? c->wb_planes_meta : c->wb_planes; pa.O = c->d.obs_dim; pa.A = c->d.act_dim; pa.NKO = nko; pa.row_sign = 1.f;
        pa.obs_absmax = a.obs_absmax; pa.vec_absmax = nullptr;       // FP16 split: the hidden_0 kernel takes the inverse of the observations' scale
        // (one copy per task even when the tasks share their parameters: the hidden_0 kernel's planes carry the task's observation scale)
        const bool per_task = theta_stride != 0 || PROMP_NT == 2;
        if (!standing) PROMP_LAUNCH(k_wb_planes, dim3((4 * (nko + 16) * 64 + 256 + 255) / 256, per_task ? c->d.n_tasks : 1), 256, 0, c->stream, pa);
        if (at_meta) {
            c->wbp.valid = true; c->wbp.theta_version = c->theta_version; c->wbp.data_version = S.data_version;
            c->wbp.sizes_version = c->sizes_version; c->wbp.step = (const void*)&S;
        }
        a.wb_theta_planes = pa.dst;
        a.wb_plane_stride = per_task ? wb_planes_words(nko) : 0;
        if (hvp) {
            // the direction's planes carry its scale: its largest entry per task first (one small launch)
            VecAbsmaxArgs va;
            va.src = c->vbuf; va.stride = c->NP; va.n = c->NP; va.out = c->vdir_absmax;
            va.n_w1 = c->d.obs_dim * c->d.hidden1; va.obs_absmax = a.obs_absmax;
            PROMP_LAUNCH(k_vec_absmax, dim3(c->d.n_tasks), 1024, 64, c->stream, va);
            a.vdir_absmax = (const float*)c->vdir_absmax;
            pa.src = c->vbuf; pa.src_stride = c->NP; pa.dst = c->wb_vplanes; pa.vec_absmax = a.vdir_absmax;
            PROMP_LAUNCH(k_wb_planes, dim3((4 * (nko + 16) * 64 + 256 + 255) / 256, c->d.n_tasks), 256, 0, c->stream, pa);
            a.wb_v_planes = c->wb_vplanes;
        }
    }
    if (c->generic) {
        if (launch_pass_generic(c, S, a, hvp, fwd_only)) return -2;
    } else if (c->wbf) {
        // two layers of 128 units on the BF16 matrix pipe (promp_kernels_wide_bf16.h)
#define PROMP_WB_CASE(CLS, NKO, NXB)                                                                                              \
    if (c->wbf == CLS) {                                                                                                          \
        if (hvp) { auto k = k_wb_hvp<NKO, NXB>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 256, c->smem_wb_hvp, c->stream, a); }            \
        else if (fwd_only) { auto k = k_wb_fwd_bwd<NKO, NXB, false>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 256, c->smem_wb_fwd, c->stream, a); } \
        else { auto k = k_wb_fwd_bwd<NKO, NXB, true>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 256, c->smem_wb_bwd, c->stream, a); }         \
    }
        PROMP_WB_ALL(PROMP_WB_CASE)
#undef PROMP_WB_CASE
    } else if (c->wide) {
        // cooperative kernels (promp_kernels_policy_wide.h): hidden 128, or hidden 64 with obs_dim > 32
        const int nob = c->d.obs_dim <= 32 ? 2 : c->d.obs_dim <= 64 ? 4 : 8;
        const size_t sm = hvp ? c->smem_hvp : c->smem_fwd;
#define PROMP_WIDE_CASE(HH, NOB)                                                                                                \
    if (c->d.hidden1 == HH && nob == NOB) {                                                                                      \
        if (hvp) { auto k = k_wide_hvp<HH, NOB>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 4 * HH, sm, c->stream, a); }                 \
        else if (fwd_only) { auto k = k_wide_fwd_bwd<HH, NOB, false>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 4 * HH, sm, c->stream, a); } \
        else { auto k = k_wide_fwd_bwd<HH, NOB, true>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 4 * HH, sm, c->stream, a); }            \
    }
        PROMP_WIDE_CASE(128, 2) PROMP_WIDE_CASE(128, 4) PROMP_WIDE_CASE(128, 8) PROMP_WIDE_CASE(64, 2) PROMP_WIDE_CASE(64, 4) PROMP_WIDE_CASE(64, 8)
#undef PROMP_WIDE_CASE
    } else if (hvp) {
        // register-chained R-operator pass: the per-task reduction happens inside the launch (last-arriving workgroup)
        const int n1 = c->d.hidden1 / 16, n2 = c->d.hidden2 / 16, ks = chain_ksteps(c->d.obs_dim);
#define PROMP_CHAIN_CASE(N1, N2, KS)                                                                                             \
    if (n1 == N1 && n2 == N2 && ks == KS) {                                                                                      \
        if (cache == 2) { auto k = k_chain_hvp<N1, N2, KS, CHAIN_NW_HVP, true>; PROMP_LAUNCH(k, dim3(S.n_chain_wg), 64 * CHAIN_NW_HVP, c->smem_hvp, c->stream, a); } \
        else { auto k = k_chain_hvp<N1, N2, KS, CHAIN_NW_HVP, false>; PROMP_LAUNCH(k, dim3(S.n_chain_wg), 64 * CHAIN_NW_HVP, c->smem_hvp, c->stream, a); }           \
    }
        PROMP_CHAIN_ALL(PROMP_CHAIN_CASE)
#undef PROMP_CHAIN_CASE
        HIPCHECK(hipGetLastError());
        if (a.fuse_reduce) return prof_end(c, id);
    } else {
        const int n1 = c->d.hidden1 / 16, n2 = c->d.hidden2 / 16;
#define PROMP_PASS_CASE(N1, N2)                                                                                                          \
    if (n1 == N1 && n2 == N2) {                                                                                                          \
        if (fwd_only) { auto k = k_pass<N1, N2, CHAIN_NW_HVP, false, false>; PROMP_LAUNCH(k, dim3(S.n_chain_wg), 64 * CHAIN_NW_HVP, c->smem_fwd, c->stream, a); }   \
        else if (cache == 1) { auto k = k_pass<N1, N2, CHAIN_NW_HVP, true, true>; PROMP_LAUNCH(k, dim3(S.n_chain_wg), 64 * CHAIN_NW_HVP, c->smem_fwd, c->stream, a); } \
        else { auto k = k_pass<N1, N2, CHAIN_NW_HVP, true, false>; PROMP_LAUNCH(k, dim3(S.n_chain_wg), 64 * CHAIN_NW_HVP, c->smem_fwd, c->stream, a); }            \
    }
        PROMP_PASS_ALL(PROMP_PASS_CASE)
#undef PROMP_PASS_CASE
    }
    HIPCHECK(hipGetLastError());
    if (prof_end(c, id)) return -2;
    ReduceArgs r;
    r.partials = c->partials; r.partial_stride = c->partial_stride;
    // the register-chained kernels write one row per segment
    r.task_wg_offsets = (c->wide || c->generic) ? S.task_wg_offsets[0] : S.chain_slot_offsets;
    r.NP = c->NP;
    r.step_sizes = c->step_sizes; r.mode = red_mode;
    r.cur = cur; r.cur_task_stride = cur_stride; r.next = next;
    r.lam = c->lam; r.v = c->vbuf; r.scal = scal;
    r.next2 = red_mode == RED_STEP ? c->pass_next2 : nullptr; r.scal2 = red_mode == RED_STEP ? c->pass_scal2 : nullptr;
    PROMP_LAUNCH(k_reduce_task, dim3((c->NP + 2 + 255) / 256, c->d.n_tasks), 256, 0, c->stream, r);
    HIPCHECK(hipGetLastError());
    return 0;
}

// (the DiCE objective's gradient is the log-likelihood objective's with the suffix-sum weights of promp_set_dice_rewards)
int loss_kind_inner(int inner_kind) { return (inner_kind == PROMP_INNER_LOGLIK || inner_kind == PROMP_INNER_DICE) ? LOSS_LOGLIK : LOSS_RATIO; }
int loss_kind_outer(int outer_kind) {
    return outer_kind == PROMP_OUTER_RATIO ? LOSS_RATIO : outer_kind == PROMP_OUTER_KL ? LOSS_KL
           : outer_kind == PROMP_OUTER_LOGLIK ? LOSS_LOGLIK : LOSS_CLIP;
}

// promp_inner_adapt has left the first inner pass's results behind (theta' in chain[1], its scalars, the primal cache) and nothing
// it read has changed since; the clip of log_std at log(min_std) -- the one difference between the two -- is not active
static bool adapt0_stands(const promp_ctx* c, int inner_kind, bool cached) {
    return c->reuse_adapt && c->adapt0.valid && c->adapt0.theta_version == c->theta_version &&
           c->adapt0.data_version == c->steps[0].data_version && c->adapt0.sizes_version == c->sizes_version &&
           c->adapt0.inner_kind == inner_kind && c->adapt0.cached == cached && c->adapt0.min_log_std == c->min_log_std &&
           c->adapt0.learn_std == c->learn_std && c->ls_known && c->ls_min >= c->min_log_std && !c->pass_adv;
}

// The one exchange of the path (meta_algos/pro_mp.py:122,151,155: the mean over tasks): the ranks' sums of n floats, in place.
//   default      ncclAllReduce -- the result is the same on every rank for a given algorithm / topology, but the ORDER of the
//                additions is RCCL's;
//   fixed_order  ncclAllGather of the ranks' vectors + k_sum_ranks adding them in rank order 0, 1, ... on every rank: replicas
//                bitwise identical by construction, and equal to what one process adding its shards in that order computes
//                (the fixed-order one-shot variant; n is ~6 k floats, the gather moves nranks x 24 KB).
static int exchange_sums_raw(promp_ctx* c, float* buf, size_t n);
static int exchange_sums(promp_ctx* c, float* buf, size_t n) {
    if (!c->comm) return 0;
    // (promp_prof_enable: HIP events around the exchange on the stream it is enqueued on -- the per-call latency bench.py reports)
    if (prof_begin(c, PROMP_KERNEL_EXCHANGE, 0)) return -2;
    const int rc = exchange_sums_raw(c, buf, n);
    if (rc) return rc;
    return prof_end(c, PROMP_KERNEL_EXCHANGE);
}
static int exchange_sums_raw(promp_ctx* c, float* buf, size_t n) {
    if (c->fixed_order) {
        const size_t need = (size_t)c->nranks * n;
        if (c->gather_len < need) {
            if (c->gather) HIPCHECK(hipFree(c->gather));
            c->gather = nullptr;
            HIPCHECK(hipMalloc((void**)&c->gather, sizeof(float) * need));
            c->gather_len = need;
        }
        ncclResult_t r = ncclAllGather(buf, c->gather, n, ncclFloat, c->comm, c->stream);
        if (r != ncclSuccess) return fail(-4, "ncclAllGather failed: %s", ncclGetErrorString(r));
        PROMP_LAUNCH(k_sum_ranks, dim3((unsigned)((n + 255) / 256)), 256, 0, c->stream, (const float*)c->gather, buf, (int)n, c->nranks);
        HIPCHECK(hipGetLastError());
        return 0;
    }
    ncclResult_t r = ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return fail(-4, "ncclAllReduce failed: %s", ncclGetErrorString(r));
    return 0;
}

// A context that holds a shard of the meta-batch (n_tasks_global > n_tasks) but no communicator: the external-collective mode.
// Its entry points return this rank's SHARE of a mean (promp_meta_grad: local sums / n_tasks_global, the sums themselves in the
// exchange buffer); the ones that would go on to USE a mean (an Adam step, a conjugate-gradient product) refuse instead.
static bool sharded_without_comm(const promp_ctx* c) { return c->d.n_tasks_global > c->d.n_tasks && c->comm == nullptr; }

// One evaluation of the meta-objective (+ gradient, + Adam) enqueued on the stream.
int enqueue_meta(promp_ctx* c, float clip_eps, const float* eta_host, int inner_kind, int outer_kind, bool want_grad,
                 bool do_adam, float lr) {
    const int K = c->d.num_inner_steps, M = c->d.n_tasks, NP = c->NP;
    const size_t MNP = (size_t)M * NP;
    for (int k = 0; k <= K; ++k)
        if (c->steps[k].n_rows == 0) return fail(-3, "step %d has no data", k);
    // a step's second-stream sample processing is waited for right in front of the first pass that reads its advantages:
    // the passes on earlier steps run while it finishes
    bool filled[PROMP_ETA_MAX] = {};      // steps whose primal cache this evaluation has written
    for (int k = 0; k < K; ++k) {
        const float* th = (k == 0) ? c->theta : c->chain + (size_t)k * MNP;
        const long long st = (k == 0) ? 0 : NP;
        if (join_side(c, c->steps[k])) return -2;
        // the R-operator pass of this step (below) runs at these parameters on this slab: it reads the activations and means
        // back instead of recomputing them (primal cache, promp_kernels_chain.h)
        const bool worth = primal_cache_worth(c, c->steps[k].n_rows);
        const bool cached = want_grad && worth && !c->wide && policy_shape_chain(&c->d);
        if (cached && !c->steps[k].hcache &&
            dev_alloc(&c->steps[k].hcache, ((size_t)c->d.max_rows + 16 * (size_t)M) * chain_cache_row(c->d.hidden1, c->d.hidden2))) return -2;
        // promp_inner_adapt has left exactly this pass's results behind (see there) if nothing it read has changed since and
        // the clip of log_std at log(min_std) -- the one difference between the two -- is not active
        const bool reuse = k == 0 && adapt0_stands(c, inner_kind, cached);
        if (reuse) {
            c->adapt_passes_skipped += 1;
            filled[k] = cached;
            continue;
        }
        c->pass_cache = cached ? 1 : 0;
        const int rc0 = launch_pass(c, c->steps[k], false, th, st, loss_kind_inner(inner_kind), clip_eps, k == 0, 0.f, false, RED_STEP, th, st,
                                    c->chain + (size_t)(k + 1) * MNP, c->scal_inner + (size_t)k * M * 2);
        c->pass_cache = 0;
        if (rc0) return -2;
        filled[k] = cached;
    }
    if (join_side(c, c->steps[K])) return -2;
    if (launch_pass(c, c->steps[K], false, c->chain + (size_t)K * MNP, NP, loss_kind_outer(outer_kind), clip_eps, 0, 0.f, !want_grad,
                    want_grad ? RED_OUTER : RED_SCAL, nullptr, 0, nullptr, c->scal_outer)) return -2;
    if (want_grad) {
        for (int k = K - 1; k >= 0; --k) {
            const float* th = (k == 0) ? c->theta : c->chain + (size_t)k * MNP;
            const long long st = (k == 0) ? 0 : NP;
            StepData& Sk = c->steps[k];
            const bool dice = inner_kind == PROMP_INNER_DICE;
            if (dice && !Sk.has_dice) return fail(-3, "step %d has no DiCE rewards: call promp_set_dice_rewards first", k);
            c->pass_row_tan = dice ? Sk.dice_c : nullptr;
            c->pass_cache = filled[k] ? 2 : 0;
            const int rc1 = launch_pass(c, Sk, true, th, st, loss_kind_inner(inner_kind), clip_eps, k == 0, dice ? 0.f : eta_host[k] / (float)K, false,
                                        RED_HVP, nullptr, 0, nullptr, c->scal_tmp);
            c->pass_row_tan = nullptr;
            c->pass_cache = 0;
            if (rc1) return -2;
            if (dice) {
                // The magic box couples the time steps of a path: H v = H_loglik(w) v + grad_loglik(u(v)), u from the row tangents
                // c_t = dlogpi_t . v of the pass above (meta_algos/dice_maml.py:245-258).  The pass ran on the direction -v, so its
                // tangents and with them u carry the sign that makes the second piece ANOTHER "lam += g" reduction.
                DiceScanArgs ds;
                ds.path_row_offsets = Sk.path_row_offsets; ds.rw = Sk.dice_rw; ds.c = Sk.dice_c; ds.out = Sk.dice_u; ds.tmp = Sk.dice_tmp;
                ds.mode = 1;
                PROMP_LAUNCH(k_dice_scan, dim3(Sk.n_paths), 64, 0, c->stream, ds);
                HIPCHECK(hipGetLastError());
                c->pass_adv = Sk.dice_u;
                const int rc2 = launch_pass(c, Sk, false, th, st, LOSS_LOGLIK, 0.f, k == 0, 0.f, false, RED_HVP, nullptr, 0, nullptr, c->scal_tmp);
                c->pass_adv = nullptr;
                if (rc2) return -2;
            }
        }
    }
    FinalArgs f;
    f.lam = c->lam; f.NP = NP; f.K = K; f.n_tasks = M;
    f.scal_inner = c->scal_inner; f.scal_outer = c->scal_outer; f.red = c->red; f.want_grad = want_grad ? 1 : 0;
    // several ranks (or an external collective: more global than local tasks): sums first, mean + Adam after the exchange
    const bool split = c->nranks > 1 || c->force_split || c->d.n_tasks_global != c->d.n_tasks;
    if (split) {
        PROMP_LAUNCH(k_reduce_final, dim3((NP + K + 2 + 63) / 64), 256, 0, c->stream, f);
        HIPCHECK(hipGetLastError());
        if (exchange_sums(c, c->red, (size_t)(NP + K + 2))) return -4;
    }
    AdamArgs ad;
    ad.theta = c->theta; ad.m = c->adam_m; ad.v = c->adam_v; ad.red = c->red; ad.grad_mean = c->grad_mean;
    ad.stats = c->stats + (size_t)c->stats_slot * (K + 2); ad.NP = NP; ad.K = K; ad.A = c->d.act_dim;
    for (int k = 0; k < PROMP_ETA_MAX; ++k) ad.eta[k] = k < K ? eta_host[k] : 0.f;
    ad.host_stats = c->publish_next ? c->stats_host : nullptr; ad.host_seq = c->stats_seq_host; ad.seq = c->stats_seq;
    ad.inv_tasks = 1.0f / (float)c->d.n_tasks_global;
    ad.do_update = do_adam ? 1 : 0;
    ad.n_trainable = c->learn_std ? NP : NP - c->d.act_dim;
    ad.lr_t = 0.f;
    if (do_adam) {
        c->theta_version = ++c->version_counter;     // (the smallest log_std entry is unknown until the next publication)
        c->ls_known = false;
        c->adam_t += 1;
        const double t = (double)c->adam_t;
        ad.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.9, t)));
    }
    if (split) PROMP_LAUNCH(k_mean_adam, dim3((NP + 1 + 255) / 256), 256, 0, c->stream, ad);
    else PROMP_LAUNCH(k_final_adam, dim3((NP + K + 2 + 63) / 64 + 1), 256, 0, c->stream, f, ad);   // one rank: nothing in between
    HIPCHECK(hipGetLastError());
    for (int k = 0; k <= K; ++k) c->steps[k].dirty = true;
    return 0;
}

int upload_eta(promp_ctx* c, const float* eta) {
    // the coefficients travel by value in the launch arguments of the final reduction; promp_adam_step reuses the last set
    for (int k = 0; k < c->d.num_inner_steps; ++k) c->eta_last[k] = eta[k];
    return 0;
}

void free_step(StepData& S) {
    void* ptrs[] = {S.obs_absmax, S.hcache, S.dice_rw, S.dice_c, S.dice_u, S.dice_tmp, S.rew64, S.obs, S.act, S.rew, S.old_mean, S.old_ls, S.ret32, S.adv32, S.ret64, S.adv64, S.path_row_offsets,
                    S.path_task, S.row_t, S.task_row_offsets, S.task_path_offsets, S.task_wg_offsets[0], S.task_wg_offsets[1], S.chain_segs, S.chain_wg_offsets,
                    S.chain_slot_offsets, S.path_ret0,
                    S.path_undisc, S.path_rsq, S.path_mom, S.coeffs, S.work[0], S.work[1]};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
}

int alloc_step(promp_ctx* c, StepData& S) {
    const promp_dims* dims = &c->d;
    const int M = dims->n_tasks;
    const size_t R = dims->max_rows, P = dims->max_paths, A = dims->act_dim, O = dims->obs_dim;
    int rc = 0;
    rc |= dev_alloc(&S.obs, R * O); rc |= dev_alloc(&S.act, R * A); rc |= dev_alloc(&S.rew, R);
    rc |= dev_alloc(&S.obs_absmax, (size_t)M);
    rc |= dev_alloc(&S.old_mean, R * A); rc |= dev_alloc(&S.old_ls, R * A);
    rc |= dev_alloc(&S.ret32, R); rc |= dev_alloc(&S.adv32, R); rc |= dev_alloc(&S.ret64, R); rc |= dev_alloc(&S.adv64, R);
    rc |= dev_alloc(&S.path_row_offsets, P + 1); rc |= dev_alloc(&S.path_task, P); rc |= dev_alloc(&S.row_t, R);
    rc |= dev_alloc(&S.task_row_offsets, (size_t)M + 1); rc |= dev_alloc(&S.task_path_offsets, (size_t)M + 1);
    rc |= dev_alloc(&S.task_wg_offsets[0], (size_t)M + 1); rc |= dev_alloc(&S.task_wg_offsets[1], (size_t)M + 1);
    rc |= dev_alloc(&S.chain_segs, (size_t)c->max_work); rc |= dev_alloc(&S.chain_wg_offsets, (size_t)c->max_work + 1);
    rc |= dev_alloc(&S.chain_slot_offsets, (size_t)M + 1);
    rc |= dev_alloc(&S.path_ret0, P); rc |= dev_alloc(&S.path_undisc, P); rc |= dev_alloc(&S.path_rsq, P);
    rc |= dev_alloc(&S.path_mom, 3 * P); rc |= dev_alloc(&S.coeffs, (size_t)M * c->coeff_stride);
    rc |= dev_alloc(&S.work[0], (size_t)c->max_work); rc |= dev_alloc(&S.work[1], (size_t)c->max_work);
    if (hipEventCreateWithFlags(&S.ev_use, hipEventDisableTiming) != hipSuccess) rc |= 1;
    if (hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming) != hipSuccess) rc |= 1;
    if (hipEventCreateWithFlags(&S.ev_ready, hipEventDisableTiming) != hipSuccess) rc |= 1;
    return rc ? -2 : 0;
}

}  // namespace

extern "C" {

const char* promp_last_error(void) { return g_err.c_str(); }
int promp_abi_version(void) { return 3; }

// The kernels are instantiated for hidden widths from {32, 64} in any combination (obs_dim <= 32) and for (64,64) / (128,128).
// Any other pair of widths up to 128 runs EMBEDDED in the next instantiated shape: the extra hidden units have zero incoming and
// outgoing weights and zero bias, so they output tanh(0) = 0, receive a zero cotangent, and every gradient / Hessian-vector entry
// that belongs to them is exactly zero -- they stay zero under the inner steps and under Adam.  Parameter vectors cross the C ABI in
// the caller's (unpadded) layout (policies/networks/mlp.py:5-62 takes any hidden_sizes; policies/base.py:271-277 fixes the order).
static void pad_dims(const promp_dims* u, promp_dims* p) {
    *p = *u;
    if (policy_shape_generic(u)) return;          // the layer-by-layer kernels take any width as it is
    auto up = [](int h) { return h <= 32 ? 32 : h <= 64 ? 64 : 128; };
    int a = up(u->hidden1), b = up(u->hidden2);
    if (u->obs_dim > 32) a = b = std::max(std::max(a, b), 64);
    else if (a == 128 || b == 128) a = b = 128;
    p->hidden1 = a;
    p->hidden2 = b;
}
// one parameter vector between the caller's layout (du) and the padded one (dp); to_padded: dst must arrive zeroed
static void remap_params(const promp_dims& du, const promp_dims& dp, const float* src, float* dst, bool to_padded) {
    const int O = du.obs_dim, A = du.act_dim, h1 = du.hidden1, h2 = du.hidden2, H1 = dp.hidden1, H2 = dp.hidden2;
    size_t ou = 0, op = 0;
    auto rows = [&](int nrows_u, int nrows_p, int cols_u, int cols_p) {
        for (int r = 0; r < nrows_u; ++r)
            for (int cc = 0; cc < cols_u; ++cc) {
                if (to_padded) dst[op + (size_t)r * cols_p + cc] = src[ou + (size_t)r * cols_u + cc];
                else dst[ou + (size_t)r * cols_u + cc] = src[op + (size_t)r * cols_p + cc];
            }
        ou += (size_t)nrows_u * cols_u;
        op += (size_t)nrows_p * cols_p;
    };
    rows(O, O, h1, H1);      // hidden_0/kernel
    rows(1, 1, h1, H1);      // hidden_0/bias
    rows(h1, H1, h2, H2);    // hidden_1/kernel
    rows(1, 1, h2, H2);      // hidden_1/bias
    rows(h2, H2, A, A);      // output/kernel
    rows(1, 1, A, A);        // output/bias
    rows(1, 1, A, A);        // log_std
}

int promp_param_count(const promp_dims* d) {
    if (!d) return fail(-1, "dims is NULL");
    return param_count(d);
}
int promp_feature_dim(const promp_dims* d, int kind) {
    if (!d) return fail(-1, "dims is NULL");
    return feature_dim(d, kind);
}

int promp_ctx_create(promp_ctx** out, int device_id, const promp_dims* user_dims) {
    if (!out) return fail(-1, "out is NULL");
    *out = nullptr;
    if (check_dims(user_dims)) return -1;
    promp_dims padded_dims;
    pad_dims(user_dims, &padded_dims);
    const promp_dims* dims = &padded_dims;        // everything below sees the instantiated shape
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(-2, "no HIP device available (%s): libpromp_hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    if (device_id < 0 || device_id >= ndev) return fail(-1, "device_id %d out of range (%d devices)", device_id, ndev);
    HIPCHECK(hipSetDevice(device_id));
    promp_ctx* c = new promp_ctx();
    c->d = *dims;
    c->du = *user_dims;
    c->NPu = param_count(user_dims);
    c->padded = c->NPu != param_count(dims);
    c->device = device_id;
    { const char* e = getenv("PROMP_FIT_ONE_LAUNCH"); c->fit_one_launch = e && e[0] == '1'; }
    { const char* e = getenv("PROMP_GRAM_UNTILED"); c->gram_untiled = e && e[0] == '1'; }
    { const char* e = getenv("PROMP_GRAMT_SINGLE"); c->gramt_single = e && e[0] == '1'; }
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device_id));
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // tests: work tables for fewer workgroups than the chip has compute units (a wave then walks several tiles of a small batch)
        const char* e = getenv("PROMP_MAX_CUS");
        const int cap = e ? atoi(e) : 0;
        if (cap > 0 && cap < c->n_cus) c->n_cus = cap;
    }
    c->clock_mhz = prop.clockRate / 1000;
    snprintf(c->dev_name, sizeof c->dev_name, "%s", prop.name[0] ? prop.name : PROMP_ARCH_NAME(prop));
    HIPCHECK(hipStreamCreate(&c->stream));
    {   // sample processing of steps >= 1 is a string of small latency-bound kernels under a chip-filling pass: with the
        // higher priority their workgroups are placed first, the string finishes before the main stream needs its result
        int lo = 0, hi = 0;
        HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // (PROMP_SIDE_PRIO=lo / none: the A/B switch)
        const char* e = getenv("PROMP_SIDE_PRIO");
        if (e && e[0] == 'n') HIPCHECK(hipStreamCreate(&c->side));
        else HIPCHECK(hipStreamCreateWithPriority(&c->side, hipStreamDefault, (e && e[0] == 'l') ? lo : hi));
    }
    const int K = dims->num_inner_steps, M = dims->n_tasks;
    c->NP = param_count(dims);
    // (observations wider than PROMP_LINFEAT_MAX_O: the partial Gram blocks of 2 obs_dim + 5 columns outgrow what a context should
    //  hold -- 3.8 MB per work item at obs_dim 480; such contexts fit LinearTimeBaseline / no baseline on the device, or take
    //  advantages through promp_set_advantages)
    c->Dmax = dims->obs_dim <= PROMP_LINFEAT_MAX_O ? 2 * dims->obs_dim + 4 : 4;
    c->coeff_stride = c->Dmax;
    c->max_work = 2 * c->n_cus + M;
    c->partial_stride = (c->NP + PROMP_PARTIAL_EXTRA + 3) & ~3;
    const int nblk_max = (c->Dmax + 1 + 15) / 16;
    c->gram_stride = nblk_max * (nblk_max + 1) / 2 * 256;
    c->wide = policy_shape_coop(dims);
    c->generic = policy_shape_generic(dims);
    if (c->generic) {
        const HiddenList L = hidden_list(dims);
        int in = dims->obs_dim, off = 0;
        c->n_lin = L.n + 1;
        c->g_maxw = dims->act_dim;
        for (int l = 0; l <= L.n; ++l) {
            const int out = l < L.n ? L.h[l] : dims->act_dim;
            c->lin[l] = GenLin{in, out, off, off + in * out};
            off += in * out + out;
            if (out > c->g_maxw) c->g_maxw = out;
            in = out;
        }
        { const char* e = getenv("PROMP_GEN_FP32"); if (e && e[0] == '1') c->gen_bf16 = false; }
        for (int l = 0; l < c->n_lin; ++l) {
            c->gb_pf_off[l] = (int)c->gb_plane_stride; c->gb_plane_stride += gb_f_elems(c->lin[l].K, c->lin[l].N);
            c->gb_pb_off[l] = (int)c->gb_plane_stride; c->gb_plane_stride += gb_b_elems(c->lin[l].K, c->lin[l].N);
        }
    } else if (c->wide) {
        const int nob = dims->obs_dim <= 32 ? 2 : dims->obs_dim <= 64 ? 4 : 8;
        c->smem_fwd = sizeof(float) * (size_t)make_layout_wide(dims->hidden1, 4, nob, false).total;
        c->smem_hvp = sizeof(float) * (size_t)make_layout_wide(dims->hidden1, 2, nob, true).total;
        // two layers of 128 units: the first-order pass on the BF16 matrix pipe (float32-equivalent 3-way split).  PROMP_WIDE_FP32=1
        // (environment, at context creation) keeps the exact-FP32 cooperative kernels -- the A/B switch of the measurements.
        const char* fp32_env = getenv("PROMP_WIDE_FP32");
        if (dims->hidden1 == 128 && dims->obs_dim <= 127 && !(fp32_env && atoi(fp32_env) != 0)) {
            c->wbf = dims->obs_dim <= 63 ? 1 : dims->obs_dim <= 111 ? 2 : 3;
            c->smem_wb_bwd = sizeof(float) * (size_t)wb_layout(false).total;
            c->smem_wb_fwd = c->smem_wb_bwd;
            c->smem_wb_hvp = sizeof(float) * (size_t)wb_layout(true).total;
        }
    } else {
        // (sized for obs_dim 32: constant offsets in the kernels; contexts with wider observations and these hidden sizes run
        //  sample processing only -- the policy passes reject them at launch -- and must not fail here on LDS they never use)
        promp_dims pd = *dims;
        if (pd.obs_dim > 32) pd.obs_dim = 32;
        c->smem_fwd = sizeof(float) * (size_t)pass_layout(dims->hidden1 / 16, dims->hidden2 / 16, CHAIN_NW_HVP, param_count(&pd)).total;
        // (one size for both instances: the cache-reading one lays LDS out with the backward planes)
        c->smem_hvp = sizeof(float) * (size_t)std::max(chain_layout(dims->hidden1 / 16, dims->hidden2 / 16, CHAIN_NW_HVP, true, param_count(&pd)).total,
                                                       chain_layout(dims->hidden1 / 16, dims->hidden2 / 16, CHAIN_NW_HVP, true, param_count(&pd), true).total);
    }
    if (c->smem_hvp > 160 * 1024 || c->smem_fwd > 160 * 1024) {
        const size_t need = c->smem_hvp > c->smem_fwd ? c->smem_hvp : c->smem_fwd;
        promp_ctx_destroy(c);
        return fail(-1, "LDS budget exceeded (%zu bytes)", need);
    }
    {
#define PROMP_CHAIN_ATTR(N1, N2, KS)                                                                                        \
    {                                                                                                                     \
        auto c2 = k_chain_hvp<N1, N2, KS, CHAIN_NW_HVP, false>; auto c3 = k_chain_hvp<N1, N2, KS, CHAIN_NW_HVP, true>;     \
        HIPCHECK(hipFuncSetAttribute((const void*)c2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)c3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
    }
        PROMP_CHAIN_ALL(PROMP_CHAIN_ATTR)
#undef PROMP_CHAIN_ATTR
#define PROMP_PASS_ATTR(N1, N2)                                                                                             \
    {                                                                                                                     \
        auto c0 = k_pass<N1, N2, CHAIN_NW_HVP, true, false>; auto c1 = k_pass<N1, N2, CHAIN_NW_HVP, false, false>;         \
        auto cs = k_pass<N1, N2, CHAIN_NW_HVP, true, true>;                                                                \
        HIPCHECK(hipFuncSetAttribute((const void*)cs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)c0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)c1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
    }
        PROMP_PASS_ALL(PROMP_PASS_ATTR)
#undef PROMP_PASS_ATTR
#define PROMP_WB_ATTR(CLS, NKO, NXB)                                                                                       \
    {                                                                                                                     \
        auto b0 = k_wb_fwd_bwd<NKO, NXB, true>; auto b1 = k_wb_fwd_bwd<NKO, NXB, false>; auto b2 = k_wb_hvp<NKO, NXB>;         \
        HIPCHECK(hipFuncSetAttribute((const void*)b2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)b0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)b1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
    }
        PROMP_WB_ALL(PROMP_WB_ATTR)
#undef PROMP_WB_ATTR
#define PROMP_WIDE_ATTR(HH, NOB)                                                                                          \
    {                                                                                                                     \
        auto w0 = k_wide_fwd_bwd<HH, NOB, true>; auto w1 = k_wide_fwd_bwd<HH, NOB, false>; auto w2 = k_wide_hvp<HH, NOB>;   \
        HIPCHECK(hipFuncSetAttribute((const void*)w0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)w1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
        HIPCHECK(hipFuncSetAttribute((const void*)w2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
    }
        PROMP_WIDE_ATTR(128, 2) PROMP_WIDE_ATTR(128, 4) PROMP_WIDE_ATTR(128, 8) PROMP_WIDE_ATTR(64, 2) PROMP_WIDE_ATTR(64, 4) PROMP_WIDE_ATTR(64, 8)
#undef PROMP_WIDE_ATTR
        auto g1 = k_gram<1>; auto g2 = k_gram<2>; auto g3 = k_gram<3>; auto g4 = k_gram<4>; auto g5 = k_gram<5>;
        HIPCHECK(hipFuncSetAttribute((const void*)g1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)g2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)g3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)g4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)g5, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)k_fit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        {
            auto f1 = k_fit_wave<12>; auto f2 = k_fit_wave<48>; auto f3 = k_fit_wave<64>; auto f4 = k_fit_wave<45>;
            HIPCHECK(hipFuncSetAttribute((const void*)f4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)f1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)f2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)f3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        HIPCHECK(hipFuncSetAttribute((const void*)k_gram_wide, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        {
            auto t1 = k_gram_tiled<GRAMT_TB, GRAMT_NWV, GRAMT_NLD>;
            HIPCHECK(hipFuncSetAttribute((const void*)t1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
#define PROMP_GEN_ATTR(NBW)                                                                                            \
    {                                                                                                                \
        auto w1 = k_gen_wgrad<1, NBW>; auto w2 = k_gen_wgrad<2, NBW>;                                                 \
        auto l0 = k_gen_linear<GEN_FWD, NBW>; auto l1 = k_gen_linear<GEN_FWD_T, NBW>;                                 \
        auto l2 = k_gen_linear<GEN_BWD, NBW>; auto l3 = k_gen_linear<GEN_BWD_T, NBW>;                                 \
        HIPCHECK(hipFuncSetAttribute((const void*)l0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)l1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)l2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)l3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)w1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)w2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        auto bw1 = k_gb_wgrad<1, NBW>; auto bw2 = k_gb_wgrad<2, NBW>;                                                 \
        auto b0 = k_gb_linear<GEN_FWD, NBW>; auto b1 = k_gb_linear<GEN_FWD_T, NBW>;                                   \
        auto b2 = k_gb_linear<GEN_BWD, NBW>; auto b3 = k_gb_linear<GEN_BWD_T, NBW>;                                   \
        HIPCHECK(hipFuncSetAttribute((const void*)b0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)b1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)b2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)b3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        HIPCHECK(hipFuncSetAttribute((const void*)bw1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
        HIPCHECK(hipFuncSetAttribute((const void*)bw2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
    }
        PROMP_GEN_ATTR(1) PROMP_GEN_ATTR(2) PROMP_GEN_ATTR(3) PROMP_GEN_ATTR(4)
#undef PROMP_GEN_ATTR
        {
            auto s0 = k_gen_loss<true, true>; auto s1 = k_gen_loss<false, true>; auto s2 = k_gen_loss<false, false>;
            HIPCHECK(hipFuncSetAttribute((const void*)s0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gen_loss_smem(GEN_MAX_A)));
            HIPCHECK(hipFuncSetAttribute((const void*)s1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gen_loss_smem(GEN_MAX_A)));
            HIPCHECK(hipFuncSetAttribute((const void*)s2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gen_loss_smem(GEN_MAX_A)));
        }
        {
            auto fw32 = k_fit_wide<32>; auto fw16 = k_fit_wide<16>;
            auto p32 = k_fitw_panel<32>; auto p16 = k_fitw_panel<16>; auto u32 = k_fitw_update<32>; auto u16 = k_fitw_update<16>;
            auto b32 = k_fitw_back<32>; auto b16 = k_fitw_back<16>;
            HIPCHECK(hipFuncSetAttribute((const void*)p32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)p16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)u32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)u16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)b32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)b16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)fw32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHECK(hipFuncSetAttribute((const void*)fw16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    }
    const size_t NP = c->NP, MNP = (size_t)M * NP;
    int rc = 0;
    rc |= dev_alloc(&c->theta, NP); rc |= dev_alloc(&c->step_sizes, NP);
    rc |= dev_alloc(&c->adam_m, NP); rc |= dev_alloc(&c->adam_v, NP);
    rc |= dev_alloc(&c->theta_tasks, MNP); rc |= dev_alloc(&c->chain, (size_t)(K + 1) * MNP);
    rc |= dev_alloc(&c->lam, MNP); rc |= dev_alloc(&c->vbuf, MNP);
    if (c->wbf) {
        const size_t pw = (size_t)M * wb_planes_words(wb_nko(c->wbf));
        rc |= dev_alloc(&c->wb_planes, pw); rc |= dev_alloc(&c->wb_vplanes, pw); rc |= dev_alloc(&c->wb_planes_meta, pw);
        rc |= dev_alloc(&c->vdir_absmax, (size_t)M);
    }
    rc |= dev_alloc(&c->partials, (size_t)c->max_work * c->partial_stride);
    rc |= dev_alloc(&c->scal_inner, (size_t)K * M * 2); rc |= dev_alloc(&c->scal_outer, (size_t)M * 2);
    rc |= dev_alloc(&c->scal_tmp, (size_t)M * 2);
    rc |= dev_alloc(&c->red, NP + K + 2); rc |= dev_alloc(&c->grad_mean, NP);
    rc |= dev_alloc(&c->stats, (size_t)2 * (K + 2));
    // (the baseline fit's partial Gram blocks and scratch matrices -- 2.7 GB per set at Humanoid's 757 columns -- are allocated by the
    //  first promp_process_samples that fits a baseline on that stream: fit_buffers())
    rc |= dev_alloc(&c->red64, 64);
    if (c->generic) {
        const size_t R = (size_t)dims->max_rows;
        for (int l = 1; l < c->n_lin; ++l) {
            rc |= dev_alloc(&c->g_act[l], R * c->lin[l].K);
            rc |= dev_alloc(&c->g_ract[l], R * c->lin[l].K);
        }
        rc |= dev_alloc(&c->g_mu, R * dims->act_dim); rc |= dev_alloc(&c->g_rmu, R * dims->act_dim);
        for (int i = 0; i < 2; ++i) { rc |= dev_alloc(&c->g_dz[i], R * c->g_maxw); rc |= dev_alloc(&c->g_qz[i], R * c->g_maxw); }
        if (c->gen_bf16) { rc |= dev_alloc(&c->gb_wplanes, (size_t)M * c->gb_plane_stride); rc |= dev_alloc(&c->gb_vplanes, (size_t)M * c->gb_plane_stride); }
    }
    rc |= dev_alloc(&c->task_counters, (size_t)M);
    rc |= dev_alloc(&c->dbg, 256 + 4 * 1024);
    rc |= dev_alloc(&c->split_events, 4);
    if (hipHostMalloc((void**)&c->stats_host, sizeof(float) * (2 * (K + 2) + 1), hipHostMallocDefault) != hipSuccess) rc |= 1;
    if (hipHostMalloc((void**)&c->stats_seq_host, sizeof(unsigned), hipHostMallocDefault) != hipSuccess) rc |= 1;
    else *c->stats_seq_host = 0;
    c->steps.resize(K + 1);
    for (int s = 0; s <= K && !rc; ++s) rc |= alloc_step(c, c->steps[s]);
    if (rc) { promp_ctx_destroy(c); return -2; }
    *out = c;
    return 0;
}

void promp_ctx_destroy(promp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->gather) (void)hipFree(c->gather);
    for (auto* set : {&c->steps, &c->back})
        for (auto& S : *set) {
            free_step(S);
            if (S.ev_use) (void)hipEventDestroy(S.ev_use);
            if (S.ev_done) (void)hipEventDestroy(S.ev_done);
            if (S.ev_ready) (void)hipEventDestroy(S.ev_ready);
        }
    void* ptrs[] = {c->cg_buf, c->cg_scal, c->vdir_absmax, c->gb_wplanes, c->gb_vplanes, c->wb_planes, c->wb_vplanes, c->wb_planes_meta, c->wbuf, c->gram_partials_side, c->fit_scratch_side, c->theta, c->step_sizes, c->adam_m, c->adam_v, c->theta_tasks, c->chain, c->lam, c->vbuf,
                    c->partials, c->scal_inner, c->scal_outer, c->scal_tmp, c->red, c->grad_mean, c->stats,
                    c->gram_partials, c->red64, c->fwd_buf, c->stage_rows, c->task_counters, c->split_events, c->dbg, c->fit_scratch, c->rollout_buf};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int l = 0; l < GEN_MAX_LIN; ++l) {
        if (c->g_act[l]) (void)hipFree(c->g_act[l]);
        if (c->g_ract[l]) (void)hipFree(c->g_ract[l]);
    }
    for (float* p : {c->g_mu, c->g_rmu, c->g_dz[0], c->g_dz[1], c->g_qz[0], c->g_qz[1]})
        if (p) (void)hipFree(p);
    if (c->stats_host) (void)hipHostFree(c->stats_host);
    if (c->small_host) (void)hipHostFree(c->small_host);
    if (c->stats_seq_host) (void)hipHostFree(c->stats_seq_host);
    for (auto& s : c->prof_slots)
        for (auto ev : s.ev) (void)hipEventDestroy(ev);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int promp_sync(promp_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    if (c->copy) HIPCHECK(hipStreamSynchronize(c->copy));
    HIPCHECK(hipStreamSynchronize(c->side));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// Offsets, time indices and the three work tables of one sampling step (everything of promp_upload_step but the data).
// `S` is the slab set to describe and `st` the stream the table copies go to: the step's current set on the main stream
// (promp_upload_step; synchronous), or its back set on the copy stream (promp_stage_step; the host-side tables are
// then kept alive in S.host_tables until the set is staged again).
static int set_step_layout(promp_ctx* c, StepData& S, hipStream_t st, bool async, int n_paths, const int32_t* tpo, const int32_t* pro) {
    if (!tpo || !pro) return fail(-1, "offsets are required");
    const int M = c->d.n_tasks;
    S.obs_range_valid = false;         // (whoever describes a slab anew is about to fill it)
    if (n_paths < 1 || n_paths > c->d.max_paths) return fail(-1, "n_paths %d outside [1, max_paths=%d]", n_paths, c->d.max_paths);
    if (tpo[0] != 0 || tpo[M] != n_paths) return fail(-1, "task_path_offsets must start at 0 and end at n_paths");
    if (pro[0] != 0) return fail(-1, "path_row_offsets must start at 0");
    const int R = pro[n_paths];
    if (R < 1 || R > c->d.max_rows) return fail(-1, "rows %d outside [1, max_rows=%d]", R, c->d.max_rows);
    // Same offsets as the batch this set held before (fixed-horizon environments: every batch): the time indices and the
    // work tables on the device are already the right ones -- nothing to rebuild on the host (0.3 ms at 160 000 rows), no
    // table copies to enqueue.  (A set is only ever re-described after its previous table copies have completed.)
    if ((int)S.lay_tpo.size() == M + 1 && (int)S.lay_pro.size() == n_paths + 1 && S.n_paths == n_paths && S.n_rows == R &&
        memcmp(S.lay_tpo.data(), tpo, sizeof(int) * (M + 1)) == 0 && memcmp(S.lay_pro.data(), pro, sizeof(int) * (n_paths + 1)) == 0) {
        S.processed = false; S.has_adv = false; S.has_rew64 = false; S.has_dice = false;
        return 0;
    }
    S.lay_tpo.clear(); S.lay_pro.clear();         // (rebuilt below; stays empty if anything fails half-way)
    std::vector<int> path_task(n_paths), row_t(R), tro(M + 1);
    for (int i = 0; i < M; ++i) {
        if (tpo[i + 1] <= tpo[i]) return fail(-1, "task %d has no paths", i);
        tro[i] = pro[tpo[i]];
        for (int p = tpo[i]; p < tpo[i + 1]; ++p) {
            if (pro[p + 1] < pro[p]) return fail(-1, "path_row_offsets must be non-decreasing");
            path_task[p] = i;
            for (int r = pro[p]; r < pro[p + 1]; ++r) row_t[r] = r - pro[p];
        }
        if (pro[tpo[i + 1]] == tro[i]) return fail(-1, "task %d has no rows", i);
    }
    tro[M] = R;
    // work tables: contiguous ranges of 16-row wave tiles, workgroups shared out over tasks in proportion to their tiles
    std::vector<int> tiles(M);
    long long total_tiles = 0;
    const int GR = 16;   // work granule = one wave tile (16 rows)
    for (int i = 0; i < M; ++i) { tiles[i] = (tro[i + 1] - tro[i] + GR - 1) / GR; total_tiles += tiles[i]; }
    std::vector<WorkItem> work[2];
    std::vector<int> two[2];
    for (int t = 0; t < 2; ++t) {
        // table 0: the cooperative pass kernels (one workgroup per CU: measured faster than two shorter ones, the parameter
        // staging and the end-of-kernel reduction amortise over twice the tiles); table 1: the sample-processing kernels
        int target = (t + 1) * c->n_cus;
        two[t].assign(M + 1, 0);
        // largest-remainder split: sum of workgroups <= target (one more would cost a whole second round on the chip),
        // every task gets at least one and at most one per tile
        std::vector<long long> nw(M), rem(M);
        long long used = 0;
        for (int i = 0; i < M; ++i) {
            const long long num = tiles[i] * (long long)target;
            nw[i] = num / total_tiles;
            rem[i] = num % total_tiles;
            if (nw[i] < 1) { nw[i] = 1; rem[i] = 0; }
            if (nw[i] > tiles[i]) { nw[i] = tiles[i]; rem[i] = 0; }
            used += nw[i];
        }
        while (used < target) {
            int best = -1;
            for (int i = 0; i < M; ++i)
                if (nw[i] < tiles[i] && rem[i] > 0 && (best < 0 || rem[i] > rem[best])) best = i;
            if (best < 0) break;
            nw[best] += 1;
            rem[best] = 0;
            used += 1;
        }
        for (int i = 0; i < M; ++i) {
            const long long w = nw[i];
            for (int g = 0; g < (int)w; ++g) {
                const int t0 = (int)((long long)tiles[i] * g / w), t1 = (int)((long long)tiles[i] * (g + 1) / w);
                WorkItem it;
                it.task = i;
                it.row_begin = tro[i] + t0 * GR;
                it.row_end = tro[i] + t1 * GR;
                if (it.row_end > tro[i + 1]) it.row_end = tro[i + 1];
                it.pad = 0;
                work[t].push_back(it);
            }
            two[t][i + 1] = (int)work[t].size();
        }
        if ((int)work[t].size() > c->max_work) return fail(-5, "internal: work table overflow (%zu > %d)", work[t].size(), c->max_work);
    }
    // chain kernels: the NW waves of a workgroup walk a segment's tiles round-robin, so a task of n tiles costs
    // ceil(n / NW) rounds; the global list of rounds is cut into equal shares, one per CU; a share that straddles task
    // boundaries becomes one segment per task (walked one after the other).  Segments are generated in task order, so a
    // task's partial rows are the contiguous segment indices [slot_off[i], slot_off[i+1]).
    struct ChainTable { std::vector<ChainSeg> segs; std::vector<int> wg_off, slot_off; };
    ChainTable T;
    {
        const int NW = CHAIN_NW_HVP;
        std::vector<long long> rounds(M);
        long long total = 0;
        for (int i = 0; i < M; ++i) { rounds[i] = (tiles[i] + NW - 1) / NW; total += rounds[i]; }
        // A segment also costs its parameter staging and end reduction, about SEGC rounds' worth: workgroups are filled
        // up to a common cost limit (rounds + SEGC per segment, in quarter rounds), the smallest limit that needs no more
        // workgroups than there are CUs.
        const long long SEGC = 2;                  // quarter rounds per segment
        auto cut = [&](long long limit, bool emit) -> long long {
            long long nwg = 0, cost = 0;
            bool open = false;
            for (int i = 0; i < M; ++i) {
                long long done = 0;
                while (done < rounds[i]) {
                    long long room = open ? (limit - cost - SEGC) / 4 : 0;   // rounds of task i that still fit
                    if (!open || room < 1) {
                        if (open && emit) T.wg_off.push_back((int)T.segs.size());
                        ++nwg; open = true; cost = 0;
                        room = (limit - SEGC) / 4;
                        if (room < 1) room = 1;
                    }
                    const long long take = std::min(room, rounds[i] - done);
                    if (emit) {
                        ChainSeg sg;
                        sg.task = i;
                        sg.tile0 = (int)(done * NW);
                        sg.ntiles = (int)std::min<long long>(tiles[i], (done + take) * NW) - sg.tile0;
                        sg.pad = 0;
                        T.segs.push_back(sg);
                        T.slot_off[i + 1] = (int)T.segs.size();
                    }
                    done += take;
                    cost += 4 * take + SEGC;
                }
            }
            if (open && emit) T.wg_off.push_back((int)T.segs.size());
            return nwg;
        };
        long long lo = 4 + SEGC, hi = 4 * total + SEGC * M + 4;
        while (lo < hi) {
            const long long mid = (lo + hi) / 2;
            if (cut(mid, false) <= c->n_cus) hi = mid; else lo = mid + 1;
        }
        T.slot_off.assign(M + 1, 0);
        T.wg_off.assign(1, 0);
        cut(lo, true);
        for (int i = 0; i < M; ++i)
            if (T.slot_off[i + 1] < T.slot_off[i]) T.slot_off[i + 1] = T.slot_off[i];
        if ((int)T.segs.size() > c->max_work) return fail(-5, "internal: segment table overflow (%zu > %d)", T.segs.size(), c->max_work);
    }
    S.n_paths = n_paths; S.n_rows = R; S.n_work[0] = (int)work[0].size(); S.n_work[1] = (int)work[1].size();
    S.processed = false; S.has_adv = false; S.has_rew64 = false; S.has_dice = false;
    // every source below lives in `keep` (asynchronous mode: until the set is staged again)
    struct Keep {
        std::vector<int> pro, tpo, path_task, row_t, tro, wg_off, slot_chain, two[2];
        std::vector<ChainSeg> segs;
        std::vector<WorkItem> work[2];
    };
    auto keep = std::make_shared<Keep>();
    keep->pro.assign(pro, pro + n_paths + 1); keep->tpo.assign(tpo, tpo + M + 1);
    keep->path_task = std::move(path_task); keep->row_t = std::move(row_t); keep->tro = std::move(tro);
    keep->wg_off = std::move(T.wg_off); keep->slot_chain = std::move(T.slot_off);
    keep->segs = std::move(T.segs);
    for (int t = 0; t < 2; ++t) { keep->two[t] = std::move(two[t]); keep->work[t] = std::move(work[t]); }
    const Keep& k = *keep;
    HIPCHECK(hipMemcpyAsync(S.path_row_offsets, k.pro.data(), sizeof(int) * (n_paths + 1), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.task_path_offsets, k.tpo.data(), sizeof(int) * (M + 1), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.path_task, k.path_task.data(), sizeof(int) * n_paths, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.row_t, k.row_t.data(), sizeof(int) * R, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.task_row_offsets, k.tro.data(), sizeof(int) * (M + 1), hipMemcpyHostToDevice, st));
    S.n_chain_wg = (int)k.wg_off.size() - 1;
    HIPCHECK(hipMemcpyAsync(S.chain_segs, k.segs.data(), sizeof(ChainSeg) * k.segs.size(), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.chain_wg_offsets, k.wg_off.data(), sizeof(int) * k.wg_off.size(), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.chain_slot_offsets, k.slot_chain.data(), sizeof(int) * (M + 1), hipMemcpyHostToDevice, st));
    for (int t = 0; t < 2; ++t) {
        HIPCHECK(hipMemcpyAsync(S.task_wg_offsets[t], k.two[t].data(), sizeof(int) * (M + 1), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(S.work[t], k.work[t].data(), sizeof(WorkItem) * k.work[t].size(), hipMemcpyHostToDevice, st));
    }
    if (async) S.host_tables = keep;
    else HIPCHECK(hipStreamSynchronize(st));  // the sources go out of scope
    S.lay_tpo.assign(tpo, tpo + M + 1);
    S.lay_pro.assign(pro, pro + n_paths + 1);
    return 0;
}

static int copy_step_data(promp_ctx* c, StepData& S, hipStream_t st, const float* obs, const float* act, const float* rew,
                          const float* old_mean, const float* old_ls, int ls_per_row) {
    const int M = c->d.n_tasks;
    const size_t R = (size_t)S.n_rows;
    const size_t O = c->d.obs_dim, A = c->d.act_dim;
    HIPCHECK(hipMemcpyAsync(S.obs, obs, sizeof(float) * R * O, hipMemcpyHostToDevice, st));
    if (enqueue_obs_range(c, S, st)) return -2;
    HIPCHECK(hipMemcpyAsync(S.rew, rew, sizeof(float) * R, hipMemcpyHostToDevice, st));
    S.has_policy = act && old_mean && old_ls;
    if (S.has_policy) {
        S.ls_per_row = ls_per_row ? 1 : 0;
        HIPCHECK(hipMemcpyAsync(S.act, act, sizeof(float) * R * A, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(S.old_mean, old_mean, sizeof(float) * R * A, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(S.old_ls, old_ls, sizeof(float) * (ls_per_row ? (size_t)R : (size_t)M) * A, hipMemcpyHostToDevice, st));
    }
    return 0;
}

int promp_upload_step(promp_ctx* c, int step, int n_paths, const int32_t* tpo, const int32_t* pro, const float* obs,
                      const float* act, const float* rew, const float* old_mean, const float* old_ls, int ls_per_row) {
    if (!c) return fail(-1, "ctx is NULL");
    if (!obs || !rew) return fail(-1, "offsets, obs and rew are required");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (set_step_layout(c, S, c->stream, false, n_paths, tpo, pro)) return -2;
    return copy_step_data(c, S, c->stream, obs, act, rew, old_mean, old_ls, ls_per_row);
}

// ---- staged uploads: the NEXT batch travels while the current one is computed ------------------------------------------
void* promp_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(-2, "hipHostMalloc of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}
void promp_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int promp_stage_step(promp_ctx* c, int step, int n_paths, const int32_t* tpo, const int32_t* pro, const float* obs,
                     const float* act, const float* rew, const float* old_mean, const float* old_ls, int ls_per_row) {
    if (!c) return fail(-1, "ctx is NULL");
    if (!obs || !rew) return fail(-1, "offsets, obs and rew are required");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (!c->copy) HIPCHECK(hipStreamCreate(&c->copy));
    if (c->back.empty()) {
        c->back.resize(c->steps.size());
        for (auto& B : c->back)
            if (alloc_step(c, B)) return -2;
    }
    StepData& B = c->back[step];
    // the set's previous life: host tables of its last staging, and whatever the compute streams still read from it
    if (B.ready_set) HIPCHECK(hipEventSynchronize(B.ev_ready));
    if (mark_use(c, B)) return -2;           // (normally settled already by the entry points that came after its last use)
    if (B.use_set) HIPCHECK(hipStreamWaitEvent(c->copy, B.ev_use, 0));
    if (B.side_pending) {
        HIPCHECK(hipStreamWaitEvent(c->copy, B.ev_done, 0));
        B.side_pending = false;
    }
    if (set_step_layout(c, B, c->copy, true, n_paths, tpo, pro)) return -2;
    if (copy_step_data(c, B, c->copy, obs, act, rew, old_mean, old_ls, ls_per_row)) return -2;
    HIPCHECK(hipEventRecord(B.ev_ready, c->copy));
    B.ready_set = true;
    B.wait_ready_main = B.wait_ready_side = true;
    B.staged = true;
    return 0;
}

int promp_commit_step(promp_ctx* c, int step) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (c->back.empty() || !c->back[step].staged) return fail(-3, "step %d has nothing staged", step);
    // uses of the outgoing set that are still unmarked get their mark now, while "everything enqueued so far" is tight
    if (mark_use(c, c->steps[step])) return -2;
    std::swap(c->steps[step], c->back[step]);
    c->steps[step].staged = false;
    c->steps[step].data_version = ++c->version_counter;
    return 0;
}

int promp_stage_wait(promp_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    if (c->copy) HIPCHECK(hipStreamSynchronize(c->copy));
    return 0;
}

// The baseline fit's buffers of one stream (main / side), allocated on first use: contexts that never fit a LinearFeatureBaseline
// (policy passes only, ZeroBaseline, advantages handed in) do not pay for them.
// k_gram_tiled, one slice (at most GRAMT_NWV squares): share the squares out over the waves so that the four SIMDs of a compute
// unit carry about the same number of matrix instructions per k-step (wave w of a workgroup runs on SIMD w mod 4).  Waves to
// spare take halves of diagonal squares (first row of the triangle / the rest: 3 + 3 products at TB = 3) -- Ant's 15 squares on
// 16 waves: 10 x 9 + 4 x 6 + 2 x 3 products = 30 per SIMD.  Longest first, each to the least loaded SIMD that still has a wave
// free.  More squares than waves: slices in list order, diagonal squares whole (the kernel ignores the map).
void gramt_balance(int nblk, int nwv, GramtMap* map) {
    const int nb = gramt_nb(nblk), nr = gramt_nrect(nblk), TB = GRAMT_TB;
    memset(map->rect, 255, sizeof map->rect);
    memset(map->part, GRAMT_DIAG, sizeof map->part);
    if (nr > nwv || nwv > 16) return;
    struct Piece { int rect, part, cost; };
    std::vector<Piece> pieces;
    int spare = nwv - nr;
    for (int bi = 0, r = 0; bi < nb; ++bi)
        for (int bj = bi; bj < nb; ++bj, ++r) {
            if (bi != bj) pieces.push_back({r, GRAMT_FULL, TB * TB});
            else if (spare > 0) {
                pieces.push_back({r, GRAMT_DIAG_TOP, TB});
                pieces.push_back({r, GRAMT_DIAG_REST, TB * (TB + 1) / 2 - TB});
                --spare;
            } else pieces.push_back({r, GRAMT_DIAG, TB * (TB + 1) / 2});
        }
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& x, const Piece& y) { return x.cost > y.cost; });
    int slots[4] = {0, 0, 0, 0}, load[4] = {0, 0, 0, 0}, next[4] = {0, 1, 2, 3};
    for (int w = 0; w < nwv; ++w) slots[w & 3]++;
    for (const Piece& pc : pieces) {
        int q = -1;
        for (int t = 0; t < 4; ++t)
            if (slots[t] > 0 && (q < 0 || load[t] < load[q])) q = t;
        map->rect[next[q]] = (unsigned char)pc.rect;
        map->part[next[q]] = (unsigned char)pc.part;
        next[q] += 4; slots[q]--; load[q] += pc.cost;
    }
}

int fit_buffers(promp_ctx* c, bool on_side) {
    double*& gp = on_side ? c->gram_partials_side : c->gram_partials;
    double*& fs = on_side ? c->fit_scratch_side : c->fit_scratch;
    if (!gp && dev_alloc(&gp, (size_t)c->max_work * c->gram_stride)) return -2;
    const int nblk_max = (c->Dmax + 1 + 15) / 16;
    if (!fs && (nblk_max > 5 || c->d.obs_dim > 32) && dev_alloc(&fs, (size_t)c->d.n_tasks * 2 * (c->Dmax + 1) * (c->Dmax + 1) + c->d.n_tasks)) return -2;   // (+ k_fitw_back's flags)
    return 0;
}

int promp_process_samples(promp_ctx* c, int step, const promp_proc_opts* o) {
    if (!c || !o) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    if (!(o->discount >= 0 && o->discount <= 1)) return fail(-1, "discount factor must be in [0,1]");      // samplers/base.py:57
    if (!(o->gae_lambda >= 0 && o->gae_lambda <= 1)) return fail(-1, "gae_lambda must be in [0,1]");       // samplers/base.py:58
    if (o->baseline_kind < 0 || o->baseline_kind > 2) return fail(-1, "unknown baseline kind %d", o->baseline_kind);
    if (o->baseline_kind == PROMP_BASELINE_LINEAR_FEATURE && c->d.obs_dim > PROMP_LINFEAT_MAX_O)
        return fail(-1, "LinearFeatureBaseline's fit is sized for obs_dim <= %d (%d here: %d feature columns); fit LinearTimeBaseline / no "
                    "baseline on the device, or hand advantages in through promp_set_advantages", PROMP_LINFEAT_MAX_O, c->d.obs_dim,
                    2 * c->d.obs_dim + 5);
    SampleArgs a;
    a.obs = S.obs; a.rew = S.rew; a.rew64 = S.has_rew64 ? S.rew64 : nullptr; a.path_row_offsets = S.path_row_offsets; a.path_task = S.path_task; a.row_t = S.row_t;
    a.task_row_offsets = S.task_row_offsets; a.task_path_offsets = S.task_path_offsets;
    a.work = S.work[0];                         // k_gram / k_fit: one workgroup per CU
    a.task_wg_offsets = S.task_wg_offsets[0];
    a.O = c->d.obs_dim; a.kind = o->baseline_kind; a.D = feature_dim(&c->d, o->baseline_kind);
    a.gamma = o->discount; a.lam = o->gae_lambda; a.reg = o->reg_coeff;
    a.normalize = o->normalize_adv; a.positive = o->positive_adv;
    a.ret64 = S.ret64; a.ret32 = S.ret32; a.adv64 = S.adv64; a.adv32 = S.adv32;
    a.path_ret0 = S.path_ret0; a.path_undisc = S.path_undisc; a.path_rsq = S.path_rsq; a.path_mom = S.path_mom;
    a.coeffs = S.coeffs; a.coeff_stride = c->coeff_stride;
    a.bl64 = nullptr;
    S.feat_dim = a.D;
    // Steps >= 1 go to the second stream (no data dependence on the step-0 work the host enqueued just before: their
    // samples are resident), behind the last main-stream work that touched this step's slabs.  Per-kernel timing
    // (promp_profile) keeps everything on the one stream it brackets.
    const bool on_side = c->overlap && !c->prof && step >= 1;
    hipStream_t st = on_side ? c->side : c->stream;
    if (o->baseline_kind != PROMP_BASELINE_ZERO && fit_buffers(c, on_side)) return -2;
    a.gram_partials = on_side ? c->gram_partials_side : c->gram_partials;
    double* fit_scratch = on_side ? c->fit_scratch_side : c->fit_scratch;
    if (on_side) {
        if (mark_use(c, S)) return -2;
        if (S.use_set) HIPCHECK(hipStreamWaitEvent(c->side, S.ev_use, 0));
        if (S.wait_ready_side) HIPCHECK(hipStreamWaitEvent(c->side, S.ev_ready, 0));
    }
    S.wait_ready_side = false;
    PROMP_LAUNCH(k_returns, dim3(S.n_paths), 64, 0, st, a);
    HIPCHECK(hipGetLastError());
    if (a.kind != BASE_ZERO) {
        const int nblk = (a.D + 1 + 15) / 16;
        if (prof_begin(c, PROMP_KERNEL_GRAM, S.n_rows)) return -2;
        // k_gram<NBLK> stages raw observation rows of at most 32 floats (LinearTimeBaseline reads no observations: any obs_dim)
        const bool small = nblk <= 5 && (a.O <= 32 || a.kind != BASE_LINFEAT);
        switch (small ? nblk : 0) {
            case 1: { auto k = k_gram<1>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 64 * GramCfg<1>::NW, GramCfg<1>::SMEM_BYTES, st, a); } break;
            case 2: { auto k = k_gram<2>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 64 * GramCfg<2>::NW, GramCfg<2>::SMEM_BYTES, st, a); } break;
            case 3: { auto k = k_gram<3>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 64 * GramCfg<3>::NW, GramCfg<3>::SMEM_BYTES, st, a); } break;
            case 4: { auto k = k_gram<4>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 64 * GramCfg<4>::NW, GramCfg<4>::SMEM_BYTES, st, a); } break;
            case 5: { auto k = k_gram<5>; PROMP_LAUNCH(k, dim3(S.n_work[0]), 64 * GramCfg<5>::NW, GramCfg<5>::SMEM_BYTES, st, a); } break;
            default:
                // 13 blocks and more (obs_dim >= 94; Ant: 15, Humanoid: 48): a square of 3 x 3 blocks per wave, operands reused in
                // registers (k_gram_tiled); fewer blocks make too few squares to fill a compute unit: k_gram_wide
                if (nblk >= GRAMT_MIN_NBLK && !c->gram_untiled) {
                    const int nr = gramt_nrect(nblk);
                    int rows, db;
                    gramt_cfg(nblk, a.O, 64 * GRAMT_NWV * GRAMT_NLD, c->gramt_single, &rows, &db);
                    if (c->gramt_map_nblk != nblk) { gramt_balance(nblk, GRAMT_NWV, &c->gramt_map); c->gramt_map_nblk = nblk; }
                    auto k = k_gram_tiled<GRAMT_TB, GRAMT_NWV, GRAMT_NLD>;
                    PROMP_LAUNCH(k, dim3(S.n_work[0], (nr + GRAMT_NWV - 1) / GRAMT_NWV), 64 * GRAMT_NWV, gramt_smem(nblk, rows, db), st, a, nblk,
                                 c->gramt_map, rows, db);
                    break;
                }
                // (more than 17 blocks -- obs_dim > 133: the pair list is cut into slices of <= 160, one workgroup per work item and slice)
                PROMP_LAUNCH(k_gram_wide, dim3(S.n_work[0], gramw_slices(nblk)), 512, gramw_smem(nblk, a.O, gramw_rows(nblk, a.O)), st, a, nblk,
                             gramw_rows(nblk, a.O));
        }
        HIPCHECK(hipGetLastError());
        if (prof_end(c, PROMP_KERNEL_GRAM)) return -2;
        const int DA = a.D + 1;
        if (small) {
            const size_t fit_smem = sizeof(double) * ((size_t)2 * DA * DA + 3 * DA + 2 + fitwv_aux(64));      // (k_fit / k_fit_wave<DT <= 64>)
            // one wave per task while a row of the work matrix fits a wave's lanes (D + 1 <= 64); else one workgroup per task
            if (DA <= 12) { auto k = k_fit_wave<12>; PROMP_LAUNCH(k, dim3(c->d.n_tasks), FITWV_NT, fit_smem, st, a, nblk); }
            else if (DA <= 45) { auto k = k_fit_wave<45>; PROMP_LAUNCH(k, dim3(c->d.n_tasks), FITWV_NT, fit_smem, st, a, nblk); }    // obs_dim 20
            else if (DA <= 48) { auto k = k_fit_wave<48>; PROMP_LAUNCH(k, dim3(c->d.n_tasks), FITWV_NT, fit_smem, st, a, nblk); }
            else if (DA <= 64) { auto k = k_fit_wave<64>; PROMP_LAUNCH(k, dim3(c->d.n_tasks), FITWV_NT, fit_smem, st, a, nblk); }
            else PROMP_LAUNCH(k_fit, dim3(c->d.n_tasks), 256, fit_smem, st, a, nblk);
        } else {
            PROMP_LAUNCH(k_gram_sum_wide, dim3(c->d.n_tasks * fitw_sum_split(nblk)), 256, 0, st, a, nblk, fit_scratch, fitw_sum_split(nblk));
            HIPCHECK(hipGetLastError());
            const int* none = nullptr;
            int* bad = (int*)(fit_scratch + (size_t)c->d.n_tasks * 2 * (c->Dmax + 1) * (c->Dmax + 1));
            const bool phases = a.D >= FITW_ML_MIN_D && !c->fit_one_launch;      // one launch per phase: all CUs in the trailing updates
#define PROMP_FITW(NB)                                                                                                              \
    if (phases) {                                                                                                                   \
        auto kp = k_fitw_panel<NB>; auto ku = k_fitw_update<NB>; auto kb = k_fitw_back<NB>; auto kf = k_fit_wide<NB>;              \
        for (int k0 = 0; k0 < a.D; k0 += NB) {                                                                                      \
            PROMP_LAUNCH(kp, dim3(c->d.n_tasks), FITW_NT, fitw_panel_smem(a.D, NB), st, a, fit_scratch, k0);                        \
            if (k0 + NB < a.D) PROMP_LAUNCH(ku, dim3(c->d.n_tasks, FITW_UPD_SPLIT), FITW_NT, fitw_panel_smem(a.D, NB), st, a, fit_scratch, k0); \
        }                                                                                                                           \
        PROMP_LAUNCH(kb, dim3(c->d.n_tasks), FITW_NT, fitw_back_smem(a.D, NB), st, a, fit_scratch, bad);                               \
        PROMP_LAUNCH(kf, dim3(c->d.n_tasks), FITW_NT, fitw_smem(a.D, NB), st, a, nblk, fit_scratch, (const int*)bad);               \
    } else { auto k = k_fit_wide<NB>; PROMP_LAUNCH(k, dim3(c->d.n_tasks), FITW_NT, fitw_smem(a.D, NB), st, a, nblk, fit_scratch, none); }
            if (fitw_nb(a.D) == 32) { PROMP_FITW(32) } else { PROMP_FITW(16) }
#undef PROMP_FITW
        }
        HIPCHECK(hipGetLastError());
    }
    PROMP_LAUNCH(k_gae, dim3(S.n_paths), 64, sizeof(double) * (size_t)(a.D > 0 ? a.D : 1), st, a);
    HIPCHECK(hipGetLastError());
    a.work = S.work[1];                         // k_normalize: two workgroups per CU
    PROMP_LAUNCH(k_normalize, dim3(S.n_work[1]), 256, 0, st, a);
    HIPCHECK(hipGetLastError());
    if (on_side) {
        HIPCHECK(hipEventRecord(S.ev_done, c->side));
        S.side_pending = true;
    }
    S.processed = true;
    S.has_adv = true;
    return 0;
}

int promp_download_processed(promp_ctx* c, int step, float* returns, float* adv, double* coeffs, double* ret0,
                             double* undisc, double* rsq) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    if (!S.processed) return fail(-3, "step %d has not been processed", step);
    hipStream_t st = c->stream;
    if (returns) HIPCHECK(hipMemcpyAsync(returns, S.ret32, sizeof(float) * S.n_rows, hipMemcpyDeviceToHost, st));
    if (adv) HIPCHECK(hipMemcpyAsync(adv, S.adv32, sizeof(float) * S.n_rows, hipMemcpyDeviceToHost, st));
    // the small results (three per-path sums, the tasks' coefficients) land in one page-locked staging area behind ONE
    // synchronisation and are copied out from there: the caller's arrays are pageable, and a device-to-pageable copy is a
    // synchronous bounce each (the plugin classes make this call once per sampling step)
    const size_t P = (size_t)S.n_paths, NC = (coeffs && S.feat_dim > 0) ? (size_t)c->d.n_tasks * c->coeff_stride : 0;
    const size_t need = 3 * P + NC;
    if (c->small_host_len < need) {
        if (c->small_host) HIPCHECK(hipHostFree(c->small_host));
        c->small_host = nullptr; c->small_host_len = 0;
        HIPCHECK(hipHostMalloc((void**)&c->small_host, sizeof(double) * need, hipHostMallocDefault));
        c->small_host_len = need;
    }
    double* h = c->small_host;
    if (ret0) HIPCHECK(hipMemcpyAsync(h, S.path_ret0, sizeof(double) * P, hipMemcpyDeviceToHost, st));
    if (undisc) HIPCHECK(hipMemcpyAsync(h + P, S.path_undisc, sizeof(double) * P, hipMemcpyDeviceToHost, st));
    if (rsq) HIPCHECK(hipMemcpyAsync(h + 2 * P, S.path_rsq, sizeof(double) * P, hipMemcpyDeviceToHost, st));
    if (NC) HIPCHECK(hipMemcpyAsync(h + 3 * P, S.coeffs, sizeof(double) * NC, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (ret0) memcpy(ret0, h, sizeof(double) * P);
    if (undisc) memcpy(undisc, h + P, sizeof(double) * P);
    if (rsq) memcpy(rsq, h + 2 * P, sizeof(double) * P);
    if (NC)
        for (int i = 0; i < c->d.n_tasks; ++i)
            memcpy(coeffs + (size_t)i * S.feat_dim, h + 3 * P + (size_t)i * c->coeff_stride, sizeof(double) * S.feat_dim);
    return 0;
}

int promp_download_raw(promp_ctx* c, int step, double* ret64, double* adv64) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    if (!S.processed) return fail(-3, "step %d has not been processed", step);
    if (ret64) HIPCHECK(hipMemcpyAsync(ret64, S.ret64, sizeof(double) * S.n_rows, hipMemcpyDeviceToHost, c->stream));
    if (adv64) HIPCHECK(hipMemcpyAsync(adv64, S.adv64, sizeof(double) * S.n_rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int promp_set_coeffs(promp_ctx* c, int step, int kind, const double* coeffs) {
    if (!c || !coeffs) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (kind < 1 || kind > 2) return fail(-1, "coefficients exist for the linear baselines only");
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    const int D = feature_dim(&c->d, kind), M = c->d.n_tasks;
    std::vector<double> tmp((size_t)M * c->coeff_stride, 0.0);
    for (int i = 0; i < M; ++i) memcpy(tmp.data() + (size_t)i * c->coeff_stride, coeffs + (size_t)i * D, sizeof(double) * D);
    HIPCHECK(hipMemcpyAsync(S.coeffs, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    S.feat_dim = D;
    return 0;
}

int promp_predict_baseline(promp_ctx* c, int step, int kind, double* out) {
    if (!c || !out) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (kind < 0 || kind > 2) return fail(-1, "unknown baseline kind %d", kind);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    SampleArgs a;
    memset(&a, 0, sizeof a);
    a.obs = S.obs; a.rew = S.rew; a.rew64 = S.has_rew64 ? S.rew64 : nullptr; a.path_row_offsets = S.path_row_offsets; a.path_task = S.path_task; a.row_t = S.row_t;
    a.task_row_offsets = S.task_row_offsets; a.task_path_offsets = S.task_path_offsets;
    a.O = c->d.obs_dim; a.kind = kind; a.D = feature_dim(&c->d, kind);
    a.gamma = 1.0; a.lam = 1.0;
    a.adv64 = S.adv64; a.path_mom = S.path_mom; a.coeffs = S.coeffs; a.coeff_stride = c->coeff_stride;
    a.bl64 = S.ret64;                              // scratch: returns of this step are recomputed by process_samples
    if (kind == PROMP_BASELINE_ZERO) HIPCHECK(hipMemsetAsync(S.ret64, 0, sizeof(double) * S.n_rows, c->stream));
    PROMP_LAUNCH(k_gae, dim3(S.n_paths), 64, sizeof(double) * (size_t)(a.D > 0 ? a.D : 1), c->stream, a);
    HIPCHECK(hipGetLastError());
    S.processed = false;
    S.has_adv = false;
    HIPCHECK(hipMemcpyAsync(out, S.ret64, sizeof(double) * S.n_rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int promp_set_advantages(promp_ctx* c, int step, const float* adv) {
    if (!c || !adv) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    HIPCHECK(hipMemcpyAsync(S.adv32, adv, sizeof(float) * S.n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    S.has_adv = true;
    return 0;
}

int promp_set_dice_rewards(promp_ctx* c, int step, const float* rw) {
    if (!c || !rw) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (c->wide || c->generic) return fail(-1, "the DiCE objective is built on the register-chained kernels (hidden sizes from {32,64}, obs_dim <= 32)");
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    if (!S.dice_rw) {
        const size_t R = c->d.max_rows;
        int rc = dev_alloc(&S.dice_rw, R);
        rc |= dev_alloc(&S.dice_c, R); rc |= dev_alloc(&S.dice_u, R); rc |= dev_alloc(&S.dice_tmp, R);
        if (rc) return -2;
    }
    HIPCHECK(hipMemcpyAsync(S.dice_rw, rw, sizeof(float) * S.n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));      // (the source may be a temporary of the caller)
    // the gradient weights w_t = sum_{t' >= t} rw_t' take the advantages' place in the log-likelihood objective
    DiceScanArgs ds;
    ds.path_row_offsets = S.path_row_offsets; ds.rw = S.dice_rw; ds.c = nullptr; ds.out = S.adv32; ds.tmp = nullptr; ds.mode = 0;
    PROMP_LAUNCH(k_dice_scan, dim3(S.n_paths), 64, 0, c->stream, ds);
    HIPCHECK(hipGetLastError());
    S.has_adv = true;
    S.has_dice = true;
    return 0;
}

static int copy_in(promp_ctx* c, float* dst, const float* src, size_t n) {
    if (!c || !src) return fail(-1, "NULL argument");
    HIPCHECK(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}
static int copy_out(promp_ctx* c, float* dst, const float* src, size_t n) {
    if (!c || !dst) return fail(-1, "NULL argument");
    HIPCHECK(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// nvec parameter vectors in the caller's layout <-> device buffers in the instantiated (zero-padded) layout
static int params_in(promp_ctx* c, float* dst, const float* src, size_t nvec) {
    if (!c || !src) return fail(-1, "NULL argument");
    if (!c->padded) return copy_in(c, dst, src, nvec * (size_t)c->NP);
    std::vector<float> tmp(nvec * (size_t)c->NP, 0.f);
    for (size_t i = 0; i < nvec; ++i) remap_params(c->du, c->d, src + i * (size_t)c->NPu, tmp.data() + i * (size_t)c->NP, true);
    return copy_in(c, dst, tmp.data(), tmp.size());
}
static int params_out(promp_ctx* c, float* dst, const float* src, size_t nvec) {
    if (!c || !dst) return fail(-1, "NULL argument");
    if (!c->padded) return copy_out(c, dst, src, nvec * (size_t)c->NP);
    std::vector<float> tmp(nvec * (size_t)c->NP);
    if (copy_out(c, tmp.data(), src, tmp.size())) return -2;
    for (size_t i = 0; i < nvec; ++i) remap_params(c->du, c->d, tmp.data() + i * (size_t)c->NP, dst + i * (size_t)c->NPu, false);
    return 0;
}

int promp_set_theta(promp_ctx* c, const float* th) {
    if (!c || !th) return fail(-1, "NULL argument");
    c->theta_version = ++c->version_counter;
    c->ls_min = th[c->NPu - c->d.act_dim];
    for (int i = 1; i < c->d.act_dim; ++i) c->ls_min = std::min(c->ls_min, th[c->NPu - c->d.act_dim + i]);
    c->ls_known = true;
    return params_in(c, c->theta, th, 1);
}
int promp_get_theta(promp_ctx* c, float* th) { return c ? params_out(c, th, c->theta, 1) : fail(-1, "ctx is NULL"); }
static int mask_log_std_step_sizes(promp_ctx* c) {
    if (c->learn_std) return 0;
    HIPCHECK(hipMemsetAsync(c->step_sizes + (c->NP - c->d.act_dim), 0, sizeof(float) * c->d.act_dim, c->stream));
    return 0;
}
int promp_set_step_sizes(promp_ctx* c, const float* s) {
    if (!c) return fail(-1, "ctx is NULL");
    c->sizes_version = ++c->version_counter;
    if (params_in(c, c->step_sizes, s, 1)) return -2;
    return mask_log_std_step_sizes(c);
}
int promp_set_min_std(promp_ctx* c, float min_std) {
    if (!c) return fail(-1, "ctx is NULL");
    if (!(min_std > 0.f)) return fail(-1, "min_std must be positive");
    c->min_log_std = logf(min_std);
    c->version_counter += 1;
    return 0;
}
int promp_set_schedule(promp_ctx* c, int stage_overlap, int fuse_min_tasks) {
    if (!c) return fail(-1, "ctx is NULL");
    if (stage_overlap >= 0) {
        HIPCHECK(hipStreamSynchronize(c->side));
        c->overlap = stage_overlap != 0;
    }
    if (fuse_min_tasks >= 0) c->fuse_min_tasks = fuse_min_tasks;
    return 0;
}
int promp_set_reuse_adapt(promp_ctx* c, int on) {
    if (!c) return fail(-1, "ctx is NULL");
    c->reuse_adapt = on != 0;
    c->adapt0.valid = false;
    return 0;
}
long long promp_adapt_passes_skipped(promp_ctx* c) { return c ? c->adapt_passes_skipped : -1; }
long long promp_constraint_hvp_cached_passes(promp_ctx* c) { return c ? c->chvp_cached_passes : -1; }
long long promp_state_version(promp_ctx* c) { return c ? (long long)c->version_counter : -1; }
int promp_set_primal_cache(promp_ctx* c, int on) {
    if (!c) return fail(-1, "ctx is NULL");
    c->primal_cache = on < 0 ? -1 : on != 0;
    return 0;
}
int promp_set_learn_std(promp_ctx* c, int on) {
    if (!c) return fail(-1, "ctx is NULL");
    if (on && !c->learn_std) return fail(-3, "learn_std cannot be switched back on: the log_std step sizes were zeroed (set the step sizes again)");
    c->learn_std = on != 0;
    c->sizes_version = ++c->version_counter;
    return mask_log_std_step_sizes(c);
}
static int tasks_materialize(promp_ctx* c);
int promp_set_task_thetas(promp_ctx* c, const float* t) {
    if (!c) return fail(-1, "ctx is NULL");
    c->tasks_shared = false;
    return params_in(c, c->theta_tasks, t, (size_t)c->d.n_tasks);
}
int promp_get_task_thetas(promp_ctx* c, float* t) {
    if (!c) return fail(-1, "ctx is NULL");
    if (tasks_materialize(c)) return -2;
    return params_out(c, t, c->theta_tasks, (size_t)c->d.n_tasks);
}

int promp_set_adam_state(promp_ctx* c, const float* m, const float* v, int64_t t) {
    if (!c) return fail(-1, "ctx is NULL");
    if (params_in(c, c->adam_m, m, 1) || params_in(c, c->adam_v, v, 1)) return -2;
    c->adam_t = t;
    return 0;
}
int promp_get_adam_state(promp_ctx* c, float* m, float* v, int64_t* t) {
    if (!c) return fail(-1, "ctx is NULL");
    if (m && params_out(c, m, c->adam_m, 1)) return -2;
    if (v && params_out(c, v, c->adam_v, 1)) return -2;
    if (t) *t = c->adam_t;
    return 0;
}

// MetaPolicy.switch_to_pre_update (policies/base.py:173-179): every task's parameters are the meta-parameters again.  Nothing is
// launched here: the inner step reads theta with a task stride of zero, and the per-task copies are only written
// (tasks_materialize) for the entry points that hand them out or index them per task.
int promp_switch_to_pre_update(promp_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    c->tasks_shared = true;
    return 0;
}
static int tasks_materialize(promp_ctx* c) {
    if (!c->tasks_shared) return 0;
    PROMP_LAUNCH(k_replicate, dim3((c->NP + 255) / 256), 256, 0, c->stream, c->theta_tasks, (const float*)c->theta, c->NP, c->d.n_tasks);
    HIPCHECK(hipGetLastError());
    c->tasks_shared = false;
    return 0;
}

int promp_inner_adapt(promp_ctx* c, int step, int inner_kind) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    // pre-update mode: all tasks start from theta itself (stride 0); the step writes every task's row of theta_tasks
    const float* cur = c->tasks_shared ? c->theta : c->theta_tasks;
    const long long st = c->tasks_shared ? 0 : c->NP;
    // From the meta-parameters on step 0 this IS the first inner pass of the meta-objective (pro_mp.py:113-128 rebuilds what
    // base.py:217-242 just ran): leave theta', the scalars and the primal cache where the first epoch looks for them.  The two
    // differ only in log_std entries below log(min_std) (raw here, gaussian_mlp_policy.py:182; clipped there, :71,163), which
    // enqueue_meta checks before it trusts the result.
    const bool leave = c->reuse_adapt && c->tasks_shared && step == 0 &&
                       (inner_kind == PROMP_INNER_RATIO || inner_kind == PROMP_INNER_LOGLIK);
    if (step == 0) c->adapt0.valid = false;       // (inner steps on later sampling steps touch nothing the record stands for)
    bool cached = false;
    if (leave) {
        const size_t MNP = (size_t)c->d.n_tasks * c->NP;
        const bool worth = primal_cache_worth(c, S.n_rows);
        cached = worth && !c->wide && policy_shape_chain(&c->d);      // (the cooperative kernels keep no primal cache: theta' and the scalars only)
        if (cached && !S.hcache &&
            dev_alloc(&S.hcache, ((size_t)c->d.max_rows + 16 * (size_t)c->d.n_tasks) * chain_cache_row(c->d.hidden1, c->d.hidden2))) return -2;
        c->pass_next2 = c->chain + MNP;
        c->pass_scal2 = c->scal_inner;
        c->pass_cache = cached ? 1 : 0;
    }
    c->tasks_shared = false;
    const int rc = launch_pass(c, S, false, cur, st, loss_kind_inner(inner_kind), 0.f, 0, 0.f, false, RED_STEP, cur, st, c->theta_tasks, c->scal_tmp);
    c->pass_next2 = nullptr; c->pass_scal2 = nullptr; c->pass_cache = 0;
    if (rc) return rc;
    if (leave) {
        c->adapt0.valid = true; c->adapt0.theta_version = c->theta_version; c->adapt0.data_version = S.data_version;
        c->adapt0.sizes_version = c->sizes_version; c->adapt0.inner_kind = inner_kind; c->adapt0.cached = cached;
        c->adapt0.min_log_std = c->min_log_std; c->adapt0.learn_std = c->learn_std;
    }
    return 0;
}

int promp_policy_forward(promp_ctx* c, const float* obs, int batch, float* mean_out) {
    if (!c || !obs || !mean_out) return fail(-1, "NULL argument");
    if (batch < 1) return fail(-1, "batch must be positive");
    const int M = c->d.n_tasks, O = c->d.obs_dim, A = c->d.act_dim;
    const size_t n_obs = (size_t)M * batch * O, n_out = (size_t)M * batch * A;
    const size_t n_scr = c->generic ? (size_t)M * batch * 2 * c->g_maxw : 0;
    if (n_obs + n_out + n_scr > c->fwd_capacity) {
        if (c->fwd_buf) (void)hipFree(c->fwd_buf);
        c->fwd_buf = nullptr;
        c->fwd_capacity = 2 * (n_obs + n_out + n_scr);
        HIPCHECK(hipMalloc((void**)&c->fwd_buf, sizeof(float) * c->fwd_capacity));
    }
    float* d_obs = c->fwd_buf;
    float* d_out = c->fwd_buf + n_obs;
    HIPCHECK(hipMemcpyAsync(d_obs, obs, sizeof(float) * n_obs, hipMemcpyHostToDevice, c->stream));
    ForwardArgs f;
    if (tasks_materialize(c)) return -2;
    f.obs = d_obs; f.theta_tasks = c->theta_tasks; f.mean = d_out;
    f.B = batch; f.O = O; f.A = A; f.H1 = c->d.hidden1; f.H2 = c->d.hidden2;
    if (c->generic) {
        GenForwardArgs gf;
        gf.obs = d_obs; gf.theta_tasks = c->theta_tasks; gf.mean = d_out; gf.scratch = d_out + n_out;
        gf.B = batch; gf.NP = c->NP; gf.n_lin = c->n_lin; gf.maxw = c->g_maxw; gf.act_kind = gen_act_kinds(&c->d);
        for (int l = 0; l < c->n_lin; ++l) gf.lin[l] = c->lin[l];
        PROMP_LAUNCH(k_gen_policy_forward, dim3(M), 256, 0, c->stream, gf);
    } else
    PROMP_LAUNCH(k_policy_forward, dim3(M), 256, 0, c->stream, f);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(mean_out, d_out, sizeof(float) * n_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// fixed-length layout of a device-side rollout: path p of task i is rows [(i B + p) T, (i B + p + 1) T)
static int begin_fixed_rollout(promp_ctx* c, int step, int B, int T) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (B < 1 || T < 1) return fail(-1, "envs_per_task and path_length must be positive");
    const int M = c->d.n_tasks;
    const long long rows = (long long)M * B * T;
    if (rows > c->d.max_rows) return fail(-1, "rollout of %lld rows exceeds max_rows = %d", rows, c->d.max_rows);
    std::vector<int32_t> tpo(M + 1), pro((size_t)M * B + 1);
    for (int i = 0; i <= M; ++i) tpo[i] = i * B;
    for (int p = 0; p <= M * B; ++p) pro[p] = p * T;
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (set_step_layout(c, S, c->stream, false, M * B, tpo.data(), pro.data())) return -2;
    S.has_policy = true;
    S.ls_per_row = 0;
    S.rollout_B = B; S.rollout_T = T;
    return 0;
}

static int ensure_rollout_buf(promp_ctx* c, size_t need) {
    if (need > c->rollout_capacity) {
        if (c->rollout_buf) (void)hipFree(c->rollout_buf);
        c->rollout_buf = nullptr;
        c->rollout_capacity = 2 * need;
        HIPCHECK(hipMalloc((void**)&c->rollout_buf, c->rollout_capacity));
    }
    return 0;
}

int promp_rollout_point_env(promp_ctx* c, int step, int envs_per_task, int path_length, const double* goals,
                            const double* start, const float* noise, const promp_point_env_opts* o) {
    if (!c || !goals || !start || !o) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (c->d.obs_dim != 2 || c->d.act_dim != 2) return fail(-1, "the point environment has obs_dim = act_dim = 2 (context: %d, %d)", c->d.obs_dim, c->d.act_dim);
    if (o->reward_type < 0 || o->reward_type > 2) return fail(-1, "unknown reward type %d", o->reward_type);
    if (begin_fixed_rollout(c, step, envs_per_task, path_length)) return -2;
    const int M = c->d.n_tasks, B = envs_per_task, T = path_length;
    const long long rows = (long long)M * B * T;
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    const size_t need = sizeof(double) * ((size_t)M * 2 + (size_t)M * B * 2) + sizeof(float) * (size_t)rows * 2;
    if (ensure_rollout_buf(c, need)) return -2;
    double* d_goals = (double*)c->rollout_buf;
    double* d_start = d_goals + (size_t)M * 2;
    float* d_noise = (float*)(d_start + (size_t)M * B * 2);
    hipStream_t st = c->stream;
    HIPCHECK(hipMemcpyAsync(d_goals, goals, sizeof(double) * M * 2, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_start, start, sizeof(double) * M * B * 2, hipMemcpyHostToDevice, st));
    if (noise) HIPCHECK(hipMemcpyAsync(d_noise, noise, sizeof(float) * rows * 2, hipMemcpyHostToDevice, st));
    PointRolloutArgs a;
    if (tasks_materialize(c)) return -2;
    a.theta_tasks = c->theta_tasks; a.NP = c->NP; a.H1 = c->d.hidden1; a.H2 = c->d.hidden2;
    a.B = B; a.T = T; a.goals = d_goals; a.start = d_start; a.noise = noise ? d_noise : nullptr;
    a.seed = o->seed; a.stream = (unsigned)step;
    a.obs = S.obs; a.act = S.act; a.rew = S.rew; a.mean = S.old_mean; a.old_ls = S.old_ls;
    a.clip_infos = o->clip_infos; a.min_log_std = c->min_log_std;
    a.normalization_scale = o->normalization_scale; a.max_step = o->max_step; a.reward_type = o->reward_type; a.sparse_radius = o->sparse_radius;
    if (c->generic) {          // any layer table: one workgroup per environment (promp_kernels_generic.h)
        GenPointRolloutArgs g;
        g.p = a; g.n_lin = c->n_lin; g.act_kind = gen_act_kinds(&c->d);
        for (int l = 0; l < c->n_lin; ++l) g.lin[l] = c->lin[l];
        PROMP_LAUNCH(k_gen_point_rollout, dim3(B, M), 256, gen_rollout_smem(2), st, g);
    } else
    PROMP_LAUNCH(k_point_rollout, dim3(M), 64, 0, st, a);
    HIPCHECK(hipGetLastError());
    S.obs_range_valid = false;
    return 0;
}

int promp_begin_rollout(promp_ctx* c, int step, int envs_per_task, int path_length) {
    if (!c) return fail(-1, "ctx is NULL");
    if (begin_fixed_rollout(c, step, envs_per_task, path_length)) return -2;
    c->steps[step].rollout_ragged = false;
    return 0;
}

int promp_begin_collection(promp_ctx* c, int step, int envs_per_task, int max_steps) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (envs_per_task < 1 || max_steps < 1) return fail(-1, "envs_per_task and max_steps must be positive");
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    const size_t need = (size_t)max_steps * c->d.n_tasks * envs_per_task * (c->d.obs_dim + 2 * c->d.act_dim);
    if (need > c->stage_capacity) {
        if (c->stage_rows) (void)hipFree(c->stage_rows);
        c->stage_rows = nullptr;
        c->stage_capacity = need;
        HIPCHECK(hipMalloc((void**)&c->stage_rows, sizeof(float) * need));
    }
    S.rollout_B = envs_per_task; S.rollout_T = max_steps; S.rollout_ragged = true;
    S.has_policy = false; S.processed = false; S.has_adv = false;
    return 0;
}

int promp_end_collection(promp_ctx* c, int step, int n_paths, const int32_t* task_path_offsets, const int32_t* path_env,
                         const int32_t* path_start, const int32_t* path_len, const float* rewards) {
    if (!c || !task_path_offsets || !path_env || !path_start || !path_len || !rewards) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    if (!S.rollout_ragged || S.rollout_B < 1) return fail(-3, "promp_begin_collection has not been called for step %d", step);
    const int M = c->d.n_tasks, B = S.rollout_B, O = c->d.obs_dim, A = c->d.act_dim;
    if (n_paths < 1 || n_paths > c->d.max_paths) return fail(-1, "%d paths outside [1, max_paths = %d]", n_paths, c->d.max_paths);
    std::vector<int32_t> pro((size_t)n_paths + 1, 0);
    for (int p = 0; p < n_paths; ++p) {
        if (path_len[p] < 1 || path_start[p] < 0 || path_start[p] + path_len[p] > S.rollout_T || path_env[p] < 0 || path_env[p] >= M * B)
            return fail(-1, "path %d (environment %d, steps [%d, %d)) lies outside the collection", p, path_env[p], path_start[p], path_start[p] + path_len[p]);
        pro[p + 1] = pro[p] + path_len[p];
    }
    if (pro[n_paths] > c->d.max_rows) return fail(-1, "%d collected rows exceed max_rows = %d", pro[n_paths], c->d.max_rows);
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (set_step_layout(c, S, c->stream, false, n_paths, task_path_offsets, pro.data())) return -2;
    // the finished episodes: staging rows -> slab rows in path order
    if (ensure_rollout_buf(c, sizeof(int32_t) * 2 * (size_t)n_paths)) return -2;
    int32_t* d_env = (int32_t*)c->rollout_buf;
    int32_t* d_start = d_env + n_paths;
    HIPCHECK(hipMemcpyAsync(d_env, path_env, sizeof(int32_t) * n_paths, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(d_start, path_start, sizeof(int32_t) * n_paths, hipMemcpyHostToDevice, c->stream));
    GatherPathsArgs g;
    const size_t n_rows = (size_t)S.rollout_T * M * B;
    g.obs_in = c->stage_rows; g.act_in = c->stage_rows + n_rows * O; g.mean_in = g.act_in + n_rows * A;
    g.obs = S.obs; g.act = S.act; g.mean = S.old_mean;
    g.path_env = d_env; g.path_start = d_start; g.path_row_offsets = S.path_row_offsets;
    g.n_envs = M * B; g.O = O; g.A = A;
    PROMP_LAUNCH(k_gather_paths, dim3(n_paths), 256, 0, c->stream, g);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(S.rew, rewards, sizeof(float) * pro[n_paths], hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    S.has_policy = true; S.ls_per_row = 0; S.has_rew64 = false; S.processed = false; S.has_adv = false;
    S.rollout_ragged = false; S.rollout_B = 0;
    S.obs_range_valid = false;
    return 0;
}

int promp_policy_step(promp_ctx* c, int step, int t, const float* obs, uint64_t seed, int clip_infos, float* actions_out) {
    if (!c || !obs || !actions_out) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (S.rollout_B < 1) return fail(-3, "promp_begin_rollout has not been called for step %d", step);
    const int M = c->d.n_tasks, B = S.rollout_B, T = S.rollout_T, O = c->d.obs_dim, A = c->d.act_dim;
    if (t < 0 || t >= T) return fail(-1, "time step %d outside the horizon %d", t, T);
    const size_t n_obs = (size_t)M * B * O, n_act = (size_t)M * B * A;
    if (ensure_rollout_buf(c, sizeof(float) * (n_obs + n_act))) return -2;
    float* d_obs = (float*)c->rollout_buf;
    float* d_act = d_obs + n_obs;
    hipStream_t st = c->stream;
    HIPCHECK(hipMemcpyAsync(d_obs, obs, sizeof(float) * n_obs, hipMemcpyHostToDevice, st));
    PolicyStepArgs a;
    if (tasks_materialize(c)) return -2;
    a.obs_in = d_obs; a.theta_tasks = c->theta_tasks;
    a.obs = S.obs; a.act = S.act; a.mean = S.old_mean; a.old_ls = S.old_ls; a.actions_out = d_act;
    a.row_env_stride = T; a.row_t_stride = 1;
    if (S.rollout_ragged) {       // (s, env) rows of the staging area
        const size_t n_rows = (size_t)T * M * B;
        a.obs = c->stage_rows; a.act = c->stage_rows + n_rows * O; a.mean = a.act + n_rows * A;
        a.row_env_stride = 1; a.row_t_stride = (long long)M * B;
    }
    a.B = B; a.T = T; a.t = t; a.O = O; a.A = A; a.H1 = c->d.hidden1; a.H2 = c->d.hidden2; a.NP = c->NP;
    a.clip_infos = clip_infos; a.min_log_std = c->min_log_std;
    a.seed = seed; a.stream = (unsigned)step;
    if (c->generic) {          // any layer table: one workgroup per environment (promp_kernels_generic.h)
        GenPolicyStepArgs g;
        g.p = a; g.n_lin = c->n_lin; g.act_kind = gen_act_kinds(&c->d);
        for (int l = 0; l < c->n_lin; ++l) g.lin[l] = c->lin[l];
        PROMP_LAUNCH(k_gen_policy_step, dim3(B, M), 256, gen_rollout_smem(O), st, g);
    } else
    PROMP_LAUNCH(k_policy_step, dim3((B + 63) / 64, M), 64, 0, st, a);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(actions_out, d_act, sizeof(float) * n_act, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    S.obs_range_valid = false;
    return 0;
}

int promp_set_rewards(promp_ctx* c, int step, const float* rew) {
    if (!c || !rew) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    HIPCHECK(hipMemcpyAsync(S.rew, rew, sizeof(float) * S.n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    S.processed = false;
    S.has_rew64 = false;
    return 0;
}

int promp_set_rewards_f64(promp_ctx* c, int step, const double* rew) {
    if (!c || !rew) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    if (!S.rew64 && dev_alloc(&S.rew64, (size_t)c->d.max_rows)) return -2;
    std::vector<float> r32((size_t)S.n_rows);
    for (int i = 0; i < S.n_rows; ++i) r32[i] = (float)rew[i];
    HIPCHECK(hipMemcpyAsync(S.rew64, rew, sizeof(double) * S.n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(S.rew, r32.data(), sizeof(float) * S.n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    S.processed = false;
    S.has_rew64 = true;
    return 0;
}

int promp_download_step(promp_ctx* c, int step, float* obs, float* act, float* rew, float* old_mean, float* old_log_std) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    if (S.n_rows == 0) return fail(-3, "step %d has no data", step);
    const size_t R = S.n_rows, O = c->d.obs_dim, A = c->d.act_dim, M = c->d.n_tasks;
    hipStream_t st = c->stream;
    if (obs) HIPCHECK(hipMemcpyAsync(obs, S.obs, sizeof(float) * R * O, hipMemcpyDeviceToHost, st));
    if (rew) HIPCHECK(hipMemcpyAsync(rew, S.rew, sizeof(float) * R, hipMemcpyDeviceToHost, st));
    if ((act || old_mean || old_log_std) && !S.has_policy) return fail(-3, "step %d holds no actions / agent_infos", step);
    if (act) HIPCHECK(hipMemcpyAsync(act, S.act, sizeof(float) * R * A, hipMemcpyDeviceToHost, st));
    if (old_mean) HIPCHECK(hipMemcpyAsync(old_mean, S.old_mean, sizeof(float) * R * A, hipMemcpyDeviceToHost, st));
    if (old_log_std) HIPCHECK(hipMemcpyAsync(old_log_std, S.old_ls, sizeof(float) * (S.ls_per_row ? R : M) * A, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

int promp_meta_grad(promp_ctx* c, float clip_eps, const float* eta, int inner_kind, int outer_kind, float* grad_out,
                    float* stats_out) {
    if (!c || !eta) return fail(-1, "NULL argument");
    if (upload_eta(c, eta)) return -2;
    if (enqueue_meta(c, clip_eps, eta, inner_kind, outer_kind, true, false, 0.f)) return -2;
    if (grad_out && params_out(c, grad_out, c->grad_mean, 1)) return -2;
    if (stats_out && copy_out(c, stats_out, c->stats, c->d.num_inner_steps + 2)) return -2;
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// Exact Hessian-vector product of the TRPO constraint, mean_i KL(pi_old || pi_{theta'_i(theta)}) on the last step's samples,
// through the adaptation: with J_k = I - diag(alpha) H_k(theta_k) (H_k = Hessian of task i's inner objective on step k),
//     H v = mean_i  J_0^T ... J_{K-1}^T  H_KL(theta_K)  J_{K-1} ... J_0  v .
// The terms that differentiate the J_k are contracted with grad_{theta_K} KL, which is zero where TRPO builds the
// product: at the parameters the samples were drawn with (old distribution == adapted policy).  2K + 1 R-operator passes.
// (the direction is in c->grad_mean, the sums over the tasks -- all-reduced -- are left in c->red)
static int enqueue_constraint_hvp(promp_ctx* c, int inner_kind, int refresh_chain) {
    if (sharded_without_comm(c))
        return fail(-3, "this context holds %d of %d tasks and has no communicator: the product would be this rank's share only "
                        "(promp_comm_init first)", c->d.n_tasks, c->d.n_tasks_global);
    const int K = c->d.num_inner_steps, M = c->d.n_tasks, NP = c->NP;
    const size_t MNP = (size_t)M * NP;
    for (int k = 0; k <= K; ++k) {
        if (c->steps[k].n_rows == 0) return fail(-3, "step %d has no data", k);
        if (join_side(c, c->steps[k])) return -2;
    }
    if (!c->wbuf && dev_alloc(&c->wbuf, MNP)) return -2;
    auto theta_of = [&](int k, long long* stride) -> const float* {
        *stride = (k == 0) ? 0 : NP;
        return (k == 0) ? c->theta : c->chain + (size_t)k * MNP;
    };
    const int lk = loss_kind_inner(inner_kind);
    long long st = 0;
    // Primal caches: the products of one conjugate-gradient solve all run at the same parameters on the same slabs, so the
    // passes that refresh the chain store their activations (and one extra storing pass with the KL objective covers step K);
    // every R-operator pass below then reads them back instead of recomputing layers 1 and 2.  Same worth-it rule as the
    // meta-objective's cache (enqueue_meta).
    bool use_cache = !c->wide && policy_shape_chain(&c->d);      // (the cooperative kernels keep no primal cache)
    for (int k = 0; k <= K && use_cache; ++k)
        use_cache = primal_cache_worth(c, c->steps[k].n_rows);
    if (refresh_chain) {
        c->chvp.valid = false;
        for (int k = 0; k <= K && use_cache; ++k)
            if (!c->steps[k].hcache &&
                dev_alloc(&c->steps[k].hcache, ((size_t)c->d.max_rows + 16 * (size_t)M) * chain_cache_row(c->d.hidden1, c->d.hidden2))) return -2;
        for (int k = 0; k < K; ++k) {
            const float* th = theta_of(k, &st);
            if (k == 0 && adapt0_stands(c, inner_kind, use_cache)) {       // promp_inner_adapt has just run exactly this pass
                c->adapt_passes_skipped += 1;
                continue;
            }
            c->pass_cache = use_cache ? 1 : 0;
            const int rc = launch_pass(c, c->steps[k], false, th, st, lk, 0.f, k == 0, 0.f, false, RED_STEP, th, st,
                                       c->chain + (size_t)(k + 1) * MNP, c->scal_inner + (size_t)k * M * 2);
            c->pass_cache = 0;
            if (rc) return -2;
        }
        if (use_cache) {
            const float* th = theta_of(K, &st);
            c->pass_cache = 1;
            const int rc = launch_pass(c, c->steps[K], false, th, st, LOSS_KL, 0.f, K == 0, 0.f, false, RED_PLAIN, nullptr, 0, nullptr, c->scal_tmp);
            c->pass_cache = 0;
            if (rc) return -2;
            c->chvp.valid = true; c->chvp.theta_version = c->theta_version; c->chvp.sizes_version = c->sizes_version;
            c->chvp.inner_kind = inner_kind; c->chvp.min_log_std = c->min_log_std;
            for (int k = 0; k <= K; ++k) {
                c->chvp.data_version[k] = c->steps[k].data_version;
                c->chvp.tag[k] = c->steps[k].cache_tag;
            }
        }
    }
    const bool rec_ok = use_cache && c->chvp.valid && c->chvp.theta_version == c->theta_version && c->chvp.sizes_version == c->sizes_version &&
                        c->chvp.inner_kind == inner_kind && c->chvp.min_log_std == c->min_log_std;
    auto cache_ok = [&](int k) {
        return rec_ok && c->steps[k].hcache && c->chvp.data_version[k] == c->steps[k].data_version && c->chvp.tag[k] == c->steps[k].cache_tag;
    };
    PROMP_LAUNCH(k_replicate, dim3((NP + 255) / 256), 256, 0, c->stream, c->vbuf, c->grad_mean, NP, M);
    const dim3 eg((NP + 255) / 256, M);
    auto pass = [&](int k, int kind) -> int {
        const float* th = theta_of(k, &st);
        const bool cached = cache_ok(k);
        c->pass_cache = cached ? 2 : 0;
        c->chvp_cached_passes += cached ? 1 : 0;
        const int rc = launch_pass(c, c->steps[k], true, th, st, kind, 0.f, k == 0, 0.f, false, RED_PLAIN, nullptr, 0, nullptr, c->scal_tmp);
        c->pass_cache = 0;
        return rc;
    };
    for (int k = 0; k < K; ++k) {                      // u = J_{K-1} ... J_0 v
        if (pass(k, lk)) return -2;
        PROMP_LAUNCH(k_jstep, eg, 256, 0, c->stream, c->vbuf, c->wbuf, c->lam, c->step_sizes, NP, 0);
    }
    if (pass(K, LOSS_KL)) return -2;                   // w = H_KL(theta_K) u
    PROMP_LAUNCH(k_jstep, eg, 256, 0, c->stream, c->vbuf, c->wbuf, c->lam, c->step_sizes, NP, 1);
    for (int k = K - 1; k >= 0; --k) {                 // w = J_k^T w
        if (pass(k, lk)) return -2;
        PROMP_LAUNCH(k_jstep, eg, 256, 0, c->stream, c->vbuf, c->wbuf, c->lam, c->step_sizes, NP, 2);
    }
    HIPCHECK(hipGetLastError());
    FinalArgs f;
    f.lam = c->wbuf; f.NP = NP; f.K = K; f.n_tasks = M;
    f.scal_inner = c->scal_inner; f.scal_outer = c->scal_outer; f.red = c->red; f.want_grad = 1;
    PROMP_LAUNCH(k_reduce_final, dim3((NP + K + 2 + 63) / 64), 256, 0, c->stream, f);
    HIPCHECK(hipGetLastError());
    if (exchange_sums(c, c->red, (size_t)NP)) return -4;
    for (int k = 0; k <= K; ++k) c->steps[k].dirty = true;
    return 0;
}

int promp_constraint_hvp(promp_ctx* c, int inner_kind, const float* v, int refresh_chain, float* out) {
    if (!c || !v || !out) return fail(-1, "NULL argument");
    if (params_in(c, c->grad_mean, v, 1)) return -2;
    const int rc = enqueue_constraint_hvp(c, inner_kind, refresh_chain);
    if (rc) return rc;
    if (params_out(c, out, c->red, 1)) return -2;
    const float inv = 1.0f / (float)c->d.n_tasks_global;
    for (int j = 0; j < c->NPu; ++j) out[j] *= inv;
    return 0;
}

// ConjugateGradientOptimizer's solve with nothing crossing to the host between its products (conjugate_gradient_optimizer.py:325-354
// conjugate_gradients(), :59-104 FiniteDifferenceHvp.Hx / build_eval, :259-264 the solve and the closing product): see include/promp_hip.h.
int promp_cg_solve(promp_ctx* c, int inner_kind, const float* b, int cg_iters, float reg_coeff, float eps, int hvp_mode,
                   float residual_tol, float* x_out, double* xhx_out) {
    if (!c || !b || !x_out || !xhx_out) return fail(-1, "NULL argument");
    if (cg_iters < 0) return fail(-1, "cg_iters must not be negative");
    if (hvp_mode < 0 || hvp_mode > 2) return fail(-1, "hvp_mode %d unknown (0 symmetric differences, 1 one-sided, 2 exact)", hvp_mode);
    if (hvp_mode != 2 && !(eps > 0.f)) return fail(-1, "the finite differences need eps > 0");
    if (sharded_without_comm(c))
        return fail(-3, "this context holds %d of %d tasks and has no communicator: the products would be this rank's share only "
                        "(promp_comm_init first)", c->d.n_tasks, c->d.n_tasks_global);
    const int NP = c->NP;
    if (!c->cg_buf) {
        if (dev_alloc(&c->cg_buf, (size_t)7 * NP)) return -2;
        if (dev_alloc(&c->cg_scal, 4)) return -2;
    }
    float *x = c->cg_buf, *r = x + NP, *d = r + NP, *hd = d + NP, *ga = hd + NP, *th0 = ga + NP;
    if (params_in(c, r, b, 1)) return -2;                       // (blocking: b is the caller's)
    float eta[PROMP_ETA_MAX] = {};
    if (upload_eta(c, eta)) return -2;
    CgArgs a;
    a.g_ahead = ga; a.g_behind = nullptr; a.div_h = 1.f; a.mul_s = 1.f; a.reg = reg_coeff;
    a.x = x; a.r = r; a.d = d; a.hd = hd; a.scal = c->cg_scal; a.tol = residual_tol; a.n = NP; a.mode = 2;
    PROMP_LAUNCH(k_cg_step, dim3(1), 1024, 0, c->stream, a);      // x = 0, d = r = b, scal = {b.b, 0, 0, 0}
    HIPCHECK(hipGetLastError());
    const bool exact = hvp_mode == 2;
    const bool ls_known = c->ls_known;
    auto gradient_at = [&](float s, const float* v) -> int {      // the constraint's gradient at theta0 + s v -> c->grad_mean
        PROMP_LAUNCH(k_cg_displace, dim3((NP + 255) / 256), 256, 0, c->stream, c->theta, th0, v, s, NP);
        c->theta_version = ++c->version_counter;
        c->ls_known = false;
        return enqueue_meta(c, 0.f, eta, inner_kind, PROMP_OUTER_KL, true, false, 0.f);
    };
    if (!exact) {
        HIPCHECK(hipMemcpyAsync(th0, c->theta, sizeof(float) * NP, hipMemcpyDeviceToDevice, c->stream));
        if (hvp_mode == 1) {                                      // one-sided: the gradient at theta0 itself, once
            if (enqueue_meta(c, 0.f, eta, inner_kind, PROMP_OUTER_KL, true, false, 0.f)) return -2;
            HIPCHECK(hipMemcpyAsync(th0 + NP, c->grad_mean, sizeof(float) * NP, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    bool fresh = false;
    auto product = [&](const float* v, int mode) -> int {        // hd = (H + reg I) v and the vector updates of `mode`
        if (exact) {
            HIPCHECK(hipMemcpyAsync(c->grad_mean, v, sizeof(float) * NP, hipMemcpyDeviceToDevice, c->stream));
            const int rc = enqueue_constraint_hvp(c, inner_kind, fresh ? 0 : 1);
            if (rc) return rc;
            fresh = true;
            a.g_ahead = c->red; a.g_behind = nullptr; a.div_h = 1.f; a.mul_s = 1.0f / (float)c->d.n_tasks_global;
        } else {
            if (gradient_at(eps, v)) return -2;
            HIPCHECK(hipMemcpyAsync(ga, c->grad_mean, sizeof(float) * NP, hipMemcpyDeviceToDevice, c->stream));
            if (hvp_mode == 0) {
                if (gradient_at(-eps, v)) return -2;
                a.g_behind = c->grad_mean; a.div_h = 2.f * eps;
            } else {
                a.g_behind = th0 + NP; a.div_h = eps;
            }
            a.g_ahead = ga; a.mul_s = 1.f;
        }
        a.mode = mode;
        PROMP_LAUNCH(k_cg_step, dim3(1), 1024, 0, c->stream, a);
        HIPCHECK(hipGetLastError());
        return 0;
    };
    int rc = 0;
    for (int it = 0; it < cg_iters && !rc; ++it) rc = product(d, 0);
    if (!rc) rc = product(x, 1);                                  // x . (H + reg I) x: the step length's denominator
    if (!exact) {                                                 // the parameters are back at theta0 when the solve returns
        HIPCHECK(hipMemcpyAsync(c->theta, th0, sizeof(float) * NP, hipMemcpyDeviceToDevice, c->stream));
        c->theta_version = ++c->version_counter;
        c->ls_known = ls_known;
    }
    if (rc) return rc;
    if (params_out(c, x_out, x, 1)) return -2;
    double sc[4];
    HIPCHECK(hipMemcpyAsync(sc, c->cg_scal, sizeof sc, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    *xhx_out = sc[3];
    return 0;
}

int promp_adam_step(promp_ctx* c, float lr) {
    if (!c) return fail(-1, "ctx is NULL");
    // red still holds the (all-reduced) sums of the last promp_meta_grad
    AdamArgs ad;
    ad.theta = c->theta; ad.m = c->adam_m; ad.v = c->adam_v; ad.red = c->red; ad.grad_mean = c->grad_mean;
    ad.stats = c->stats; ad.NP = c->NP; ad.K = c->d.num_inner_steps;
    for (int k = 0; k < PROMP_ETA_MAX; ++k) ad.eta[k] = k < ad.K ? c->eta_last[k] : 0.f;
    ad.host_stats = nullptr; ad.host_seq = nullptr; ad.seq = 0;
    ad.inv_tasks = 1.0f / (float)c->d.n_tasks_global;
    ad.do_update = 1;
    ad.n_trainable = c->learn_std ? c->NP : c->NP - c->d.act_dim;
    ad.A = c->d.act_dim;
    c->theta_version = ++c->version_counter;
    c->ls_known = false;
    c->adam_t += 1;
    const double t = (double)c->adam_t;
    ad.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.9, t)));
    PROMP_LAUNCH(k_mean_adam, dim3((c->NP + 1 + 255) / 256), 256, 0, c->stream, ad);
    HIPCHECK(hipGetLastError());
    return 0;
}

int promp_optimize_begin(promp_ctx* c, int num_epochs, float lr, float clip_eps, const float* eta, int inner_kind, int outer_kind) {
    if (!c || !eta) return fail(-1, "NULL argument");
    if (num_epochs < 0) return fail(-1, "num_epochs must be >= 0");
    if (c->opt_pending) return fail(-1, "promp_optimize_begin: the previous optimisation has not been collected (promp_optimize_end)");
    if (sharded_without_comm(c))
        return fail(-3, "this context holds %d of %d tasks and has no communicator: promp_optimize would apply this rank's sums as if they "
                        "were the meta-batch's.  Attach one (promp_comm_init), or run the exchange yourself: promp_meta_grad -> "
                        "promp_reduced_get -> all-reduce -> promp_reduced_set -> promp_adam_step", c->d.n_tasks, c->d.n_tasks_global);
    if (upload_eta(c, eta)) return -2;
    const int K = c->d.num_inner_steps;
    for (int e = 0; e < num_epochs; ++e) {
        // the loss evaluated by the first epoch, before its update, stays on the device (second statistics slot) until the
        // end: a download here would hold the host back until the epoch has run, and the next one would start late
        c->stats_slot = (e == 0) ? 1 : 0;
        const int rc = enqueue_meta(c, clip_eps, eta, inner_kind, outer_kind, true, true, lr);
        c->stats_slot = 0;
        if (rc) return -2;
    }
    // compute_stats.  Its final launch stores both statistics slots into page-locked host memory and then a sequence number
    // (system-scope release): no copy operation and no event on the queue, and nothing waits here -- the host can enqueue
    // the next batch's sample processing while this optimisation still runs
    c->stats_seq += 1;
    c->publish_next = true;
    const int rc_stats = enqueue_meta(c, clip_eps, eta, inner_kind, outer_kind, false, false, 0.f);
    c->publish_next = false;
    if (rc_stats) return -2;
    (void)K;
    c->opt_pending = true;
    c->opt_epochs = num_epochs;
    c->opt_theta_version = c->theta_version;
    return 0;
}

int promp_optimize_end(promp_ctx* c, float* loss_before, float* stats_after) {
    if (!c) return fail(-1, "ctx is NULL");
    if (!c->opt_pending) return fail(-1, "promp_optimize_end without promp_optimize_begin");
    c->opt_pending = false;
    // poll the sequence number (the stream is queried now and then: a failed launch must not turn into an endless wait)
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n(c->stats_seq_host, __ATOMIC_ACQUIRE) == c->stats_seq) break;
        if ((spins & 0xfff) == 0xfff) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(c->stats_seq_host, __ATOMIC_ACQUIRE) == c->stats_seq) break;
                return fail(-2, "promp_optimize_end: the stream drained without publishing the statistics");
            }
            if (q != hipErrorNotReady) return fail(-2, "promp_optimize_end: %s", hipGetErrorString(q));
        }
    }
    const int K = c->d.num_inner_steps;
    // the smallest log_std entry the publishing launch saw: current only if nothing replaced theta between begin and end
    // (promp_set_theta / promp_adam_step in the window leave it unknown, as they do anywhere else)
    if (c->theta_version == c->opt_theta_version) {
        c->ls_min = c->stats_host[2 * (K + 2)];
        c->ls_known = true;
    }
    if (stats_after) memcpy(stats_after, c->stats_host, sizeof(float) * (K + 2));
    if (loss_before) *loss_before = c->opt_epochs > 0 ? c->stats_host[K + 2] : c->stats_host[0];
    return 0;
}

int promp_optimize(promp_ctx* c, int num_epochs, float lr, float clip_eps, const float* eta, int inner_kind, int outer_kind,
                   float* loss_before, float* stats_after) {
    if (promp_optimize_begin(c, num_epochs, lr, clip_eps, eta, inner_kind, outer_kind)) return -2;
    return promp_optimize_end(c, loss_before, stats_after);
}

int promp_eval_loss_grad(promp_ctx* c, int step, int kind, float clip_eps, int clip_ls, float* grads_out, float* loss_out,
                         float* kl_out) {
    if (!c) return fail(-1, "ctx is NULL");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    if (kind < 0 || kind > 3) return fail(-1, "unknown objective kind %d", kind);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    const int M = c->d.n_tasks;
    if (tasks_materialize(c)) return -2;
    if (launch_pass(c, S, false, c->theta_tasks, c->NP, kind, clip_eps, clip_ls, 0.f, false, RED_PLAIN, nullptr, 0, nullptr, c->scal_tmp)) return -2;
    if (grads_out && params_out(c, grads_out, c->lam, (size_t)M)) return -2;
    std::vector<float> sc((size_t)M * 2);
    if (copy_out(c, sc.data(), c->scal_tmp, sc.size())) return -2;
    for (int i = 0; i < M; ++i) {
        if (loss_out) loss_out[i] = sc[2 * i];
        if (kl_out) kl_out[i] = sc[2 * i + 1];
    }
    return 0;
}

int promp_eval_hvp(promp_ctx* c, int step, int inner_kind, int clip_ls, float klw, const float* v, float* out) {
    if (!c || !v || !out) return fail(-1, "NULL argument");
    if (step < 0 || step > c->d.num_inner_steps) return fail(-1, "step %d out of range", step);
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    const int M = c->d.n_tasks, NP = c->NP;
    if (params_in(c, c->vbuf, v, (size_t)M)) return -2;
    HIPCHECK(hipMemsetAsync(c->lam, 0, sizeof(float) * (size_t)M * NP, c->stream));
    if (tasks_materialize(c)) return -2;
    if (launch_pass(c, S, true, c->theta_tasks, NP, loss_kind_inner(inner_kind), 0.f, clip_ls, klw, false, RED_PLAIN, nullptr, 0, nullptr, c->scal_tmp)) return -2;
    return params_out(c, out, c->lam, (size_t)M);
}

int promp_comm_unique_id(void* id_out, size_t id_bytes) {
    if (!id_out || id_bytes < sizeof(ncclUniqueId)) return fail(-1, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(-4, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
    memset(id_out, 0, id_bytes);
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int promp_comm_init(promp_ctx* c, int rank, int nranks, const void* id, size_t id_bytes) {
    if (!c) return fail(-1, "ctx is NULL");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "bad rank %d / nranks %d", rank, nranks);
    if (!id || id_bytes < sizeof(ncclUniqueId)) return fail(-1, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    HIPCHECK(hipSetDevice(c->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) return fail(-4, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    c->rank = rank;
    c->nranks = nranks;
    return 0;
}

int promp_comm_info(promp_ctx* c, int32_t* nranks, int32_t* rank, int32_t* fixed_order, char* bus_id_out, size_t bus_id_bytes) {
    if (!c) return fail(-1, "ctx is NULL");
    int n = 1, r = 0;
    if (c->comm) {        // what the communicator itself says, not what the caller passed to promp_comm_init
        ncclResult_t e = ncclCommCount(c->comm, &n);
        if (e == ncclSuccess) e = ncclCommUserRank(c->comm, &r);
        if (e != ncclSuccess) return fail(-4, "ncclCommCount / ncclCommUserRank failed: %s", ncclGetErrorString(e));
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (fixed_order) *fixed_order = c->fixed_order ? 1 : 0;
    if (bus_id_out && bus_id_bytes) {
        bus_id_out[0] = 0;
        HIPCHECK(hipDeviceGetPCIBusId(bus_id_out, (int)bus_id_bytes, c->device));
    }
    return 0;
}

int promp_comm_fixed_order(promp_ctx* c, int on) {
    if (!c) return fail(-1, "ctx is NULL");
    c->fixed_order = on != 0;
    return 0;
}

int promp_comm_split_path(promp_ctx* c, int on) {
    if (!c) return fail(-1, "ctx is NULL");
    c->force_split = on != 0;
    return 0;
}

int promp_comm_move(promp_ctx* dst, promp_ctx* src) {
    if (!dst || !src) return fail(-1, "ctx is NULL");
    if (dst == src) return 0;
    if (dst->comm) return fail(-3, "destination context already holds a communicator");
    if (dst->device != src->device) return fail(-1, "contexts live on different devices (%d, %d)", dst->device, src->device);
    HIPCHECK(hipStreamSynchronize(src->stream));    // nothing of the old context may still be using it
    dst->comm = src->comm;
    src->comm = nullptr;
    dst->fixed_order = src->fixed_order;
    dst->rank = src->rank; dst->nranks = src->nranks;
    src->rank = 0; src->nranks = 1;
    return 0;
}

int promp_reduced_get(promp_ctx* c, float* out) {
    if (!c || !out) return fail(-1, "NULL argument");
    // [grad Theta | K + 2 scalars]: the gradient part in the caller's layout, like every parameter vector that crosses the ABI
    if (params_out(c, out, c->red, 1)) return -2;
    return copy_out(c, out + c->NPu, c->red + c->NP, (size_t)c->d.num_inner_steps + 2);
}
int promp_reduced_set(promp_ctx* c, const float* in) {
    if (!c || !in) return fail(-1, "NULL argument");
    if (params_in(c, c->red, in, 1)) return -2;
    return copy_in(c, c->red + c->NP, in + c->NPu, (size_t)c->d.num_inner_steps + 2);
}

int promp_allreduce_f64(promp_ctx* c, double* buf, int n, int op) {
    if (!c || !buf) return fail(-1, "NULL argument");
    if (n < 1 || n > 64) return fail(-1, "n must be in [1,64]");
    if (c->nranks == 1) return 0;
    HIPCHECK(hipMemcpyAsync(c->red64, buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = ncclAllReduce(c->red64, c->red64, (size_t)n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return fail(-4, "ncclAllReduce failed: %s", ncclGetErrorString(r));
    HIPCHECK(hipMemcpyAsync(buf, c->red64, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// Developer tooling (not part of include/promp_hip.h): run one chain-kernel launch on the current per-task parameters
// with the cycle stamps of workgroup 0 enabled, and return the 256 raw stamps (needs a -DPROMP_DEV_STAMPS build).
int promp_debug_phase_stamps(promp_ctx* c, int step, int hvp, unsigned long long* out) {
    if (!c || !out) return fail(-1, "NULL argument");
    if (!PROMP_STAMPS_ON) return fail(-3, "phase stamps need a build with -DPROMP_DEV_STAMPS");
    StepData& S = c->steps[step];
    StepScope scope_(c, S, false);
    if (scope_.rc) return -2;
    if (tasks_materialize(c)) return -2;
    if (hvp == 2) {               // the cache-reading R-operator pass: fill the step's primal cache first (unstamped)
        if (c->wide || !policy_shape_chain(&c->d)) return fail(-1, "no primal cache for this shape");
        if (!S.hcache && dev_alloc(&S.hcache, ((size_t)c->d.max_rows + 16 * (size_t)c->d.n_tasks) * chain_cache_row(c->d.hidden1, c->d.hidden2))) return -2;
        c->pass_cache = 1;
        const int rc0 = launch_pass(c, S, false, c->theta_tasks, c->NP, LOSS_RATIO, 0.3f, 0, 0.f, false, RED_PLAIN, nullptr, 0, nullptr, c->scal_tmp);
        c->pass_cache = 0;
        if (rc0) return rc0;
    }
    HIPCHECK(hipMemsetAsync(c->dbg, 0, sizeof(unsigned long long) * (256 + 4 * 1024), c->stream));
    c->dbg_enabled = true;
    c->pass_cache = hvp == 2 ? 2 : 0;
    const int rc = launch_pass(c, S, hvp != 0, c->theta_tasks, c->NP, LOSS_RATIO, 0.3f, 0, 0.f, false, RED_PLAIN, nullptr, 0, nullptr, c->scal_tmp);
    c->pass_cache = 0;
    c->dbg_enabled = false;
    if (rc) return rc;
    HIPCHECK(hipMemcpyAsync(out, c->dbg, sizeof(unsigned long long) * (256 + 4 * 1024), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int promp_prof_enable(promp_ctx* c, int on) {
    if (!c) return fail(-1, "ctx is NULL");
    if (prof_collect(c)) return -2;
    c->prof = on != 0;
    if (on)
        for (auto& s : c->prof_slots) { s.total_ms = 0.0; s.launches = 0; s.rows = 0; }
    return 0;
}

int promp_prof_read(promp_ctx* c, int id, double* total_ms, int64_t* launches, int64_t* rows) {
    if (!c) return fail(-1, "ctx is NULL");
    if (id < 0 || id >= PROMP_KERNEL_COUNT) return fail(-1, "kernel id %d out of range", id);
    if (prof_collect(c)) return -2;
    if (total_ms) *total_ms = c->prof_slots[id].total_ms;
    if (launches) *launches = c->prof_slots[id].launches;
    if (rows) *rows = c->prof_slots[id].rows;
    return 0;
}

int promp_split_events(promp_ctx* c, int64_t* out2) {
    if (!c || !out2) return fail(-1, "NULL argument");
    int h[2];
    HIPCHECK(hipMemcpyAsync(h, c->split_events, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemsetAsync(c->split_events, 0, sizeof h, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 2; ++i) out2[i] = h[i];
    return 0;
}

int promp_device_info(promp_ctx* c, char* name_out, size_t name_bytes, int32_t* n_cus, int32_t* clock_mhz) {
    if (!c) return fail(-1, "ctx is NULL");
    if (name_out && name_bytes) snprintf(name_out, name_bytes, "%s", c->dev_name);
    if (n_cus) *n_cus = c->n_cus;
    if (clock_mhz) *clock_mhz = c->clock_mhz;
    return 0;
}

}  // extern "C"
