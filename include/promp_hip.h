/*
 * promp_hip.h -- C ABI of libpromp_hip.so, the MI355X (gfx950) implementation of the ProMP
 * meta-RL training hot path:
 *
 *     MetaSampleProcessor.process_samples -> MAMLAlgo._adapt -> ProMP.optimize_policy
 *
 * The reference (jonasrothfuss/ProMP) is pure Python/TF1 and has no FFI; this header is the
 * boundary a maintainer would bind with ctypes from the reference's plugin classes (see
 * INTEGRATION.md).  Each entry point cites the reference code it replaces, paths relative to
 * meta_policy_search/ in the reference tree.
 *
 * Conventions
 *   - return 0 on success, negative on failure; promp_last_error() has the message
 *     (thread-local, valid until the next call on the thread).
 *   - the caller owns every host buffer (C-contiguous); the library owns device memory in ctx.
 *   - float32 unless noted; parameters cross the boundary as ONE flat vector [Theta] in the
 *     reference's OrderedDict order (policies/base.py:271-277): hidden_0/kernel [O,H1] row-major
 *     (x @ W convention, policies/networks/mlp.py:101), hidden_0/bias, hidden_1/kernel,
 *     hidden_1/bias, output/kernel [H2,A], output/bias, log_std_var [A].
 *   - work is enqueued on the context's HIP stream; calls that hand data back to the host
 *     synchronise before returning, the others return once the work is enqueued.
 *   - one ctx per GPU per process; not thread-safe (the reference's host is single-threaded).
 *   - there is NO CPU fallback: with no usable HIP device promp_ctx_create fails.
 */
#ifndef PROMP_HIP_H
#define PROMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct promp_ctx promp_ctx;

typedef struct promp_dims {
    int32_t n_tasks;            /* meta-tasks resident on this GPU (M_local)                       */
    int32_t n_tasks_global;     /* meta_batch_size over all ranks: the task-mean divides by this   */
    int32_t obs_dim;            /* O, 1..1024.  The fused kernels serve O <= 32 (hidden widths from {32,64}) and O <= 128
                                 * ((64,64) / (128,128)); everything else runs on the layer-by-layer kernels.
                                 * LinearFeatureBaseline's fit (2 O + 4 features) exists for O <= 480                        */
    int32_t act_dim;            /* A, 1..64 (fused kernels: A <= 8)                                */
    int32_t hidden1, hidden2;   /* the first two of hidden_sizes (policies/networks/mlp.py:5-62 takes any tuple).  Two tanh
                                 * layers of 1..128 units run on the fused kernels -- instantiated for {32,64} x {32,64}
                                 * (obs_dim <= 32) and (64,64) / (128,128), other widths zero-padded on the next of those;
                                 * wider layers (<= 256) and other depths run on the layer-by-layer kernels.  Every parameter
                                 * vector crosses this ABI in the caller's layout (promp_param_count floats,
                                 * policies/base.py:271-277 order)                                                          */
    int32_t num_inner_steps;    /* K = num_inner_grad_steps (>= 1)                                 */
    int32_t max_rows;           /* capacity: rows (env steps) per sampling step over local tasks   */
    int32_t max_paths;          /* capacity: paths per sampling step over local tasks              */
    /* ---- ABI 3 ---- */
    int32_t n_hidden;           /* len(hidden_sizes), 1..4; 0 means 2 (hidden1, hidden2)           */
    int32_t hidden3, hidden4;   /* widths of the third / fourth hidden layer (n_hidden >= 3 / 4)    */
    int32_t hidden_act;         /* hidden nonlinearity (policies/networks/mlp.py:47, policies/base.py:31): PROMP_ACT_TANH (0, the
                                 * reference's default and every run script's), PROMP_ACT_RELU, PROMP_ACT_IDENTITY (the reference's
                                 * hidden_nonlinearity=None: linear hidden layers).  Anything but tanh runs on the layer-by-layer kernels.
                                 * Bits 8..15: output_nonlinearity (policies/networks/mlp.py:53-60, 114-117: applied to the mean network's
                                 * last layer; None in every run script of the reference): PROMP_OUT_ACT_NONE (0), _TANH, _RELU, e.g.
                                 * PROMP_ACT_TANH | (PROMP_OUT_ACT_TANH << PROMP_OUT_ACT_SHIFT); layer-by-layer kernels as well */
} promp_dims;
enum { PROMP_ACT_TANH = 0, PROMP_ACT_RELU = 1, PROMP_ACT_IDENTITY = 2 };
enum { PROMP_OUT_ACT_NONE = 0, PROMP_OUT_ACT_TANH = 1, PROMP_OUT_ACT_RELU = 2, PROMP_OUT_ACT_SHIFT = 8 };

enum { PROMP_BASELINE_ZERO = 0, PROMP_BASELINE_LINEAR_FEATURE = 1, PROMP_BASELINE_LINEAR_TIME = 2 };
enum { PROMP_INNER_RATIO = 0,   /* -mean(ratio*adv)   meta_algos/pro_mp.py:59-65   */
       PROMP_INNER_LOGLIK = 1,  /* -mean(logpi*adv)   meta_algos/trpo_maml.py:58-62 */
       PROMP_INNER_DICE = 2     /* -mean(magic_box(cumsum logpi) * adjusted_reward * mask), meta_algos/dice_maml.py:39-45,245-258;
                                   needs promp_set_dice_rewards on every step; outer kind PROMP_OUTER_LOGLIK */ };
enum { PROMP_OUTER_CLIP = 0,    /* PPO clipped surrogate, meta_algos/pro_mp.py:141-145 */
       PROMP_OUTER_RATIO = 1,   /* unclipped,            meta_algos/trpo_maml.py:135    */
       PROMP_OUTER_KL = 2       /* mean KL(old || new) of the last step itself: the TRPO constraint
                                   (meta_algos/trpo_maml.py:133,147,158); its gradient through the
                                   adaptation feeds the finite-difference HVP of
                                   optimizers/conjugate_gradient_optimizer.py:59-89 */,
       PROMP_OUTER_LOGLIK = 3   /* -mean(logpi*adv) at the adapted parameters: VPG-MAML, meta_algos/vpg_maml.py:122-124 */ };

/* SampleProcessor.__init__ arguments (samplers/base.py:48-65) + baseline choice */
typedef struct promp_proc_opts {
    double discount;
    double gae_lambda;
    double reg_coeff;          /* LinearBaseline reg_coeff, baselines/linear_baseline.py:12,68 */
    int32_t normalize_adv;
    int32_t positive_adv;
    int32_t baseline_kind;     /* PROMP_BASELINE_* */
    int32_t reserved;
} promp_proc_opts;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int promp_ctx_create(promp_ctx** out, int device_id, const promp_dims* dims);
void promp_ctx_destroy(promp_ctx* ctx);
const char* promp_last_error(void);
/* 3.  The version moves when an existing entry point changes its meaning or signature: 3 = promp_dims grew by n_hidden,
 * hidden3, hidden4 (hidden_sizes of any length 1..4; a v2 caller's two-layer struct is the same prefix, but the library reads the
 * new fields, so bindings must be rebuilt).  Entry points added since the first v2 header -- round 3: promp_set_reuse_adapt,
 * promp_begin_collection / promp_end_collection, promp_state_version, the two pass counters; round 4: promp_comm_fixed_order --
 * were additions and did not move it. */
int promp_abi_version(void);
/* Theta = O*H1+H1 + H1*H2+H2 + H2*A+A + A */
int promp_param_count(const promp_dims* dims);
/* LinearFeatureBaseline: 2*O+4 (baselines/linear_baseline.py:101-106); LinearTime: 4; Zero: 0 */
int promp_feature_dim(const promp_dims* dims, int baseline_kind);
int promp_sync(promp_ctx* ctx);

/* ---- trajectories ---------------------------------------------------------------------------
 * Upload one sampling step's paths (what MetaSampler.obtain_samples returns,
 * samplers/meta_sampler.py:59-137), flattened task-major, path-major, time-major:
 *   task_path_offsets [n_tasks+1] : paths of task i are [tpo[i], tpo[i+1])
 *   path_row_offsets  [n_paths+1] : rows of path p are  [pro[p], pro[p+1])   (ragged allowed)
 *   obs [rows,O], act [rows,A], rew [rows], old_mean [rows,A]  (agent_infos['mean'])
 *   old_log_std: [rows,A] when log_std_per_row != 0 (the reference's layout,
 *                policies/meta_gaussian_mlp_policy.py:135-136), else [n_tasks,A].
 * act / old_mean / old_log_std may be NULL when only process_samples is needed. */
int promp_upload_step(promp_ctx* ctx, int step, int n_paths,
                      const int32_t* task_path_offsets, const int32_t* path_row_offsets,
                      const float* obs, const float* act, const float* rew,
                      const float* old_mean, const float* old_log_std, int log_std_per_row);

/* Staged upload: the same arguments, but the copies go to the step's SECOND slab set on a copy stream and return at once;
 * promp_commit_step then makes that set the step's current one (a pointer swap on the host; the first launch that reads
 * it waits for the copies).  A trainer whose next batch exists while the current one is still being optimised (replayed
 * or asynchronously collected samples; bench.py's upload-inclusive number) hides the transfer under the compute:
 *     commit(0), commit(1), process_samples / adapt ..., stage(next 0), stage(next 1), optimize
 * Host arrays from promp_host_alloc (pinned) are read by DMA after the call returns: leave them untouched until
 * promp_stage_wait or the step's next commit + use; pageable arrays work too (the call then returns when they are read).
 * The reference has no counterpart (its feed_dict copies are synchronous, meta_algos/base.py:245-301). */
int promp_stage_step(promp_ctx* ctx, int step, int n_paths,
                     const int32_t* task_path_offsets, const int32_t* path_row_offsets,
                     const float* obs, const float* act, const float* rew,
                     const float* old_mean, const float* old_log_std, int log_std_per_row);
int promp_commit_step(promp_ctx* ctx, int step);
int promp_stage_wait(promp_ctx* ctx);
/* page-locked host memory for the arrays above (hipHostMalloc); NULL on failure */
void* promp_host_alloc(size_t bytes);
void promp_host_free(void* p);

/* ---- rows a1-a7: MetaSampleProcessor.process_samples (samplers/meta_sample_processor.py:8-49,
 * samplers/base.py:99-173, utils/utils.py:59-81, baselines/linear_baseline.py:17-106).
 * Computes on the device, per task: returns -> baseline fit -> GAE -> normalise; results stay
 * resident for promp_inner_adapt / promp_optimize.  Asynchronous. */
int promp_process_samples(promp_ctx* ctx, int step, const promp_proc_opts* opts);

/* Copy results of promp_process_samples to the host (any pointer may be NULL):
 *   returns [rows], advantages [rows] float32;  coeffs [n_tasks, feature_dim] float64
 *   (LinearBaseline._coeffs per task);  path_returns0 [n_paths] = returns[0] of each path,
 *   path_undiscounted [n_paths] = sum(rewards), path_reward_sumsq [n_paths] = sum(rewards^2)
 *   (float64; inputs of _log_path_stats, samplers/base.py:135-149, and of the adj_avg_rewards
 *   moments, samplers/meta_sample_processor.py:40-44). */
int promp_download_processed(promp_ctx* ctx, int step, float* returns, float* advantages, double* coeffs,
                             double* path_returns0, double* path_undiscounted, double* path_reward_sumsq);

/* float64 views of the same step: returns [rows] and the RAW (pre-normalisation) GAE advantages [rows], i.e. what the
 * reference stores back into every path dict (samplers/base.py:104,159).  Either pointer may be NULL. */
int promp_download_raw(promp_ctx* ctx, int step, double* returns64, double* raw_advantages64);

/* LinearBaseline.set_params / predict (baselines/linear_baseline.py:17-53) for the paths uploaded in `step`:
 * promp_set_coeffs installs coefficients [n_tasks, feature_dim(kind)] (float64); promp_predict_baseline evaluates
 * b = Phi . w per row with them -> baselines_out [rows] float64 (kind ZERO gives zeros). */
int promp_set_coeffs(promp_ctx* ctx, int step, int baseline_kind, const double* coeffs);
int promp_predict_baseline(promp_ctx* ctx, int step, int baseline_kind, double* baselines_out);

/* Use caller-provided advantages for a step instead of promp_process_samples' (float32 [rows]). */
int promp_set_advantages(promp_ctx* ctx, int step, const float* advantages);
/* DICE-MAML (meta_algos/dice_maml.py:39-45, 84-152, 245-258): the step's per-row rewards of the DiCE objective
 *   -mean_{p,t}( magic_box(sum_{t'<=t} logpi_{p,t'}) * adjusted_reward_{p,t} * mask_{p,t} ),
 * rewards [rows] float32 = adjusted_reward of every VALID row (path by path, as uploaded), already multiplied by
 * rows / (paths * max_path_length) of its task so that the slab mean equals the reference's mean over the zero-padded
 * [paths, max_path_length] array (samplers/dice_sample_processor.py:165-191).  Computes the suffix sums within each path
 * that weight the log-likelihood gradient and installs them as the step's advantages; PROMP_INNER_DICE additionally
 * couples the time steps of a path in the second-order term (two extra launches per R-operator pass).
 * Register-chained kernels only (hidden sizes from {32,64}, obs_dim <= 32). */
int promp_set_dice_rewards(promp_ctx* ctx, int step, const float* rewards);

/* ---- parameters (rows a14: policies/base.py:173-203, 234-240, 262-286) ---------------------- */
int promp_set_theta(promp_ctx* ctx, const float* theta);                 /* [Theta] meta-parameters     */
int promp_get_theta(promp_ctx* ctx, float* theta);
int promp_set_step_sizes(promp_ctx* ctx, const float* step_sizes);       /* [Theta], base.py:303-313     */
/* GaussianMLPPolicy(learn_std=False) (policies/gaussian_mlp_policy.py:63-69: log_std_var created with trainable=False):
 * the trailing act_dim parameters are neither adapted by the inner step (their step sizes are forced to 0) nor updated
 * by Adam.  Default: learned. */
int promp_set_learn_std(promp_ctx* ctx, int learn_std);
/* GaussianMLPPolicy(min_std) (policies/gaussian_mlp_policy.py:31,35,71): log_std evaluated from the shared variables is
 * max(log_std_var, log min_std), gradient iff log_std_var >= log min_std (tf.maximum).  Default 1e-6. */
int promp_set_min_std(promp_ctx* ctx, float min_std);
/* Launch scheduling knobs (no reference counterpart; results are identical for every setting, -1 keeps a value):
 *   stage_overlap  != 0 (default): promp_process_samples of steps >= 1 is enqueued on a second stream, ordered behind the
 *                  last main-stream work on that step's slabs, and joined in front of the first launch that reads its
 *                  outputs -- it then runs under process_samples(0) + the inner step the host enqueued just before.
 *   fuse_min_tasks (default: never): from this many local tasks on, the Hessian-vector pass sums each task's partial rows
 *                  inside its own launch (last-arriving workgroup) instead of leaving it to the grid-wide reduction that
 *                  otherwise follows the launch (measured slower at every task count since that reduction was tuned). */
int promp_set_schedule(promp_ctx* ctx, int stage_overlap, int fuse_min_tasks);
/* Primal cache (no reference counterpart; hidden widths from {32, 64} only).  In one evaluation of the meta-gradient
 * (meta_algos/pro_mp.py:113-155) the inner gradient pass and the second-order pass of an adaptation step run at the same
 * parameters on the same slab; with the cache on, the former writes its hidden activations, means and first-layer
 * cotangents to HBM (4 (2 H1 + H2 + 8) bytes per row) and the latter reads them back instead of recomputing them.
 * Results agree with the recomputing path to float32 rounding (both evaluate the same tanh network; the two kernels
 * contract in different orders).  on = 1 always, 0 never, -1 (default) = always too since round 6 (the cache-reading pass is
 * less than half the recomputing one per tile and wins on 3-task shards as well; through round 5: from two rounds of 16-row
 * tiles per compute unit on). */
int promp_set_primal_cache(promp_ctx* ctx, int on);
/* The inner step and the first epoch of the optimisation that follows it evaluate the same pass.  MAMLAlgo._adapt
 * (meta_algos/base.py:217-242) runs the inner gradient step of every task from the meta-parameters; the first thing
 * ProMP.optimize_policy's graph (pro_mp.py:113-128) then evaluates is that very step again.  With reuse on (default),
 * promp_inner_adapt(step 0) from the pre-update state also leaves theta', the inner scalars and the primal cache where the
 * meta-objective's first evaluation looks for them, and that evaluation skips its first pass as long as nothing the pass reads
 * has changed since (meta-parameters, step-0 slab and advantages, step sizes, inner objective, min_std, learn_std; tracked by
 * version counters) and no log_std entry is below log(min_std) -- the one place where the two differ (raw log_std in the inner
 * step, policies/gaussian_mlp_policy.py:182; clipped in the graph's first step, :71,163).  The library knows the smallest
 * log_std entry from promp_set_theta or from the statistics the last optimisation published; otherwise it does not skip.
 * Bit-identical results either way.  promp_adapt_passes_skipped counts the skipped passes (for tests / logs). */
int promp_set_reuse_adapt(promp_ctx* ctx, int on);
long long promp_adapt_passes_skipped(promp_ctx* ctx);
/* R-operator passes of promp_constraint_hvp that read a step's primal cache instead of recomputing layers 1 and 2 (the
 * products of one conjugate-gradient solve run at the same parameters on the same slabs: the passes that refresh the chain
 * store, every pass of every product reads; governed by promp_set_primal_cache; for tests / logs) */
long long promp_constraint_hvp_cached_passes(promp_ctx* ctx);
/* A counter that moves whenever something an evaluation of the objectives depends on is replaced: the parameters (set /
 * Adam update), the inner step sizes, min_std / learn_std, a step's slabs or advantages.  Equal values before two evaluations
 * with the same arguments mean equal results: the host side uses it to answer TRPO's separate loss / constraint queries at
 * one parameter vector (optimizers/conjugate_gradient_optimizer.py:227-236 evaluates them one after the other) from one pass. */
long long promp_state_version(promp_ctx* ctx);
int promp_set_adam_state(promp_ctx* ctx, const float* m, const float* v, int64_t t);
int promp_get_adam_state(promp_ctx* ctx, float* m, float* v, int64_t* t);
/* MetaPolicy.switch_to_pre_update: replicate theta into every task's parameter slot */
int promp_switch_to_pre_update(promp_ctx* ctx);
/* MetaPolicy.update_task_parameters / policies_params_vals: per-task parameters [n_tasks,Theta] */
int promp_set_task_thetas(promp_ctx* ctx, const float* theta_tasks);
int promp_get_task_thetas(promp_ctx* ctx, float* theta_tasks);

/* ---- rows a8-a10: MAMLAlgo._adapt (meta_algos/base.py:217-242): for every local task
 *   theta_i <- theta_i - step_sizes * grad_theta L_i(theta_i)   on step `step`'s data,
 * explicit per-task parameters => raw log_std (policies/gaussian_mlp_policy.py:182).  Asynchronous. */
int promp_inner_adapt(promp_ctx* ctx, int step, int inner_kind);

/* ---- rollout-side inference: MetaGaussianMLPPolicy.get_actions
 * (policies/meta_gaussian_mlp_policy.py:99-157): mean network of every task's CURRENT parameters
 * (theta replicated after promp_switch_to_pre_update, adapted after promp_inner_adapt) on
 * obs [n_tasks, batch, O] -> mean_out [n_tasks, batch, A].  The Gaussian noise is added by the caller. */
int promp_policy_forward(promp_ctx* ctx, const float* obs, int batch, float* mean_out);

/* ---- rollouts that fill the step slab on the device -------------------------------
 * A device-side rollout lays the sampling step out as n_tasks * envs_per_task fixed-length paths:
 * path p of task i = rows [(i B + p) T, (i B + p + 1) T).
 *
 * (a) environments stepped by the host (MuJoCo ...): promp_begin_rollout once per sampling step, then per
 *     environment step t = 0 .. T-1 promp_policy_step: observations [n_tasks][B][O] in, actions [n_tasks][B][A] out.
 *     The device evaluates each task's mean network under its current parameters, draws the exploration noise
 *     (Philox4x32-10 keyed by `seed`, counter = slab row, Box-Muller), and writes observation, action and mean
 *     into the slab at row (task, env, t): what policies/meta_gaussian_mlp_policy.py:99-157 +
 *     samplers/meta_sampler.py:87-125 do with one sess.run and a Python loop per environment step.  The rewards
 *     follow once at the end (promp_set_rewards, float32 [rows]); promp_process_samples then needs no upload.
 *     clip_infos != 0: the log_std recorded for agent_infos is max(log_std, log 1e-6) (pre-update policy,
 *     policies/gaussian_mlp_policy.py:71); the noise scale always uses the raw value (:74).
 *     Every policy shape the context accepts is served (layer-by-layer shapes: one workgroup per environment). */
int promp_begin_rollout(promp_ctx* ctx, int step, int envs_per_task, int path_length);
/* (a') the same for environments whose episodes end early (samplers/meta_sampler.py:100-125: an environment that reports
 *     `done` is reset and keeps collecting; gym-style termination, e.g. envs/mujoco_envs/ant_rand_direc.py:32-46).  The slab
 *     cannot be laid out before the episode lengths are known, so promp_policy_step files the rows of vectorised step
 *     s = 0 .. max_steps-1 (its `t` argument) under (s, environment) in a staging area -- the Philox counter is that staging
 *     row, s * n_tasks * envs_per_task + environment -- and promp_end_collection copies the FINISHED episodes into the slab in
 *     path order once the host knows them: path p = steps [path_start[p], path_start[p] + path_len[p]) of environment
 *     path_env[p] (= task * envs_per_task + b); task_path_offsets [n_tasks + 1] as in promp_upload_step; rewards [rows] in path
 *     order.  Unfinished episodes are simply not listed (meta_sampler.py:114-125 drops them too).  2 * max_path_length - 1
 *     vectorised steps always suffice to finish n_tasks * envs_per_task * max_path_length environment steps. */
int promp_begin_collection(promp_ctx* ctx, int step, int envs_per_task, int max_steps);
int promp_end_collection(promp_ctx* ctx, int step, int n_paths, const int32_t* task_path_offsets, const int32_t* path_env,
                         const int32_t* path_start, const int32_t* path_len, const float* rewards);
int promp_policy_step(promp_ctx* ctx, int step, int t, const float* obs, uint64_t seed, int clip_infos, float* actions_out);
int promp_set_rewards(promp_ctx* ctx, int step, const float* rewards);
/* The same in the environment's float64: the reference scans float64 rewards (utils/utils.py:74-81 on the arrays the env
 * returned) and LinearBaseline.fit can be given arbitrary float64 targets; returns, GAE deltas and path statistics then
 * read these instead of the float32 copy (which is refreshed too, for promp_download_step). */
int promp_set_rewards_f64(promp_ctx* ctx, int step, const double* rewards);

/* (b) the 2-D point-mass meta-environment of BASELINE config 1, run_scripts/pro-mp_run_point_mass.py:
 *     normalize(MetaPointEnvCorner()) -- whole rollouts in one launch, one thread per environment:
 *       e = lb + (a + s)(ub - lb)/(2 s), clipped to [lb, ub]             envs/normalized_env.py:109-123: the normalize wrapper,
 *                                                                        s = normalization_scale (10), [lb, ub] = -+max_step;
 *                                                                        s = 0: bare environment, e = a
 *       state += clip(e, -max_step, max_step)                            envs/point_envs/point_env_2d_corner.py:37
 *       reward_type 0 dense -|s' - goal|, 1 dense_squared, 2 sparse      point_env_2d_corner.py:62-81
 *       no early termination                                             point_env_2d_corner.py:39
 *     goals [n_tasks][2], start [n_tasks][envs_per_task][2] (float64 like the NumPy environment);
 *     noise [n_tasks][envs_per_task][path_length][2] standard normals drawn by the caller, or NULL: drawn on the device
 *     from `seed` as in (a).  Asynchronous. */
typedef struct {
    double normalization_scale, max_step, sparse_radius;
    int32_t reward_type, clip_infos;
    uint64_t seed;
} promp_point_env_opts;
int promp_rollout_point_env(promp_ctx* ctx, int step, int envs_per_task, int path_length, const double* goals,
                            const double* start, const float* noise, const promp_point_env_opts* opts);

/* Step slab back on the host (any pointer may be NULL): obs [rows][O], act [rows][A], rew [rows],
 * old_mean [rows][A], old_log_std [rows | n_tasks][A] as uploaded / produced by a device rollout. */
int promp_download_step(promp_ctx* ctx, int step, float* obs, float* act, float* rew, float* old_mean, float* old_log_std);

/* ---- rows a11-a13 ---------------------------------------------------------------------------
 * One evaluation of the ProMP meta-objective (meta_algos/pro_mp.py:67-163) and of its exact
 * gradient (what tf.gradients yields through meta_algos/base.py:206, i.e. including the
 * second-order MAML term), over steps 0..K of resident data, reduced over local tasks and, when a
 * communicator is attached, all-reduced over ranks; leaves the task-MEAN gradient on the device.
 *   stats_out [K+2] = { loss, inner_kl[0..K-1], outer_kl };  grad_out [Theta] (may be NULL). */
int promp_meta_grad(promp_ctx* ctx, float clip_eps, const float* inner_kl_coeff, int inner_kind, int outer_kind,
                    float* grad_out, float* stats_out);
/* ConjugateGradientOptimizer's Hessian-vector product of the constraint (optimizers/conjugate_gradient_optimizer.py:59-89
 * builds it by finite differences of the constraint gradient, hvp_approach=FiniteDifferenceHvp; SURVEY 8f row 2 asks
 * for the exact one): out [Theta] = H v, H = Hessian wrt theta of mean_i KL(pi_old || pi_{theta'_i(theta)}) on the last
 * step's samples (trpo_maml.py:146-158), through the K inner steps (Gauss-Newton form J^T H_KL J: exact where the old
 * distribution is the adapted policy's, i.e. at the parameters TRPO evaluates it).  The mean is over the GLOBAL
 * meta-batch (all-reduce when a communicator is attached).  refresh_chain != 0 recomputes the adapted parameters
 * theta_k from the current theta first (needed once per theta).  No reg_coeff term: the caller adds reg_coeff * v.
 * Every supported policy shape (register-chained and cooperative kernels); the register-chained ones keep primal caches
 * over the products of one solve (promp_set_primal_cache, promp_constraint_hvp_cached_passes). */
int promp_constraint_hvp(promp_ctx* ctx, int inner_kind, const float* v, int refresh_chain, float* out);
/* ConjugateGradientOptimizer's whole solve on the device (optimizers/conjugate_gradient_optimizer.py:325-354 conjugate_gradients();
 * :59-104 FiniteDifferenceHvp.Hx / build_eval -- the product x -> (H + reg_coeff I) x of the constraint; :259-264 the solve inside
 * optimize() and the closing product that sizes the step): cg_iters conjugate-gradient iterations on (H + reg_coeff I) x = b from x = 0, then x . (H + reg_coeff I) x.
 *   hvp_mode 0  H v = (grad c(theta + eps v) - grad c(theta - eps v)) / (2 eps)   (the reference's default: symmetric, eps 1e-5)
 *            1  H v = (grad c(theta + eps v) - grad c(theta)) / eps
 *            2  the exact product of promp_constraint_hvp (eps unused)
 * with grad c the gradient of the mean outer KL through the adaptation (promp_meta_grad with PROMP_OUTER_KL).  Same arithmetic
 * as the host loop over promp_set_theta / promp_meta_grad (float32 vectors; the dot products are summed in float64 in a fixed
 * order), but nothing crosses to the host between the products: 2 cg_iters + 2 gradient evaluations enqueued back to back, one
 * synchronisation at the end.  The iteration stops updating once r.r < residual_tol (1e-10 in the reference).  The parameters
 * are back at theta when the call returns.  b [Theta] = the loss gradient; x_out [Theta]; *xhx_out = x . (H + reg_coeff I) x. */
int promp_cg_solve(promp_ctx* ctx, int inner_kind, const float* b, int cg_iters, float reg_coeff, float eps, int hvp_mode,
                   float residual_tol, float* x_out, double* xhx_out);
/* tf.train.AdamOptimizer step on theta with the gradient left by promp_meta_grad
 * (optimizers/maml_first_order_optimizer.py:24,64; b1=.9 b2=.999 eps=1e-8, bias-corrected lr). */
int promp_adam_step(promp_ctx* ctx, float learning_rate);
/* ProMP.optimize_policy's numerical core (meta_algos/pro_mp.py:165-199 +
 * optimizers/maml_first_order_optimizer.py:82-115,146-163): num_epochs x (meta_grad, Adam), then
 * compute_stats.  loss_before = loss of the first epoch; stats_after [K+2] as in promp_meta_grad.
 * Everything is enqueued back-to-back; one synchronisation at the end. */
int promp_optimize(promp_ctx* ctx, int num_epochs, float learning_rate, float clip_eps,
                   const float* inner_kl_coeff, int inner_kind, int outer_kind,
                   float* loss_before, float* stats_after);
/* The same in two halves.  promp_optimize_begin enqueues everything (epochs, compute_stats, an asynchronous copy of
 * the statistics to page-locked memory) and returns without waiting; promp_optimize_end waits for that copy and hands
 * out loss_before / stats_after.  Between the two the caller may enqueue work that does not depend on the statistics --
 * Trainer.train's next process_samples / _adapt (meta_trainer.py:105-116); the KL-coefficient rule (pro_mp.py:201-214)
 * needs them only in front of the next optimize_policy.  One optimisation may be pending at a time. */
int promp_optimize_begin(promp_ctx* ctx, int num_epochs, float learning_rate, float clip_eps,
                         const float* inner_kl_coeff, int inner_kind, int outer_kind);
int promp_optimize_end(promp_ctx* ctx, float* loss_before, float* stats_after);

/* ---- multi-GPU: task-sharded data parallelism, one process per GPU, RCCL over xGMI.
 * The only exchange on the path is the task-mean of the meta-objective / its gradient
 * (meta_algos/pro_mp.py:122,151,155): ONE all-reduce of [Theta+K+2] floats per epoch. */
int promp_comm_unique_id(void* id_out, size_t id_bytes);                  /* rank 0; 128 bytes  */
int promp_comm_init(promp_ctx* ctx, int rank, int nranks, const void* id, size_t id_bytes);
int promp_allreduce_f64(promp_ctx* ctx, double* host_buf, int n, int op /*0 sum, 1 max*/);
/* Hand the communicator of `src` over to `dst` (same device): a context that is re-created with more capacity
 * keeps its communicator instead of running a new rendezvous -- ranks with ragged batches regrow at different
 * times, a rendezvous per regrow would deadlock.  `src` is left single-rank. */
int promp_comm_move(promp_ctx* dst, promp_ctx* src);
/* Take the several-rank launch sequence (per-rank sums -> [all-reduce] -> mean + Adam as separate launches) even on
 * one rank; numerically identical to the fused single-rank launch (the parity tests assert bitwise equality). */
int promp_comm_split_path(promp_ctx* ctx, int on);
/* What the attached communicator says about itself (ncclCommCount, ncclCommUserRank; 1 / 0 without one), whether the exchange is
 * the fixed-order one, and the PCI bus id of the context's device: evidence for a benchmark line that N ranks on N distinct GPUs
 * took part (meta_algos/pro_mp.py:122,151,155 is the mean the exchange computes).  Any output may be NULL. */
int promp_comm_info(promp_ctx* ctx, int32_t* nranks, int32_t* rank, int32_t* fixed_order, char* bus_id_out, size_t bus_id_bytes);
/* The exchange as ncclAllGather + a sum in rank order (0, 1, ...) on every rank instead of ncclAllReduce: the replicas'
 * parameters are bitwise identical by construction -- not by the grace of RCCL choosing the same reduction order on every
 * rank -- and equal to one process adding the ranks' shards in that order.  The buffer is ~6 k floats:
 * the gather moves nranks x 24 KB, latency as the all-reduce's.  Set it the same on every rank; default off. */
int promp_comm_fixed_order(promp_ctx* ctx, int on);
/* The buffer the all-reduce acts on, [Theta + K + 2] floats = { grad sums | J sum | inner-KL sums [K] | outer-KL sum }
 * over the LOCAL tasks after promp_meta_grad (when n_tasks_global > n_tasks and no communicator is attached, the
 * exchange is the caller's: get, reduce over ranks with any collective, set, then promp_adam_step -- which divides by
 * n_tasks_global). */
int promp_reduced_get(promp_ctx* ctx, float* out);
int promp_reduced_set(promp_ctx* ctx, const float* in);

/* ---- evaluation hooks used by the parity tests and by alternative optimizers (TRPO-MAML's
 * conjugate-gradient loop calls these per evaluation): per-task objective, mean-KL and their
 * gradient at the CURRENT per-task parameters on step `step`'s data.
 *   kind: 0 ratio surrogate, 1 clipped surrogate, 2 log-likelihood, 3 mean KL(old||new)
 *   grads_out [n_tasks,Theta], loss_out [n_tasks], kl_out [n_tasks] (any may be NULL). */
int promp_eval_loss_grad(promp_ctx* ctx, int step, int kind, float clip_eps, int clip_log_std,
                         float* grads_out, float* loss_out, float* kl_out);
/* per task: out_i = -H_i v_i + kl_weight * grad KL_i, H_i = Hessian of the inner objective at the
 * current per-task parameters on step `step`'s data; v [n_tasks,Theta]; out [n_tasks,Theta]. */
int promp_eval_hvp(promp_ctx* ctx, int step, int inner_kind, int clip_log_std, float kl_weight,
                   const float* v, float* out);

/* ---- measurement: HIP-event timing of the pass kernels on the context's stream ------------- */
enum { PROMP_KERNEL_FWD_BWD = 0, PROMP_KERNEL_HVP = 1, PROMP_KERNEL_GRAM = 2, PROMP_KERNEL_FWD = 3 /* forward-only k_pass */,
       PROMP_KERNEL_EXCHANGE = 4 /* the ranks' exchange of [Theta+K+2] floats (all-reduce, or all-gather + ordered sum) */,
       PROMP_KERNEL_COUNT = 5 };
int promp_prof_enable(promp_ctx* ctx, int on);
int promp_prof_read(promp_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches, int64_t* rows);
/* The fused pass kernels run their float32-equivalent products as a two-term FP16 split (promp_amd/csrc/promp_device.h); FP16 has a
 * range, so cotangents carry per-wave powers of two that follow the data.  out2 receives, and the call clears, how many segments
 * (a workgroup's share of a task's rows) were walked a second time since the last call because a cotangent left the format at the
 * scale its wave had chosen: k_pass [0], k_chain_hvp [1].  Diagnostics only -- heavy-tailed importance ratios (a policy far from the
 * one that sampled) cost time, never accuracy; the reference has no counterpart. */
int promp_split_events(promp_ctx* ctx, int64_t* out2);
int promp_device_info(promp_ctx* ctx, char* name_out, size_t name_bytes, int32_t* n_cus, int32_t* clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* PROMP_HIP_H */
