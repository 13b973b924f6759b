/*
 * libtonic_hip.so — C ABI of the MI355X (gfx950) rollout-collect + learner-update engine
 * that sits under Tonic's duck-typed Python API (tonic.torch.agents.Agent / tonic.Trainer /
 * tonic.environments.distribute / tonic.replays.Segment,Buffer).
 *
 * The reference (fabiopardo/tonic) has NO native code and NO FFI: every entry point below
 * replaces arithmetic that the reference runs in NumPy / torch-CPU; the "replaces" line of
 * each declaration cites that reference code (paths relative to the reference checkout).
 * The Python binding a Tonic maintainer would add is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / HIP C++ types in signatures.
 *     `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *   - Every pointer marked `d_` is DEVICE memory owned by the CALLER (e.g. a torch ROCm
 *     tensor's data_ptr()); row-major contiguous float32 unless stated.  The library
 *     allocates nothing and never synchronises: every call only enqueues kernels on
 *     `stream` (hipGraph-capturable).  Scratch space is passed in by the caller; its size
 *     comes from the matching *_workspace_bytes() query.
 *   - Return value: 0 on success, negative tonic_status otherwise; tonic_last_error()
 *     returns a thread-local message for the last failure on the calling thread.
 *   - Shapes: T = Segment time steps, W = workers, N = T*W samples (time-major flattening,
 *     tonic/replays/utils.py:22-25), O = observation size, A = action size, hidden = 64.
 *   - Parameter blocks are ONE flat float32 buffer per network in the reference's
 *     `model.parameters()` order:
 *       PPO actor : W1[64,O] b1[64] W2[64,64] b2[64] log_scale[1,A] W3[A,64] b3[A]
 *       V critic  : W1[64,O] b1[64] W2[64,64] b2[64] w3[1,64] b3[1]
 *     Gradient / Adam-moment buffers use the same layout and length.
 */
#ifndef TONIC_HIP_H
#define TONIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tonic_status {
  TONIC_OK = 0,
  TONIC_ERR_INVALID_ARGUMENT = -1,   /* bad shape / NULL pointer / unsupported size   */
  TONIC_ERR_UNSUPPORTED_SHAPE = -2,  /* PPO path: O > 384, A > 32, a torso outside 1..4 layers of 4..384; the */
                                     /* workspace-less forwards: O > 32 or A > 8          */
  TONIC_ERR_LAUNCH = -3,             /* a HIP runtime call / kernel launch failed       */
  TONIC_ERR_WORKSPACE = -4,          /* workspace too small                             */
  TONIC_ERR_TIMEOUT = -5             /* collector: workers / actions did not arrive     */
} tonic_status;

/* ---- library ------------------------------------------------------------------------ */
const char* tonic_last_error(void);
/* ABI version (bumped on any signature or layout change; 2 = padded off-policy parameter blocks,
 * 3 = pinned-host collector, gradient / normaliser clipping, 4 = distributional critic entries,
 * 5 = tonic_collector_arm / _ring / _claim / _block_carry_over, `ring` argument of
 * tonic_collector_synthetic_step, 6 = `max_workgroups` argument of tonic_ppo_actor_grad /
 * tonic_value_regression_grad, 7 = tonic_stream_gate, 8 = the tonic_*_torso entries, tonic_mlp_hidden, `min_log_dual` of the MPO entries,  tonic_q_iteration_t.phase,
 * 9 = collector transport 3 + tonic_collector_transport, 10 = tonic_q_iteration_t.refresh_images (fp16x2 weight
 * images of the off-policy passes in the workspaces: tonic_offpolicy_workspace_bytes / tonic_q_iteration_workspace_bytes
 * grow), tonic_mlp_hidden packs with bit 30 set (plain widths of any size pass as they are), tonic_collector_q_act)
 * and the gfx target the kernels were built for.  TONIC_ABI_VERSION is what a binding was compiled against:
 * tonic_amd/_fastcall (csrc/fastcall.c) and tonic_amd/_lib.py compare it with the loaded library's answer. */
#define TONIC_ABI_VERSION 11
int32_t tonic_abi_version(void);
const char* tonic_target_arch(void);


/* Sizes of the flat parameter blocks described above. */
int64_t tonic_ppo_actor_param_count(int32_t O, int32_t A);
int64_t tonic_v_critic_param_count(int32_t O);

/* ---- GAE / lambda-returns (HBM-bound scan) --------------------------------------------
 * replaces: tonic/replays/utils.py:4-19 (lambda_returns) + tonic/replays/segments.py:41-46
 *           (raw advantages and their global mean / population std).
 * Inputs  [T,W]: d_next_values, d_rewards, d_resets, d_terminations, d_values (float32; the
 *           0/1 flags are stored as float32 exactly like Segment.store, segments.py:33).
 * Outputs [T,W]: d_returns, d_advantages (RAW = returns - values; the normalisation
 *           (adv-mean)/std is applied in-register by the PPO actor kernel from d_adv_stats).
 *         d_adv_stats[4] float32 = {mean, std, all_zero_flag (1.0 if every raw advantage is
 *           0, actors.py:71), normalise_flag (1.0 if std != 0, segments.py:44)}.
 * `chunks`: 1 = ONE chain per worker column over the whole T axis in the reference's float32
 *           operation order (bit-exact returns); > 1 = T is cut into 128-row segments (64 from W = 8192; whatever the
 *           number) that run concurrently, each segment held in registers by one workgroup and
 *           the carries between segments composed as affine maps: one pass over the
 *           data, 28 B per transition, returns agree to ~1e-6 relative; 0 = the library chooses
 *           (segments unless W alone fills the chip).
 */
int64_t tonic_gae_workspace_bytes(int64_t T, int64_t W, int32_t chunks);
int tonic_gae_lambda_returns(const float* d_next_values, const float* d_rewards,
                             const float* d_resets, const float* d_terminations,
                             const float* d_values, float* d_returns, float* d_advantages,
                             float* d_adv_stats, double* d_adv_moments, int64_t T, int64_t W,
                             double discount_factor, double trace_decay, int32_t chunks,
                             void* d_workspace, int64_t workspace_bytes, void* stream);
/* Multi-GPU: d_adv_moments (float64[5], may be NULL) receives this rank's
 * {sum, sum_sq, -min, max, count} of the raw advantages.  Ranks all-reduce it (SUM on
 * [0,1,4], MAX on [2,3]) and then rebuild the GLOBAL d_adv_stats with the call below, so the
 * normalisation matches the single-process full batch (segments.py:43-46). */
int tonic_advantage_stats_from_moments(const double* d_adv_moments, float* d_adv_stats,
                                       void* stream);

/* ---- acting: policy forward + sample + log-prob ----------------------------------------
 * replaces: tonic/torch/agents/a2c.py:75-85 (A2C._step) = Actor.forward
 *           (tonic/torch/models/actors.py:60-66,134-137; MLP utils.py:22-23) + Normal.sample
 *           + log_prob.sum(-1).  d_eps [n,A] are standard-normal draws made by the host
 *           torch CPU generator so the RNG stream is the reference's;
 *           actions = loc + scale*eps.  Passing d_eps = NULL returns the mode (loc) instead.
 * Outputs: d_actions [n,A], d_log_probs [n] (may be NULL).
 */
int tonic_ppo_act(const float* d_actor_params, const float* d_observations,
                  const float* d_eps, float* d_actions, float* d_log_probs, int64_t n,
                  int32_t O, int32_t A, void* stream);

/* ---- critic forward ---------------------------------------------------------------------
 * replaces: tonic/torch/agents/a2c.py:92-99 (A2C._evaluate) = Critic.forward
 *           (tonic/torch/models/critics.py:15-20,87-90; encoders.py:13-16;
 *            tonic/torch/normalizers/mean_stds.py:34-39 with clip=None).
 * d_norm_mean / d_norm_std: [O] (the MeanStd `_mean` / `_std` parameters); norm_clip: MeanStd's
 *   `clip` (mean_stds.py:37-38: the normalised input clamped to [-clip, clip]); <= 0 = None.
 *   The same pair + clip travels into every entry point that normalises observations.
 */
int tonic_value_forward(const float* d_critic_params, const float* d_norm_mean,
                        const float* d_norm_std, double norm_clip, const float* d_observations,
                        float* d_values, int64_t n, int32_t O, void* stream);

/* ---- learner: fused forward + loss + backward over the whole batch ------------------------
 * Both calls write per-workgroup partial sums into d_workspace, then reduce them into
 * d_grad_sums (SUMS over the n local samples, not means: the 1/N_global scaling happens in
 * tonic_adam_step so that multi-GPU ranks can all-reduce d_grad_sums first).
 * d_grad_sums layout: [P gradient sums | 8 statistic sums], P = param count:
 *   actor  stats: {loss_sum, kl_sum, clipped_count, n*entropy, n*mean_sigma, n, 0, 0}
 *   critic stats: {squared_error_sum, value_sum, 0, 0, 0, n, 0, 0}        (n = local samples)
 * d_skip_flag (int32, may be NULL): when *d_skip_flag != 0 the kernels exit immediately
 * and d_grad_sums is zero-filled (device-side replacement of the per-iteration
 * `stop.numpy()` host sync of tonic/torch/agents/ppo.py:45-46).
 *
 * tonic_ppo_actor_grad replaces: tonic/torch/updaters/actors.py:70-99 (ClippedRatio
 *   forward, surrogate loss, entropy, backward).   d_adv_stats as produced by
 *   tonic_gae_lambda_returns.  ratio_clip < 0 selects the plain policy gradient of
 *   StochasticPolicyGradient (A2C; actors.py:20-51): loss_sum = -sum(adv * logp), no ratio, no
 *   clipping; kl_sum keeps its meaning (old - new log-probabilities).
 * tonic_value_regression_grad replaces: tonic/torch/updaters/critics.py:18-24 (VRegression).
 * max_workgroups: 0 = the kernel's own width (one workgroup per compute unit, fewer for small
 *   n); k > 0 = at most k workgroups — for a network whose iterations run beside another kernel
 *   that must keep its compute units (agents.PPO: the critic under the next rollout's resident
 *   collect kernel).  The float32 partial sums are grouped per workgroup, so the same
 *   (n, max_workgroups) gives the same bits on every call; different widths differ at rounding
 *   level.  The layer-by-layer kernels of the wide shapes ignore it.
 *
 * Shapes: O <= 32 and A <= 8 run in the fused kernels (a whole network per 16-sample tile in
 * registers).  Wider ones (O <= 384, A <= 32: Ant-v3, Humanoid, humanoid-walk ...) run layer by
 * layer with the activations in the workspace (csrc/mlpwide.hip), same arguments, same outputs;
 * tonic_ppo_workspace_bytes sizes the workspace for either kind (the *_wide forwards fall back to
 * the fused kernels for narrow shapes, so callers may always use them).
 */
int64_t tonic_mlp64_grad_workspace_bytes(int64_t n, int64_t param_count);
int64_t tonic_ppo_workspace_bytes(int64_t n, int32_t O, int32_t A, int32_t actor);
int tonic_ppo_act_wide(const float* d_actor_params, const float* d_observations,
                       const float* d_eps, float* d_actions, float* d_log_probs, int64_t n,
                       int32_t O, int32_t A, void* d_workspace, int64_t workspace_bytes,
                       void* stream);
int tonic_value_forward_wide(const float* d_critic_params, const float* d_norm_mean,
                             const float* d_norm_std, double norm_clip,
                             const float* d_observations, float* d_values, int64_t n, int32_t O,
                             void* d_workspace, int64_t workspace_bytes, void* stream);
int tonic_ppo_actor_grad(const float* d_actor_params, const float* d_observations,
                         const float* d_actions, const float* d_advantages,
                         const float* d_adv_stats, const float* d_old_log_probs,
                         float* d_grad_sums, int64_t n, int32_t O, int32_t A,
                         double ratio_clip, double entropy_coeff, const int32_t* d_skip_flag,
                         int32_t max_workgroups, void* d_workspace, int64_t workspace_bytes,
                         void* stream);
int tonic_value_regression_grad(const float* d_critic_params, const float* d_norm_mean,
                                const float* d_norm_std, double norm_clip,
                                const float* d_observations,
                                const float* d_returns, float* d_grad_sums, int64_t n,
                                int32_t O, int32_t max_workgroups, void* d_workspace,
                                int64_t workspace_bytes, void* stream);

/* ---- PPO / A2C networks with ANY torso --------------------------------------------------------
 * replaces: the same reference code as the five entries above for actor / critic networks built with
 *   tonic.torch.models.MLP(sizes, activation) other than the default (tonic/torch/models/utils.py:4-23;
 *   the default MLP((64, 64), Tanh) of a2c.py:7-17 is served by the entries above): `layers` hidden
 *   layers (1 .. 4) of `sizes[l]` units (4 .. 384, multiples of 4), activation 1 = torch.nn.Tanh,
 *   2 = torch.nn.ReLU; heads as above (DetachedScaleGaussianPolicyHead / ValueHead).  Layer by layer on
 *   fp32 MFMA tiles with the activations in the workspace (csrc/mlpwide.hip); O <= 384, A <= 32.
 * Parameter blocks: dense, parameters() order — per layer W [size, fan_in] then b [size]; actor: then
 *   log_scale [A], W_loc [A, last], b_loc [A]; critic: w_v [last], b_v [1].  tonic_ppo_torso_param_count
 *   returns the float count (-1: unsupported torso / shape), tonic_ppo_torso_workspace_bytes the scratch.
 * Same argument meaning, outputs (gradient SUMS + 8 statistic sums) and error behaviour as the
 *   entries they mirror.
 */
int64_t tonic_ppo_torso_param_count(int32_t O, int32_t A, int32_t actor, int32_t layers,
                                    const int32_t* sizes);
int64_t tonic_ppo_torso_workspace_bytes(int64_t n, int32_t O, int32_t A, int32_t actor, int32_t layers,
                                        const int32_t* sizes);
int tonic_ppo_act_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                        const float* d_actor_params, const float* d_observations, const float* d_eps,
                        float* d_actions, float* d_log_probs, int64_t n, int32_t O, int32_t A,
                        void* d_workspace, int64_t workspace_bytes, void* stream);
int tonic_value_forward_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                              const float* d_critic_params, const float* d_norm_mean,
                              const float* d_norm_std, double norm_clip, const float* d_observations,
                              float* d_values, int64_t n, int32_t O, void* d_workspace,
                              int64_t workspace_bytes, void* stream);
int tonic_ppo_actor_grad_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                               const float* d_actor_params, const float* d_observations,
                               const float* d_actions, const float* d_advantages,
                               const float* d_adv_stats, const float* d_old_log_probs,
                               float* d_grad_sums, int64_t n, int32_t O, int32_t A,
                               double ratio_clip, double entropy_coeff, const int32_t* d_skip_flag,
                               void* d_workspace, int64_t workspace_bytes, void* stream);
int tonic_value_regression_grad_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                                      const float* d_critic_params, const float* d_norm_mean,
                                      const float* d_norm_std, double norm_clip,
                                      const float* d_observations, const float* d_returns,
                                      float* d_grad_sums, int64_t n, int32_t O, void* d_workspace,
                                      int64_t workspace_bytes, void* stream);

/* ---- optimizer ---------------------------------------------------------------------------
 * replaces: torch.optim.Adam single-tensor path (torch/optim/adam.py:395-547) as
 *   constructed at tonic/torch/updaters/actors.py:58-59 and critics.py:9-10, applied to the
 *   flat parameter block: grad = d_grad_sums[i] * grad_scale (grad_scale = 1/N_global).
 * d_state: int32[4] = {step_count, stop_flag, reserved, arrival counter (zero between calls)};
 *   step_count is incremented on the device so the call is graph-replayable.
 * Statistic extras (stats_kind: 0 none, 1 PPO actor, 2 V critic, 3 twin Q critics
 *   {loss, q1 mean, q2 mean}, 4 Q-gradient actor {loss}): finalises
 *   the 8 statistic sums into d_info_row[8] and, for the actor, sets stop_flag when
 *   kl > kl_threshold (actors.py:102-112).  Actor rows: {loss, kl, entropy, clip_fraction,
 *   std, stop, ran(1.0), 0}; critic rows: {loss, v_mean, ...}.  When d_adv_stats says every
 *   advantage is zero the actor step is skipped and loss=kl=clip_fraction=0 (actors.py:71-78).
 * The whole call is skipped when d_skip_flag (may alias &d_state[1]) is non-zero.
 */
int tonic_adam_step(float* d_params, const float* d_grad_sums, float* d_exp_avg,
                    float* d_exp_avg_sq, int32_t* d_state, int64_t param_count,
                    double grad_scale, double lr, double beta1, double beta2, double eps,
                    int32_t stats_kind, double kl_threshold, double entropy_coeff,
                    const float* d_adv_stats, float* d_info_row, const int32_t* d_skip_flag,
                    void* stream);
/* replaces: torch.nn.utils.clip_grad_norm_(variables, gradient_clip) as every updater calls it
 *   between backward() and optimizer.step() (tonic/torch/updaters/actors.py:40-42,96-98,182-184,
 *   260-262; critics.py:24-25,79-80,176-177,229-230): scales the n gradient SUMS in place by
 *   min(1, max_norm / (||grad_scale * sums||_2 + 1e-6)).  The norm is reduced in float64 in a
 *   fixed order (bit-reproducible; with several ranks it is taken AFTER the all-reduce, so every
 *   rank clips alike).  Call it before tonic_adam_step*; d_skip_flag as in the grad kernels.
 *   The workspace ends with two floats {factor, norm} for inspection. */
int64_t tonic_clip_workspace_bytes(int64_t n);
int tonic_clip_grad_norm(float* d_grad_sums, int64_t n, double grad_scale, double max_norm,
                         const int32_t* d_skip_flag, void* d_workspace, int64_t workspace_bytes,
                         void* stream);
/* Two independent optimizer steps in ONE launch pair (PPO: actor, with its KL early-stop flag and
 * statistics, and critic — ppo.py:33-46 steps them back to back); same arithmetic as two
 * tonic_adam_step calls sharing grad_scale / betas / eps. */
int tonic_adam_step_pair(
    float* d_params_a, const float* d_grad_sums_a, float* d_exp_avg_a, float* d_exp_avg_sq_a,
    int32_t* d_state_a, int64_t param_count_a, double lr_a, int32_t stats_kind_a,
    double kl_threshold, double entropy_coeff, const float* d_adv_stats, float* d_info_row_a,
    const int32_t* d_skip_flag_a,
    float* d_params_b, const float* d_grad_sums_b, float* d_exp_avg_b, float* d_exp_avg_sq_b,
    int32_t* d_state_b, int64_t param_count_b, double lr_b, int32_t stats_kind_b,
    float* d_info_row_b,
    double grad_scale, double beta1, double beta2, double eps, void* stream);

/* ---- replay: HBM-resident Segment -----------------------------------------------------------
 * replaces: tonic/replays/segments.py:27-36 (Segment.store, one time row for all W workers)
 *   and tonic/torch/normalizers/mean_stds.py:44-48 (MeanStd.record: sequential float32
 *   accumulation in worker order, square then add — bit-exact by construction).
 * d_norm_acc: float32[2*O] = {new_sum[O], new_sum_sq[O]} (may be NULL to skip recording).
 */
int tonic_segment_store(float* d_seg_observations, float* d_seg_actions,
                        float* d_seg_next_observations, float* d_seg_rewards,
                        float* d_seg_resets, float* d_seg_terminations, float* d_seg_log_probs,
                        const float* d_observations, const float* d_actions,
                        const float* d_next_observations, const float* d_rewards,
                        const float* d_resets, const float* d_terminations,
                        const float* d_log_probs, float* d_norm_acc, int64_t row, int64_t W,
                        int32_t O, int32_t A, void* stream);

/* replaces: tonic/torch/normalizers/mean_stds.py:44-48 (MeanStd.record) on its own: advances
 *   d_norm_acc = {new_sum[size], new_sum_sq[size]} by `rows` rows of d_values [rows, size] in row
 *   order with the reference's float32 operation sequence (bit-exact).  For callers that keep
 *   the record off the critical path of their step kernel: one launch over all T*W rows of a
 *   rollout runs the chain at ~5 cycles per row, against ~11 when it shares a step launch. */
int tonic_meanstd_record(const float* d_values, float* d_norm_acc, int64_t rows, int32_t size,
                         void* stream);

/* replaces: tonic/replays/segments.py:58-65 (Segment.get with batch_size: fancy-index every
 *   learner input by a shuffled index vector).  Gathers `count` flattened transitions
 *   d_indices[i] in [0, segment_rows) of observations [N,O], actions [N,A], raw advantages,
 *   log-probs and returns [N] into contiguous outputs; one call per shuffled epoch, the
 *   minibatches are then [start, start + batch_size) slices of the outputs.  Bit-exact copy.
 */
int tonic_segment_gather(const int64_t* d_indices, const float* d_seg_observations,
                         const float* d_seg_actions, const float* d_seg_advantages,
                         const float* d_seg_log_probs, const float* d_seg_returns,
                         float* d_observations, float* d_actions, float* d_advantages,
                         float* d_log_probs, float* d_returns, int64_t count,
                         int64_t segment_rows, int32_t O, int32_t A, void* stream);

/* ---- fused on-policy collect step (device-resident collectors) ---------------------------------
 * replaces: tonic/torch/agents/a2c.py:41-52 (A2C.step: forward + sample + log-prob) and
 *   a2c.py:58-69 (Segment.store + MeanStd.record) for ONE environment step in ONE launch:
 *   tonic_ppo_act + tonic_segment_store fused, actions / log-probs written straight into row
 *   `row` of the Segment.  d_actions_out [W,A] (may be NULL) receives a copy of the actions for
 *   the environment.  d_eps / d_norm_acc may be NULL as in the unfused calls.
 */
int tonic_ppo_collect_step(const float* d_actor_params, const float* d_observations,
                           const float* d_eps, const float* d_next_observations,
                           const float* d_rewards, const float* d_resets,
                           const float* d_terminations, float* d_seg_observations,
                           float* d_seg_actions, float* d_seg_next_observations,
                           float* d_seg_rewards, float* d_seg_resets, float* d_seg_terminations,
                           float* d_seg_log_probs, float* d_norm_acc, float* d_actions_out,
                           int64_t row, int64_t W, int32_t O, int32_t A, void* stream);

/* Latency-optimised form of tonic_ppo_collect_step for device-resident collectors: the actor's
 * MFMA operand images are packed ONCE per learner update (tonic_ppo_pack_actor, a float32 buffer
 * of tonic_ppo_packed_actor_floats(O, A) elements) and every per-step launch streams them from
 * L2 instead of re-staging 21 KB of weights through LDS.  Same mathematics on 16x16x4 tiles
 * (results equal tonic_ppo_collect_step's up to float32 summation order). */
int64_t tonic_ppo_packed_actor_floats(int32_t O, int32_t A);
int tonic_ppo_pack_actor(const float* d_actor_params, float* d_packed, int32_t O, int32_t A,
                         void* stream);
int tonic_ppo_collect_step_packed(const float* d_packed_actor, const float* d_observations,
                                  const float* d_eps, const float* d_next_observations,
                                  const float* d_rewards, const float* d_resets,
                                  const float* d_terminations, float* d_seg_observations,
                                  float* d_seg_actions, float* d_seg_next_observations,
                                  float* d_seg_rewards, float* d_seg_resets,
                                  float* d_seg_terminations, float* d_seg_log_probs,
                                  float* d_norm_acc, float* d_actions_out, int64_t row,
                                  int64_t W, int32_t O, int32_t A, void* stream);

/* `steps` consecutive environment steps of a device-resident collector whose inputs are laid out
 * step-major: d_observations [steps + 1, W, O] (row t + 1 is the next observation of step t, as
 * in tonic/utils/trainer.py:44-56), d_eps [steps, W, A] (may be NULL), d_rewards / d_resets /
 * d_terminations [steps, W].  Enqueues ONE tonic_ppo_collect_step_packed launch per step into
 * Segment rows row0 .. row0 + steps - 1; knowing where the next step's inputs live, each launch
 * also touches them so that the following one does not start with HBM round trips.
 * d_norm_acc may be NULL when the caller records the rollout with one tonic_meanstd_record. */
int tonic_ppo_collect_steps_packed(const float* d_packed_actor, const float* d_observations,
                                   const float* d_eps, const float* d_rewards,
                                   const float* d_resets, const float* d_terminations,
                                   float* d_seg_observations, float* d_seg_actions,
                                   float* d_seg_next_observations, float* d_seg_rewards,
                                   float* d_seg_resets, float* d_seg_terminations,
                                   float* d_seg_log_probs, float* d_norm_acc, int64_t row0,
                                   int64_t steps, int64_t W, int32_t O, int32_t A, void* stream);

/* ---- pinned-host batched collector --------------------------------------------------------------
 * replaces: tonic/environments/distributed.py:82-95,136-155 (Parallel: one pickled Pipe message per
 *   worker group and step, one shared Queue back) and the host <-> device hops of
 *   tonic/torch/agents/a2c.py:41-73 (torch.as_tensor / .numpy() around every forward pass).
 *
 * One page-aligned shared BLOCK (the caller maps it MAP_SHARED | MAP_ANONYMOUS before forking its
 * workers; tonic_collector_block_bytes gives the size) carries one environment step of all W workers
 * as float32 fields [W, ...] — the enum below — plus uint8 copies of the two flags for NumPy bool
 * views.  It is the ONLY exception to "the library owns no memory and never blocks": the collector
 * handle owns a stream, events and a few KB of device scratch, and the wait calls block.
 *
 * Environment side (no HIP; parent and forked workers): the parent publishes the actions with
 * tonic_collector_submit_actions (one futex wake for ALL worker groups), every group answers with
 * tonic_collector_worker_done after writing its rows, the parent blocks in tonic_collector_wait_obs
 * (TONIC_ERR_TIMEOUT instead of the reference's dead-worker hang).  tonic_collector_worker_wait
 * returns the new step sequence number (>= 0), -1 after tonic_collector_shutdown, -2 on timeout.
 *
 * Agent side: tonic_collector_create page-locks the block (hipHostRegister) so that the workers'
 * memory is the DMA source.  transport 0: the fused act kernel reads observations / noise / the
 * previous outcome from the mapped block over PCIe and writes actions + completion words back;
 * transport 1: hipMemcpyAsync H2D + kernel + hipMemcpyAsync D2H on the collector's own stream and
 * an event; transport 2: as 0, but the kernel is RESIDENT for a rollout — launched by the first
 * step, it waits for the host's next command word in pinned memory instead of being launched
 * again, and parks itself (the next step starts it again) after TONIC_AMD_COLLECTOR_PARK_US
 * (default 200) microseconds without a command, e.g. under a slow simulator or a test episode.
 * transport 3: as 2, but the HOST pushes the step's command word, observation rows and noise rows into a
 * window of fine-grained device memory (write-combined stores through the GPU's BAR, a store fence, then the
 * command) — the kernel polls and reads its own HBM instead of pulling those bytes over PCIe: one PCIe read
 * round trip and the poll's leave the step's critical path (6.2 -> 4.9 us for the bare exchange at W = 256,
 * O = 28; scripts/ubench/pingpong.hip).  Outcome fields and actions travel as in transport 2.  Only the
 * process that created the collector can store through the window: a forked worker group's ring leaves an
 * armed command to the parent (tonic_collector_ring / _claim).  Falls back to transport 2 when the device
 * memory is not CPU-visible (hipDeviceAttributeIsLargeBar), beyond 48 KB of observations per step (the
 * kernel's pull is faster than the host's stores there), when another collector already pushes into the
 * block, or with TONIC_AMD_COLLECTOR_PUSH=0; tonic_collector_transport tells.
 * Shapes beyond the fused act kernel (32 < O <= 384 or 8 < A <= 32) always run as transport 0 with
 * five launches per step (ingest, three dense layers, sample + store + completion words).
 * A step is idempotent — issuing the same row again overwrites the same rows and recomputes the same
 * MeanStd.record sums (kept as one entry per row between begin_rollout and end_rollout) — so a caller
 * may issue the next step before it knows that the block's contents are final, and repeat it.
 * Per environment step t (tonic/utils/trainer.py:44-56):
 *   tonic_collector_ppo_step(row t)   ONE launch: policy forward + sample + log-prob of the block's
 *                                     observations (a2c.py:75-85) -> Segment row t and the block's
 *                                     actions; MeanStd.record of the observations (a2c.py:66-69);
 *                                     and, with store_previous, the block's outcome fields
 *                                     (next_observations, rewards, resets, terminations — still
 *                                     those of step t-1) -> Segment row t-1 (segments.py:27-36).
 *                                     eps_slot selects the noise field (-1: the mode, no noise).
 *   tonic_collector_wait_actions      returns when the actions are in the block and every block
 *                                     field may be overwritten (no stream synchronisation).
 * tonic_collector_begin_rollout orders the collector's stream behind `learner_stream` (the
 * parameters were just updated there) and packs the actor; tonic_collector_end_rollout stores the
 * outcome of the last step (last_row; < 0: none pending), makes `learner_stream` wait for the
 * collector and returns once the block may be reused.
 */
typedef struct tonic_collector tonic_collector_t;
enum {
  TONIC_COLLECTOR_EPS0 = 0,            /* [W,A] standard-normal draws, slot 0               */
  TONIC_COLLECTOR_OBSERVATIONS = 1,    /* [W,O] post-reset observations the agent acts on   */
  TONIC_COLLECTOR_NEXT_OBSERVATIONS = 2, /* [W,O] pre-reset next observations (infos)       */
  TONIC_COLLECTOR_REWARDS = 3,         /* [W]                                               */
  TONIC_COLLECTOR_RESETS = 4,          /* [W] 0/1 as float32 (segments.py:33)               */
  TONIC_COLLECTOR_TERMINATIONS = 5,    /* [W] 0/1 as float32                                */
  TONIC_COLLECTOR_EPS1 = 6,            /* [W,A] noise slot 1                                */
  TONIC_COLLECTOR_ACTIONS = 7,         /* [W,A] written by the agent side                   */
  TONIC_COLLECTOR_RESETS_U8 = 8,       /* [W] uint8 copies of the flags (NumPy bool views)  */
  TONIC_COLLECTOR_TERMINATIONS_U8 = 9,
  TONIC_COLLECTOR_DONE_FLAGS = 10,     /* uint32 completion words of the act launch (internal) */
  TONIC_COLLECTOR_FIELD_COUNT = 11
};
int64_t tonic_collector_block_bytes(int64_t W, int32_t O, int32_t A);
int tonic_collector_block_init(void* block, int64_t bytes, int64_t W, int32_t O, int32_t A,
                               int32_t worker_groups);
int64_t tonic_collector_block_offset(const void* block, int32_t field);   /* bytes; < 0: error */
/* The environment's promise about the rows it writes (tonic/environments/distributed.py:41-57: a
 * worker's `observations` row of step t + 1 IS its `next_observations` row of step t unless it reset,
 * in which case it is the reset observation): with it, steps of many workers (W * O >= 16 384 floats,
 * where the step is bound by PCIe bytes) fetch each observation row ONCE — the act launch writes the
 * rows of the workers whose reset flag is 0 to the previous Segment row's next_observations from the
 * copy it holds anyway, only the rows of workers that reset are read from NEXT_OBSERVATIONS.  Off by
 * default: a block filled by anybody else keeps the two fields independent. */
int tonic_collector_block_carry_over(void* block, int32_t promised);
/* A vectorised simulator's step record in one host call (no reference counterpart: the per-worker
 * loop of tonic/environments/distributed.py:28-58 for the zero-cost synthetic benchmark
 * environment of SURVEY.md §8d): next_observations [W,O] -> the NEXT_OBSERVATIONS and OBSERVATIONS
 * fields, rewards[w] = -sum_a actions[w][a]^2 (float32, left to right).  actions == NULL: the
 * block's ACTIONS field.  Flags are the caller's business; ring != 0: they are final as well, the
 * record is complete -> tonic_collector_ring from inside the call. */
int tonic_collector_synthetic_step(void* block, const float* next_observations,
                                   const float* actions, int32_t ring);
/* The environment's side of an ARMED step (tonic_collector_arm below): the step record in the block
 * is complete — issue the command the agent has prepared for exactly this moment.  Host stores only
 * (no HIP): callable from a forked worker; tonic_collector_worker_done of the LAST worker group does
 * it itself, before it wakes the parent.  Returns 1 if a command went out, 0 if nothing was armed. */
int tonic_collector_ring(void* block);
int64_t tonic_collector_worker_wait(void* block, int64_t seen_sequence, double timeout_s);
int tonic_collector_worker_done(void* block);
int tonic_collector_submit_actions(void* block);
int tonic_collector_wait_obs(void* block, double timeout_s);
int tonic_collector_shutdown(void* block);

int tonic_collector_create(tonic_collector_t** out, void* block, int32_t transport);
/* The address at which kernels see page-locked host memory (hipHostMalloc'ed, e.g. a pinned torch
 * tensor, or registered), NULL if it is not mapped into the current device's address space.  The
 * agents' per-step staging passes such pointers to the ordinary entry points (tonic_ppo_act_wide,
 * tonic_policy_forward, tonic_segment_store, tonic_buffer_store ...): the kernels then read the
 * step's inputs and write its results in host memory, no copy in either direction. */
void* tonic_host_device_pointer(void* pinned_host);
int tonic_collector_destroy(tonic_collector_t* collector);
void* tonic_collector_stream(tonic_collector_t* collector);      /* the collector's hipStream_t */
int32_t tonic_collector_transport(tonic_collector_t* collector); /* the transport in effect (see above) */
int tonic_collector_bind_segment(tonic_collector_t* collector, float* d_seg_observations,
                                 float* d_seg_actions, float* d_seg_next_observations,
                                 float* d_seg_rewards, float* d_seg_resets,
                                 float* d_seg_terminations, float* d_seg_log_probs,
                                 float* d_norm_acc, int64_t segment_rows);
int tonic_collector_begin_rollout(tonic_collector_t* collector, const float* d_actor_params,
                                  void* learner_stream);
int tonic_collector_ppo_step(tonic_collector_t* collector, int64_t row, int32_t eps_slot,
                             int32_t store_previous);
int tonic_collector_wait_actions(tonic_collector_t* collector, double timeout_s);
/* tonic_collector_arm: the command of tonic_collector_ppo_step(row, eps_slot, store_previous), NOT
 * issued but left in the block's header for the environment side to issue (tonic_collector_ring) the
 * moment its step record is complete — the trainer's path from `environment.step` returning to
 * `agent.update` / `agent.step` (tonic/utils/trainer.py:44-56) then runs while the GPU works instead
 * of in front of it.  Returns 1 (armed), 0 (not possible now: only a resident kernel, transport 2,
 * past the first step of its rollout, can be commanded by a process that cannot launch), < 0 errors.
 * tonic_collector_claim: 1 = the environment issued it, the step is in flight (wait_actions next);
 * 0 = it did not (the command is withdrawn; issue tonic_collector_ppo_step yourself).  Every other
 * entry point claims first, so an armed command can never be issued behind the handle's back. */
int tonic_collector_arm(tonic_collector_t* collector, int64_t row, int32_t eps_slot,
                        int32_t store_previous);
int tonic_collector_claim(tonic_collector_t* collector);
int tonic_collector_end_rollout(tonic_collector_t* collector, int64_t last_row,
                                void* learner_stream);

/* ---- a gate in a stream ---------------------------------------------------------------------------
 * replaces: nothing in the reference (ppo.py:33-46 runs the critic's iterations between the actor's;
 *   tonic_amd runs them on a second stream under the NEXT rollout, agents.PPO._update).  Work enqueued
 *   on `stream` behind the gate starts when the caller stores a value >= `value` into `host_word`
 *   (page-locked host memory: a plain CPU store opens it, no launch) or after `timeout_seconds`,
 *   whichever comes first: one wave polls the word at system scope.  PPO enqueues the critic's 160
 *   launches while the host has nothing else to do and opens the gate at the Segment row from which
 *   they just finish before the rollout does — the next update's first launches then find the device at
 *   its working point (profiles/r04_clock_ramp.md).  Timing only: what the gated work computes does not
 *   depend on when the gate opens. */
int tonic_stream_gate(const uint32_t* host_word, uint32_t value, double timeout_seconds, void* stream);

/* ---- one-shot all-reduce between the GPUs of a node -----------------------
 * replaces: nothing in the reference (single process); it is the exchange step of the sharded
 *   learner: after the fused grad kernels every rank holds gradient SUMS over its worker shard
 *   (+ 8 statistic sums), the sum over ranks is what tonic_adam_step scales by 1 / N_global.
 * For the 44 KB - 711 KB buffers of this path a ring all-reduce is latency-bound (2 (G - 1) hops
 * over point-to-point xGMI).  tonic_allreduce_f32 is ONE launch per rank: each rank writes its
 * buffer into a window of every peer (G - 1 links used concurrently, one hop), raises a flag
 * there, waits for the flags in its own window and adds the G contributions IN RANK ORDER, in
 * place — identical bits on every rank, independent of arrival order.  n <= max_floats.
 * Set-up (once): tonic_comm_init allocates this rank's window; tonic_comm_export gives its
 * hipIpcMemHandle_t (tonic_comm_handle_bytes() bytes) which the caller carries to the other
 * processes by any means; tonic_comm_connect takes all `world` handles in rank order.
 * A peer that never arrives makes the kernel give up (TONIC_AMD_ALLREDUCE_TIMEOUT_S, default 120 s;
 * tonic_comm_set_timeout overrides it for one communicator, 0 = back to the default) instead of
 * hanging: tonic_comm_status (synchronous, call it where the host reads results anyway) reports it
 * (and keeps reporting it: a communicator that missed a peer is not to be used again).
 * tonic_comm_can_access_peer: hipDeviceCanAccessPeer(device, peer) as 1 / 0 (negative: error) —
 * what tonic_amd.parallel asks of every pair of ranks before it picks this exchange over RCCL. */
typedef struct tonic_comm tonic_comm_t;
int64_t tonic_comm_handle_bytes(void);
int tonic_comm_init(tonic_comm_t** out, int32_t rank, int32_t world, int64_t max_floats);
int tonic_comm_export(tonic_comm_t* comm, void* handle_out);
int tonic_comm_connect(tonic_comm_t* comm, const void* all_handles);
int tonic_allreduce_f32(tonic_comm_t* comm, float* d_buffer, int64_t n, void* stream);
int tonic_comm_status(tonic_comm_t* comm);
int tonic_comm_set_timeout(tonic_comm_t* comm, double seconds);
int tonic_comm_can_access_peer(int32_t device, int32_t peer);
int tonic_comm_destroy(tonic_comm_t* comm);

/* ---- target networks (SAC / TD3) ---------------------------------------------------------------
 * replaces: tonic/torch/models/actor_critics.py:126-130 (update_targets): per element
 *   t = fl(fl(t*(1-coeff)) + fl(coeff*o)) — three roundings, no FMA — on flat buffers.
 */
int tonic_polyak_update(float* d_target, const float* d_online, int64_t n, double coeff,
                        void* stream);
/* replaces: the optimizer step of the actor followed by update_targets() (ddpg.py:105-112,
 *   td3.py:43-46, sac via ddpg.py) in ONE launch: Adam (as tonic_adam_step, stats_kind 0 / 4, no
 *   skip flag) on the block [block_offset, block_offset + param_count) of the online buffer, then
 *   the polyak update of ALL total_count target entries — each thread updates the target of the
 *   entry it just stepped, extra workgroups cover the entries outside the block (whose online
 *   values are final: their optimizer ran in an earlier launch).  Same roundings as the two calls. */
int tonic_adam_polyak_step(float* d_online, const float* d_grad_sums, float* d_exp_avg,
                           float* d_exp_avg_sq, int32_t* d_state, int64_t block_offset,
                           int64_t param_count, int64_t total_count, double grad_scale, double lr,
                           double beta1, double beta2, double eps, int32_t stats_kind,
                           float* d_info_row, float* d_target, double coeff, void* stream);

/* ======================= off-policy path: SAC / TD3 (2 hidden ReLU layers of width H) =========
 * Flat parameter blocks (reference `parameters()` order):
 *   deterministic actor (TD3): W1[H,O] b1[H] W2[H,H] b2[H] Wa[A,H] ba[A]              (heads = 1)
 *   Gaussian actor (SAC)     : W1 b1 W2 b2 Wloc[A,H] bloc[A] Wscale[A,H] bscale[A]    (heads = 2)
 *   Q critic                 : W1[H,O+A] b1[H] W2[H,H] b2[H] w3[1,H] b3[1]
 *   twin critics             : [critic_1 | critic_2] contiguous (one Adam over both, as
 *                              tonic/torch/updaters/critics.py:148-153,195-200 build it).
 * Off-policy parameter layout (these blocks only; the PPO blocks above are dense): tensors in
 *   that order, every tensor starts on a 16-byte boundary (a 1-D tensor of n floats occupies
 *   ceil(n/4)*4), and the rows of a [rows, cols] weight are tonic_mlp_weight_stride(cols) floats
 *   apart — cols rounded up to a multiple of 4, plus 4 if that is a multiple of 32 (256 -> 260,
 *   111 -> 112, 119 -> 120).  The kernels stream weight rows straight from L2 into MFMA
 *   operands, one row per lane: 16-byte aligned rows keep every load one request, and a stride
 *   that is not a multiple of 128 bytes doubles the stream rate (38 -> 70-84 B/ns per CU,
 *   profiles/r01_ubench_row_stride.md).  Padding floats must be zero; they stay zero under
 *   tonic_adam_step (zero gradient) and tonic_polyak_update.  The *_param_count queries return the
 *   padded block length; gradient / Adam-moment buffers use the same layout and length.
 * All scratch comes from ONE caller-provided workspace (tonic_offpolicy_workspace_bytes).
 * Torsos other than the reference's (tonic/torch/models/utils.py:4-23 accepts any MLP(sizes, activation)): for
 *   the SAC / TD3 / DDPG entries — tonic_policy_forward, tonic_twin_q_grad, tonic_actor_q_grad, the three size
 *   queries below — `H` may be tonic_mlp_hidden(H1, H2, activation): two hidden layers of H1 and H2 units
 *   (1 .. 4095: the (400, 300) class; a plain width — two ReLU layers of H units — is passed as is, any width), activation 1 = torch.nn.ReLU, 2 = Tanh, 3 = ELU; W2 is then [H2, H1],
 *   heads / w3 are H2 wide, same padding rules.  Such torsos run layer by layer (csrc/gemm16.hip) instead of in
 *   the fused kernels; tonic_q_iteration and the D4PG / MPO entries take plain widths only.
 */
int64_t tonic_offpolicy_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H);
int32_t tonic_mlp_weight_stride(int32_t cols);
int32_t tonic_mlp_hidden(int32_t H1, int32_t H2, int32_t activation);
int64_t tonic_mlp_actor_param_count(int32_t O, int32_t H, int32_t A, int32_t heads);
int64_t tonic_q_critic_param_count(int32_t O, int32_t A, int32_t H);

/* replaces: tonic/replays/buffers.py:33-56 (Buffer.store: one row of the circular buffers incl.
 *   discounts = float32(1 - terminations) * discount_factor) + mean_stds.py:44-48 (record). */
int tonic_buffer_store(float* d_buf_observations, float* d_buf_actions,
                       float* d_buf_next_observations, float* d_buf_rewards, float* d_buf_resets,
                       float* d_buf_terminations, float* d_buf_discounts,
                       const float* d_observations, const float* d_actions,
                       const float* d_next_observations, const float* d_rewards,
                       const float* d_resets, const float* d_terminations, float* d_norm_acc,
                       int64_t row, int64_t W, int32_t O, int32_t A, double discount_factor,
                       void* stream);

/* replaces: tonic/replays/buffers.py:58-79 (Buffer.accumulate_n_steps, return_steps > 1): call
 *   right after tonic_buffer_store wrote row `row`, with `size` = the number of filled rows
 *   BEFORE this store.  Folds the new reward / discount / next observation into the previous
 *   min(size, return_steps - 1) rows until a stored reset cuts the chain.  Bit-exact. */
int tonic_buffer_accumulate_n_steps(float* d_buf_next_observations, float* d_buf_rewards,
                                    float* d_buf_discounts, const float* d_buf_resets,
                                    const float* d_next_observations, const float* d_rewards,
                                    const float* d_terminations, int64_t row, int64_t size,
                                    int64_t max_size, int64_t W, int32_t O, int32_t return_steps,
                                    double discount_factor, void* stream);

/* replaces: tonic/replays/buffers.py:84-91 (Buffer.get: rows = idx // W, cols = idx % W, fancy-
 *   index gather of 5 keys).  d_indices: int64[B] drawn by the host RandomState (bit-exact
 *   stream); one wavefront copies one sampled transition. */
int tonic_buffer_gather(const int64_t* d_indices, const float* d_buf_observations,
                        const float* d_buf_actions, const float* d_buf_next_observations,
                        const float* d_buf_rewards, const float* d_buf_discounts,
                        float* d_observations, float* d_actions, float* d_next_observations,
                        float* d_rewards, float* d_discounts, int64_t W, int32_t B, int32_t O,
                        int32_t A, void* stream);

/* replaces: DDPG._greedy_actions (tonic/torch/agents/ddpg.py:78-81; kind 0 = tanh head,
 *   models/actors.py:113-115) and SAC._stochastic_actions / _greedy_actions
 *   (tonic/torch/agents/sac.py:40-51; kind 1 = tanh(loc + sigma * eps), eps NULL -> tanh(loc)). */
int tonic_policy_forward(const float* d_actor_params, const float* d_observations,
                         const float* d_eps, float* d_actions, int32_t kind, int32_t B, int32_t O,
                         int32_t H, int32_t A, void* d_workspace, int64_t workspace_bytes,
                         void* stream);

/* replaces: kind 0 TwinCriticDeterministicQLearning.__call__ (tonic/torch/updaters/
 *   critics.py:156-175, TargetActionNoise :125-134; d_policy_params = TARGET actor),
 *   kind 1 TwinCriticSoftQLearning.__call__ (critics.py:202-227; d_policy_params = ONLINE
 *   actor, quirk Q9), kind 2 DeterministicQLearning.__call__ (critics.py:68-86, DDPG: ONE critic
 *   block in d_target_critics / d_critics, target actor, no noise, d_eps may be NULL; output
 *   [critic sums | 8 statistics {sq_err_sum, q_sum, 0, ...}]).
 *   d_eps [B,A]: host-drawn standard normals (torch CPU generator order).
 *   Output: gradient SUMS over the batch for [critic_1 | critic_2] + 8 statistics
 *   {sq_err_sum (both critics), q1_sum, q2_sum, 0, 0, B, 0, 0}; follow with
 *   tonic_adam_step(grad_scale = 1/B). */
int tonic_twin_q_grad(int32_t kind, const float* d_policy_params, const float* d_target_critics,
                      const float* d_critics, const float* d_norm_mean, const float* d_norm_std,
                      double norm_clip, const float* d_observations, const float* d_actions,
                      const float* d_next_observations, const float* d_rewards,
                      const float* d_discounts, const float* d_eps, float* d_grad_sums, int32_t B,
                      int32_t O, int32_t H, int32_t A, double entropy_coeff, double noise_scale,
                      double noise_clip, void* d_workspace, int64_t workspace_bytes, void* stream);

/* replaces: kind 0 DeterministicPolicyGradient.__call__ on critic_1 (tonic/torch/updaters/
 *   actors.py:170-189, td3.py:36), kind 1 TwinCriticSoftDeterministicPolicyGradient.__call__
 *   (actors.py:238-267).  Critics are frozen (no weight gradients).  Output: gradient SUMS for
 *   the actor + 8 statistics {loss_sum, 0, 0, 0, 0, B, 0, 0}. */
int tonic_actor_q_grad(int32_t kind, const float* d_actor_params, const float* d_critics,
                       const float* d_norm_mean, const float* d_norm_std, double norm_clip,
                       const float* d_observations, const float* d_eps, float* d_grad_sums,
                       int32_t B, int32_t O, int32_t H, int32_t A, double entropy_coeff,
                       void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- one whole learner iteration of DDPG / TD3 / SAC in 8 launches (4 when the actor is not due).
 * replaces: the body of DDPG._update's loop (tonic/torch/agents/ddpg.py:95-112; td3.py:38-55 with
 *   delay_steps) for one batch: critic_updater(**batch) -> [actor_updater(observations) ->
 *   model.update_targets()], i.e. tonic_twin_q_grad + tonic_adam_step [+ tonic_actor_q_grad +
 *   tonic_adam_polyak_step] with the launches of the two steps merged where they do not depend on
 *   each other (both policy passes are one launch; forward, loss and input-gradient chain of a step
 *   are one launch whose workgroups hand the few floats that cross them over through an exchange
 *   area polled with agent-scope loads: 5 launches with the actor due, 3 without) and torch.optim.Adam [+ the polyak update] applied in the epilogue of the
 *   weight-gradient launches (same expressions as tonic_adam_step / tonic_polyak_update).
 *   Single rank and no gradient clipping: several ranks (an all-reduce sits between the gradients
 *   and the step) and `gradient_clip` use the split entry points.  kind as tonic_twin_q_grad
 *   (0 TD3, 1 SAC, 2 DDPG).  Statistics rows as tonic_adam_step writes them (stats_kind 3 / 4).
 *   All pointers are device memory unless noted; asynchronous on `stream`. */
typedef struct tonic_q_optimizer {
  float* d_grad_sums;          /* [count + 8]: the gradient sums + statistic slots (written) */
  float* d_exp_avg;            /* Adam moments of the block (same layout as the parameters)   */
  float* d_exp_avg_sq;
  int32_t* d_state;            /* {step_count, -, -, arrivals}                                */
  float* d_info_row;           /* [8] logged statistics of this step (may be NULL)            */
  const float* d_step_constants; /* [2] {lr / (1 - beta1^t), sqrt(1 - beta2^t)} of THIS step t as
                                  float32 of the float64 values (adam.py:530-536), formed by the
                                  host; NULL: formed on the device from d_state's counter       */
  double lr, beta1, beta2, eps;
} tonic_q_optimizer;

typedef struct tonic_q_iteration_t {
  int32_t kind, actor_due;     /* actor_due: this iteration also steps the actor and the targets */
  int32_t B, O, H, A;
  int64_t global_batch;        /* gradient scale 1 / global_batch (0: B)                      */
  float* d_actor;              /* online parameter blocks (padded off-policy layout)          */
  float* d_critics;            /* [critic_1 | critic_2] (DDPG: one critic)                    */
  float* d_target_actor;
  float* d_target_critics;
  const float* d_norm_mean; const float* d_norm_std; double norm_clip;
  const float* d_observations; const float* d_actions; const float* d_next_observations;
  const float* d_rewards; const float* d_discounts;
  const float* d_eps_critic;   /* [B, A] TD3: target-action noise, SAC: the next action's draw */
  const float* d_eps_actor;    /* [B, A] SAC: the actor step's draw (else NULL)               */
  double critic_entropy_coeff, actor_entropy_coeff, noise_scale, noise_clip, target_coeff;
  tonic_q_optimizer critic, actor;
  void* d_workspace; int64_t workspace_bytes;   /* ZERO-FILLED when first handed over: its first
                                  word is the failure flag of the chained launches (critic step:
                                  targets -> online critics' loss and chain; actor step: critics ->
                                  objective -> chain -> actor chain), whose workgroups hand the few
                                  floats that cross them over through an exchange area in the
                                  workspace: written once per iteration with agent-scope stores, read
                                  with agent-scope loads until they are no longer the "empty" pattern
                                  the iteration's first launch fills the area with                  */
  int32_t phase;               /* 0: the whole iteration, optimizer steps in the weight-gradient launches'
                                  epilogues (one rank, no gradient clipping).  Several ranks / clipping
                                  need the COMPLETE gradient sums between gradients and step — the same
                                  chained launches in two halves, gradient SUMS + statistic slots only:
                                  1: policy passes + critic step + the critics' weight gradients
                                  -> critic.d_grad_sums; the caller exchanges / clips them and steps the
                                  critics (tonic_adam_step), then (actor_due) 2: the actor step on the
                                  UPDATED critics + the actor's weight gradients -> actor.d_grad_sums, the
                                  caller steps the actor and the targets (tonic_adam_polyak_step).  In
                                  these phases the failure word is only ever SET (the caller clears it
                                  ahead of an update and looks at it after it)                         */
  int32_t refresh_images;      /* The fused passes read their weights from fp16x2 operand-order IMAGES kept in the
                                  workspace (csrc/mlpimg.h: two binary16 terms per weight in MFMA operand order;
                                  products = three fp16 MFMAs, fp32 accumulation — float32-class accuracy).  The
                                  float32 parameter blocks stay the authority: 1 = rebuild every image from them
                                  first — on the first iteration of an update call and whenever parameters were
                                  written by anything but this entry's own optimizer epilogues since the workspace
                                  last served (the phases always rebuild); 0 = the images the previous iteration's
                                  epilogues left (they follow every Adam / polyak write).                          */
  int32_t stage;               /* 0: the whole iteration; 2: everything BEHIND the policy passes (launch 1), which the
                                  previous call has run ahead (see `ahead`): their outputs are in set `slot`.  phase 0. */
  int32_t slot;                /* 0 / 1: which of the workspace's two sets of launch-1 outputs, exchange area and
                                  failure word this iteration works on                                              */
  const struct tonic_q_iteration_t* ahead;   /* NULL, or (host memory) the NEXT iteration's arguments: its policy
                                  passes — they read the actor, the target actor, ITS batch and noise, none of which
                                  a critic step writes — run as more workgroups of THIS iteration's critic-step
                                  launch, into set ahead->slot (the other one).  Only when this iteration does not
                                  step the actor (actor_due = 0: delayed updates, td3.py:43-46) and
                                  tonic_q_iteration_ahead_supported says so; the next call then passes stage = 2.   */
} tonic_q_iteration_t;

int64_t tonic_q_iteration_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H);
/* 1 when tonic_q_iteration serves these shapes (heads: 1 deterministic, 2 Gaussian policy) */
int tonic_q_iteration_supported(int32_t O, int32_t H, int32_t A, int32_t heads);
/* 1 when the next iteration's `passes` (1, 2 = it steps the actor) policy passes fit beside a critic step of `nets`
 * twin / single critics on B rows (the image passes, every workgroup of the launch resident at once) */
int tonic_q_iteration_ahead_supported(int32_t B, int32_t O, int32_t H, int32_t A, int32_t nets, int32_t passes);
int tonic_q_iteration(const tonic_q_iteration_t* iteration, void* stream);

/* ---- off-policy acting on a collector block --------------------------------------------------------------
 * replaces: tonic/torch/agents/ddpg.py:45-52 (`_greedy_actions`), sac.py:40-51 (`_stochastic_actions` /
 *   `_greedy_actions`) for observations that the environment has written into a collector block (tonic_collector_*
 *   above; tonic_amd.environments do): ONE launch on `stream` reads the block's W observation rows in place
 *   (page-locked host memory), runs the policy of tonic_policy_forward (kind 0: deterministic tanh head, 1: squashed
 *   Gaussian — eps_slot 0: the block's first noise field holds the host's standard-normal draws, -1: the greedy loc),
 *   writes the actions into the block's SECOND noise field (TONIC_COLLECTOR_EPS1: host-visible) and one completion
 *   word per 16 rows at system scope; tonic_collector_wait_actions is the wait.  d_rows_out [W, O] (may be NULL)
 *   receives a device copy of the observation rows (Buffer.store of the transition runs after the environment's
 *   step has overwritten the block's).  d_workspace: tonic_offpolicy_workspace_bytes(W, O, A, H).  The forward runs
 *   on the actor's fp16x2 weight images (csrc/mlpimg.h; the input rows are fetched over PCIe in ONE round trip):
 *   d_actor_images = tonic_mlp_actor_image_bytes(O, H, A, heads) bytes owned by the caller, rebuild_images = 1
 *   whenever the parameters changed since the last call (after every learner update).  Shapes with 0 image bytes:
 *   TONIC_ERR_UNSUPPORTED_SHAPE (callers keep tonic_policy_forward for those). */
/* store (may be NULL): the transition of the step BEFORE rides in the same launch — one more workgroup runs
 *   tonic_buffer_store's body (Buffer.store + MeanStd.record, replays/buffers.py:33-52, mean_stds.py:44-48) with the
 *   block's outcome fields (executed actions, next observations, rewards, resets, terminations: read in place) and
 *   d_observations = the device copy of that step's observation rows (the d_rows_out of ITS acting call: callers
 *   alternate two buffers), into row `row` of the HBM Buffer; its completion word follows the tiles'.  The
 *   environment may overwrite the block once tonic_collector_wait_actions has returned. */
typedef struct tonic_q_store_t {
  float* d_buf_observations; float* d_buf_actions; float* d_buf_next_observations; float* d_buf_rewards;
  float* d_buf_resets; float* d_buf_terminations; float* d_buf_discounts;
  const float* d_observations;   /* [W, O] */
  float* d_norm_acc;             /* MeanStd running sums [2 O] (may be NULL) */
  int64_t row;
  double discount_factor;
} tonic_q_store_t;
int64_t tonic_mlp_actor_image_bytes(int32_t O, int32_t H, int32_t A, int32_t heads);   /* 0: not served */
int tonic_collector_q_act(tonic_collector_t* collector, const float* d_actor_params, void* d_actor_images,
                          int32_t rebuild_images, int32_t kind, int32_t H, int32_t eps_slot, float* d_rows_out,
                          const tonic_q_store_t* store, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- D4PG: distributional critic (tonic/torch/models/critics.py:23-66, agents/d4pg.py).
 *   The critic is an actor-shaped network on the encoded input [normalised observation | action]:
 *   parameters in the layout of tonic_mlp_actor_param_count(O + A, H, NA, 1) — W1 [H, O + A], b1,
 *   W2, b2, distributional_layer [NA, H], bias [NA]; 2 <= NA <= 64 atoms; d_values = the support
 *   (DistributionalValueHead.values, float32 [NA], ascending) — of the TARGET critic's head for
 *   tonic_distributional_q_grad (returns and projection are taken from the target distribution,
 *   critics.py:104-109), of the online critic's for tonic_distributional_actor_grad.
 *
 * tonic_distributional_q_grad — DistributionalDeterministicQLearning.__call__ (updaters/critics.py:
 *   100-122) up to the optimizer step: a' = target_actor(s'), the target critic's distribution at
 *   (s', a') projected onto the support at r + discount * z (CategoricalWithSupport.project,
 *   critics.py:32-46), cross-entropy against log_softmax of the online critic's logits at (s, a).
 *   Output: gradient SUMS of the critic + 8 statistics {loss_sum, 0, 0, 0, 0, B, 0, 0}.
 * tonic_distributional_actor_grad — DistributionalDeterministicPolicyGradient.__call__
 *   (updaters/actors.py:203-224): loss = -mean(sum_i softmax(critic(s, actor(s)))_i z_i), critic
 *   frozen.  Output: gradient SUMS of the actor + the same 8 statistics. */
int64_t tonic_distributional_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H, int32_t NA);
int tonic_distributional_q_grad(const float* d_target_actor, const float* d_target_critic,
                                const float* d_critic, const float* d_norm_mean,
                                const float* d_norm_std, double norm_clip,
                                const float* d_observations, const float* d_actions,
                                const float* d_next_observations, const float* d_rewards,
                                const float* d_discounts, const float* d_values,
                                float* d_grad_sums, int32_t B, int32_t O, int32_t H, int32_t A,
                                int32_t NA, void* d_workspace, int64_t workspace_bytes,
                                void* stream);
int tonic_distributional_actor_grad(const float* d_actor_params, const float* d_critic,
                                    const float* d_norm_mean, const float* d_norm_std,
                                    double norm_clip, const float* d_observations,
                                    const float* d_values, float* d_grad_sums, int32_t B,
                                    int32_t O, int32_t H, int32_t A, int32_t NA,
                                    void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- MPO (tonic/torch/agents/mpo.py): Gaussian policy head with a tanh loc and
 *   sigma = clamp(softplus(.), 1e-4, 1) (models/actors.py:69-98, two heads), ONE critic with
 *   targets (layouts as for DDPG / SAC); S <= 64 sampled actions per state, tiled like
 *   updaters.tile + merge_first_two_dims (row s * B + m); d_eps = the standard-normal draws
 *   [S, B, A] in the order Normal.rsample / Normal.sample consume them.
 *   tonic_policy_forward kind 2 acts with this head (a = loc + sigma * eps; eps NULL = loc).
 *
 * tonic_expected_sarsa_grad — ExpectedSARSA.__call__ (updaters/critics.py:253-282) up to the
 *   optimizer step: returns = r + discount * mean_s target_critic(s', a'_s), a'_s ~ target_actor(s');
 *   MSE against critic(s, a).  Output: gradient SUMS of the critic + {sq_err_sum, q_sum, 0, 0, 0, B, 0, 0}.
 * tonic_mpo_actor_grad — MaximumAPosterioriPolicyOptimization.__call__ (updaters/actors.py:318-464)
 *   up to the optimizer steps, per-dimension KL constraints: E-step weights softmax_s(Q / temperature)
 *   (+ the action-bound penalty weights), decomposed fixed-std / fixed-mean policy losses, the
 *   alpha-weighted KL terms and the dual losses.  d_duals [2 A + 2] = {log_temperature,
 *   log_alpha_mean[A], log_alpha_std[A], log_penalty_temperature}: floored IN PLACE at min_log_dual by
 *   this call, as the reference clamps them at the head of its own (actors.py:347-356) — every read goes
 *   through the floor, the last kernel ahead of the duals' optimizer step writes the floored values back
 *   (the penalty temperature only with action_penalization).  Outputs: d_grad_sums = gradient SUMS of the actor + {B * (policy + KL losses), 0, 0,
 *   0, 0, B, 0, 0}; d_dual_grads [2 A + 2 + 8] = d loss / d log-duals + {0, 0, 0, 0, 0, 1, 0, 0};
 *   d_stats [9 + 2 A] = {policy_mean_loss, policy_std_loss, kl_mean_loss, kl_std_loss, alpha_mean_loss,
 *   alpha_std_loss, temperature_loss, temperature, alpha_mean[A], alpha_std[A], penalty_temperature}. */
int64_t tonic_mpo_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H, int32_t S);
int tonic_expected_sarsa_grad(const float* d_target_actor, const float* d_target_critic,
                              const float* d_critic, const float* d_norm_mean,
                              const float* d_norm_std, double norm_clip,
                              const float* d_observations, const float* d_actions,
                              const float* d_next_observations, const float* d_rewards,
                              const float* d_discounts, const float* d_eps, float* d_grad_sums,
                              int32_t B, int32_t O, int32_t H, int32_t A, int32_t S,
                              void* d_workspace, int64_t workspace_bytes, void* stream);
int tonic_mpo_actor_grad(const float* d_actor_params, const float* d_target_actor,
                         const float* d_target_critic, float* d_duals, double min_log_dual,
                         const float* d_norm_mean, const float* d_norm_std, double norm_clip,
                         const float* d_observations, const float* d_eps, float* d_grad_sums,
                         float* d_dual_grads, float* d_stats, int32_t B, int32_t O, int32_t H,
                         int32_t A, int32_t S, double epsilon, double epsilon_penalty,
                         double epsilon_mean, double epsilon_std, int32_t action_penalization,
                         void* d_workspace, int64_t workspace_bytes, void* stream);
/* The same step when the batch is spread over several ranks.  The dual losses and gradients are
 * functions of batch MEANS of per-state terms, so the step runs in two halves around one all-reduce:
 * tonic_mpo_actor_grad_shard — everything of tonic_mpo_actor_grad for this rank's B states (the
 *   actor's gradient SUMS -> d_grad_sums, statistics slot left to the second half) and, instead of the
 *   dual step, the local column sums d_column_sums [6 + 2 A] (float64: the six per-state partials,
 *   kl_mean[A], kl_std[A]); the caller sum-all-reduces them;
 * tonic_mpo_dual_step — dual gradients, logged losses (d_stats) and the actor's statistics slot
 *   (d_actor_stats = d_grad_sums + actor parameter count: {B * losses, 0, 0, 0, 0, B, 0, 0} with this
 *   rank's B, possibly 0) from the all-reduced sums and the global batch size. */
int tonic_mpo_actor_grad_shard(const float* d_actor_params, const float* d_target_actor,
                               const float* d_target_critic, float* d_duals, double min_log_dual,
                               const float* d_norm_mean, const float* d_norm_std, double norm_clip,
                               const float* d_observations, const float* d_eps, float* d_grad_sums,
                               double* d_column_sums, int32_t B, int32_t O, int32_t H, int32_t A,
                               int32_t S, int32_t action_penalization, void* d_workspace,
                               int64_t workspace_bytes, void* stream);
int tonic_mpo_dual_step(const double* d_column_sums, float* d_duals, double min_log_dual,
                        float* d_dual_grads,
                        float* d_stats, float* d_actor_stats, int32_t B, int32_t B_global,
                        int32_t A, int32_t S, double epsilon, double epsilon_penalty,
                        double epsilon_mean, double epsilon_std, int32_t action_penalization,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TONIC_HIP_H */
