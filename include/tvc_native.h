/*
 * tvc_native.h -- C ABI of the MI355X-native rocket-TVC hot path (libtvc_hip.so).
 *
 * The reference (NIKHILSAI71/TVC-AI) has no FFI/plugin boundary: it is 100 % Python and its hot
 * path is the Gymnasium surface of env/enhanced_rocket_tvc_env.py plus agent.update() of
 * agent/multi_algorithm_agent.py.  This header is the boundary a maintainer binds with ctypes
 * (INTEGRATION.md shows the stub); every entry point cites the reference method it replaces.
 *
 * Conventions
 *   - return 0 on success, negative TVC_E* code otherwise; tvc_last_error() gives the message
 *     (thread-local).
 *   - every *_dev pointer is caller-owned DEVICE memory (e.g. a torch tensor's data_ptr()); the
 *     library owns only the SoA state inside its handles.
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on it, no hidden
 *     synchronisation, graph-capturable (no allocation inside step/reset/update calls).
 *   - a handle belongs to one GPU; calls on one handle must be serialised by the caller.
 *   - there is NO CPU fallback: without a visible gfx950 device create() fails with TVC_ENODEV.
 */
#ifndef TVC_NATIVE_H
#define TVC_NATIVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVC_ABI_VERSION 2

enum {
    TVC_OK = 0,
    TVC_EINVAL = -1,  /* bad argument (shape, null pointer, range) */
    TVC_ENODEV = -2,  /* no usable HIP device */
    TVC_EHIP = -3,    /* a HIP runtime call failed (message in tvc_last_error) */
    TVC_ENOMEM = -4
};

const char* tvc_last_error(void);
int tvc_abi_version(void);

/* ------------------------------------------------------------------ vector env (K1/K2) */

typedef struct tvc_env tvc_env;

/* Plain-old-data configuration.  Defaults (tvc_env_default_cfg) are the constants of
 * env/enhanced_rocket_tvc_env.py:324-352 (_setup_physics) and :409-464 (_create_enhanced_rocket). */
typedef struct tvc_env_cfg {
    double mass;            /* 2.0 kg                                   ref env/...:412 */
    double inertia_xx;      /* (1/12) m (3 r^2 + l^2) = Iyy             ref :431 */
    double inertia_zz;      /* (1/2) m r^2                              ref :432 */
    double thrust;          /* 35 N                                     ref :463 */
    double half_len;        /* 0.5 m  (thrust applied at body z=-0.5)   ref :550 */
    double radius;          /* 0.05 m                                   ref :414 */
    double lin_damp;        /* 0.01                                     ref :453 */
    double ang_damp;        /* 0.02                                     ref :454 */
    double gravity;         /* 9.81, applied twice like the reference   ref :338,:524-527 */
    double dt_sub;          /* 0.02/4                                   ref :339-345 */
    int32_t n_sub;          /* 4 */
    int32_t max_episode_steps; /* 1000, <= 65535                        ref :282 */
    int32_t distinct_window;/* reward "diversity" window: 1000 = reference-exact, 10 = fast (default) */
    int32_t contact;        /* 1 = build-defined ground contact model, 0 = free flight only */
    int32_t auto_reset;     /* 1 = same-step auto-reset (vector env), 0 = caller resets (N=1 wrapper) */
    double mu;              /* 0.8*0.3 combined friction                ref :350,:456 */
    double erp;             /* 0.2 */
    double cop_s0;          /* 0.02 centre-of-pressure blend width */
    double init_pos[3];     /* (0,0,1)                                  ref :438 */
    double init_quat[4];    /* (0,0,0,1) x,y,z,w                        ref :439 */
    /* build-defined domain randomisation, sampled per env at every reset from Philox4x32-10
       keyed by (seed, global env id, episode); all zero = reference behaviour.
       Ranges follow config/config.yaml:340-349. */
    int32_t dr_enabled;
    int32_t _pad0;
    double dr_mass_var;     /* mass *= 1 + U(-v, v) */
    double dr_thrust_std;   /* thrust *= 1 + N(0, s^2), clamped to [0.5, 1.5] */
    double dr_cg_max;       /* cg offset along body z: U(-m, m) metres */
    double dr_wind_std;     /* constant world-frame wind force, x and y ~ N(0, s^2) newtons */
    double dr_init_tilt_max;/* initial tilt about x and y: U(-t, t) rad (curriculum) */
    double dr_obs_noise_std;/* N(0, s^2) added to obs[0:7] each step */
    uint64_t seed;
    int64_t env_id_offset;  /* global id of local env 0 (rank * n_envs): results independent of sharding */
} tvc_env_cfg;

void tvc_env_default_cfg(tvc_env_cfg* cfg);

/* EnhancedRocketTVCEnv.__init__ (env/...:279-322) for n_envs independent env objects on GPU
 * `device`.  Persistent per-env state (success window, reward history, previous action) starts
 * empty, dynamic state = reset(). */
int tvc_env_create(const tvc_env_cfg* cfg, int32_t n_envs, int32_t device, tvc_env** out);
void tvc_env_destroy(tvc_env* env);
int32_t tvc_env_num_envs(const tvc_env* env);

/* Updates the DR ranges of a live handle (curriculum stage change, scripts/curriculum_manager.py:248-290 -> the env); takes
 * effect at each env's next reset.  The step kernel reads the ranges from a device-resident record, so the change also reaches
 * steps replayed from a captured hipGraph.  tvc_env_set_dr synchronises the device around the change (steps enqueued before the
 * call on any stream keep the old ranges); tvc_env_set_dr_async enqueues it on `stream` without synchronising: steps enqueued on
 * that stream before the call keep the old ranges, later ones (graph launches included) see the new ones.  dr_enabled itself
 * selects the kernel instantiation at launch: do not flip it under a captured graph. */
int tvc_env_set_dr(tvc_env* env, const tvc_env_cfg* cfg);
int tvc_env_set_dr_async(tvc_env* env, const tvc_env_cfg* cfg, void* stream);

/* EnhancedRocketTVCEnv.reset (env/...:381-407).  mask_dev: uint8[N] or NULL (= all).
 * hard != 0 also clears the state the reference keeps across resets (== constructing a new
 * env object).  obs_dev: float[N,10] or NULL; rows of unmasked envs are left untouched. */
int tvc_env_reset(tvc_env* env, const uint8_t* mask_dev, int32_t hard, float* obs_dev, void* stream);

/* EnhancedRocketTVCEnv.step (env/...:466-518) for all N envs.
 *   act_dev   float[N,2] row-major, any range (clipped to [-1,1] like ref :470)
 *   obs_dev   float[N,10] row-major  (quat xyzw, omega xyz, fuel, phase/7, progress) ref :587-606
 *   rew_dev   float[N]
 *   term_dev, trunc_dev  uint8[N]
 *   final_obs_dev float[N,10] or NULL: with auto_reset, the terminal observation of envs that
 *             finished this step (rows of other envs = obs).  obs_dev then holds the first
 *             observation of the next episode for those envs.
 */
int tvc_env_step(tvc_env* env, const float* act_dev, float* obs_dev, float* rew_dev, uint8_t* term_dev,
                 uint8_t* trunc_dev, float* final_obs_dev, void* stream);

/* T consecutive steps in ONE launch with pre-supplied actions (the procedure of the reference's
 * tests/benchmark.py:40-60: pre-sampled actions, auto-reset on done).  State stays in registers;
 * every step's outputs are still written.
 *   act_dev float[T,N,2]; obs_dev float[T,N,10]; rew_dev float[T,N]; term/trunc uint8[T,N]
 */
int tvc_env_step_many(tvc_env* env, int32_t n_steps, const float* act_dev, float* obs_dev, float* rew_dev,
                      uint8_t* term_dev, uint8_t* trunc_dev, void* stream);

/* State exchange for parity tests and checkpoints (row-major, device pointers).
 *   dyn_dev  float[N,13]  pos3, quat4 (xyzw), vel3, omega3 (world)
 *   aux_dev  int32[N,8]   step, phase, mission_successful, success_run, hist_len, has_prev_action,
 *                         distinct, episode
 *   pa_dev   float[N,2]   previous (clipped) action
 *   par_dev  float[N,8]   mass_scale, thrust_scale, cg_offset, wind_x, wind_y, wind_z, 0, 0
 *   hist_dev float[N,W]   reward window, oldest first (W = distinct_window), unused tail = 0
 * any pointer may be NULL.
 */
int tvc_env_export_state(tvc_env* env, float* dyn_dev, int32_t* aux_dev, float* pa_dev, float* par_dev,
                         float* hist_dev, void* stream);
int tvc_env_import_state(tvc_env* env, const float* dyn_dev, const int32_t* aux_dev, const float* pa_dev,
                         const float* par_dev, const float* hist_dev, void* stream);

/* info['reward_components'] of step() (env/...:491-493, 514; MultiObjectiveReward.compute_reward :97-112): when comps_dev is
 * non-NULL every following tvc_env_step also writes comps_dev float[N,12] =
 *   mission_completion, safety_compliance, fuel_efficiency, stability_bonus, control_smoothness, altitude_maintenance,
 *   crash_penalty, excessive_tilt, control_saturation (0 when the penalty did not fire), anti-hacking adjustment (:209-224),
 *   unclipped total, penalty presence mask (bit 0 crash, 1 tilt, 2 saturation: the reference's dict only holds fired penalties).
 * NULL (default) switches it off.  The pointer is caller-owned device memory that must outlive the steps. */
int tvc_env_set_components_out(tvc_env* env, float* comps_dev);

/* On-device episode statistics: what StateOfTheArtTrainer keeps per episode on the host (scripts/train.py:594-616: episode
 * reward, length, success) and feeds to the curriculum manager (:458-460), accumulated by the step kernel so that N envs need no
 * host round trip.  ep_return_dev float[N] (running extrinsic return of each env's current episode, zero it before use);
 * sums_dev double[TVC_EP_SLOTS * 16]: TVC_EP_SLOTS partial records, one 128-byte line each, record s at sums_dev[16 s + k] with
 * k = 0 episodes finished, 1 of which mission_successful, 2 sum of their returns, 3 sum of their lengths; the totals are the sums
 * over the records (workgroups add into different records so that the atomics do not serialise on one address; the caller
 * zeroes / reads them).  NULL, NULL switches it off (default). */
#define TVC_EP_SLOTS 256
int tvc_env_set_episode_stats(tvc_env* env, float* ep_return_dev, double* sums_dev);

/* info dict of _get_enhanced_info (env/...:723-742) as tensors:
 *   info_dev float[N,8]: x, y, altitude, tilt_angle_deg, angular_velocity_mag, fuel, phase idx,
 *                        success_criteria_met(last 10) */
int tvc_env_info(tvc_env* env, float* info_dev, void* stream);

/* Fuel bookkeeping table the kernel uses (fuel is a pure function of the step count in the
 * reference, env/...:530-533): number of decrements until empty / until < 0.8 / until <= 0.1. */
int tvc_env_fuel_thresholds(const tvc_env* env, int32_t* k_empty, int32_t* k_coast, int32_t* k_low);

/* Philox4x32-10 (Random123) of n (counter[4], key[2]) pairs on the device: in_dev uint32[6n], out_dev uint32[4n].
 * The generator behind domain randomisation, observation noise and replay sampling; exposed for known-answer tests. */
int tvc_debug_philox(const uint32_t* in_dev, uint32_t* out_dev, int32_t n, void* stream);

/* Placement probe: out_dev uint32[2 * n_blocks] = {XCD id, HW_ID register} of every workgroup of one launch on `stream`
 * (which compute units a CU-masked stream owns; used by the CU-partitioned train loop and its tests). */
int tvc_debug_hwid(uint32_t* out_dev, int32_t n_blocks, void* stream);

/* ------------------------------------------------------------------ SAC learner (K3-K7, K11) */

typedef struct tvc_sac tvc_sac;

/* Network families:
 *   0 = the shipped reference shapes (agent/multi_algorithm_agent.py:587-627): actor =
 *       TransformerPolicyNetwork at sequence length 1 (== embed, +PE, n_layers x {V-proj, O-proj, +res,
 *       LN, FFN GELU, +res, LN}, LN, policy head d->h1->h2->2A with GELU+LN); critics
 *       (obs+A)->c1->c2->1 with GELU+LN.  Dead compute of the reference (Q/K projections, value head;
 *       SURVEY F8) is not executed; those tensors stay with the caller for checkpoints.
 *   1 = the "256x256 MLP" of BASELINE.json / the legacy SACAgent: actor obs->m1->m2->2A ReLU,
 *       critics (obs+A)->c1->c2->1 ReLU. */
typedef struct tvc_sac_cfg {
    int32_t obs_dim, act_dim;       /* 10, 2 */
    int32_t family;
    int32_t d_model, n_layers, ff_dim, head1, head2; /* family 0: 256, 4, 512, 512, 512 */
    int32_t mlp1, mlp2;             /* family 1 actor hidden: 256, 256 */
    int32_t critic1, critic2;       /* 512, 256 (family 0) or 256, 256 (family 1) */
    int32_t batch_size;             /* update batch rows (256) */
    int32_t max_act_rows;           /* largest batch tvc_sac_act will see */
    int32_t pe_rows;                /* 1: PE(0) on every row (== reference at B=1); >1: the reference's
                                       batch-row-indexed table, row index taken modulo pe_rows */
    float gamma, alpha, tau, lr;    /* 0.99, 0.2, 0.005, 3e-4  (agent/...:971,998,1005,623-625) */
    float adam_b1, adam_b2, adam_eps; /* torch.optim.Adam defaults */
    int32_t use_se;                 /* 1: SqueezeExcitation(d_model, 16) after feature_norm (agent/...:104-118,151-152,214-215): the
                                       NetworkConfig() default of the hierarchical low-level policy; acting only */
    float dropout_p;                /* 0 (default): nets as in .eval(), what the goldens pin.  0.1 = the reference's train-mode
                                       update (it never calls .eval()): Dropout after the LayerNorms of the policy head and the
                                       critics, and the encoder layers' attention-weight / dropout1 / FFN / dropout2 sites, in
                                       every forward of tvc_sac_update (family 0).  Masks are a counter-based hash, not torch's
                                       Philox stream: statistically equivalent.  tvc_sac_act stays deterministic. */
    int32_t nhead;                  /* 8: attention heads (only the granularity of the attention-weight dropout at seq len 1) */
    uint32_t dropout_seed;          /* mixed into every dropout mask key: handles (policies, ranks, seeds) with different values draw
                                       different mask sequences; 0 = the sequence the golden tests pin */
} tvc_sac_cfg;

void tvc_sac_default_cfg(tvc_sac_cfg* cfg, int32_t family);

/* Parameter table (host-only queries, no GPU needed).  One flat fp32 buffer holds
 * [policy | q1 | q2 | target_q1 | target_q2]; gradients / Adam moments cover the trainable prefix
 * [policy | q1 | q2].  Tensor names follow the reference's state_dict keys where a 1:1 tensor exists
 * ("policy.input_embedding.weight", "q1.0.weight", ...); "policy.layers.<l>.v_proj.*" are rows
 * 2d..3d of the reference's self_attn.in_proj_*. */
int64_t tvc_sac_param_count(const tvc_sac_cfg* cfg);      /* floats in the full parameter buffer */
int64_t tvc_sac_trainable_count(const tvc_sac_cfg* cfg);  /* floats in grads / adam_m / adam_v */
int32_t tvc_sac_num_tensors(const tvc_sac_cfg* cfg);
int tvc_sac_tensor_info(const tvc_sac_cfg* cfg, int32_t idx, char* name, int32_t name_cap, int64_t* offset,
                        int32_t* rows, int32_t* cols);

/* _create_sac_agent (agent/...:587-627).  The four flat buffers are caller-owned device memory
 * (torch tensors): params[param_count], grads / adam_m / adam_v [trainable_count].  pe_table_host:
 * float[pe_rows, d_model] (family 0) or NULL. */
int tvc_sac_create(const tvc_sac_cfg* cfg, int32_t device, float* params_dev, float* grads_dev, float* adam_m_dev,
                   float* adam_v_dev, const float* pe_table_host, tvc_sac** out);
void tvc_sac_destroy(tvc_sac* sac);

/* Re-derives what the library caches from the parameter buffer (the folded attention weights W_o W_v of the
 * acting net).  tvc_sac_actor_apply does this itself; call it after writing parameters from the host side
 * (initialisation, checkpoint load, parameter broadcast). */
int tvc_sac_sync_derived(tvc_sac* sac, void* stream);

/* Copies the policy parameters (and the folded acting weights) into a library-owned snapshot; tvc_sac_act with flags
 * bit 1 reads that snapshot.  Lets the learner keep stepping the live parameters on another stream while the acting
 * pass of the same vector step runs (the policy that acts during step t is the one left by step t-1 either way). */
int tvc_sac_snapshot_policy(tvc_sac* sac, void* stream);

/* Adam step counters {critics, actor} (device-resident so that a captured update keeps counting); checkpoints only,
 * both calls synchronise. */
int tvc_sac_get_adam_steps(tvc_sac* sac, int32_t out[2]);
int tvc_sac_set_adam_steps(tvc_sac* sac, const int32_t in[2]);
/* Call counter of the train-mode acting passes (tvc_sac_act flags bit 3): it keys their dropout masks and advances by one per
 * call; carried in checkpoints so that a resumed run does not replay the masks from call 0.  0 / no-op for handles without
 * train-mode acting.  Both calls synchronise. */
int tvc_sac_get_act_counter(tvc_sac* sac, int32_t* out);
int tvc_sac_set_act_counter(tvc_sac* sac, int32_t value);

/* Packs the split-operand weight stream of the acting net (tvc_sac_act flags bit 4) from the current folded weights; from then on
 * every policy update re-packs it and every snapshot copies it.  tvc_sac_act does this itself on first use; call it explicitly
 * before a loop whose updates run on another stream.  Idempotent. */
int tvc_sac_enable_x3(tvc_sac* sac, void* stream);

/* Policy part of get_action (agent/...:765-789) for n rows: mean/log_std (clamped to [-20,2]) and
 * action = clamp(mean + exp(log_std) * eps, -1, 1); eps_dev NULL = deterministic (action = clamp(mean)).
 * flags bit 0: do not clamp (the safety layer sees the raw sample, agent/...:780-789); bit 1: act with the snapshot;
 * bit 2: share the CUs -- the one-launch acting kernel (n >= 12 288 rows, reference shapes) then occupies half of each CU's
 * registers instead of all of them, so that kernels on other streams (a SAC update) run beside it instead of after it;
 * bit 3: act in TRAIN mode like the reference's get_action (no .eval() anywhere, agent/...:765): Dropout active at every site of
 * the policy with fresh masks per call (needs family 0 and dropout_p > 0; per-layer kernels, attention not folded);
 * bit 4: split-operand arithmetic for the one-launch kernel (n >= 16 384 rows, reference shapes; ignored otherwise): every Linear on the
 * bf16 matrix pipe with both operands written as three bf16 terms and six products accumulated in fp32 -- the same fp32 result to
 * rounding (error against an fp64 sum equal to the f32-input MFMA's, tests/test_acting_x3_gpu.py) at 6 / 16 of its MFMA cycles.  With
 * bit 3: the train-mode instantiation of that kernel (the net as trained, every Dropout live, the same masks as the other train-mode paths).
 * obs_dev float[n,obs]; act_dev float[n,A]; mean_dev / logstd_dev float[n,A] or NULL. */
int tvc_sac_act(tvc_sac* sac, const float* obs_dev, int32_t n, const float* eps_dev, float* act_dev, float* mean_dev,
                float* logstd_dev, int32_t flags, void* stream);

/* Diagnostics of the one-launch acting kernel (used for n >= 12 288 rows with the reference shapes): the shader clock the chip
 * holds inside it (Delta s_memtime / Delta s_memrealtime x 100 MHz, median over workgroups of the last of `launches` back-to-back
 * launches) -- out[0] = MHz, out[1] = median workgroup lifetime in microseconds, out[2] = workgroups.  Synchronises. */
int tvc_debug_rows_clock(tvc_sac* sac, const float* obs_dev, int32_t n, int32_t launches, double* out, void* stream);
/* ... and the raw stamps of the last of those launches: 6 x uint64 per workgroup {s_memtime, s_memrealtime (100 MHz) at start, the
 * same at the end, XCC_ID, HW_ID} into a HOST buffer of ceil(n / 64) * 6 entries; flags bit 2 as in tvc_sac_act.  Synchronises. */
int tvc_debug_rows_stamps(tvc_sac* sac, const float* obs_dev, int32_t n, int32_t launches, int32_t flags, uint64_t* out_host,
                          void* stream);

/* One _update_sac (agent/...:950-1016) on a batch of batch_size rows, in four phases so that the caller can
 * all-reduce gradients between them (data parallel, K10):
 *   critic_grads: target y, q1/q2 forward + backward -> grads[q1|q2], losses[0..1]
 *   critic_apply: Adam on q1, q2 (gradients multiplied by grad_scale first, e.g. 1/world_size)
 *   actor_grads : policy forward (shared with critic_grads: one pass over [s ; s'] when both phases of an update
 *                 see the same s), q(s, a_new) through the UPDATED critics, backward -> grads[policy], losses[2]
 *   actor_apply : Adam on the policy, Polyak update of both targets
 * s, s2 float[B,obs]; a float[B,A]; r, d float[B] (d = done as 0/1 float; the reference's BoolTensor
 * `dones` raises inside its update); eps_next / eps_new float[B,A] standard-normal draws;
 * losses_dev float[4] = q1_loss, q2_loss, policy_loss, physics_loss (PhysicsInformedLoss, agent/...:236-285,
 * reported only). */
int tvc_sac_critic_grads(tvc_sac* sac, const float* s, const float* a, const float* r, const float* s2, const float* d,
                         const float* eps_next, float* losses_dev, void* stream);
int tvc_sac_critic_apply(tvc_sac* sac, float grad_scale, void* stream);
int tvc_sac_actor_grads(tvc_sac* sac, const float* s, const float* eps_new, float* losses_dev, void* stream);
int tvc_sac_actor_apply(tvc_sac* sac, float grad_scale, void* stream);
/* all four phases back to back (single GPU) */
int tvc_sac_update(tvc_sac* sac, const float* s, const float* a, const float* r, const float* s2, const float* d,
                   const float* eps_next, const float* eps_new, float* losses_dev, void* stream);

/* The fused nn.Linear kernel on its own: Y[M,N] = act(X[M,K] W[N,K]^T + b), act 0 none / 1 GELU / 2 ReLU
 * (numerics tests against torch.nn.functional.linear, kernel benchmarks).  variant 0 = automatic choice,
 * 1 = 64x64 LDS-tiled, 3 = skinny split-K. */
int tvc_nn_linear_forward(const float* X, const float* W, const float* b, float* Y, int32_t M, int32_t N, int32_t K,
                          int32_t act, int32_t variant, void* stream);

/* The fused acting-pass kernel on its own: Y[M,N] = LayerNorm(act(X W^T + b) + R) * gamma + beta (eps 1e-5), i.e. one
 * nn.Linear + activation + residual + nn.LayerNorm of the policy (agent/multi_algorithm_agent.py:137-170) in one launch,
 * 32 complete rows per workgroup.  N in {256, 512}, K a multiple of 16; b, R may be NULL. */
int tvc_nn_linear_ln_forward(const float* X, const float* W, const float* b, const float* R, const float* gamma,
                             const float* beta, float* Y, int32_t M, int32_t N, int32_t K, int32_t act, void* stream);

/* critic forward q1(s,a), q2(s,a) for n <= batch_size rows (tests / diagnostics): q_dev float[2,n] */
int tvc_sac_q_values(tvc_sac* sac, const float* s, const float* a, int32_t n, int32_t use_target, float* q_dev, void* stream);

/* ------------------------------------------------------------------ small MLPs of the acting path (K9, K12) */

typedef struct tvc_mlp tvc_mlp;
/* dims[0] -> dims[1] -> ... -> dims[n_layers], hidden activation act (1 GELU, 2 ReLU), last layer linear.
 * Parameters: caller-owned flat fp32 device buffer, layer l at tvc_mlp_tensor_offset (weight [out,in] row-major,
 * then bias), every tensor 16-byte aligned; tvc_mlp_param_count floats in total (host-only queries). */
int64_t tvc_mlp_param_count(const int32_t* dims, int32_t n_layers);
int tvc_mlp_tensor_offset(const int32_t* dims, int32_t n_layers, int32_t layer, int64_t* w_off, int64_t* b_off);
/* act may carry TVC_MLP_LAYERNORM: a LayerNorm (eps 1e-5) behind every hidden activation, i.e. Linear-act-LN-...-Linear, the
 * shape of HierarchicalAgent.high_level_policy (agent/multi_algorithm_agent.py:366-374).  tvc_mlp_layout is the
 * parameter table for that case: offsets of layer `layer` (weight, bias, LayerNorm gamma / beta or -1), returns the total
 * float count (layer = -1: count only), -1 on error. */
#define TVC_MLP_LAYERNORM 0x100
int64_t tvc_mlp_layout(const int32_t* dims, int32_t n_layers, int32_t act_flags, int32_t layer, int64_t* w_off, int64_t* b_off,
                       int64_t* ln_w_off, int64_t* ln_b_off);
int tvc_mlp_create(const int32_t* dims, int32_t n_layers, int32_t act, int32_t max_rows, int32_t device,
                   const float* params_dev, tvc_mlp** out);
void tvc_mlp_destroy(tvc_mlp* mlp);
/* out[n, dims_last] = MLP([x[:, :k1] | x2[:, :dims0-k1]]); x row stride x_ld, x2 row stride x2_ld (x2 may be NULL
 * when k1 == dims0). */
int tvc_mlp_forward(tvc_mlp* mlp, const float* x, int32_t x_ld, int32_t k1, const float* x2, int32_t x2_ld, int32_t n,
                    float* out, void* stream);
/* CuriosityModule.compute_intrinsic_reward (env/enhanced_rocket_tvc_env.py:257-269) for n envs with the forward
 * model in `mlp` (D+A -> ... -> D): rew[m] += 0.01 * mean((f([prev_obs[m,:D] | act[m]]) - obs[m,:D])^2) unless
 * skip[m] != 0 (first step of an episode, env/...:496).  prev_obs / obs have row stride obs_ld. */
int tvc_curiosity_add(tvc_mlp* mlp, const float* prev_obs, int32_t obs_ld, const float* act, int32_t act_dim,
                      const float* obs, const uint8_t* skip, float* rew, int32_t n, void* stream);
/* SafetyLayer.forward (agent/multi_algorithm_agent.py:304-351) followed by get_action's clamp (:789), with the
 * correction net in `mlp` (state_dim+A -> ... -> A): out = clamp(violates ? net([state | proposed]) : proposed). */
int tvc_safety_apply(tvc_mlp* mlp, const float* state, int32_t state_dim, const float* proposed, float* out, int32_t n,
                     float max_tilt, float max_angular_velocity, float max_control_effort, void* stream);

/* HierarchicalAgent.select_goal + the input of its low-level policy (agent/multi_algorithm_agent.py:396-413): softmax over
 * logits[n, n_goals], one categorical draw per row from the uniform u[n] in [0,1) (torch.multinomial's role), then
 * state_goal_out[n, state_dim + n_goals] = [state[:, :state_dim] | one_hot(goal)]; goal_idx_out int32[n] or NULL. */
int tvc_goal_sample(const float* logits, const float* u, const float* state, int32_t state_ld, int32_t state_dim,
                    int32_t n_goals, int32_t n, float* state_goal_out, int32_t* goal_idx_out, void* stream);

/* ------------------------------------------------------------------ replay buffer (K8) */

typedef struct tvc_replay tvc_replay;
/* Device-resident uniform replay (BASELINE.json: capacity 1M, batch 256; the shipped reference has none,
 * SURVEY a21; legacy surface: store_transition / len(replay_buffer), tests/test_agent.py:99-108).
 * Row = {s[obs], a[A], r, s2[obs], d} stored row-major ("array of rows": a sampled batch is 256 random
 * 96-byte rows, each one contiguous). */
/* The buffer starts zero-filled: sampling before the first insert returns all-zero transitions, not uninitialised memory. */
int tvc_replay_create(int64_t capacity, int32_t obs_dim, int32_t act_dim, int32_t device, tvc_replay** out);
void tvc_replay_destroy(tvc_replay* rb);
int64_t tvc_replay_size(const tvc_replay* rb);
/* append n transitions (ring overwrite). done = terminated|truncated as uint8 flags. */
int tvc_replay_insert(tvc_replay* rb, const float* s, const float* a, const float* r, const float* s2,
                      const uint8_t* term, const uint8_t* trunc, int32_t n, void* stream);
/* uniform sample of `batch` rows with Philox4x32-10 keyed by (seed, counter); counter == UINT64_MAX uses an
 * auto-incrementing device-resident counter (so a captured hipGraph draws fresh rows at every replay).
 * Head / size live on the device too; tvc_replay_size() synchronises.  Outputs are dense device arrays. */
int tvc_replay_sample(tvc_replay* rb, int32_t batch, uint64_t seed, uint64_t counter, float* s, float* a, float* r,
                      float* s2, float* d, void* stream);

/* Snapshot for a true resume (the reference's --resume is a stub, scripts/train.py:904-907): meta = {head, size, sample
 * counter}; rows_dev float[size, 2*obs+A+2] in storage order (may be NULL on export to query meta only).  Both calls
 * synchronise the device. */
int tvc_replay_export(tvc_replay* rb, float* rows_dev, int64_t meta[3]);
int tvc_replay_import(tvc_replay* rb, const float* rows_dev, const int64_t meta[3]);

#ifdef __cplusplus
}
#endif
#endif /* TVC_NATIVE_H */
