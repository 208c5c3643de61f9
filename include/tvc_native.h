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

#define TVC_ABI_VERSION 1

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

/* Updates the DR ranges of a live handle (curriculum stage change); takes effect at the next reset. */
int tvc_env_set_dr(tvc_env* env, const tvc_env_cfg* cfg);

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

/* info dict of _get_enhanced_info (env/...:723-742) as tensors:
 *   info_dev float[N,8]: x, y, altitude, tilt_angle_deg, angular_velocity_mag, fuel, phase idx,
 *                        success_criteria_met(last 10) */
int tvc_env_info(tvc_env* env, float* info_dev, void* stream);

/* Fuel bookkeeping table the kernel uses (fuel is a pure function of the step count in the
 * reference, env/...:530-533): number of decrements until empty / until < 0.8 / until <= 0.1. */
int tvc_env_fuel_thresholds(const tvc_env* env, int32_t* k_empty, int32_t* k_coast, int32_t* k_low);

#ifdef __cplusplus
}
#endif
#endif /* TVC_NATIVE_H */
