/*
 * tvc_oracle.c -- CPU fp64 restatement of one rocket-TVC env step.
 * TEST INFRASTRUCTURE ONLY (see tvc_oracle.h for the parity status header).
 *
 * Every function cites the reference lines it follows.  "ref:" paths are
 * relative to /root/reference.  ASSUMPTION(bullet) marks arithmetic that lives
 * in the un-vendored pybullet 3.2.x dependency and is restated from its
 * published algorithm (parity unpinned).
 */
#include "tvc_oracle.h"

#include <math.h>
#include <string.h>

#define GIMBAL_RAD 0.3141592653589793 /* np.radians(18.0), ref: env/enhanced_rocket_tvc_env.py:471 */
#define PI_D 3.141592653589793

/* ------------------------------------------------------------------ helpers */

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void mat_vec(const double m[9], const double v[3], double o[3]) {
    o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    o[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    o[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}
static void matT_vec(const double m[9], const double v[3], double o[3]) {
    o[0] = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
    o[1] = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
    o[2] = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
}
static double norm3(const double a[3]) { return sqrt(dot3(a, a)); }

/* ASSUMPTION(bullet) btVector3::safeNorm(): scaled norm, 0 for the zero vector. */
static double safe_norm3(const double a[3]) {
    double ax = fabs(a[0]), ay = fabs(a[1]), az = fabs(a[2]);
    double mx = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
    if (mx > 0.0) {
        double s[3] = {a[0] / mx, a[1] / mx, a[2] / mx};
        return mx * norm3(s);
    }
    return 0.0;
}

/* ASSUMPTION(bullet) pybullet.getMatrixFromQuaternion == btMatrix3x3::setRotation, row-major.
   Used at ref: env/enhanced_rocket_tvc_env.py:546. */
void tvc_oracle_quat_to_matrix(const double q[4], double m[9]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double d = x * x + y * y + z * z + w * w;
    double s = 2.0 / d;
    double xs = x * s, ys = y * s, zs = z * s;
    double wx = w * xs, wy = w * ys, wz = w * zs;
    double xx = x * xs, xy = x * ys, xz = x * zs;
    double yy = y * ys, yz = y * zs, zz = z * zs;
    m[0] = 1.0 - (yy + zz); m[1] = xy - wz;         m[2] = xz + wy;
    m[3] = xy + wz;         m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
    m[6] = xz - wy;         m[7] = yz + wx;         m[8] = 1.0 - (xx + yy);
}

/* ASSUMPTION(bullet) pybullet.getEulerFromQuaternion (pybullet.c), used at
   ref: env/enhanced_rocket_tvc_env.py:614,728.  Returns roll,pitch,yaw. */
void tvc_oracle_quat_to_euler(const double q[4], double rpy[3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
    double sarg = -2.0 * (x * z - w * y);
    if (sarg <= -0.99999) {
        rpy[0] = 0.0;
        rpy[1] = -0.5 * PI_D;
        rpy[2] = 2.0 * atan2(x, -y);
    } else if (sarg >= 0.99999) {
        rpy[0] = 0.0;
        rpy[1] = 0.5 * PI_D;
        rpy[2] = 2.0 * atan2(-x, y);
    } else {
        rpy[0] = atan2(2.0 * (y * z + w * x), squ - sqx - sqy + sqz);
        rpy[1] = asin(sarg);
        rpy[2] = atan2(2.0 * (x * y + w * z), squ + sqx - sqy - sqz);
    }
}

/* ref: env/enhanced_rocket_tvc_env.py:530-533: fuel = max(0, fuel - 0.001), sequential in fp64. */
double tvc_oracle_fuel_after(int k) {
    double f = 1.0;
    for (int i = 0; i < k; ++i) {
        if (!(f > 0.0)) break;
        f = f - 0.001;
        if (f < 0.0) f = 0.0;
    }
    return f;
}

/* ------------------------------------------------------------------ params / reset */

/* ref: env/enhanced_rocket_tvc_env.py:409-464 (_create_enhanced_rocket), :324-352 (_setup_physics) */
void tvc_oracle_default_params(tvc_oracle_params* p) {
    memset(p, 0, sizeof(*p));
    double mass = 2.0, length = 1.0, radius = 0.05;
    p->mass = mass;
    /* ref :431-432, python evaluation order: ((1/12)*mass) * (3*radius**2 + length**2) */
    p->inertia[0] = p->inertia[1] = ((1.0 / 12.0) * mass) * (3.0 * (radius * radius) + length * length);
    p->inertia[2] = ((1.0 / 2.0) * mass) * (radius * radius);
    p->thrust = 35.0;
    p->half_len = 0.5;
    p->radius = radius;
    p->lin_damp = 0.01;
    p->ang_damp = 0.02;
    p->gravity = 9.81;
    p->dt_sub = 0.02 / 4.0; /* ASSUMPTION(bullet): stepSimulation(fixedTimeStep, numSubSteps, fixedTimeStep/numSubSteps) */
    p->n_sub = 4;
    p->max_episode_steps = 1000;
    p->distinct_window = TVC_ORACLE_HIST_MAX;
    p->contact = 0;
    p->auto_reset = 0;
    p->cg_offset = 0.0;
    p->mu = 0.8 * 0.3;
    p->erp = 0.2;
    p->cop_s0 = 0.02;
    p->init_pos[2] = 1.0;
    p->init_quat[3] = 1.0;
}

/* build-defined DR: scaling the mass scales the (uniform-density) inertia with it */
void tvc_oracle_scale_mass(tvc_oracle_params* p, double scale) {
    p->mass *= scale;
    p->inertia[0] *= scale;
    p->inertia[1] *= scale;
    p->inertia[2] *= scale;
}

/* ref: env/enhanced_rocket_tvc_env.py:381-407 reset(): phase, success flag, step, body, fuel,
   state_history are reset; criteria_history / reward_history / previous_action are NOT (:61,82,178). */
void tvc_oracle_reset(tvc_oracle_env* e, const tvc_oracle_params* p) {
    for (int i = 0; i < 3; ++i) {
        e->pos[i] = p->init_pos[i];
        e->vel[i] = 0.0;
        e->omega[i] = 0.0;
    }
    for (int i = 0; i < 4; ++i) e->quat[i] = p->init_quat[i];
    e->fuel = 1.0;
    e->step = 0;
    e->phase = TVC_PHASE_BOOST;
    e->mission_successful = 0;
    e->has_prev_obs = 0;
}

void tvc_oracle_init(tvc_oracle_env* e, const tvc_oracle_params* p) {
    memset(e, 0, sizeof(*e));
    tvc_oracle_reset(e, p);
}

/* ref: env/enhanced_rocket_tvc_env.py:587-606 _get_enhanced_observation */
void tvc_oracle_observe(const tvc_oracle_env* e, const tvc_oracle_params* p, float obs[10]) {
    double phase_value = (double)e->phase / 7.0; /* list(MissionPhase).index / len(MissionPhase) */
    double progress = (double)e->step / (double)p->max_episode_steps;
    if (progress > 1.0) progress = 1.0;
    obs[0] = (float)e->quat[0];
    obs[1] = (float)e->quat[1];
    obs[2] = (float)e->quat[2];
    obs[3] = (float)e->quat[3];
    obs[4] = (float)e->omega[0];
    obs[5] = (float)e->omega[1];
    obs[6] = (float)e->omega[2];
    obs[7] = (float)e->fuel;
    obs[8] = (float)phase_value;
    obs[9] = (float)progress;
}

/* ------------------------------------------------------------------ physics half */

/* Build-defined ground contact vs the z=0 plane (NOT Bullet's manifold/solver).
 * Stateless velocity-level impulses, once per substep, one candidate point per end cap:
 * the cap centre shifted towards the lowest rim point by r*min(1, sin(tilt)/s0) (a continuous
 * centre-of-pressure), normal impulse with e=0 and Baumgarte bias erp*depth/h, then a
 * Coulomb-clamped friction impulse with mu = 0.8*0.3 (ref: env/...:350,456 combined by product).
 */
static void apply_impulse(tvc_oracle_env* e, const tvc_oracle_params* p, const double R[9], const double rel[3],
                          const double dir[3], double j) {
    double rxn[3], loc[3], wl[3], dw[3];
    for (int i = 0; i < 3; ++i) e->vel[i] += j * dir[i] / p->mass;
    cross3(rel, dir, rxn);
    matT_vec(R, rxn, loc);
    for (int i = 0; i < 3; ++i) wl[i] = loc[i] / p->inertia[i];
    mat_vec(R, wl, dw);
    for (int i = 0; i < 3; ++i) e->omega[i] += j * dw[i];
}
static double eff_mass_inv(const tvc_oracle_params* p, const double R[9], const double rel[3], const double dir[3]) {
    double rxn[3], loc[3], wl[3], dw[3], c[3];
    cross3(rel, dir, rxn);
    matT_vec(R, rxn, loc);
    for (int i = 0; i < 3; ++i) wl[i] = loc[i] / p->inertia[i];
    mat_vec(R, wl, dw);
    cross3(dw, rel, c);
    return 1.0 / p->mass + dot3(dir, c);
}
static void ground_contact(tvc_oracle_env* e, const tvc_oracle_params* p, double h) {
    double R[9];
    tvc_oracle_quat_to_matrix(e->quat, R);
    double a[3] = {R[2], R[5], R[8]}; /* body z axis in world */
    double reach = fabs(a[2]) * (p->half_len + fabs(p->cg_offset)) + p->radius;
    if (e->pos[2] - reach > 0.0) return;
    /* d = -(ez - (ez.a) a): "down" direction inside the cap plane, |d| = sin(tilt of axis) */
    double d[3] = {a[2] * a[0], a[2] * a[1], a[2] * a[2] - 1.0};
    double dn = norm3(d);
    double sc = dn > p->cop_s0 ? p->radius / dn : p->radius / p->cop_s0;
    const double n[3] = {0.0, 0.0, 1.0};
    for (int end = 0; end < 2; ++end) {
        double L = end == 0 ? -(p->half_len + p->cg_offset) : (p->half_len - p->cg_offset);
        double rel[3];
        for (int i = 0; i < 3; ++i) rel[i] = L * a[i] + sc * d[i];
        double pz = e->pos[2] + rel[2];
        if (pz >= 0.0) continue;
        double depth = -pz;
        double wxr[3], vp[3];
        cross3(e->omega, rel, wxr);
        for (int i = 0; i < 3; ++i) vp[i] = e->vel[i] + wxr[i];
        double vn = vp[2];
        double kn = eff_mass_inv(p, R, rel, n);
        double jn = (-vn + p->erp * depth / h) / kn;
        if (jn <= 0.0) continue;
        apply_impulse(e, p, R, rel, n, jn);
        cross3(e->omega, rel, wxr);
        for (int i = 0; i < 3; ++i) vp[i] = e->vel[i] + wxr[i];
        double vt[3] = {vp[0], vp[1], 0.0};
        double vtn = norm3(vt);
        if (vtn > 1e-9) {
            double t[3] = {vt[0] / vtn, vt[1] / vtn, 0.0};
            double kt = eff_mass_inv(p, R, rel, t);
            double jt = vtn / kt;
            double jmax = p->mu * jn;
            if (jt > jmax) jt = jmax;
            apply_impulse(e, p, R, rel, t, -jt);
        }
    }
}

/* One Bullet internal substep of a 0-link floating-base btMultiBody.
 * ASSUMPTION(bullet): btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof with no
 * links (damping form k+k|v|, gyroscopic term on), applyDeltaVeeMultiDof with the +-100
 * coordinate-velocity clamp, then stepPositionsMultiDof (exponential map, |w|h clamp pi/4,
 * Taylor branch below 1e-3, renormalise).  F and tau are the world-frame wrench held constant
 * over the control step (external forces + world gravity).
 */
static void substep(tvc_oracle_env* e, const tvc_oracle_params* p, const double F[3], const double tau[3], double h) {
    double R[9];
    tvc_oracle_quat_to_matrix(e->quat, R);
    double wb[3], vb[3], tb[3], fb[3];
    matT_vec(R, e->omega, wb);
    matT_vec(R, e->vel, vb);
    matT_vec(R, tau, tb);
    matT_vec(R, F, fb);
    double zaA[3], zaL[3];
    for (int i = 0; i < 3; ++i) {
        zaA[i] = -tb[i];
        zaL[i] = -fb[i];
    }
    double nw = safe_norm3(wb), nv = safe_norm3(vb);
    double Iw[3];
    for (int i = 0; i < 3; ++i) Iw[i] = p->inertia[i] * wb[i];
    for (int i = 0; i < 3; ++i) {
        zaA[i] += Iw[i] * (p->ang_damp + p->ang_damp * nw);
        zaL[i] += p->mass * vb[i] * (p->lin_damp + p->lin_damp * nv);
    }
    double gy[3], wxv[3];
    cross3(wb, Iw, gy); /* m_useGyroTerm */
    cross3(wb, vb, wxv);
    for (int i = 0; i < 3; ++i) {
        zaA[i] += gy[i];
        zaL[i] += p->mass * wxv[i];
    }
    double accA[3], accL[3], tmp[3], wdot[3], vdot[3];
    for (int i = 0; i < 3; ++i) {
        accA[i] = -zaA[i] / p->inertia[i];
        accL[i] = -zaL[i] / p->mass;
    }
    mat_vec(R, accA, wdot);
    for (int i = 0; i < 3; ++i) tmp[i] = accL[i] + wxv[i];
    mat_vec(R, tmp, vdot);
    for (int i = 0; i < 3; ++i) {
        e->omega[i] += wdot[i] * h;
        if (e->omega[i] > 100.0) e->omega[i] = 100.0;
        if (e->omega[i] < -100.0) e->omega[i] = -100.0;
        e->vel[i] += vdot[i] * h;
        if (e->vel[i] > 100.0) e->vel[i] = 100.0;
        if (e->vel[i] < -100.0) e->vel[i] = -100.0;
    }

    if (p->contact) ground_contact(e, p, h);

    for (int i = 0; i < 3; ++i) e->pos[i] += h * e->vel[i];

    /* exponential-map orientation update, q' = normalize(dq (x) q) */
    double fAngle = norm3(e->omega);
    if (fAngle * h > 0.25 * PI_D) fAngle = 0.5 * (0.5 * PI_D) / h;
    double ax[3];
    if (fAngle < 0.001) {
        double k = 0.5 * h - (h * h * h) * 0.020833333333 * fAngle * fAngle;
        for (int i = 0; i < 3; ++i) ax[i] = e->omega[i] * k;
    } else {
        double k = sin(0.5 * fAngle * h) / fAngle;
        for (int i = 0; i < 3; ++i) ax[i] = e->omega[i] * k;
    }
    double dw = cos(fAngle * h * 0.5);
    double x1 = ax[0], y1 = ax[1], z1 = ax[2], w1 = dw;
    double x2 = e->quat[0], y2 = e->quat[1], z2 = e->quat[2], w2 = e->quat[3];
    double qn[4];
    qn[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    qn[1] = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2;
    qn[2] = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2;
    qn[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    double l = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; ++i) e->quat[i] = qn[i] / l;
}

/* ref: env/enhanced_rocket_tvc_env.py:520-585 (_apply_enhanced_control, _apply_aerodynamics): the EXTERNAL wrench
   the reference hands to pybullet through applyExternalForce / applyExternalTorque, summed about the COM
   (world gravity of p.setGravity is NOT part of it).  Does not touch the env (fuel is decremented by the caller).
   PINNED by tests/golden/step_ref_*.npz: the reference's own step() driven over a recording pybullet stand-in
   (tests/golden/gen_step_golden.py).  action is already clipped to [-1,1]. */
void tvc_oracle_wrench(const tvc_oracle_env* e, const tvc_oracle_params* p, const double action[2], double F[3],
                       double tau[3]) {
    double R[9];
    tvc_oracle_quat_to_matrix(e->quat, R);
    F[0] = 0.0; F[1] = 0.0; F[2] = -p->gravity * p->mass; /* explicit gravity force -9.81*mass at COM, ref :524-527 */
    tau[0] = tau[1] = tau[2] = 0.0;

    if (e->fuel > 0.0) { /* ref :530 (tested BEFORE the decrement of :533) */
        double pitch = action[0] * GIMBAL_RAD; /* pitch_angle, yaw_angle = self.gimbal_angles ref :537 */
        double yaw = action[1] * GIMBAL_RAD;
        double Tl[3] = {p->thrust * sin(yaw), p->thrust * sin(pitch), p->thrust * cos(pitch) * cos(yaw)}; /* ref :539-543 */
        double Tw[3], rl[3] = {0.0, 0.0, -(p->half_len + p->cg_offset)}, rel[3], tq[3];
        mat_vec(R, Tl, Tw);   /* ref :547 */
        mat_vec(R, rl, rel);  /* thrust_position - pos, ref :550 */
        cross3(rel, Tw, tq);  /* ASSUMPTION(bullet) applyExternalForce(WORLD_FRAME): torque = (p - com) x F */
        for (int i = 0; i < 3; ++i) {
            F[i] += Tw[i];
            tau[i] += tq[i];
        }
    }
    /* ref :561-585 aerodynamics, on the pre-step pose/velocity */
    double rho = 1.225 * exp(-e->pos[2] / 8400.0);
    double vmag = norm3(e->vel);
    if (vmag > 0.1) {
        double frontal_area = PI_D * (0.05 * 0.05);
        double drag_mag = 0.5 * rho * (vmag * vmag) * 0.47 * frontal_area;
        for (int i = 0; i < 3; ++i) F[i] += drag_mag * (-e->vel[i] / vmag);
    }
    double adamp = 0.02 * rho;
    for (int i = 0; i < 3; ++i) tau[i] += -adamp * e->omega[i];
    /* build-defined DR: constant wind force */
    for (int i = 0; i < 3; ++i) F[i] += p->wind[i];
}

/* wrench (above), fuel bookkeeping ref :530-533, then :477 p.stepSimulation (n_sub substeps). */
void tvc_oracle_physics(tvc_oracle_env* e, const tvc_oracle_params* p, const double action[2]) {
    double F[3], tau[3];
    tvc_oracle_wrench(e, p, action, F, tau);
    if (e->fuel > 0.0) {
        double f = e->fuel - 0.001;
        e->fuel = f > 0.0 ? f : 0.0; /* max(0, fuel - 0.001) ref :533 */
    }
    /* world gravity p.setGravity(0,0,-9.81) ref :338 -- ASSUMPTION(bullet): added as m*g base force */
    F[2] += -p->gravity * p->mass;

    for (int s = 0; s < p->n_sub; ++s) substep(e, p, F, tau, p->dt_sub);
}

/* ref: env/enhanced_rocket_tvc_env.py:608-633 _get_state_dict */
void tvc_oracle_scalars_from_state(const tvc_oracle_env* e, tvc_oracle_scalars* sc) {
    double rpy[3];
    tvc_oracle_quat_to_euler(e->quat, rpy);
    sc->tilt = sqrt(rpy[1] * rpy[1] + rpy[2] * rpy[2]); /* sqrt(pitch**2 + yaw**2) ref :616 */
    sc->altitude = e->pos[2];
    sc->omega_mag = norm3(e->omega);
    sc->v_h = sqrt(e->vel[0] * e->vel[0] + e->vel[1] * e->vel[1]);
    sc->v_z_abs = fabs(e->vel[2]);
    sc->x = e->pos[0];
    sc->y = e->pos[1];
    sc->crashed = e->pos[2] < 0.1;
}

/* ------------------------------------------------------------------ logic half */

/* numpy add.reduce over 10 contiguous doubles: pairwise-sum block form (8 partial sums,
   tree-combined, then the 2-element tail), as np.var uses it (ref: env/...:216). */
static double np_sum10(const double* a) {
    double res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    res += a[8];
    res += a[9];
    return res;
}
static double np_var10(const double* a) {
    double mean = np_sum10(a) / 10.0;
    double d[10];
    for (int i = 0; i < 10; ++i) {
        double x = a[i] - mean;
        d[i] = x * x;
    }
    return np_sum10(d) / 10.0;
}

/* len(set(last w entries)); w = whole history in the reference (env/...:221) */
static int count_distinct(const tvc_oracle_env* e, int w) {
    int n = 0;
    int start = e->hist_len - w;
    for (int i = start; i < e->hist_len; ++i) {
        double v = e->hist[(e->hist_head + i) % TVC_ORACLE_HIST_MAX];
        int seen = 0;
        for (int j = start; j < i; ++j)
            if (e->hist[(e->hist_head + j) % TVC_ORACLE_HIST_MAX] == v) {
                seen = 1;
                break;
            }
        if (!seen) ++n;
    }
    return n;
}

void tvc_oracle_logic(tvc_oracle_env* e, const tvc_oracle_params* p, const tvc_oracle_scalars* sc,
                      const double action[2], tvc_oracle_out* out) {
    /* state_dict is built BEFORE the phase/success update (ref :481 vs :485,:488): the reward sees
       the previous step's phase and success flag. */
    int phase_old = e->phase;
    int success_old = e->mission_successful;
    double alt = sc->altitude, tilt = sc->tilt, wmag = sc->omega_mag;
    double fuel = e->fuel;

    tvc_oracle_observe(e, p, out->obs); /* ref :482, before :485 */
    out->sc = *sc;

    /* ref :635-657 _update_mission_phase */
    if (e->phase == TVC_PHASE_BOOST && fuel < 0.8) {
        e->phase = TVC_PHASE_COAST;
    } else if (e->phase == TVC_PHASE_COAST && alt < 5.0) {
        e->phase = TVC_PHASE_LANDING;
    } else if (e->phase == TVC_PHASE_LANDING && alt < 1.0) {
        e->phase = TVC_PHASE_TOUCHDOWN;
    } else if (e->phase == TVC_PHASE_TOUCHDOWN && alt < 0.5) {
        if (tilt < 0.087 && wmag < 0.1) {
            e->phase = TVC_PHASE_COMPLETE;
            e->mission_successful = 1;
        }
    }

    /* ref :659-695 _check_mission_success; thresholds ref :43-50 */
    if (!e->mission_successful) {
        int pass = (tilt < 0.087) && (sc->v_z_abs < 2.0 && sc->v_h < 0.5) && (0.2 <= alt && alt <= 2.0) && (wmag < 0.1);
        e->success_run = pass ? (e->success_run < 100 ? e->success_run + 1 : 100) : 0;
        if (e->success_run >= 100) e->mission_successful = 1;
    }

    /* ref :86-125 compute_reward and helpers :127-224 */
    double mc = success_old ? 1.0 : (phase_old == TVC_PHASE_LANDING ? 0.1 : 0.0);
    double tilt_pen = exp(-10.0 * fmax(0.0, tilt - 0.087));
    double ang_pen = exp(-5.0 * fmax(0.0, wmag - 0.1));
    double alt_pen = (0.2 <= alt && alt <= 20.0) ? 1.0 : 0.5;
    double safe = (tilt_pen + ang_pen + alt_pen) / 3.0;
    double ce = sqrt(action[0] * action[0] + action[1] * action[1]); /* np.linalg.norm(action) */
    double fe = (fuel > 0.1 && ce < 0.5) ? fuel * (1.0 - ce) : 0.0;
    double stab = (tilt < 0.05 && wmag < 0.1) ? 1.0 : ((tilt < 0.1 && wmag < 0.2) ? 0.5 : 0.0);
    double smooth;
    if (e->has_prev_action) {
        double d0 = action[0] - e->prev_action[0], d1 = action[1] - e->prev_action[1];
        smooth = exp(-5.0 * sqrt(d0 * d0 + d1 * d1));
    } else {
        smooth = 1.0;
    }
    e->prev_action[0] = action[0];
    e->prev_action[1] = action[1];
    e->has_prev_action = 1;
    double altm = exp(-2.0 * fabs(alt - 3.0));

    double c_mc = mc * 100.0, c_safe = safe * 50.0, c_fe = fe * 20.0;
    double c_stab = stab * 10.0, c_smooth = smooth * 5.0, c_alt = altm * 5.0;
    double total = 0.0; /* python sum(): left to right from 0 */
    total += c_mc;
    total += c_safe;
    total += c_fe;
    total += c_stab;
    total += c_smooth;
    total += c_alt;
    double crash_pen = 0.0, tp = 0.0, sp = 0.0;
    if (sc->crashed) {
        crash_pen = -1000.0;
        total += crash_pen;
    }
    if (tilt > 0.52) {
        tp = -500.0 * (tilt - 0.52);
        total += tp;
    }
    if (ce > 0.9) {
        sp = -50.0 * (ce - 0.9);
        total += sp;
    }
    /* ref :209-224 anti-hacking; weights default 0.1 / 0.05 (config lookup at :83-84 never finds the
       nested YAML keys) */
    double adj = 0.0;
    if (e->hist_len > 10) {
        double last10[10];
        for (int i = 0; i < 10; ++i)
            last10[i] = e->hist[(e->hist_head + e->hist_len - 10 + i) % TVC_ORACLE_HIST_MAX];
        double var = np_var10(last10);
        if (var > 10000.0) adj -= 0.1 * var;
    }
    {
        int w = e->hist_len < p->distinct_window ? e->hist_len : p->distinct_window;
        int distinct = count_distinct(e, w);
        if ((double)distinct > (double)w * 0.8) adj += 0.05;
    }
    total = total + adj;
    out->components[10] = total;
    if (total < -1000.0) total = -1000.0; /* np.clip ref :121 */
    if (total > 200.0) total = 200.0;
    /* append, deque(maxlen=1000) */
    {
        int cap = TVC_ORACLE_HIST_MAX;
        if (e->hist_len >= cap) {
            e->hist_head = (e->hist_head + 1) % TVC_ORACLE_HIST_MAX;
            e->hist_len = cap - 1;
        }
        e->hist[(e->hist_head + e->hist_len) % TVC_ORACLE_HIST_MAX] = total;
        e->hist_len += 1;
    }
    out->reward = total;
    out->components[0] = c_mc;
    out->components[1] = c_safe;
    out->components[2] = c_fe;
    out->components[3] = c_stab;
    out->components[4] = c_smooth;
    out->components[5] = c_alt;
    out->components[6] = crash_pen;
    out->components[7] = tp;
    out->components[8] = sp;
    out->components[9] = adj;
    out->components[11] = 0.0;

    /* bookkeeping of state_history (ref :505), used only by the curiosity term */
    for (int i = 0; i < 8; ++i) e->prev_obs8[i] = (double)out->obs[i];
    e->has_prev_obs = 1;

    /* ref :697-721 _check_termination */
    int terminated = 0, truncated = 0;
    if (e->mission_successful) {
        terminated = 1;
    } else {
        if (sc->crashed) terminated = 1;
        else if (tilt > 0.52) terminated = 1;
        else if (alt > 20.0) terminated = 1;
        else if (sqrt(sc->x * sc->x + sc->y * sc->y) > 50.0) terminated = 1;
        if (e->step >= p->max_episode_steps) truncated = 1;
    }
    out->terminated = terminated;
    out->truncated = truncated;
}

/* ref: env/enhanced_rocket_tvc_env.py:466-518 step() */
void tvc_oracle_step(tvc_oracle_env* e, const tvc_oracle_params* p, const double action_in[2], tvc_oracle_out* out) {
    double a[2];
    for (int i = 0; i < 2; ++i) { /* np.clip(action, -1, 1) ref :470 */
        a[i] = action_in[i];
        if (a[i] < -1.0) a[i] = -1.0;
        if (a[i] > 1.0) a[i] = 1.0;
    }
    tvc_oracle_physics(e, p, a);
    e->step += 1; /* ref :478 */
    tvc_oracle_scalars sc;
    tvc_oracle_scalars_from_state(e, &sc);
    tvc_oracle_logic(e, p, &sc, a, out);
    if (p->auto_reset && (out->terminated || out->truncated)) {
        /* build-defined vector-env semantics: same-step auto-reset, returned obs = first obs of the
           next episode (reward/flags belong to the finished one) */
        tvc_oracle_reset(e, p);
        tvc_oracle_observe(e, p, out->obs);
        e->episodes += 1;
    }
}

long tvc_oracle_run(tvc_oracle_env* envs, const tvc_oracle_params* p, int n, int T, const float* actions,
                    double* reward_sum) {
    long steps = 0;
    double rs = 0.0;
    tvc_oracle_out out;
    for (int t = 0; t < T; ++t) {
        const float* at = actions + (long)t * n * 2;
        for (int i = 0; i < n; ++i) {
            double a[2] = {(double)at[2 * i], (double)at[2 * i + 1]};
            tvc_oracle_step(&envs[i], p, a, &out);
            rs += out.reward;
            ++steps;
        }
    }
    if (reward_sum) *reward_sum = rs;
    return steps;
}

int tvc_oracle_sizeof_env(void) { return (int)sizeof(tvc_oracle_env); }
int tvc_oracle_sizeof_params(void) { return (int)sizeof(tvc_oracle_params); }
