// Device-side arithmetic of one rocket-TVC env step (fp32, one lane per env).
//
// Restates, for gfx950, the op sequence of the reference's
//   EnhancedRocketTVCEnv.step            env/enhanced_rocket_tvc_env.py:466-518
//   _apply_enhanced_control / _apply_aerodynamics          :520-585
//   p.stepSimulation (4 x 5 ms Bullet floating-base substeps)  :477
//   _get_state_dict / _get_enhanced_observation            :587-633
//   _update_mission_phase / _check_mission_success / _check_termination  :635-721
//   MultiObjectiveReward.compute_reward                    :86-224
// The Bullet half is a restatement of btMultiBody's 0-link floating-base update (see DESIGN.md
// "Physics restatement"); it is algebraically simplified for Ixx == Iyy (the m*w x v term of the
// articulated-body form cancels against the frame-acceleration term).
#pragma once
#ifndef TVC_RING_BATCH
#define TVC_RING_BATCH 32u  // loads in flight per thread in the 1000-entry reward-history scan (16 / 32 / 64: 57 / 49 / 53 us at 65 536 envs, tools/env_ring_bench.py)
#endif
#include <hip/hip_runtime.h>

namespace tvcdev {

constexpr int kPhaseBoost = 0, kPhaseCoast = 1, kPhaseLanding = 2, kPhaseTouchdown = 3, kPhaseComplete = 5;
constexpr int kCellGroups = 9;

// hot constants (every env-step); kept small: kernel arguments live in SGPRs
struct DevCfg {
    float inv_mass, inv_ixx, inv_izz, gyro_r;  // gyro_r = (Izz - Ixx) / Ixx
    float thrust, lever, radius, half_len;
    float kl, ka, g2, h;
    float mu, erp_over_h, cop_s0, inv_max_steps;
    int nsub, max_steps;
    int k_empty, k_coast, k_low;  // fuel is a function of the step count (ref :530-533)
    int contact, auto_reset, W;
    float init[7];                // pos3, quat4 (velocities start at zero)
};
// cold constants: domain randomisation (build-defined), only read by the DR instantiations
struct DrCfg {
    float mass_var, thrust_std, cg_max, wind_std, tilt_max, noise_std;
    unsigned seed_lo, seed_hi;
    long long id_off;
};

// State in HBM: array-of-struct-of-arrays of 16-byte cells, so that every access of a wavefront is one
// contiguous 1 KiB segment (16 B per lane, the widest coalesced access):
//   group 0: px py pz  aux0      aux0: step[0:16) phase[16:19) msucc[19] has_pa[20] run[21:28)
//   group 1: qx qy qz qw
//   group 2: vx vy vz  aux1      aux1: hist_len[0:10) head[10:20) distinct[20:30)
//   group 3: wx wy wz  episode
//   group 4: pa0 pa1 mass_scale thrust_scale
//   group 5: cg windx windy windz            (read only by DR instantiations)
//   groups 6,7,8: reward window slots 0-3, 4-7, 8-9 (+2 spare)   (distinct_window == 10)
// ring1000: [1000][np] floats, only for distinct_window == 1000 (reference-exact mode).
struct EnvBuf {
    float4* cells;    // [kCellGroups][np]
    float* ring1000;  // [1000][np] or nullptr
    int n, np;
};

// Optional on-device episode statistics (curriculum driver / logging without host round trips):
// ret[N] = running extrinsic return of the current episode; sums = kEpSlots partial records {episodes, successes, return sum,
// length sum}, one 128-byte line per record (sums[16 * slot + k]); a workgroup adds into slot blockIdx.x % kEpSlots and the
// reader sums the slots.  (One shared record serialised: at 4 M envs ~160 000 same-address double atomics per launch were
// 3/4 of the kernel's time.)
constexpr int kEpSlots = 256;
struct EpStats {
    float* ret;
    double* sums;
};

struct Regs {
    float px, py, pz, qx, qy, qz, qw, vx, vy, vz, wx, wy, wz;
    float pa0, pa1;
    float ms, ts, cg, windx, windy, windz;
    unsigned step, phase, msucc, has_pa, run;
    unsigned hist_len, head, distinct;
    unsigned episode;
};

struct StepOut {
    float obs[10];
    float reward;
    unsigned term, trunc;
    // reward_components of compute_reward (ref :97-112) + anti-hacking adjustment + unclipped total; only stored when the
    // caller asked for them (tvc_env_set_components_out)
    float comps[12];
};

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ------------------------------------------------------------------ Philox4x32-10 (Random123)
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
    constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f; }
__device__ __forceinline__ float usym(unsigned x) { return 2.0f * u01(x) - 1.0f; }
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
    float r = __builtin_sqrtf(-2.0f * __logf(u01(a)));
    float th = 6.283185307179586f * u01(b);
    z0 = r * __cosf(th);
    z1 = r * __sinf(th);
}

// ------------------------------------------------------------------ small math
// sin(x)/x and cos(x) for |x| <= ~0.4 (substep half-angle <= pi/8, gimbal <= 0.3142): Taylor to
// x^8 is below fp32 epsilon there, and has no 0/0 at x -> 0 (covers Bullet's small-angle branch).
__device__ __forceinline__ float sinc_small(float x2) {
    return 1.0f + x2 * (-1.6666667e-1f + x2 * (8.3333333e-3f + x2 * (-1.9841270e-4f + x2 * 2.7557319e-6f)));
}
__device__ __forceinline__ float cos_small(float x2) {
    return 1.0f + x2 * (-0.5f + x2 * (4.1666667e-2f + x2 * (-1.3888889e-3f + x2 * 2.4801587e-5f)));
}

struct Mat3 {
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;
};
// btMatrix3x3::setRotation (pybullet.getMatrixFromQuaternion, ref :546)
__device__ __forceinline__ Mat3 rotmat(float x, float y, float z, float w) {
    float d = x * x + y * y + z * z + w * w;
    float s = 2.0f * rcp(d);
    float xs = x * s, ys = y * s, zs = z * s;
    float wx = w * xs, wy = w * ys, wz = w * zs;
    float xx = x * xs, xy = x * ys, xz = x * zs;
    float yy = y * ys, yz = y * zs, zz = z * zs;
    Mat3 R;
    R.m00 = 1.0f - (yy + zz); R.m01 = xy - wz;          R.m02 = xz + wy;
    R.m10 = xy + wz;          R.m11 = 1.0f - (xx + zz); R.m12 = yz - wx;
    R.m20 = xz - wy;          R.m21 = yz + wx;          R.m22 = 1.0f - (xx + yy);
    return R;
}
__device__ __forceinline__ float clamp100(float v) { return fminf(fmaxf(v, -100.0f), 100.0f); }

// Build-defined ground contact, same model as oracle/tvc_oracle.c ground_contact().
__device__ __forceinline__ void contact_impulse(Regs& r, const Mat3& R, float rx, float ry, float rz, float dx, float dy,
                                                float dz, float j, float inv_m, float inv_ixx, float inv_izz) {
    r.vx += j * dx * inv_m; r.vy += j * dy * inv_m; r.vz += j * dz * inv_m;
    float cx = ry * dz - rz * dy, cy = rz * dx - rx * dz, cz = rx * dy - ry * dx;
    float lx = (R.m00 * cx + R.m10 * cy + R.m20 * cz) * inv_ixx;
    float ly = (R.m01 * cx + R.m11 * cy + R.m21 * cz) * inv_ixx;
    float lz = (R.m02 * cx + R.m12 * cy + R.m22 * cz) * inv_izz;
    r.wx += j * (R.m00 * lx + R.m01 * ly + R.m02 * lz);
    r.wy += j * (R.m10 * lx + R.m11 * ly + R.m12 * lz);
    r.wz += j * (R.m20 * lx + R.m21 * ly + R.m22 * lz);
}
__device__ __forceinline__ float contact_kinv(const Mat3& R, float rx, float ry, float rz, float dx, float dy, float dz,
                                              float inv_m, float inv_ixx, float inv_izz) {
    float cx = ry * dz - rz * dy, cy = rz * dx - rx * dz, cz = rx * dy - ry * dx;
    float lx = (R.m00 * cx + R.m10 * cy + R.m20 * cz) * inv_ixx;
    float ly = (R.m01 * cx + R.m11 * cy + R.m21 * cz) * inv_ixx;
    float lz = (R.m02 * cx + R.m12 * cy + R.m22 * cz) * inv_izz;
    float wx = R.m00 * lx + R.m01 * ly + R.m02 * lz;
    float wy = R.m10 * lx + R.m11 * ly + R.m12 * lz;
    float wz = R.m20 * lx + R.m21 * ly + R.m22 * lz;
    float ex = wy * rz - wz * ry, ey = wz * rx - wx * rz, ez = wx * ry - wy * rx;
    return inv_m + dx * ex + dy * ey + dz * ez;
}
__device__ __forceinline__ void ground_contact(Regs& r, const DevCfg& c, const Mat3& R, float inv_m, float inv_ixx,
                                               float inv_izz) {
    float ax = R.m02, ay = R.m12, az = R.m22;
    float dx = az * ax, dy = az * ay, dz = az * az - 1.0f;
    float dn = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    float sc = c.radius * rcp(fmaxf(dn, c.cop_s0));
#pragma unroll
    for (int end = 0; end < 2; ++end) {
        float L = end == 0 ? -(c.half_len + r.cg) : (c.half_len - r.cg);
        float rx = L * ax + sc * dx, ry = L * ay + sc * dy, rz = L * az + sc * dz;
        float pz = r.pz + rz;
        if (pz >= 0.0f) continue;
        float depth = -pz;
        float vn = r.vz + (r.wx * ry - r.wy * rx);
        float kn = contact_kinv(R, rx, ry, rz, 0.0f, 0.0f, 1.0f, inv_m, inv_ixx, inv_izz);
        float jn = (-vn + c.erp_over_h * depth) * rcp(kn);
        if (jn <= 0.0f) continue;
        contact_impulse(r, R, rx, ry, rz, 0.0f, 0.0f, 1.0f, jn, inv_m, inv_ixx, inv_izz);
        float tx = r.vx + (r.wy * rz - r.wz * ry);
        float ty = r.vy + (r.wz * rx - r.wx * rz);
        float vtn = __builtin_sqrtf(tx * tx + ty * ty);
        if (vtn > 1e-9f) {
            float iv = rcp(vtn);
            tx *= iv; ty *= iv;
            float kt = contact_kinv(R, rx, ry, rz, tx, ty, 0.0f, inv_m, inv_ixx, inv_izz);
            float jt = fminf(vtn * rcp(kt), c.mu * jn);
            contact_impulse(r, R, rx, ry, rz, tx, ty, 0.0f, -jt, inv_m, inv_ixx, inv_izz);
        }
    }
}

// ------------------------------------------------------------------ reset (ref :381-407 + build-defined DR)
template <bool DR>
__device__ __forceinline__ void reset_dynamic(Regs& r, const DevCfg& c, const DrCfg& d, long long gid) {
    r.px = c.init[0]; r.py = c.init[1]; r.pz = c.init[2];
    r.qx = c.init[3]; r.qy = c.init[4]; r.qz = c.init[5]; r.qw = c.init[6];
    r.vx = r.vy = r.vz = 0.0f;
    r.wx = r.wy = r.wz = 0.0f;
    r.step = 0; r.phase = kPhaseBoost; r.msucc = 0;
    r.ms = 1.0f; r.ts = 1.0f; r.cg = 0.0f; r.windx = r.windy = r.windz = 0.0f;
    if (DR) {
        unsigned u[4], v[4];
        unsigned long long g = (unsigned long long)gid;
        philox4x32_10((unsigned)g, (unsigned)(g >> 32), r.episode, 0u, d.seed_lo, d.seed_hi, u);
        philox4x32_10((unsigned)g, (unsigned)(g >> 32), r.episode, 1u, d.seed_lo, d.seed_hi, v);
        r.ms = 1.0f + d.mass_var * usym(u[0]);
        r.cg = d.cg_max * usym(u[1]);
        float tx = d.tilt_max * usym(u[2]), ty = d.tilt_max * usym(u[3]);
        float z0, z1, z2, z3;
        box_muller(v[0], v[1], z0, z1);
        box_muller(v[2], v[3], z2, z3);
        r.ts = fminf(fmaxf(1.0f + d.thrust_std * z0, 0.5f), 1.5f);
        r.windx = d.wind_std * z1;
        r.windy = d.wind_std * z2;
        if (d.tilt_max > 0.0f) {
            float sx = __sinf(0.5f * tx), cx = __cosf(0.5f * tx), sy = __sinf(0.5f * ty), cy = __cosf(0.5f * ty);
            // qt = qy(ty) (x) qx(tx); q = qt (x) init
            float x1 = cy * sx, y1 = sy * cx, z1q = -sy * sx, w1 = cy * cx;
            float x2 = r.qx, y2 = r.qy, z2q = r.qz, w2 = r.qw;
            float nx = w1 * x2 + x1 * w2 + y1 * z2q - z1q * y2;
            float ny = w1 * y2 + y1 * w2 + z1q * x2 - x1 * z2q;
            float nz = w1 * z2q + z1q * w2 + x1 * y2 - y1 * x2;
            float nw = w1 * w2 - x1 * x2 - y1 * y2 - z1q * z2q;
            float il = __builtin_amdgcn_rsqf(nx * nx + ny * ny + nz * nz + nw * nw);
            r.qx = nx * il; r.qy = ny * il; r.qz = nz * il; r.qw = nw * il;
        }
    }
}

__device__ __forceinline__ float fuel_value(const DevCfg& c, unsigned k) {
    // fuel after k sequential fp64 decrements of 0.001 (ref :533); identical after the fp32 cast
    return k >= (unsigned)c.k_empty ? 0.0f : (float)__builtin_fma(-0.001, (double)k, 1.0);
}

__device__ __forceinline__ float phase_value(unsigned p) {
    // float32(phase_idx / 7.0), ref :593 (list(MissionPhase).index / len(MissionPhase))
    return p == 0u ? 0.0f : p == 1u ? 0.14285715f : p == 2u ? 0.2857143f : p == 3u ? 0.42857143f
         : p == 4u ? 0.5714286f : p == 5u ? 0.71428573f : 0.85714287f;
}

__device__ __forceinline__ void observe(const Regs& r, const DevCfg& c, float o[10]) {
    unsigned k = r.step < (unsigned)c.k_empty ? r.step : (unsigned)c.k_empty;
    o[0] = r.qx; o[1] = r.qy; o[2] = r.qz; o[3] = r.qw;
    o[4] = r.wx; o[5] = r.wy; o[6] = r.wz;
    o[7] = fuel_value(c, k);
    o[8] = phase_value(r.phase);
    o[9] = fminf(1.0f, (float)r.step * c.inv_max_steps);
}

template <bool DR>
__device__ __forceinline__ void add_obs_noise(const Regs& r, const DrCfg& d, long long gid, float o[10]) {
    if (DR) {
        if (d.noise_std > 0.0f) {
            unsigned u[4], v[4];
            unsigned long long g = (unsigned long long)gid;
            philox4x32_10((unsigned)g, (unsigned)(g >> 32), r.episode, 0x10000u + r.step * 2u, d.seed_lo, d.seed_hi, u);
            philox4x32_10((unsigned)g, (unsigned)(g >> 32), r.episode, 0x10001u + r.step * 2u, d.seed_lo, d.seed_hi, v);
            float z[8];
            box_muller(u[0], u[1], z[0], z[1]);
            box_muller(u[2], u[3], z[2], z[3]);
            box_muller(v[0], v[1], z[4], z[5]);
            box_muller(v[2], v[3], z[6], z[7]);
#pragma unroll
            for (int i = 0; i < 7; ++i) o[i] += d.noise_std * z[i];
        }
    }
}

// ------------------------------------------------------------------ physics half
template <bool DR>
__device__ __forceinline__ void physics(Regs& r, const DevCfg& c, float a0, float a1) {
    float inv_m = c.inv_mass, inv_ixx = c.inv_ixx, inv_izz = c.inv_izz;
    if (DR) {
        float ims = rcp(r.ms);
        inv_m *= ims; inv_ixx *= ims; inv_izz *= ims;
    }
    Mat3 R = rotmat(r.qx, r.qy, r.qz, r.qw);
    // force other than gravity (world) and torque (world), held for the whole control step
    float fx = 0.0f, fy = 0.0f, fz = 0.0f;
    if (DR) { fx = r.windx; fy = r.windy; fz = r.windz; }
    float tqx = 0.0f, tqy = 0.0f, tqz = 0.0f;
    if (r.step < (unsigned)c.k_empty) {  // fuel > 0 before the decrement (ref :530)
        float pitch = a0 * 0.3141592653589793f, yaw = a1 * 0.3141592653589793f;
        float p2 = pitch * pitch, y2 = yaw * yaw;
        float sp = pitch * sinc_small(p2), cp = cos_small(p2);
        float sy = yaw * sinc_small(y2), cy = cos_small(y2);
        float T = DR ? c.thrust * r.ts : c.thrust;
        float tlx = T * sy, tly = T * sp, tlz = T * cp * cy;  // ref :539-543
        fx += R.m00 * tlx + R.m01 * tly + R.m02 * tlz;
        fy += R.m10 * tlx + R.m11 * tly + R.m12 * tlz;
        fz += R.m20 * tlx + R.m21 * tly + R.m22 * tlz;
        // (0,0,-L) x Tl in the body frame = (L*Tly, -L*Tlx, 0)   (ref :550-556)
        float L = DR ? c.lever + r.cg : c.lever;
        float bx = L * tly, by = -L * tlx;
        tqx += R.m00 * bx + R.m01 * by;
        tqy += R.m10 * bx + R.m11 * by;
        tqz += R.m20 * bx + R.m21 * by;
    }
    {  // aerodynamics on the pre-step state (ref :561-585)
        float rho = 1.225f * __expf(r.pz * (-1.0f / 8400.0f));
        float v2 = r.vx * r.vx + r.vy * r.vy + r.vz * r.vz;
        float vmag = __builtin_sqrtf(v2);
        if (vmag > 0.1f) {
            float k = -(0.5f * rho * vmag * 0.47f * 7.853981633974483e-3f);  // drag_mag / vmag
            fx += k * r.vx; fy += k * r.vy; fz += k * r.vz;
        }
        float ad = 0.02f * rho;
        tqx -= ad * r.wx; tqy -= ad * r.wy; tqz -= ad * r.wz;
    }
    const float acx = fx * inv_m, acy = fy * inv_m, acz = fz * inv_m - c.g2;  // gravity twice (ref :338,:524-527)
    const float h = c.h;
    for (int s = 0; s < c.nsub; ++s) {
        if (s > 0) R = rotmat(r.qx, r.qy, r.qz, r.qw);
        // angular: body frame
        float wbx = R.m00 * r.wx + R.m10 * r.wy + R.m20 * r.wz;
        float wby = R.m01 * r.wx + R.m11 * r.wy + R.m21 * r.wz;
        float wbz = R.m02 * r.wx + R.m12 * r.wy + R.m22 * r.wz;
        float tbx = R.m00 * tqx + R.m10 * tqy + R.m20 * tqz;
        float tby = R.m01 * tqx + R.m11 * tqy + R.m21 * tqz;
        float tbz = R.m02 * tqx + R.m12 * tqy + R.m22 * tqz;
        float nw = __builtin_sqrtf(wbx * wbx + wby * wby + wbz * wbz);
        float da = c.ka + c.ka * nw;
        float abx = tbx * inv_ixx - c.gyro_r * wby * wbz - wbx * da;
        float aby = tby * inv_ixx + c.gyro_r * wbx * wbz - wby * da;
        float abz = tbz * inv_izz - wbz * da;
        r.wx = clamp100(r.wx + h * (R.m00 * abx + R.m01 * aby + R.m02 * abz));
        r.wy = clamp100(r.wy + h * (R.m10 * abx + R.m11 * aby + R.m12 * abz));
        r.wz = clamp100(r.wz + h * (R.m20 * abx + R.m21 * aby + R.m22 * abz));
        // linear
        float nv = __builtin_sqrtf(r.vx * r.vx + r.vy * r.vy + r.vz * r.vz);
        float dl = c.kl + c.kl * nv;
        r.vx = clamp100(r.vx + h * (acx - r.vx * dl));
        r.vy = clamp100(r.vy + h * (acy - r.vy * dl));
        r.vz = clamp100(r.vz + h * (acz - r.vz * dl));
        if (c.contact) {
            float reach = fabsf(R.m22) * (c.half_len + (DR ? fabsf(r.cg) : 0.0f)) + c.radius;
            if (r.pz - reach <= 0.0f) ground_contact(r, c, R, inv_m, inv_ixx, inv_izz);
        }
        r.px += h * r.vx; r.py += h * r.vy; r.pz += h * r.vz;
        // exponential map, q' = normalize(dq (x) q)
        float wn2 = r.wx * r.wx + r.wy * r.wy + r.wz * r.wz;
        float wn = __builtin_sqrtf(wn2);
        float fa = fminf(wn, 0.7853981633974483f * rcp(h));  // Bullet: clamp |w| h to pi/4
        float xh = 0.5f * fa * h, xh2 = xh * xh;
        float k = 0.5f * h * sinc_small(xh2);  // sin(xh)/fa
        float x1 = r.wx * k, y1 = r.wy * k, z1 = r.wz * k, w1 = cos_small(xh2);
        float x2 = r.qx, y2 = r.qy, z2 = r.qz, w2 = r.qw;
        float nx = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
        float ny = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2;
        float nz = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2;
        float nq = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
        float l2 = nx * nx + ny * ny + nz * nz + nq * nq;
        float il = __builtin_amdgcn_rsqf(l2);
        il = il * (1.5f - 0.5f * l2 * il * il);  // one Newton step: |q| = 1 to fp32 rounding
        r.qx = nx * il; r.qy = ny * il; r.qz = nz * il; r.qw = nq * il;
    }
}

// ------------------------------------------------------------------ logic half + reward
// W10: the reward window lives in hw (registers, static indexing only); else in ringp (this env's column of ring1000).
template <bool W10>
__device__ __forceinline__ void epilogue(Regs& r, float (&hw)[12], const DevCfg& c, float a0, float a1,
                                         float* __restrict__ ringp, int np, StepOut& out, int& ring_slot) {
    // derived scalars (ref :608-633)
    float x = r.qx, y = r.qy, z = r.qz, w = r.qw;
    float sarg = -2.0f * (x * z - w * y);
    float pitch, yaw;
    if (sarg <= -0.99999f) {
        pitch = -1.5707963267948966f;
        yaw = 2.0f * atan2f(x, -y);
    } else if (sarg >= 0.99999f) {
        pitch = 1.5707963267948966f;
        yaw = 2.0f * atan2f(-x, y);
    } else {
        pitch = asinf(sarg);
        yaw = atan2f(2.0f * (x * y + w * z), w * w + x * x - y * y - z * z);
    }
    const float tilt = __builtin_sqrtf(pitch * pitch + yaw * yaw);
    const float alt = r.pz;
    const float wmag = __builtin_sqrtf(r.wx * r.wx + r.wy * r.wy + r.wz * r.wz);
    const float vh = __builtin_sqrtf(r.vx * r.vx + r.vy * r.vy);
    const float vza = fabsf(r.vz);
    const bool crashed = alt < 0.1f;
    const unsigned k = r.step < (unsigned)c.k_empty ? r.step : (unsigned)c.k_empty;  // fuel decrements so far
    const float fuel = fuel_value(c, k);

    const unsigned phase_old = r.phase, msucc_old = r.msucc;
    observe(r, c, out.obs);  // before the phase update (ref :482 vs :485)

    // ref :635-657
    if (r.phase == kPhaseBoost && k >= (unsigned)c.k_coast) {
        r.phase = kPhaseCoast;
    } else if (r.phase == kPhaseCoast && alt < 5.0f) {
        r.phase = kPhaseLanding;
    } else if (r.phase == kPhaseLanding && alt < 1.0f) {
        r.phase = kPhaseTouchdown;
    } else if (r.phase == kPhaseTouchdown && alt < 0.5f) {
        if (tilt < 0.087f && wmag < 0.1f) {
            r.phase = kPhaseComplete;
            r.msucc = 1;
        }
    }
    // ref :659-695 (deque(maxlen=100) of criteria == saturating run length)
    if (!r.msucc) {
        bool pass = (tilt < 0.087f) && (vza < 2.0f && vh < 0.5f) && (0.2f <= alt && alt <= 2.0f) && (wmag < 0.1f);
        r.run = pass ? (r.run < 100u ? r.run + 1u : 100u) : 0u;
        if (r.run >= 100u) r.msucc = 1;
    }

    // ref :86-224
    float mc = msucc_old ? 1.0f : (phase_old == (unsigned)kPhaseLanding ? 0.1f : 0.0f);
    float tilt_pen = __expf(-10.0f * fmaxf(0.0f, tilt - 0.087f));
    float ang_pen = __expf(-5.0f * fmaxf(0.0f, wmag - 0.1f));
    float alt_pen = (0.2f <= alt && alt <= 20.0f) ? 1.0f : 0.5f;
    float safe = (tilt_pen + ang_pen + alt_pen) * (1.0f / 3.0f);
    float ce = __builtin_sqrtf(a0 * a0 + a1 * a1);
    float fe = (k < (unsigned)c.k_low && ce < 0.5f) ? fuel * (1.0f - ce) : 0.0f;
    float stab = (tilt < 0.05f && wmag < 0.1f) ? 1.0f : ((tilt < 0.1f && wmag < 0.2f) ? 0.5f : 0.0f);
    float smooth = 1.0f;
    if (r.has_pa) {
        float d0 = a0 - r.pa0, d1 = a1 - r.pa1;
        smooth = __expf(-5.0f * __builtin_sqrtf(d0 * d0 + d1 * d1));
    }
    r.pa0 = a0; r.pa1 = a1; r.has_pa = 1;
    float altm = __expf(-2.0f * fabsf(alt - 3.0f));
    float total = mc * 100.0f + safe * 50.0f + fe * 20.0f + stab * 10.0f + smooth * 5.0f + altm * 5.0f;
    const float crash_pen = crashed ? -1000.0f : 0.0f;
    const float tilt_excess = tilt > 0.52f ? -500.0f * (tilt - 0.52f) : 0.0f;
    const float sat_pen = ce > 0.9f ? -50.0f * (ce - 0.9f) : 0.0f;
    total += crash_pen;
    total += tilt_excess;
    total += sat_pen;
    out.comps[0] = mc * 100.0f; out.comps[1] = safe * 50.0f; out.comps[2] = fe * 20.0f; out.comps[3] = stab * 10.0f;
    out.comps[4] = smooth * 5.0f; out.comps[5] = altm * 5.0f;
    out.comps[6] = crash_pen; out.comps[7] = tilt_excess; out.comps[8] = sat_pen;

    // anti-hacking (ref :209-224): variance of the last 10 when len > 10; distinct fraction > 0.8
    const unsigned W = W10 ? 10u : (unsigned)c.W;
    const unsigned wl = r.hist_len < W ? r.hist_len : W;
    const bool full = wl == W;
    const unsigned slot = full ? r.head : wl;
    float adj = 0.0f;
    if (r.hist_len > 10u) {
        float s = 0.0f, q = 0.0f;
        if (W10) {
#pragma unroll
            for (int i = 0; i < 10; ++i) s += hw[i];
            float mean = s * 0.1f;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                float d = hw[i] - mean;
                q += d * d;
            }
        } else {
            unsigned base = (full ? r.head : 0u) + wl - 10u;
            float l[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                unsigned p = base + i;
                p = p >= W ? p - W : p;
                l[i] = ringp[(size_t)p * np];
                s += l[i];
            }
            float mean = s * 0.1f;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                float d = l[i] - mean;
                q += d * d;
            }
        }
        float var = q * 0.1f;
        if (var > 10000.0f) adj -= 0.1f * var;
    }
    // distinct > 0.8 * len  <=>  5 * distinct > 4 * len for len <= 1000 (checked in tests)
    if (5u * r.distinct > 4u * wl) adj += 0.05f;
    total += adj;
    out.comps[9] = adj; out.comps[10] = total;
    // presence mask of the penalty keys (the reference's dict only holds the penalties that fired, ref :189-207)
    out.comps[11] = (float)((crashed ? 1 : 0) | (tilt > 0.52f ? 2 : 0) | (ce > 0.9f ? 4 : 0));
    total = fminf(fmaxf(total, -1000.0f), 200.0f);

    // append to the window, maintain the distinct count incrementally
    {
        bool e_dup = false, v_dup = false;
        if (W10) {
            float ev = 0.0f;
#pragma unroll
            for (int i = 0; i < 10; ++i) ev = ((unsigned)i == slot) ? hw[i] : ev;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                bool valid = (unsigned)i < wl && (unsigned)i != slot;
                e_dup |= valid && (hw[i] == ev);
                v_dup |= valid && (hw[i] == total);
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) hw[i] = ((unsigned)i == slot) ? total : hw[i];
        } else {
            float ev = full ? ringp[(size_t)slot * np] : 0.0f;
            // One pass over the env's ring, TVC_RING_BATCH independent loads at a time (the loop used to issue one load per iteration
            // and wait for it -- with the rings full, 1000 steps into a run, that was 1000 serial round trips per step, +0.4 ms on the
            // 65 536-env train step; found in round 3, both earlier rounds quoted the first few hundred steps).  Matches are COUNTED in
            // two VGPRs over the whole window, the slot included (when full it holds ev, once): per-element `valid && equal`
            // booleans are 64-bit lane masks in SGPRs, and beyond 16 of them hipcc serialised the loads again to save SGPRs
            // (batch 32: 318 us instead of 81 at 65 536 envs, tools/env_ring_bench.py).
            constexpr unsigned RB = TVC_RING_BATCH;
            unsigned ce = 0u, cv = 0u;
            const float* rp = ringp;
            unsigned i0 = 0u;
            for (; i0 + RB <= wl; i0 += RB) {
                float hv[RB];
#pragma unroll
                for (unsigned u = 0; u < RB; ++u) { hv[u] = *rp; rp += np; }
#pragma unroll
                for (unsigned u = 0; u < RB; ++u) {
                    ce += hv[u] == ev ? 1u : 0u;
                    cv += hv[u] == total ? 1u : 0u;
                }
            }
            if (i0 < wl) {  // the last, partial batch: clamped addresses, the entries past the window do not count
                float hv[RB];
#pragma unroll
                for (unsigned u = 0; u < RB; ++u) hv[u] = ringp[(size_t)min(i0 + u, wl - 1u) * np];
#pragma unroll
                for (unsigned u = 0; u < RB; ++u) {
                    const unsigned in = (i0 + u) < wl ? 1u : 0u;
                    ce += hv[u] == ev ? in : 0u;
                    cv += hv[u] == total ? in : 0u;
                }
            }
            e_dup = ce >= 2u;                                       // (only read when full: the slot itself is one match)
            v_dup = cv > ((full && ev == total) ? 1u : 0u);         // the slot is about to be overwritten: it does not count
            ringp[(size_t)slot * np] = total;
        }
        ring_slot = (int)slot;
        unsigned d = r.distinct;
        if (full && !e_dup) d -= 1u;
        if (!v_dup) d += 1u;
        r.distinct = d;
        if (full) r.head = (r.head + 1u == W) ? 0u : r.head + 1u;
        if (r.hist_len < 1000u) r.hist_len += 1u;
    }
    out.reward = total;

    // ref :697-721
    unsigned term = 0, trunc = 0;
    if (r.msucc) {
        term = 1;
    } else {
        if (crashed || tilt > 0.52f || alt > 20.0f || (r.px * r.px + r.py * r.py) > 2500.0f) term = 1;
        if (r.step >= (unsigned)c.max_steps) trunc = 1;
    }
    out.term = term;
    out.trunc = trunc;
}

// ------------------------------------------------------------------ state <-> registers
template <bool DR, bool W10>
__device__ __forceinline__ void load_regs(Regs& r, float (&hw)[12], const EnvBuf& b, int i) {
    const float4* c = b.cells + i;
    const size_t np = b.np;
    float4 g0 = c[0], g1 = c[np], g2 = c[2 * np], g3 = c[3 * np];
    // cell group 4 is two float2 planes: {previous action} (every instantiation) and {mass scale, thrust scale} (DR only):
    // the nominal kernel neither reads nor writes the second plane (16 B per env-step less than one float4 cell)
    const float2* p4 = reinterpret_cast<const float2*>(b.cells + 4 * np);
    const float2 pa = p4[i];
    float4 g4 = make_float4(pa.x, pa.y, 1.0f, 1.0f);
    if (DR) { const float2 mt = p4[np + i]; g4.z = mt.x; g4.w = mt.y; }
    r.px = g0.x; r.py = g0.y; r.pz = g0.z;
    r.qx = g1.x; r.qy = g1.y; r.qz = g1.z; r.qw = g1.w;
    r.vx = g2.x; r.vy = g2.y; r.vz = g2.z;
    r.wx = g3.x; r.wy = g3.y; r.wz = g3.z;
    r.pa0 = g4.x; r.pa1 = g4.y; r.ms = g4.z; r.ts = g4.w;
    unsigned a0 = __float_as_uint(g0.w), a1 = __float_as_uint(g2.w);
    r.step = a0 & 0xFFFFu; r.phase = (a0 >> 16) & 7u; r.msucc = (a0 >> 19) & 1u; r.has_pa = (a0 >> 20) & 1u;
    r.run = (a0 >> 21) & 127u;
    r.hist_len = a1 & 1023u; r.head = (a1 >> 10) & 1023u; r.distinct = (a1 >> 20) & 1023u;
    r.episode = __float_as_uint(g3.w);
    if (DR) {
        float4 g5 = c[5 * np];
        r.cg = g5.x; r.windx = g5.y; r.windy = g5.z; r.windz = g5.w;
    } else {
        r.cg = 0.0f; r.windx = r.windy = r.windz = 0.0f;
    }
    if (W10) {
        float4 h0 = c[6 * np], h1 = c[7 * np], h2 = c[8 * np];
        hw[0] = h0.x; hw[1] = h0.y; hw[2] = h0.z; hw[3] = h0.w;
        hw[4] = h1.x; hw[5] = h1.y; hw[6] = h1.z; hw[7] = h1.w;
        hw[8] = h2.x; hw[9] = h2.y; hw[10] = h2.z; hw[11] = h2.w;
    }
}
// ring_slot: the window slot written this step (W10; -1 = none, 100 = all three groups)
// write_params: the six per-episode parameters changed (reset / import); a plain step leaves their planes alone
template <bool DR, bool W10>
__device__ __forceinline__ void store_regs(const Regs& r, const float (&hw)[12], const EnvBuf& b, int i, int ring_slot,
                                           bool write_params = true) {
    float4* c = b.cells + i;
    const size_t np = b.np;
    unsigned a0 = (r.step & 0xFFFFu) | (r.phase << 16) | (r.msucc << 19) | (r.has_pa << 20) | (r.run << 21);
    unsigned a1 = r.hist_len | (r.head << 10) | (r.distinct << 20);
    c[0] = make_float4(r.px, r.py, r.pz, __uint_as_float(a0));
    c[np] = make_float4(r.qx, r.qy, r.qz, r.qw);
    c[2 * np] = make_float4(r.vx, r.vy, r.vz, __uint_as_float(a1));
    c[3 * np] = make_float4(r.wx, r.wy, r.wz, __uint_as_float(r.episode));
    float2* p4 = reinterpret_cast<float2*>(b.cells + 4 * np);
    p4[i] = make_float2(r.pa0, r.pa1);
    if (DR && write_params) {
        p4[np + i] = make_float2(r.ms, r.ts);
        c[5 * np] = make_float4(r.cg, r.windx, r.windy, r.windz);
    }
    if (W10) {
        if (ring_slot == 100) {
            c[6 * np] = make_float4(hw[0], hw[1], hw[2], hw[3]);
            c[7 * np] = make_float4(hw[4], hw[5], hw[6], hw[7]);
            c[8 * np] = make_float4(hw[8], hw[9], hw[10], hw[11]);
        } else if (ring_slot >= 0) {
            // only the 16-byte cell that holds the slot appended this step goes back to HBM
            const int g = ring_slot >> 2;
            float4 v = g == 0 ? make_float4(hw[0], hw[1], hw[2], hw[3])
                     : g == 1 ? make_float4(hw[4], hw[5], hw[6], hw[7])
                              : make_float4(hw[8], hw[9], hw[10], hw[11]);
            c[(size_t)(6 + g) * np] = v;
        }
    }
}

}  // namespace tvcdev
