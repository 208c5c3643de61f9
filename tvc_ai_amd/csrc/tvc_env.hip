// libtvc_hip.so -- vector rocket-TVC env: kernels K1 (step) / K2 (reset) and their C ABI.
//
// One lane integrates one env; state lives in HBM as struct-of-arrays [field][N] so that every
// field access of a wavefront is one contiguous 256-byte segment.  The row-major [N,10]
// observation / [N,2] action tensors the policy side wants are staged through LDS so they leave
// (enter) the CU as contiguous 16-byte-per-lane stores.
//
// Replaces: EnhancedRocketTVCEnv.step/reset, env/enhanced_rocket_tvc_env.py:381-407, 466-518.
#include <cmath>
#include <cstring>
#include <new>

#include "tvc_common.h"
#include "tvc_env_device.h"

namespace tvc {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace tvc

using namespace tvcdev;

namespace {

constexpr int kMaxBlock = 256;

// ------------------------------------------------------------------ kernels

// Cooperative store of a [rows,10] fp32 tile (LDS, row-major) to global row-major memory.
__device__ __forceinline__ void store_tile10(const float* tile, float* __restrict__ gbase, int row0, int n_rows_total,
                                             int rows_in_block) {
    const int valid_rows = min(rows_in_block, n_rows_total - row0);
    if (valid_rows <= 0) return;
    const int n_f = valid_rows * 10;
    float* g = gbase + (size_t)row0 * 10;
    const int n_v4 = n_f >> 2;  // row0*10*4 B is 16-B aligned whenever row0 % 2 == 0 (blocks are multiples of 64 rows)
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    float4* g4 = reinterpret_cast<float4*>(g);
    for (int v = threadIdx.x; v < n_v4; v += blockDim.x) g4[v] = t4[v];
    for (int f = (n_v4 << 2) + threadIdx.x; f < n_f; f += blockDim.x) g[f] = tile[f];
}

template <bool W10, bool DR>
__global__ void __launch_bounds__(kMaxBlock)
env_step_kernel(EnvBuf b, DevCfg c, const DrCfg* __restrict__ dp, const float* __restrict__ act, float* __restrict__ obs,
                float* __restrict__ rew, unsigned char* __restrict__ term, unsigned char* __restrict__ trunc,
                float* __restrict__ final_obs, float* __restrict__ comps, EpStats ep, int n_steps) {
    __shared__ __attribute__((aligned(16))) float tile[kMaxBlock * 10];
    const int row0 = blockIdx.x * blockDim.x;
    const int i = row0 + threadIdx.x;
    const bool valid = i < b.n;
    // the DR ranges live in a small DEVICE record (uniform address: scalar loads) that tvc_env_set_dr rewrites in stream order,
    // so a curriculum stage change also reaches launches replayed from a captured hipGraph (kernel arguments are frozen there)
    DrCfg d{};
    if (DR) d = *dp;
    const long long gid = DR ? d.id_off + i : 0;
    Regs r;
    float hw[12];
    float* ringp = W10 ? nullptr : b.ring1000 + (valid ? i : 0);
    int ring_slot = -1;
    bool params_dirty = false;
    if (valid) load_regs<DR, W10>(r, hw, b, i);
    for (int t = 0; t < n_steps; ++t) {
        const size_t toff = (size_t)t * b.n;
        StepOut o;
        bool done = false;
        if (valid) {
            const float2 a = reinterpret_cast<const float2*>(act)[toff + i];
            const float a0 = fminf(fmaxf(a.x, -1.0f), 1.0f), a1 = fminf(fmaxf(a.y, -1.0f), 1.0f);  // ref :470
            physics<DR>(r, c, a0, a1);
            r.step += 1u;  // ref :478
            epilogue<W10>(r, hw, c, a0, a1, ringp, b.np, o, ring_slot);
            add_obs_noise<DR>(r, d, gid, o.obs);
            done = (o.term | o.trunc) != 0u;
            rew[toff + i] = o.reward;
            term[toff + i] = (unsigned char)o.term;
            trunc[toff + i] = (unsigned char)o.trunc;
#pragma unroll
            for (int k = 0; k < 10; ++k) tile[threadIdx.x * 10 + k] = o.obs[k];
            if (comps != nullptr) {  // uniform branch; info['reward_components'] of the N = 1 surface (ref :514)
                float4* cp = reinterpret_cast<float4*>(comps + (toff + i) * 12);
                cp[0] = make_float4(o.comps[0], o.comps[1], o.comps[2], o.comps[3]);
                cp[1] = make_float4(o.comps[4], o.comps[5], o.comps[6], o.comps[7]);
                cp[2] = make_float4(o.comps[8], o.comps[9], o.comps[10], o.comps[11]);
            }
        }
        if (ep.ret != nullptr) {  // uniform branch: episode statistics, one atomic set per wavefront that saw an episode end
            float R = 0.0f;
            if (valid) R = ep.ret[i] + o.reward;
            const bool fin = valid && done;
            if (__any(fin)) {
                float c = fin ? 1.0f : 0.0f, sc = (fin && r.msucc) ? 1.0f : 0.0f, rs = fin ? R : 0.0f, ls = fin ? (float)r.step : 0.0f;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    c += __shfl_xor(c, m); sc += __shfl_xor(sc, m); rs += __shfl_xor(rs, m); ls += __shfl_xor(ls, m);
                }
                if ((threadIdx.x & 63) == 0) {
                    double* rec = ep.sums + 16 * (blockIdx.x & (kEpSlots - 1));
                    atomicAdd(rec, (double)c);
                    if (sc > 0.0f) atomicAdd(rec + 1, (double)sc);
                    atomicAdd(rec + 2, (double)rs);
                    atomicAdd(rec + 3, (double)ls);
                }
            }
            if (valid) ep.ret[i] = fin ? 0.0f : R;
        }
        if (final_obs != nullptr) {
            __syncthreads();
            store_tile10(tile, final_obs + toff * 10, row0, b.n, blockDim.x);
            __syncthreads();
        }
        if (valid && done && c.auto_reset) {
            r.episode += 1u;
            params_dirty = true;
            reset_dynamic<DR>(r, c, d, gid);
            observe(r, c, o.obs);
            add_obs_noise<DR>(r, d, gid, o.obs);
#pragma unroll
            for (int k = 0; k < 10; ++k) tile[threadIdx.x * 10 + k] = o.obs[k];
        }
        __syncthreads();
        store_tile10(tile, obs + toff * 10, row0, b.n, blockDim.x);
        if (t + 1 < n_steps) __syncthreads();
    }
    if (valid) store_regs<DR, W10>(r, hw, b, i, n_steps == 1 ? ring_slot : 100, params_dirty);
}

__global__ void env_set_dr_kernel(DrCfg v, DrCfg* __restrict__ dst) { *dst = v; }

__global__ void __launch_bounds__(kMaxBlock)
env_reset_kernel(EnvBuf b, DevCfg c, DrCfg d, int dr, const unsigned char* __restrict__ mask, int hard,
                 float* __restrict__ obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    if (mask != nullptr && mask[i] == 0) return;
    Regs r;
    float hw[12];
    load_regs<true, true>(r, hw, b, i);
    if (hard) {
        r.run = 0; r.has_pa = 0; r.pa0 = r.pa1 = 0.0f;
        r.hist_len = 0; r.head = 0; r.distinct = 0;
        r.episode = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) hw[k] = 0.0f;
        if (b.ring1000)
            for (int k = 0; k < 1000; ++k) b.ring1000[(size_t)k * b.np + i] = 0.0f;
    } else {
        r.episode += 1u;
    }
    const long long gid = d.id_off + i;
    float o[10];
    if (dr) {
        reset_dynamic<true>(r, c, d, gid);
        observe(r, c, o);
        add_obs_noise<true>(r, d, gid, o);
    } else {
        reset_dynamic<false>(r, c, d, gid);
        observe(r, c, o);
    }
    store_regs<true, true>(r, hw, b, i, 100);
    if (obs != nullptr) {
#pragma unroll
        for (int k = 0; k < 10; ++k) obs[(size_t)i * 10 + k] = o[k];
    }
}

// aux layout of the C ABI: step, phase, mission_successful, success_run, hist_len, has_prev_action, distinct, episode
__global__ void env_export_kernel(EnvBuf b, int W, float* dyn, int* aux, float* pa, float* par, float* hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    Regs r;
    float hw[12];
    load_regs<true, true>(r, hw, b, i);
    if (dyn) {
        float* d = dyn + (size_t)i * 13;
        d[0] = r.px; d[1] = r.py; d[2] = r.pz; d[3] = r.qx; d[4] = r.qy; d[5] = r.qz; d[6] = r.qw;
        d[7] = r.vx; d[8] = r.vy; d[9] = r.vz; d[10] = r.wx; d[11] = r.wy; d[12] = r.wz;
    }
    if (aux) {
        int* a = aux + (size_t)i * 8;
        a[0] = r.step; a[1] = r.phase; a[2] = r.msucc; a[3] = r.run; a[4] = r.hist_len; a[5] = r.has_pa;
        a[6] = r.distinct; a[7] = r.episode;
    }
    if (pa) { pa[(size_t)i * 2] = r.pa0; pa[(size_t)i * 2 + 1] = r.pa1; }
    if (par) {
        float* p = par + (size_t)i * 8;
        p[0] = r.ms; p[1] = r.ts; p[2] = r.cg; p[3] = r.windx; p[4] = r.windy; p[5] = r.windz; p[6] = 0.f; p[7] = 0.f;
    }
    if (hist) {
        const unsigned wl = r.hist_len < (unsigned)W ? r.hist_len : (unsigned)W;
        const unsigned start = wl == (unsigned)W ? r.head : 0u;
        const float* cellf = reinterpret_cast<const float*>(b.cells);
        for (unsigned k = 0; k < (unsigned)W; ++k) {
            float v = 0.0f;
            if (k < wl) {
                unsigned p = start + k;
                p = p >= (unsigned)W ? p - W : p;
                v = (W == 10) ? cellf[((size_t)(6 + (p >> 2)) * b.np + i) * 4 + (p & 3u)] : b.ring1000[(size_t)p * b.np + i];
            }
            hist[(size_t)i * W + k] = v;
        }
    }
}

__global__ void env_import_kernel(EnvBuf b, int W, const float* dyn, const int* aux, const float* pa, const float* par,
                                  const float* hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    Regs r;
    float hw[12];
    load_regs<true, true>(r, hw, b, i);
    if (dyn) {
        const float* d = dyn + (size_t)i * 13;
        r.px = d[0]; r.py = d[1]; r.pz = d[2]; r.qx = d[3]; r.qy = d[4]; r.qz = d[5]; r.qw = d[6];
        r.vx = d[7]; r.vy = d[8]; r.vz = d[9]; r.wx = d[10]; r.wy = d[11]; r.wz = d[12];
    }
    if (aux) {
        const int* a = aux + (size_t)i * 8;
        r.step = a[0]; r.phase = a[1]; r.msucc = a[2]; r.run = a[3]; r.hist_len = a[4]; r.has_pa = a[5];
        r.distinct = a[6]; r.episode = a[7];
    }
    if (pa) { r.pa0 = pa[(size_t)i * 2]; r.pa1 = pa[(size_t)i * 2 + 1]; }
    if (par) {
        const float* p = par + (size_t)i * 8;
        r.ms = p[0]; r.ts = p[1]; r.cg = p[2]; r.windx = p[3]; r.windy = p[4]; r.windz = p[5];
    }
    if (hist) {  // oldest first -> physical slots 0.., head = 0
        if (W == 10) {
#pragma unroll
            for (int k = 0; k < 10; ++k) hw[k] = hist[(size_t)i * 10 + k];
        } else {
            for (int k = 0; k < W; ++k) b.ring1000[(size_t)k * b.np + i] = hist[(size_t)i * W + k];
        }
        r.head = 0;
    }
    store_regs<true, true>(r, hw, b, i, 100);
}

__global__ void hwid_debug_kernel(unsigned* __restrict__ out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the block alive for a moment so that blocks spread over the allowed CUs instead of reusing the first one
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
__global__ void philox_debug_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned o[4];
    philox4x32_10(in[6 * i], in[6 * i + 1], in[6 * i + 2], in[6 * i + 3], in[6 * i + 4], in[6 * i + 5], o);
    for (int k = 0; k < 4; ++k) out[4 * i + k] = o[k];
}

// info dict of _get_enhanced_info (ref :723-742) as a tensor
__global__ void env_info_kernel(EnvBuf b, DevCfg c, float* info) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    Regs r;
    float hw[12];
    load_regs<false, false>(r, hw, b, i);
    float x = r.qx, y = r.qy, z = r.qz, w = r.qw;
    float sarg = -2.0f * (x * z - w * y);
    float pitch, yaw;
    if (sarg <= -0.99999f) { pitch = -1.5707963267948966f; yaw = 2.0f * atan2f(x, -y); }
    else if (sarg >= 0.99999f) { pitch = 1.5707963267948966f; yaw = 2.0f * atan2f(-x, y); }
    else { pitch = asinf(sarg); yaw = atan2f(2.0f * (x * y + w * z), w * w + x * x - y * y - z * z); }
    float tilt = sqrtf(pitch * pitch + yaw * yaw);
    float* o = info + (size_t)i * 8;
    unsigned k = r.step < (unsigned)c.k_empty ? r.step : (unsigned)c.k_empty;
    o[0] = r.px; o[1] = r.py; o[2] = r.pz;
    o[3] = tilt * 57.29577951308232f;
    o[4] = sqrtf(r.wx * r.wx + r.wy * r.wy + r.wz * r.wz);
    o[5] = fuel_value(c, k);
    o[6] = (float)r.phase;
    o[7] = r.run >= 10u ? 1.0f : 0.0f;
}

}  // namespace

// ------------------------------------------------------------------ host side

struct tvc_env {
    tvc_env_cfg cfg;
    DevCfg dc;
    DrCfg dr;
    EnvBuf buf;
    int n;
    int device;
    int W;
    void* slab;
    DrCfg* dr_dev;     // device copy of `dr` read by the step kernel (rewritten in stream order by tvc_env_set_dr_async)
    float* comps_out;  // optional [N,12] reward-component sink (tvc_env_set_components_out)
    EpStats ep;        // optional episode statistics (tvc_env_set_episode_stats)
};

static int fuel_threshold(double thr, bool strict_less) {
    // first k such that fuel_k < thr (strict_less) or fuel_k <= thr, fuel_k by sequential fp64 subtraction (ref :533)
    double f = 1.0;
    for (int k = 0; k <= 100000; ++k) {
        if (strict_less ? (f < thr) : (f <= thr)) return k;
        if (!(f > 0.0)) return k;
        f = f - 0.001;
        if (f < 0.0) f = 0.0;
    }
    return 100000;
}

static int build_devcfg(const tvc_env_cfg& g, int W, DevCfg& d, DrCfg& r) {
    memset(&d, 0, sizeof(d));
    memset(&r, 0, sizeof(r));
    d.inv_mass = (float)(1.0 / g.mass);
    d.inv_ixx = (float)(1.0 / g.inertia_xx);
    d.inv_izz = (float)(1.0 / g.inertia_zz);
    d.gyro_r = (float)((g.inertia_zz - g.inertia_xx) / g.inertia_xx);
    d.thrust = (float)g.thrust; d.lever = (float)g.half_len; d.radius = (float)g.radius; d.half_len = (float)g.half_len;
    d.kl = (float)g.lin_damp; d.ka = (float)g.ang_damp; d.g2 = (float)(2.0 * g.gravity); d.h = (float)g.dt_sub;
    d.mu = (float)g.mu; d.erp_over_h = (float)(g.erp / g.dt_sub); d.cop_s0 = (float)g.cop_s0;
    d.inv_max_steps = (float)(1.0 / (double)g.max_episode_steps);
    d.nsub = g.n_sub;
    d.max_steps = g.max_episode_steps;
    d.k_empty = fuel_threshold(0.0, false);
    d.k_coast = fuel_threshold(0.8, true);
    d.k_low = fuel_threshold(0.1, false);
    d.contact = g.contact; d.auto_reset = g.auto_reset; d.W = W;
    for (int i = 0; i < 3; ++i) d.init[i] = (float)g.init_pos[i];
    for (int i = 0; i < 4; ++i) d.init[3 + i] = (float)g.init_quat[i];
    r.mass_var = (float)g.dr_mass_var; r.thrust_std = (float)g.dr_thrust_std; r.cg_max = (float)g.dr_cg_max;
    r.wind_std = (float)g.dr_wind_std; r.tilt_max = (float)g.dr_init_tilt_max;
    r.noise_std = (float)g.dr_obs_noise_std;
    r.seed_lo = (unsigned)(g.seed & 0xFFFFFFFFull); r.seed_hi = (unsigned)(g.seed >> 32);
    r.id_off = g.env_id_offset;
    return 0;
}

static int validate_cfg(const tvc_env_cfg* c) {
    if (!c) return tvc::set_error(TVC_EINVAL, "cfg is NULL");
    if (!(c->mass > 0) || !(c->inertia_xx > 0) || !(c->inertia_zz > 0)) return tvc::set_error(TVC_EINVAL, "mass/inertia must be > 0");
    if (c->n_sub < 1 || c->n_sub > 64) return tvc::set_error(TVC_EINVAL, "n_sub out of range");
    if (!(c->dt_sub > 0)) return tvc::set_error(TVC_EINVAL, "dt_sub must be > 0");
    if (c->max_episode_steps < 1 || c->max_episode_steps > 65535) return tvc::set_error(TVC_EINVAL, "max_episode_steps must be in [1, 65535]");
    if (c->distinct_window != 10 && c->distinct_window != 1000) return tvc::set_error(TVC_EINVAL, "distinct_window must be 10 or 1000");
    return 0;
}

extern "C" {

const char* tvc_last_error(void) { return tvc::last_error_buf(); }
int tvc_abi_version(void) { return TVC_ABI_VERSION; }

void tvc_env_default_cfg(tvc_env_cfg* c) {
    if (!c) return;
    memset(c, 0, sizeof(*c));
    const double mass = 2.0, length = 1.0, radius = 0.05;
    c->mass = mass;
    c->inertia_xx = ((1.0 / 12.0) * mass) * (3.0 * (radius * radius) + length * length);
    c->inertia_zz = ((1.0 / 2.0) * mass) * (radius * radius);
    c->thrust = 35.0; c->half_len = 0.5; c->radius = radius;
    c->lin_damp = 0.01; c->ang_damp = 0.02; c->gravity = 9.81;
    c->dt_sub = 0.02 / 4.0; c->n_sub = 4;
    c->max_episode_steps = 1000;
    c->distinct_window = 10;
    c->contact = 1; c->auto_reset = 1;
    c->mu = 0.8 * 0.3; c->erp = 0.2; c->cop_s0 = 0.02;
    c->init_pos[2] = 1.0; c->init_quat[3] = 1.0;
    c->seed = 42;
}

int tvc_env_create(const tvc_env_cfg* cfg, int32_t n_envs, int32_t device, tvc_env** out) {
    if (!out) return tvc::set_error(TVC_EINVAL, "out is NULL");
    *out = nullptr;
    if (int e = validate_cfg(cfg)) return e;
    if (n_envs < 1) return tvc::set_error(TVC_EINVAL, "n_envs must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tvc::set_error(TVC_ENODEV, "no HIP device visible: libtvc_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return tvc::set_error(TVC_EINVAL, "device %d out of range (%d visible)", device, ndev);
    TVC_HIP_CHECK(hipSetDevice(device));
    tvc_env* e = new (std::nothrow) tvc_env();
    if (!e) return tvc::set_error(TVC_ENOMEM, "host allocation failed");
    e->cfg = *cfg; e->n = n_envs; e->device = device; e->W = cfg->distinct_window; e->comps_out = nullptr;
    e->ep.ret = nullptr; e->ep.sums = nullptr;
    build_devcfg(*cfg, e->W, e->dc, e->dr);
    const int np = tvc::ceil_div(n_envs, 64) * 64;
    const size_t state_bytes = (size_t)np * kCellGroups * sizeof(float4) + (e->W == 1000 ? (size_t)np * 1000 * sizeof(float) : 0);
    const size_t bytes = state_bytes + 256;  // + the DR record
    hipError_t he = hipMalloc(&e->slab, bytes);
    if (he != hipSuccess) {
        delete e;
        return tvc::set_error(TVC_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(he));
    }
    e->buf.cells = (float4*)e->slab;
    e->buf.ring1000 = e->W == 1000 ? (float*)((char*)e->slab + (size_t)np * kCellGroups * sizeof(float4)) : nullptr;
    e->buf.n = n_envs; e->buf.np = np;
    e->dr_dev = (DrCfg*)((char*)e->slab + state_bytes);
    he = hipMemset(e->slab, 0, bytes);
    if (he == hipSuccess) he = hipMemcpy(e->dr_dev, &e->dr, sizeof(DrCfg), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        (void)hipFree(e->slab);
        delete e;
        return tvc::set_error(TVC_EHIP, "hipMemset / hipMemcpy failed: %s", hipGetErrorString(he));
    }
    *out = e;
    int rc = tvc_env_reset(e, nullptr, 1, nullptr, nullptr);
    if (rc == 0) {
        he = hipStreamSynchronize(nullptr);
        if (he != hipSuccess) rc = tvc::set_error(TVC_EHIP, "reset kernel failed: %s", hipGetErrorString(he));
    }
    if (rc != 0) {
        (void)hipFree(e->slab);
        delete e;
        *out = nullptr;
    }
    return rc;
}

void tvc_env_destroy(tvc_env* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipFree(e->slab);
    delete e;
}

int32_t tvc_env_num_envs(const tvc_env* e) { return e ? e->n : 0; }

int tvc_env_set_dr_async(tvc_env* e, const tvc_env_cfg* cfg, void* stream) {
    if (!e || !cfg) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    e->cfg.dr_enabled = cfg->dr_enabled; e->cfg.dr_mass_var = cfg->dr_mass_var; e->cfg.dr_thrust_std = cfg->dr_thrust_std;
    e->cfg.dr_cg_max = cfg->dr_cg_max; e->cfg.dr_wind_std = cfg->dr_wind_std;
    e->cfg.dr_init_tilt_max = cfg->dr_init_tilt_max; e->cfg.dr_obs_noise_std = cfg->dr_obs_noise_std;
    build_devcfg(e->cfg, e->W, e->dc, e->dr);
    // the record travels as a kernel argument (captured at enqueue) and lands in stream order: steps enqueued on `stream`
    // before this call keep the old ranges, later ones -- graph replays included -- see the new ones
    hipLaunchKernelGGL(env_set_dr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, e->dr, e->dr_dev);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_env_set_dr(tvc_env* e, const tvc_env_cfg* cfg) {
    if (!e || !cfg) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    TVC_HIP_CHECK(hipDeviceSynchronize());  // whatever stream the caller steps on: everything enqueued so far keeps the old ranges
    if (int rc = tvc_env_set_dr_async(e, cfg, nullptr)) return rc;
    TVC_HIP_CHECK(hipStreamSynchronize(nullptr));
    return 0;
}

static inline int block_for(int n) { return n >= 65536 ? 256 : 64; }

int tvc_env_reset(tvc_env* e, const uint8_t* mask_dev, int32_t hard, float* obs_dev, void* stream) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    const int blk = 256;
    hipLaunchKernelGGL(env_reset_kernel, dim3(tvc::ceil_div(e->n, blk)), dim3(blk), 0, (hipStream_t)stream, e->buf, e->dc,
                       e->dr, (int)(e->cfg.dr_enabled != 0), mask_dev, (int)hard, obs_dev);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

static int launch_step(tvc_env* e, int n_steps, const float* act, float* obs, float* rew, uint8_t* term, uint8_t* trunc,
                       float* final_obs, void* stream) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    if (!act || !obs || !rew || !term || !trunc) return tvc::set_error(TVC_EINVAL, "act/obs/rew/term/trunc must be non-NULL");
    if (n_steps < 1) return tvc::set_error(TVC_EINVAL, "n_steps must be >= 1");
    if ((reinterpret_cast<uintptr_t>(obs) & 15) || (reinterpret_cast<uintptr_t>(act) & 7) ||
        (final_obs && (reinterpret_cast<uintptr_t>(final_obs) & 15)))
        return tvc::set_error(TVC_EINVAL, "obs/final_obs must be 16-byte aligned, act 8-byte aligned");
    if (n_steps > 1 && (e->n % 2) != 0) return tvc::set_error(TVC_EINVAL, "step_many needs an even number of envs (row alignment)");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    const int blk = block_for(e->n);
    dim3 grid(tvc::ceil_div(e->n, blk)), block(blk);
    const bool w10 = e->W == 10, dr = e->cfg.dr_enabled != 0;
    hipStream_t st = (hipStream_t)stream;
#define TVC_LAUNCH_STEP(A, B)                                                                                         \
    hipLaunchKernelGGL((env_step_kernel<A, B>), grid, block, 0, st, e->buf, e->dc, e->dr_dev, act, obs, rew, term, trunc, \
                       final_obs, n_steps == 1 ? e->comps_out : nullptr, e->ep, n_steps)
    if (w10 && !dr) TVC_LAUNCH_STEP(true, false);
    else if (w10 && dr) TVC_LAUNCH_STEP(true, true);
    else if (!w10 && !dr) TVC_LAUNCH_STEP(false, false);
    else TVC_LAUNCH_STEP(false, true);
#undef TVC_LAUNCH_STEP
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_env_step(tvc_env* e, const float* act_dev, float* obs_dev, float* rew_dev, uint8_t* term_dev, uint8_t* trunc_dev,
                 float* final_obs_dev, void* stream) {
    return launch_step(e, 1, act_dev, obs_dev, rew_dev, term_dev, trunc_dev, final_obs_dev, stream);
}

int tvc_env_step_many(tvc_env* e, int32_t n_steps, const float* act_dev, float* obs_dev, float* rew_dev, uint8_t* term_dev,
                      uint8_t* trunc_dev, void* stream) {
    return launch_step(e, n_steps, act_dev, obs_dev, rew_dev, term_dev, trunc_dev, nullptr, stream);
}

int tvc_env_export_state(tvc_env* e, float* dyn, int32_t* aux, float* pa, float* par, float* hist, void* stream) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    hipLaunchKernelGGL(env_export_kernel, dim3(tvc::ceil_div(e->n, 256)), dim3(256), 0, (hipStream_t)stream, e->buf, e->W,
                       dyn, aux, pa, par, hist);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_env_import_state(tvc_env* e, const float* dyn, const int32_t* aux, const float* pa, const float* par,
                         const float* hist, void* stream) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    hipLaunchKernelGGL(env_import_kernel, dim3(tvc::ceil_div(e->n, 256)), dim3(256), 0, (hipStream_t)stream, e->buf, e->W,
                       dyn, aux, pa, par, hist);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_env_set_components_out(tvc_env* e, float* comps_dev) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    if (comps_dev && (reinterpret_cast<uintptr_t>(comps_dev) & 15)) return tvc::set_error(TVC_EINVAL, "comps_dev must be 16-byte aligned");
    e->comps_out = comps_dev;
    return 0;
}

int tvc_env_set_episode_stats(tvc_env* e, float* ep_return_dev, double* sums_dev) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    if ((ep_return_dev == nullptr) != (sums_dev == nullptr)) return tvc::set_error(TVC_EINVAL, "pass both buffers or neither");
    e->ep.ret = ep_return_dev;
    e->ep.sums = sums_dev;
    return 0;
}

int tvc_env_info(tvc_env* e, float* info_dev, void* stream) {
    if (!e || !info_dev) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(e->device));
    hipLaunchKernelGGL(env_info_kernel, dim3(tvc::ceil_div(e->n, 256)), dim3(256), 0, (hipStream_t)stream, e->buf, e->dc,
                       info_dev);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// Philox4x32-10 of n (counter, key) pairs on the device: in[6n] = c0 c1 c2 c3 k0 k1, out[4n] (known-answer tests of the
// generator the reset / observation-noise / replay kernels draw from)
int tvc_debug_philox(const uint32_t* in_dev, uint32_t* out_dev, int32_t n, void* stream) {
    if (!in_dev || !out_dev || n < 1) return tvc::set_error(TVC_EINVAL, "bad argument");
    hipLaunchKernelGGL(philox_debug_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in_dev, out_dev, n);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// Where a launch's workgroups ran: out[2b] = XCD id (HW_REG_XCC_ID), out[2b + 1] = HW_REG_HW_ID (CU / SE bits) of block b.
// Used to find out which compute units a CU-masked stream (hipExtStreamCreateWithCUMask) really owns.
int tvc_debug_hwid(uint32_t* out_dev, int32_t n_blocks, void* stream) {
    if (!out_dev || n_blocks < 1) return tvc::set_error(TVC_EINVAL, "bad argument");
    hipLaunchKernelGGL(hwid_debug_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, out_dev);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_env_fuel_thresholds(const tvc_env* e, int32_t* k_empty, int32_t* k_coast, int32_t* k_low) {
    if (!e) return tvc::set_error(TVC_EINVAL, "env is NULL");
    if (k_empty) *k_empty = e->dc.k_empty;
    if (k_coast) *k_coast = e->dc.k_coast;
    if (k_low) *k_low = e->dc.k_low;
    return 0;
}

}  // extern "C"
