// Acting pass of the reference-shape policy as ONE launch: the "row-owner wave" kernel.
//
// Replaces, for N rows per call, the policy part of MultiAlgorithmAgent.get_action (agent/multi_algorithm_agent.py:765-789):
// TransformerPolicyNetwork.forward (:192-227) at sequence length 1 -- embedding + PE(0) + 4 x {attention (= one folded
// 256x256 Linear), +residual, LayerNorm, FFN 256 -> 512 GELU -> 256, +residual, LayerNorm}, feature LayerNorm,
// policy head 256 -> 512 -> 512 (GELU, LayerNorm) -> 2A -- followed by the Gaussian sample and clamp.
//
// Why one kernel: measured per layer, the 64x64 / 32-row LDS-tiled kernels take about (MFMA time + HBM time of the layer's
// activations): every layer reads and writes a [N, 256..512] fp32 activation (2.9 GB per pass at 65 536 rows) and the two phases
// do not overlap.  Here no activation ever leaves the CU.
//
// Design (gfx950): a wavefront owns 16 rows for the whole network.  Every Linear is computed TRANSPOSED,
//     out^T[n, m] = sum_k W[n, k] x[m, k],
// with v_mfma_f32_16x16x4_f32: A operand = a weight fragment (lane l: W[n = l % 16][k = l / 16]), B operand = the activation
// (lane l: x[m = l % 16][k = l / 16]).  The accumulator of that product holds out[m = l % 16][n = 16 t + 4 (l / 16) + r]
// (tile t, register r) -- which is exactly the B-operand layout of the NEXT Linear when its k-tile t / k-step r consumes register
// (t, r).  So activations chain from accumulator registers to operand registers with no shuffle, no LDS round trip and no
// cross-wave exchange; residual adds, GELU and LayerNorm are lane-local (+ two xor-shuffles per row statistic).
// Only weights move: all passes read one pre-packed stream of 16 KB tiles (image[q][n] = W[n0 + n][k0 + 4 q .. + 3], built by
// pack_actor_kernel after every policy update) through a two-buffer LDS ring filled by global_load_lds_dwordx4 one tile ahead,
// one s_barrier per tile; the four waves of a workgroup (64 rows) share each tile.  A tile is 16 conflict-free ds_read_b128 and
// 64 MFMAs per wave.
#pragma once
#include "tvc_nn_kernels.h"

namespace tvcnn {

constexpr int AR_TILE_F4 = 1024;    // float4 per weight tile: 4 k-quads x 256 output features
constexpr int AR_LAYER_VEC = 2048;  // floats per encoder layer in the vector section
// vector section, encoder layer l at l * AR_LAYER_VEC:
//   +0 attention bias b_ov (layer 0: b' of the folded embedding)   +256 norm1 gamma   +512 norm1 beta
//   +768 linear1 bias (512)   +1280 linear2 bias   +1536 norm2 gamma   +1792 norm2 beta
// tail at n_layers * AR_LAYER_VEC:
//   +0 feature_norm gamma  +256 beta  +512 head.0 bias (512)  +1024 head.2 gamma  +1536 head.2 beta  +2048 head.4 bias
//   +2560 head.6 gamma  +3072 head.6 beta  +3584 gamma6 x head.8 weight [4][512] (rows >= 2A zero)  +5632 E[4]  +5636 G[4]
//   (pack_head_kernel: the input-independent parts of the folded LayerNorm + output Linear)
constexpr int AR_TAIL_VEC = 5648;
// SqueezeExcitation block of the hierarchical low-level policy (use_se), behind the tail: +0 fc1 bias (16)  +16 fc2 bias (256)
constexpr int AR_SE_VEC = 272;

struct ActRowsArgs {
    const float* obs; const float* eps;          // [M, obs_dim], [M, A] or nullptr (deterministic)
    float* act; float* mean; float* logstd;      // [M, A]; mean / logstd may be nullptr
    const float4* tiles; const float* vec;       // packed weights
    int M, obs_dim, A, clamp_act, n_layers, n_tiles;
    int use_se;                  // SqueezeExcitation(256, reduction 16) between feature_norm and the head (agent/...:104-118, 206-209)
    unsigned long long* stamps;  // diagnostics (tvc_debug_rows_clock): per workgroup {s_memtime, s_memrealtime} at start and end, XCC_ID, HW_ID
    // train-mode acting (actor_split_kernel<true> only): second vector section (b_e, per layer b_v / b_o, beta6 x W8, b8), PE(0),
    // and the dropout hash parameters (tvc_nn_kernels.h: DropArgs)
    const float* tvec; const float* pe0; const int* drop_ctr; unsigned drop_thresh, drop_seed; float drop_scale;
};

#ifdef AR_TRACE
#define AR_STAMPS 96
#define AR_T() do { if (a.stamps && tid == 0 && tr < AR_STAMPS) a.stamps[(long)AR_STAMPS * blockIdx.x + tr++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AR_STAMPS 6
#define AR_T() do {} while (0)
#endif
struct ArPipe {
    const float4* tiles; float4* Bs;
    int ti, n_tiles;
    int wave;        // wave index in the workgroup, wave-uniform (readfirstlane): LDS destinations and M0 stay in SGPRs
    unsigned goff;   // this lane's byte offset inside a tile, (wave * 256 + lane) * 16: ONE 32-bit VGPR next to a scalar tile base
};
#ifndef AR_NW
#define AR_NW 4     // waves (16 rows each) per workgroup sharing one tile stream: 4 = 64 rows; 8 = 128 rows (experiment, profiles/r03_d_actor_rows.md)
#endif
#ifndef AR_NBUF
#define AR_NBUF 2   // LDS tile buffers: 2 = copy one tile ahead; 4 = copy two tiles ahead (experiment, profiles/r02_c_actor_rows.md)
#endif
__device__ __forceinline__ void ar_issue_tile(const ArPipe& p, int ti) {
    // 16 KB = 4 waves x 4 wave-instructions x 1 KB, lane-linear image: LDS byte i of the tile = global byte i
    // scalar tile base + 32-bit lane offset (the saddr form of the load): a per-lane 64-bit address would be one more live VGPR
    // pair in a kernel that sits at the register limit, and it was being spilled and reloaded once per tile
    const char* src = reinterpret_cast<const char*>(p.tiles) + ((size_t)__builtin_amdgcn_readfirstlane(ti) << 14) + p.goff;
    float4* dst = p.Bs + (ti & (AR_NBUF - 1)) * AR_TILE_F4 + p.wave * (1024 / AR_NW);
    // one address pair; the four 1 KB pieces go through the instruction's immediate offset, which advances both the global and
    // the LDS address (checked by tools/micro/mfma_lds.hip) -- separate pointers cost an M0 write + readfirstlane per piece (-3 %)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 1024, 0);
#if AR_NW == 4
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 3072, 0);
#endif
}
// Make tile p.ti readable, start the copy of tile p.ti + 1 into the buffer that tile p.ti - 1 just vacated, return the readable
// tile.  __syncthreads() here is: wait for my quarter of tile ti (vmcnt(0): the LDS-DMA is a pending LDS write) and for my
// fragment reads of tile ti - 1 (lgkmcnt(0)), then s_barrier -- everyone's quarter has landed and everyone is done with the
// other buffer.  (An inline-asm wait instead hides the counters from hipcc's waitcnt pass, which then answers every later
// fragment use with a full lgkmcnt(0); three buffers / two tiles ahead measured the same, profiles/r02_c_actor_rows.md.)
__device__ __forceinline__ const float4* ar_next(ArPipe& p) {
#if defined(AR_ABL) && AR_ABL == 1   // timing-only ablation: no workgroup barrier (waits kept)
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_sched_barrier(0);
    const float4* cur = p.Bs + (p.ti & 1) * AR_TILE_F4;
    if (p.ti + 1 < p.n_tiles) ar_issue_tile(p, p.ti + 1);
#elif AR_NBUF == 2
    __syncthreads();
    const float4* cur = p.Bs + (p.ti & 1) * AR_TILE_F4;
    if (p.ti + 1 < p.n_tiles) ar_issue_tile(p, p.ti + 1);
#else
    // two tiles ahead: tile ti + 2 goes to the buffer of tile ti - 2, whose fragment reads every wave consumed long ago (no LDS
    // counter wait needed); of the copies in flight only tile ti's must have landed, tile ti + 1's four may stay outstanding
    if (p.ti + 1 < p.n_tiles) __builtin_amdgcn_s_waitcnt(0x0F74);  // vmcnt(4), expcnt / lgkmcnt untouched
    else __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    const float4* cur = p.Bs + (p.ti & (AR_NBUF - 1)) * AR_TILE_F4;
    if (p.ti + 2 < p.n_tiles) ar_issue_tile(p, p.ti + 2);
#endif
    p.ti += 1;
    return cur;
}
// KT k-tiles of one Linear: acc[t] (t = 0..15: output features 16 t + 4 q + r of this lane's row) += W . x^T.
// The weight fragments come in sets of TWO n-tiles (two ds_read_b128 = 8 registers, 8 MFMAs: the two tiles x the four k-steps); the
// stream of sets is software-pipelined ACROSS tile boundaries with two register sets: the reads of set i + 1 are issued in the
// middle of the MFMAs of set i (4 MFMAs, 2 reads, 4 MFMAs by sched_group_barrier: hipcc answers a fragment's first use with a full
// lgkmcnt(0), so reads issued right before it would expose an LDS round trip), and the tile barrier sits between the loads of a
// tile's last set and that set's MFMAs, so the first reads of the next tile are in flight while the matrix pipe still works on the
// previous one.  (Sets of four n-tiles, 16 registers each: +2 % at one wave per SIMD, profiles/r02_c_actor_rows.md.)
__device__ __forceinline__ void ar_mfma8(const float4 (&w)[2], const f32x4 xk, f32x4* __restrict__ acc) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float wv = c == 0 ? w[j].x : (c == 1 ? w[j].y : (c == 2 ? w[j].z : w[j].w));
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xk[c], acc[j], 0, 0, 0);
        }
}
__device__ __forceinline__ void ar_frag2(float4 (&w)[2], const float4* __restrict__ base, int hg) {
#pragma unroll
    for (int j = 0; j < 2; ++j) w[j] = base[(2 * hg + j) * 16];
}
#define AR_SCHED_HALF()                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0)
template <int KT>
__device__ __forceinline__ void ar_pass(ArPipe& p, const f32x4* __restrict__ x, f32x4* __restrict__ acc, int l15, int q) {
    float4 wa[2], wb[2];
    const float4* base = ar_next(p) + q * 256 + l15;
    ar_frag2(wa, base, 0);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const f32x4 xk = x[kt];
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {  // half-group pairs (2 hp, 2 hp + 1)
            ar_frag2(wb, base, 2 * hp + 1);
            ar_mfma8(wa, xk, acc + 4 * hp);
            AR_SCHED_HALF();
            if (hp < 3) {
                ar_frag2(wa, base, 2 * hp + 2);
            } else if (kt + 1 < KT) {
                base = ar_next(p) + q * 256 + l15;
                ar_frag2(wa, base, 0);
            }
            ar_mfma8(wb, xk, acc + 4 * hp + 2);
            AR_SCHED_HALF();
        }
    }
}
// The same for a "deep" tile: 32 k x 128 outputs (two 8 KB halves, half a = k-steps 16 a .. 16 a + 15 of the 128 output rows), used
// where only 128 outputs are wanted at a time: acc[t] (t = 0..7) += W[128 rows] . x^T over KT2 tiles = 32 KT2 inputs; the tile's
// first half multiplies x[2 kt], its second half x[2 kt + 1], into the SAME eight accumulators.  64 MFMAs per tile as before.
template <int KT2>
__device__ __forceinline__ void ar_pass_q(ArPipe& p, const f32x4* __restrict__ x, f32x4* __restrict__ acc, int l15, int q) {
    float4 wa[2], wb[2];
    const float4* base = ar_next(p) + q * 128 + l15;
    auto frag = [&](float4 (&w)[2], const float4* b, int s) {  // set s of a tile: half s / 4, n-tiles 2 (s % 4), 2 (s % 4) + 1
#pragma unroll
        for (int j = 0; j < 2; ++j) w[j] = b[(s >> 2) * 512 + (2 * (s & 3) + j) * 16];
    };
    frag(wa, base, 0);
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt) {
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {  // sets 2 hp, 2 hp + 1
            frag(wb, base, 2 * hp + 1);
            ar_mfma8(wa, x[2 * kt + ((2 * hp) >> 2)], acc + 2 * ((2 * hp) & 3));
            AR_SCHED_HALF();
            if (hp < 3) {
                frag(wa, base, 2 * hp + 2);
            } else if (kt + 1 < KT2) {
                base = ar_next(p) + q * 128 + l15;
                frag(wa, base, 0);
            }
            ar_mfma8(wb, x[2 * kt + ((2 * hp + 1) >> 2)], acc + 2 * ((2 * hp + 1) & 3));
            AR_SCHED_HALF();
        }
    }
}
__device__ __forceinline__ f32x4 ar_vec4(const float* __restrict__ v, int t, int q) {
    return *reinterpret_cast<const f32x4*>(v + 16 * t + 4 * q);
}
template <int NT>
__device__ __forceinline__ void ar_zero(f32x4* u) {
#pragma unroll
    for (int t = 0; t < NT; ++t) u[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// nn.LayerNorm over the 16 * NT features of each row (eps 1e-5, two-pass like torch); a row lives in the 4 lanes l15 + 16 q
template <int NT>
__device__ __forceinline__ void ar_layernorm(f32x4* __restrict__ u, const float* __restrict__ gamma, const float* __restrict__ beta, int q) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += (u[t][0] + u[t][1]) + (u[t][2] + u[t][3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (16.0f * NT));
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = u[t][r] - mean; v = fmaf(d, d, v); }
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    const float rstd = rsqrtf(v * (1.0f / (16.0f * NT)) + 1e-5f);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 g4 = ar_vec4(gamma, t, q), b4 = ar_vec4(beta, t, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) u[t][r] = (u[t][r] - mean) * rstd * g4[r] + b4[r];
        // the 512-wide norm holds 128 registers of activation: without a fence every 8 tiles hipcc hoists all 64 vector loads
        // (256 more registers) and spills half the activation around them
        if (NT > 16 && (t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}


#ifndef AR_HEADPK
#define AR_HEADPK 23  // packed fp32 + v_rcp_f32 in the epilogues. bit 0: GELU of the 512-wide head activation (146 -> 54 spilled registers, 1 725 -> 1 692 us at 65 536 rows); bit 1: its LayerNorm (no change); bit 2: GELU of the FFN (1 655 us); bit 4: the 256-wide LayerNorms (with bits 0-2: NO spilled register left, 1 629 us = 132 TFLOP/s, the CU-sharing form 991 -> 909 us per 32 768 rows)
#endif
__device__ __forceinline__ f32x4 ar_splat(float c) { return (f32x4){c, c, c, c}; }
// GELU of four values with packed fp32 instructions and the rational erf's quotient as p x rcp(q) (see fast_erff)
__device__ __forceinline__ f32x4 ar_gelu4(const f32x4 x) {
    f32x4 z = x * 0.7071067811865476f;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = fminf(fmaxf(z[r], -4.0f), 4.0f);
    const f32x4 x2 = z * z;
    f32x4 p = ar_splat(-2.72614225801306e-10f);
    p = __builtin_elementwise_fma(p, x2, ar_splat(2.77068142495902e-08f));
    p = __builtin_elementwise_fma(p, x2, ar_splat(-2.10102402082508e-06f));
    p = __builtin_elementwise_fma(p, x2, ar_splat(-5.69250639462346e-05f));
    p = __builtin_elementwise_fma(p, x2, ar_splat(-7.34990630326855e-04f));
    p = __builtin_elementwise_fma(p, x2, ar_splat(-2.95459980854025e-03f));
    p = __builtin_elementwise_fma(p, x2, ar_splat(-1.60960333262415e-02f));
    f32x4 q = ar_splat(-1.45660718464996e-05f);
    q = __builtin_elementwise_fma(q, x2, ar_splat(-2.13374055278905e-04f));
    q = __builtin_elementwise_fma(q, x2, ar_splat(-1.68282697438203e-03f));
    q = __builtin_elementwise_fma(q, x2, ar_splat(-7.37332916720468e-03f));
    q = __builtin_elementwise_fma(q, x2, ar_splat(-1.42647390514189e-02f));
    f32x4 rq;
#pragma unroll
    for (int r = 0; r < 4; ++r) rq[r] = __builtin_amdgcn_rcpf(q[r]);
    const f32x4 e = z * p * rq;
    const f32x4 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
}
template <int NT>
__device__ __forceinline__ void ar_layernorm_pk(f32x4* __restrict__ u, const float* __restrict__ gamma, const float* __restrict__ beta, int q) {
    f32x4 s4 = u[0];
#pragma unroll
    for (int t = 1; t < NT; ++t) s4 += u[t];
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (16.0f * NT));
    const f32x4 m4 = ar_splat(mean);
    f32x4 v4 = ar_splat(0.0f);
#pragma unroll
    for (int t = 0; t < NT; ++t) { const f32x4 d = u[t] - m4; v4 = __builtin_elementwise_fma(d, d, v4); }
    float v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    const float rstd = rsqrtf(v * (1.0f / (16.0f * NT)) + 1e-5f);
    const f32x4 r4 = ar_splat(rstd);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 g4 = ar_vec4(gamma, t, q), b4 = ar_vec4(beta, t, q);
        u[t] = __builtin_elementwise_fma((u[t] - m4) * r4, g4, b4);
        if (NT > 16 && (t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

#if AR_HEADPK & 16
#define AR_LN16 ar_layernorm_pk<16>
#else
#define AR_LN16 ar_layernorm<16>
#endif
#if AR_NW == 4
__global__ void __launch_bounds__(256, 2) actor_rows_kernel(ActRowsArgs a) {
#else
__global__ void __launch_bounds__(64 * AR_NW) actor_rows_kernel(ActRowsArgs a) {
#endif
    __shared__ __attribute__((aligned(16))) float4 Bs[AR_NBUF * AR_TILE_F4];  // the ONLY LDS object: the 16 KB weight-tile buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * (16 * AR_NW) + wave * 16 + l15;
    const int rowc = min(row, a.M - 1);
    if (a.stamps && tid == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        a.stamps[AR_STAMPS * blockIdx.x] = __builtin_amdgcn_s_memtime();
        a.stamps[AR_STAMPS * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        a.stamps[AR_STAMPS * blockIdx.x + 4] = xcc;
        a.stamps[AR_STAMPS * blockIdx.x + 5] = hw;
    }
    int tr = 6; (void)tr;  // AR_TRACE: stamps 6.. = s_memtime after every pass / epilogue of wave 0
    ArPipe p{a.tiles, Bs, 0, a.n_tiles, __builtin_amdgcn_readfirstlane(wave), (unsigned)((wave * (1024 / AR_NW) + lane) * 16)};
    ar_issue_tile(p, 0);
#if AR_NBUF > 2
    ar_issue_tile(p, 1);
#endif

    // observation as the first B operand: x[m][k = 4 q + r], zero beyond obs_dim (clamped address, selected after the load)
    f32x4 xin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 4 * q + r;
        const float v = a.obs[(long)rowc * a.obs_dim + min(k, a.obs_dim - 1)];
        xin[r] = k < a.obs_dim ? v : 0.0f;
    }
    const float* vec = a.vec;
    f32x4 x[16];
    // ---- layer 0, first sublayer: embedding + PE(0) + folded attention + residual as ONE obs -> 256 Linear, then norm1
    {   // (the stream carries an all-zero second tile behind W': every pass is an even number of tiles, so the fragment
        //  register sets and the two LDS buffers are in the same phase at every pass boundary)
        ar_zero<16>(x);
        f32x4 xin2[2] = {xin, xin};
        ar_pass<2>(p, xin2, x, l15, q); AR_T();
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] += ar_vec4(vec, t, q);
        AR_LN16(x, vec + 256, vec + 512, q); AR_T();
    }
    for (int l = 0; l < a.n_layers; ++l) {
        const float* lv = vec + l * AR_LAYER_VEC;
        if (l > 0) {  // x = norm1(x + W_ov x + b_ov)
            f32x4 acc[16];
            ar_zero<16>(acc);
            ar_pass<16>(p, x, acc, l15, q); AR_T();
#pragma unroll
            for (int t = 0; t < 16; ++t) x[t] += acc[t] + ar_vec4(lv, t, q);
            AR_LN16(x, lv + 256, lv + 512, q); AR_T();
        }
        // x = norm2(x + W2 gelu(W1 x + b1) + b2), the 512 hidden units in four quarters of 128: a quarter of the hidden activation is
        // 32 registers per lane, so x (64) + the output accumulators (64) + the quarter + the fragment sets stay clear of the
        // 256-VGPR budget -- with halves the kernel spilled ~180 dwords per lane, and every spill is a round trip to the
        // Infinity Cache (scratch is 190 MB at 65 536 rows: 0.5 GB of write-backs and as much reloaded per launch)
        f32x4 acc2[16];
        ar_zero<16>(acc2);
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            f32x4 h[8];
            ar_zero<8>(h);
            ar_pass_q<8>(p, x, h, l15, q); AR_T();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 b4 = ar_vec4(lv + 768 + 128 * quarter, t, q);
#if AR_HEADPK & 4
                h[t] = ar_gelu4(h[t] + b4);
#else
#pragma unroll
                for (int r = 0; r < 4; ++r) h[t][r] = gelu_f(h[t][r] + b4[r]);
#endif
            }
            AR_T();
            ar_pass<8>(p, h, acc2, l15, q); AR_T();
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] += acc2[t] + ar_vec4(lv + 1280, t, q);
        AR_LN16(x, lv + 1536, lv + 1792, q); AR_T();
    }
    const float* tv = vec + a.n_layers * AR_LAYER_VEC;
    AR_LN16(x, tv, tv + 256, q); AR_T();  // feature_norm
    if (a.use_se) {  // x *= sigmoid(fc2(relu(fc1(x)))): pooling a [B, C, 1] tensor over its last axis is the identity
        const float* sv = tv + AR_TAIL_VEC;
        // fc1 (256 -> 16) as ONE tile: n-block kt of the image holds the 16 output rows' weights of k-tile kt, so the 64 MFMAs
        // of the tile accumulate into a single 16 x 16 accumulator (out[m][4 q + r]); an all-zero tile keeps the ring's parity
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const float4* base = ar_next(p) + q * 256 + l15;
#pragma unroll
            for (int kt = 0; kt < 16; ++kt) {
                const float4 w = base[kt * 16];
                s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x[kt][0], s4, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x[kt][1], s4, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x[kt][2], s4, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x[kt][3], s4, 0, 0, 0);
            }
            (void)ar_next(p);
        }
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sv + 4 * q);
        f32x4 yy[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) yy[0][r] = fmaxf(s4[r] + b1[r], 0.0f);
        yy[1] = yy[0];
        f32x4 g[16];
        ar_zero<16>(g);
        ar_pass<2>(p, yy, g, l15, q);  // fc2 (16 -> 256): one zero-padded 16-deep tile + an all-zero one, like the embedding
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 b2 = ar_vec4(sv + 16, t, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[t][r] *= 1.0f / (1.0f + __expf(-(g[t][r] + b2[r])));
        }
    }
    // ---- policy head: 256 -> 512 GELU LayerNorm
    f32x4 pp[32];
    ar_zero<32>(pp);
    ar_pass<16>(p, x, pp, l15, q); AR_T();
    ar_pass<16>(p, x, pp + 16, l15, q); AR_T();
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const f32x4 b4 = ar_vec4(tv + 512, t, q);
#if AR_HEADPK & 1
        pp[t] = ar_gelu4(pp[t] + b4);
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[t][r] = gelu_f(pp[t][r] + b4[r]);
#endif
        if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
#if AR_HEADPK & 2
    ar_layernorm_pk<32>(pp, tv + 1024, tv + 1536, q); AR_T();
#else
    ar_layernorm<32>(pp, tv + 1024, tv + 1536, q); AR_T();
#endif
    // ---- 512 -> 512 GELU LayerNorm -> 2A outputs.  The LayerNorm and the output Linear are folded into running sums:
    //   out[o] = rstd (sum_n g_n gamma_n W[o,n] - mean sum_n gamma_n W[o,n]) + sum_n beta_n W[o,n] + b[o],  g = gelu(.)
    // (gamma_n W[o,n] and the two input-independent sums come ready-made from pack_head_kernel)
    // so the second 512-wide activation is never held (one-pass variance E[g^2] - mean^2 on O(1) values)
    float s1 = 0.0f, s2 = 0.0f, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 a2[16];
        ar_zero<16>(a2);
        ar_pass<32>(p, pp, a2, l15, q); AR_T();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int tt = 16 * half + t;
            const f32x4 b4 = ar_vec4(tv + 2048, tt, q);
            f32x4 gw[4];  // gamma6[n] W8[o][n], made by pack_head_kernel
#pragma unroll
            for (int o = 0; o < 4; ++o) gw[o] = ar_vec4(tv + 3584 + 512 * o, tt, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = gelu_f(a2[t][r] + b4[r]);  // (IEEE division here: v_rcp_f32 in THIS epilogue tips the allocation over, 569 spills)
                s1 += v;
                s2 = fmaf(v, v, s2);
#pragma unroll
                for (int o = 0; o < 4; ++o) d[o] = fmaf(v, gw[o][r], d[o]);
            }
        }
    }
#define AR_RED(v) v += __shfl_xor(v, 16); v += __shfl_xor(v, 32)
    AR_RED(s1); AR_RED(s2);
#pragma unroll
    for (int o = 0; o < 4; ++o) { AR_RED(d[o]); }
#undef AR_RED
    const float mean = s1 * (1.0f / 512.0f);
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / 512.0f) - mean * mean, 0.0f) + 1e-5f);
    float out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) out[o] = rstd * (d[o] - mean * tv[5636 + o]) + tv[5632 + o];
    if (a.stamps && tid == 0) {
        a.stamps[AR_STAMPS * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        a.stamps[AR_STAMPS * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
    // mean, clamped log_std, action = mean + exp(log_std) eps  (agent/...:224-225, 780-782, 789)
    if (q == 0 && row < a.M) {
        for (int j = 0; j < a.A; ++j) {
            const float mu = out[j];
            const float ls = fminf(fmaxf(out[a.A + j], -20.0f), 2.0f);
            const long i = (long)row * a.A + j;
            float av = a.eps ? mu + expf(ls) * a.eps[i] : mu;
            if (a.clamp_act) av = fminf(fmaxf(av, -1.0f), 1.0f);
            a.act[i] = av;
            if (a.mean) a.mean[i] = mu;
            if (a.logstd) a.logstd[i] = ls;
        }
    }
}

// ------------------------------------------------------------------ weight packing
struct PackTile { long src; int ld, k0, kvalid, from_ov, blocked; };   // src: float offset of W[n0][0] in the parameter / derived buffer;
// blocked = 1: a 16-row matrix, n-block kt of the image = its 16 rows at k-tile kt (image[q][16 kt + j] = W[j][16 kt + 4 q ..]);
// blocked = 2: a deep tile, 32 k x 128 rows: image[a][q][n] = W[n][k0 + 16 a + 4 q ..] for n < 128
struct PackVec { long src; int dst, count, from_ov; };
// The output head behind the second LayerNorm of the policy head is folded into running sums (actor_rows_body): the parts of that
// fold that do not depend on the input are made by the LAST workgroup of pack_actor_kernel, once per policy update, into the
// vector tail:
//   +3584  gW[o][n] = gamma6[n] W8[o][n]   (o < 2A, else 0)      +5632  E[o] = sum_n beta6[n] W8[o][n] + b8[o]      +5636  G[o] = sum_n gW[o][n]
struct HeadPack { const float* gamma6; const float* beta6; const float* W8; const float* b8; int n_out; float* tail;
                  float* tail_t; };  // tail_t (train-mode stream, may be null): bW[o][n] = beta6[n] W8[o][n] at +512 o, b8[o] at +2048
struct PackSet { const PackTile* tiles; int n_tiles; const PackVec* vecs; int n_vecs; float4* out_tiles; float* out_vec; };
__device__ __forceinline__ void pack_head(const HeadPack& hp, float (*red)[4][256]) {
    const int tid = threadIdx.x;
    for (int o = 0; o < 4; ++o) {
        float g = 0.0f, e = 0.0f;
        for (int n = tid; n < 512; n += 256) {
            const float w = o < hp.n_out ? hp.W8[o * 512 + n] : 0.0f;
            const float gw = hp.gamma6[n] * w;
            hp.tail[3584 + 512 * o + n] = gw;
            if (hp.tail_t) hp.tail_t[512 * o + n] = hp.beta6[n] * w;
            g += gw;
            e = fmaf(hp.beta6[n], w, e);
        }
        red[0][o][tid] = g; red[1][o][tid] = e;
    }
    __syncthreads();
    if (tid < 8) {
        const int o = tid & 3, which = tid >> 2;
        float s = 0.0f;
        for (int i = 0; i < 256; ++i) s += red[which][o][i];
        if (which == 0) hp.tail[5636 + o] = s;
        else hp.tail[5632 + o] = s + (o < hp.n_out ? hp.b8[o] : 0.0f);
        if (which == 1 && hp.tail_t) hp.tail_t[2048 + o] = o < hp.n_out ? hp.b8[o] : 0.0f;
    }
}
// ------------------------------------------------------------------ packing of the split-operand stream (tvc_actor_x3.h)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int X3_TRI = 8;                    // split-operand stream (tvc_actor_x3.h): triples per tile
constexpr int X3_TILE_BYTES = X3_TRI * 3072;  // 24 KB
// One descriptor per 24 KB tile: triples g0 .. g0 + 7 of a pass over NT n-tiles (triple g: k-block g / NT, n-tile g % NT) of the
// matrix at `src` (float offset of W[n0][0], row stride ld); input feature of k-slot (kb, q, c) = k0 + 32 kb + 16 (c / 4) + 4 q + (c % 4),
// zero at and beyond kvalid; rows at and beyond nvalid (relative to n0) are zero (the 16-row SE matrix).  kvalid = 0: an all-zero tile.
struct PackTile3 { long src; int ld, k0, kvalid, nvalid, g0, NT, from_ov; };
__device__ __forceinline__ unsigned x3_rne_bits(float x) {  // the bf16 nearest to x, in the upper half of the word
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u & 0xFFFF0000u;
}
__device__ __forceinline__ void pack_x3_tile(const PackTile3& t, const float* __restrict__ P, const float* __restrict__ OV,
                                             u32x4_t* __restrict__ out) {
    const float* base = (t.from_ov ? OV : P) + t.src;
    for (int it = threadIdx.x; it < X3_TRI * 64; it += 256) {  // (triple j, lane): 8 weights -> three 16-byte fragments pieces
        const int j = it >> 6, lane = it & 63, n = lane & 15, q = lane >> 4;
        const int g = t.g0 + j, kb = g / t.NT, nt = g % t.NT;
        const int nrow = 16 * nt + n;
        const float* r = base + (long)nrow * t.ld;
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = t.k0 + 32 * kb + 16 * (c >> 2) + 4 * q + (c & 3);
            float w = 0.0f;
            if (k < t.kvalid && nrow < t.nvalid) w = r[k];
            h[c] = x3_rne_bits(w);
            const float r1 = w - __uint_as_float(h[c]);
            m[c] = x3_rne_bits(r1);
            const float r2 = r1 - __uint_as_float(m[c]);
            l[c] = x3_rne_bits(r2);
        }
        u32x4_t H, M, L;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            H[pp] = (h[2 * pp] >> 16) | h[2 * pp + 1];
            M[pp] = (m[2 * pp] >> 16) | m[2 * pp + 1];
            L[pp] = (l[2 * pp] >> 16) | l[2 * pp + 1];
        }
        out[(3 * j + 0) * 64 + lane] = H;
        out[(3 * j + 1) * 64 + lane] = M;
        out[(3 * j + 2) * 64 + lane] = L;
    }
}
struct PackSet3 { const PackTile3* tiles; int n_tiles; char* out; };
// up to three streams per launch: the acting net (attention and embedding folded), when train-mode acting is live the net as trained,
// and when split-operand acting is live the folded net again as bf16 triples
__global__ void __launch_bounds__(256) pack_actor_kernel(const float* __restrict__ P, const float* __restrict__ OV, PackSet s0, PackSet s1,
                                                         HeadPack hp, Ticks tk, PackSet3 s3, PackSet3 s4) {
    __shared__ float red[2][4][256];
    int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int n0 = s0.n_tiles + s0.n_vecs, n1 = s1.n_tiles + s1.n_vecs;
    if (b > n0 + n1) {  // behind the head's workgroup: the split-operand stream (when it is live), one 24 KB tile per workgroup
        int i = b - (n0 + n1 + 1);
        const bool fourth = i >= s3.n_tiles;   // ... and of the net as trained (train-mode acting with the split-operand kernel)
        if (fourth) i -= s3.n_tiles;
        const PackSet3& sx = fourth ? s4 : s3;
        pack_x3_tile(sx.tiles[i], P, OV, reinterpret_cast<u32x4_t*>(sx.out + (size_t)i * X3_TILE_BYTES));
        return;
    }
    if (b == n0 + n1) {  // last workgroup: the folded output head (+ the riders of this launch)
        if (tid == 0) run_ticks(tk);
        pack_head(hp, red);
        return;
    }
    const bool second = b >= n0;
    if (second) b -= n0;
    const PackTile* __restrict__ tiles = second ? s1.tiles : s0.tiles;
    const PackVec* __restrict__ vecs = second ? s1.vecs : s0.vecs;
    const int n_tiles = second ? s1.n_tiles : s0.n_tiles;
    float4* __restrict__ out_tiles = second ? s1.out_tiles : s0.out_tiles;
    float* __restrict__ out_vec = second ? s1.out_vec : s0.out_vec;
    if (b < n_tiles) {
        const PackTile t = tiles[b];
        const float* base = (t.from_ov ? OV : P) + t.src;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 256 + tid, qq = idx >> 8, n = idx & 255;
            int k = t.k0 + 4 * qq, nrow = n;
            if (t.blocked == 1) { k = 16 * (n >> 4) + 4 * qq; nrow = n & 15; }
            if (t.blocked == 2) {  // deep tile: idx = a * 512 + q * 128 + n, 128 output rows, k0 + 16 a + 4 q
                const int a = idx >> 9, rem = idx & 511;
                k = t.k0 + 16 * a + 4 * (rem >> 7); nrow = rem & 127;
            }
            const float* r = base + (long)nrow * t.ld;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float w = r[min(k + c, t.kvalid - 1)];
                v[c] = (k + c) < t.kvalid ? w : 0.0f;
            }
            out_tiles[(long)b * AR_TILE_F4 + idx] = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
        const PackVec e = vecs[b - n_tiles];
        const float* src = (e.from_ov ? OV : P) + e.src;
        for (int i = tid; i < e.count; i += 256) out_vec[e.dst + i] = src[i];
    }
}

}  // namespace tvcnn
