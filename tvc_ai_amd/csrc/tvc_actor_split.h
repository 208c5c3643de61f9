// Acting pass of the reference-shape policy as ONE launch for SMALL row counts (BASELINE's per-GPU shards: 4 096 - 8 192 envs):
// the "split" sibling of actor_rows_kernel (tvc_actor_rows.h), same network, same packed weight stream, same outputs.
//
// Replaces the policy part of MultiAlgorithmAgent.get_action (agent/multi_algorithm_agent.py:765-789, forward :192-227) for N rows.
//
// Why a second kernel: actor_rows_kernel gives a workgroup 64 rows (a wave owns 16 rows for the whole network), so 4 096 rows are
// 64 workgroups -- a quarter of the chip, 564 us -- and below 12 288 rows the per-layer kernels (316 us at 4 096 rows) were faster.
// Here a WORKGROUP owns 16 rows and its four waves split every Linear, so 4 096 rows are 256 workgroups = one per CU:
//   * Linears whose input every wave holds in full (folded attention 256 -> 256, FFN up 256 -> 512, head 256 -> 512) are split over
//     the OUTPUT features (wave w computes a quarter of them);
//   * the FFN's down projection and the head's 512 -> 512 are split over the INPUT features: the wave contracts over the 128
//     features it has just produced itself (they never leave its registers) and the four partial sums meet in LDS;
//   * activations keep the transposed-MFMA register chaining of actor_rows_kernel (the accumulator layout of one Linear is the
//     B-operand layout of the next: lane (m = l % 16, q = l / 16) holds features 16 t + 4 q + r of row m), so the exchanges through
//     LDS are plain 16-byte tile copies; LayerNorm runs redundantly in every wave on the complete row.
// Weights: every wave streams exactly the fragments it multiplies, straight from the packed tile stream (L2) into registers
// (global_load_dwordx4 issued AS_PF fragments ahead through inline asm) -- no LDS staging, no per-tile workgroup barrier; the only barriers are the ~3
// exchanges per encoder layer.  MFMA work per workgroup = 25.7 k v_mfma_f32_16x16x4_f32 over four SIMDs = 86 us at 2.4 GHz.
#pragma once
#include "tvc_actor_rows.h"

#ifndef AS_PF
#define AS_PF 8   // weight fragments (1 KB per wave each) in flight per wave
#endif

namespace tvcnn {

// four MFMAs of one weight fragment pair (two n-tiles) against one k-tile of the activation, c outermost so that consecutive
// MFMAs go to different accumulators (a dependent v_mfma_f32_16x16x4_f32 needs 40 cycles, an independent one issues after 32)
__device__ __forceinline__ void as_mfma_pair(const f32x4 w0, const f32x4 w1, const f32x4 xk, f32x4& a0, f32x4& a1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[c], xk[c], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[c], xk[c], a1, 0, 0, 0);
    }
}
// One weight fragment (this lane's 16 bytes of a 16 x 16 n-by-k block) from the packed stream: uniform base (SGPR pair) + this
// lane's byte offset (one VGPR) + an immediate.  Issued through inline asm, and waited for by as_wait below, because hipcc's
// scheduler otherwise sinks every load to just before its use (two fragments in flight, whatever the source says: the first build
// of this kernel ran at 0.3 of the MFMA rate) -- the fragment stream must run AS_PF fragments ahead of the MFMAs.
template <int IMM>
__device__ __forceinline__ f32x4 as_load(const float4* sbase, unsigned voff) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM));
    return v;
}
// wait until at most N of this wave's vector-memory operations are outstanding (they complete in order), and tie the two
// fragments about to be consumed to the wait so that their MFMAs cannot be scheduled above it
template <int N>
__device__ __forceinline__ void as_wait(f32x4& w0, f32x4& w1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w0), "+v"(w1) : "n"(N));
}
// One Linear (or one wave's share of it): KT k-tiles x NT n-tiles of weight fragments, fragment (kt, j) at float4 index
// kt * 1024 + j * 16 of `base` plus this lane's offset `voff` (bytes), multiplied into acc[j] against x[kt].  The fragment stream
// runs AS_PF fragments ahead of the MFMAs through a register ring.
// DEEP: k-tile kt lives in "deep" tile kt / 2, half kt % 2 (32 k x 128 n images, tvc_actor_rows.h PackTile.blocked == 2).
template <int KT, int NT, bool DEEP, int I>  // issue the load of fragment I (nothing beyond the last one)
__device__ __forceinline__ void as_issue(f32x4 (&buf)[AS_PF], const float4* __restrict__ base, unsigned voff) {
    if constexpr (I < KT * NT) {
        constexpr int kt = I / NT, j = I % NT;
        buf[I % AS_PF] = as_load<j * 256>(base + (DEEP ? (kt >> 1) * AR_TILE_F4 + (kt & 1) * 512 : kt * AR_TILE_F4), voff);
    }
}
template <int KT, int NT, bool DEEP, int I>  // consume fragments I, I + 1, keep the stream AS_PF ahead, recurse
__device__ __forceinline__ void as_steps(f32x4 (&buf)[AS_PF], const float4* __restrict__ base, unsigned voff, const f32x4* __restrict__ x,
                                         f32x4* __restrict__ acc) {
    if constexpr (I < KT * NT) {
        constexpr int TOT = KT * NT, PF = AS_PF;
        // loads issued so far: 0 .. min(TOT, I + PF) - 1; the ones younger than fragment I + 1 may stay outstanding
        constexpr int issued = I + PF < TOT ? I + PF : TOT;
        f32x4 w0 = buf[I % PF], w1 = buf[(I + 1) % PF];
        as_wait<issued - (I + 2)>(w0, w1);
        as_issue<KT, NT, DEEP, I + PF>(buf, base, voff);
        as_issue<KT, NT, DEEP, I + 1 + PF>(buf, base, voff);
        as_mfma_pair(w0, w1, x[I / NT], acc[I % NT], acc[(I + 1) % NT]);
        as_steps<KT, NT, DEEP, I + 2>(buf, base, voff, x, acc);
    }
}
template <int KT, int NT, bool DEEP, int I>
__device__ __forceinline__ void as_prologue(f32x4 (&buf)[AS_PF], const float4* __restrict__ base, unsigned voff) {
    if constexpr (I < AS_PF) {
        as_issue<KT, NT, DEEP, I>(buf, base, voff);
        as_prologue<KT, NT, DEEP, I + 1>(buf, base, voff);
    }
}
template <int KT, int NT, bool DEEP>
__device__ __forceinline__ void as_pass(const float4* __restrict__ base, unsigned voff, const f32x4* __restrict__ x, f32x4* __restrict__ acc) {
    static_assert((NT & 1) == 0 && (AS_PF & 1) == 0, "fragments are consumed in pairs");
    f32x4 buf[AS_PF];
    as_prologue<KT, NT, DEEP, 0>(buf, base, voff);
    as_steps<KT, NT, DEEP, 0>(buf, base, voff, x, acc);
}

constexpr int AS_XBUF_F4 = 4096;  // 64 KB of LDS: four waves x 16 partial-sum tiles of 1 KB, or one gathered row block

// every wave ends up with the complete 256-feature rows: wave w contributes tiles 4 w .. 4 w + 3 (`mine`), reads the other twelve
__device__ __forceinline__ void as_allgather16(float4* xb, const f32x4 (&mine)[4], f32x4 (&all)[16], int wave, int lane) {
    __syncthreads();  // every wave is done reading the buffer's previous content
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(xb + (4 * wave + j) * 64 + lane) = mine[j];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) all[t] = *reinterpret_cast<const f32x4*>(xb + t * 64 + lane);
}
// four partial sums of the complete 256-feature rows (input-split Linear): every wave ends up with their total
__device__ __forceinline__ void as_reduce16(float4* xb, const f32x4 (&part)[16], f32x4 (&sum)[16], int wave, int lane) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(xb + wave * 1024 + t * 64 + lane) = part[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        f32x4 s = *reinterpret_cast<const f32x4*>(xb + t * 64 + lane);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(xb + w * 1024 + t * 64 + lane);
        sum[t] = s;
    }
}
// per-row scalars summed over the four waves (v already summed over the row's four lanes): red[k][wave][row]
template <int NV>
__device__ __forceinline__ void as_rowsum(float* red, float (&v)[NV], int wave, int l15, int q) {
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[(k * 4 + wave) * 16 + l15] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = red[(k * 4 + 0) * 16 + l15] + red[(k * 4 + 1) * 16 + l15] + red[(k * 4 + 2) * 16 + l15] + red[(k * 4 + 3) * 16 + l15];
}

__global__ void __launch_bounds__(256, 2) actor_split_kernel(ActRowsArgs a) {
    __shared__ __attribute__((aligned(16))) float4 xb[AS_XBUF_F4];
    float* red = reinterpret_cast<float*>(xb);  // (the small per-row reductions reuse the head of the buffer, between barriers)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + l15;
    const int rowc = min(row, a.M - 1);
    const float4* __restrict__ tiles = a.tiles;
    const float* __restrict__ vec = a.vec;
    // this lane's byte offset inside a standard tile image[q][n] (+ 256 j for n-tile j, + 1024 w for the wave's quarter of the
    // outputs) and inside one half of a deep tile image[a][q][n < 128]
    const unsigned lb = (unsigned)(q * 256 + l15) * 16u, lbd = (unsigned)(q * 128 + l15) * 16u;

    // observation as the first B operand: x[m][k = 4 q + r], zero beyond obs_dim (clamped address, selected after the load)
    f32x4 xin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 4 * q + r;
        const float v = a.obs[(long)rowc * a.obs_dim + min(k, a.obs_dim - 1)];
        xin[r] = k < a.obs_dim ? v : 0.0f;
    }
    f32x4 x[16];
    int t0 = 0;  // index of the next tile of the packed stream (order: rows_tables() in tvc_sac.hip)
    // ---- layer 0, first sublayer: embedding + PE(0) + folded attention + residual as ONE obs -> 256 Linear, then norm1
    {
        f32x4 o[4];
        ar_zero<4>(o);
        as_pass<1, 4, false>(tiles + (long)t0 * AR_TILE_F4, lb + 1024u * wave, &xin, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += ar_vec4(vec, 4 * wave + j, q);
        as_allgather16(xb, o, x, wave, lane);
        ar_layernorm<16>(x, vec + 256, vec + 512, q);
        t0 += 2;  // (the stream carries an all-zero second tile behind W', tvc_actor_rows.h)
    }
    for (int l = 0; l < a.n_layers; ++l) {
        const float* lv = vec + l * AR_LAYER_VEC;
        if (l > 0) {  // x = norm1(x + W_ov x + b_ov): output-split, then gathered
            f32x4 o[4];
            ar_zero<4>(o);
            as_pass<16, 4, false>(tiles + (long)t0 * AR_TILE_F4, lb + 1024u * wave, x, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += ar_vec4(lv, 4 * wave + j, q);
            f32x4 g[16];
            as_allgather16(xb, o, g, wave, lane);
#pragma unroll
            for (int t = 0; t < 16; ++t) x[t] += g[t];
            ar_layernorm<16>(x, lv + 256, lv + 512, q);
            t0 += 16;
        }
        // x = norm2(x + W2 gelu(W1 x + b1) + b2): this wave's 128 hidden units (deep tiles of quarter `wave`) stay in its registers
        // and are the input slice of its share of W2 (tiles of quarter `wave`, all 256 outputs); the partial sums meet in LDS
        f32x4 h[8];
        ar_zero<8>(h);
        as_pass<16, 8, true>(tiles + (long)(t0 + 16 * wave) * AR_TILE_F4, lbd, x, h);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 b4 = ar_vec4(lv + 768 + 128 * wave, t, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) h[t][r] = gelu_f(h[t][r] + b4[r]);
        }
        f32x4 part[16];
        ar_zero<16>(part);
        as_pass<8, 16, false>(tiles + (long)(t0 + 16 * wave + 8) * AR_TILE_F4, lb, h, part);
        f32x4 y[16];
        as_reduce16(xb, part, y, wave, lane);
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] += y[t] + ar_vec4(lv + 1280, t, q);
        ar_layernorm<16>(x, lv + 1536, lv + 1792, q);
        t0 += 64;
    }
    const float* tv = vec + a.n_layers * AR_LAYER_VEC;
    ar_layernorm<16>(x, tv, tv + 256, q);  // feature_norm
    // ---- policy head: 256 -> 512 GELU LayerNorm, output-split: this wave's 128 features = n-tiles 8 (w % 2) .. + 7 of half w / 2
    f32x4 pp[8];
    ar_zero<8>(pp);
    as_pass<16, 8, false>(tiles + (long)(t0 + 16 * (wave >> 1)) * AR_TILE_F4, lb + 2048u * (wave & 1), x, pp);
    t0 += 32;
    {
        float st[1] = {0.0f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 b4 = ar_vec4(tv + 512 + 128 * wave, t, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) { pp[t][r] = gelu_f(pp[t][r] + b4[r]); st[0] += pp[t][r]; }
        }
        st[0] += __shfl_xor(st[0], 16); st[0] += __shfl_xor(st[0], 32);
        as_rowsum<1>(red, st, wave, l15, q);
        const float mean = st[0] * (1.0f / 512.0f);
        float sv[1] = {0.0f};
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = pp[t][r] - mean; sv[0] = fmaf(d, d, sv[0]); }
        sv[0] += __shfl_xor(sv[0], 16); sv[0] += __shfl_xor(sv[0], 32);
        as_rowsum<1>(red, sv, wave, l15, q);
        const float rstd = rsqrtf(sv[0] * (1.0f / 512.0f) + 1e-5f);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 g4 = ar_vec4(tv + 1024 + 128 * wave, t, q), b4 = ar_vec4(tv + 1536 + 128 * wave, t, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[t][r] = (pp[t][r] - mean) * rstd * g4[r] + b4[r];
        }
    }
    // ---- 512 -> 512 GELU LayerNorm -> 2A outputs, input-split over the wave's own 128 features, in two halves of 256 outputs; the
    // LayerNorm and the output Linear are folded into running sums (actor_rows_kernel): of each half's total, this wave finishes
    // the 64 outputs 64 w .. 64 w + 63
    float s1 = 0.0f, s2 = 0.0f, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        f32x4 part[16];
        ar_zero<16>(part);
        as_pass<8, 16, false>(tiles + (long)(t0 + 32 * half + 8 * wave) * AR_TILE_F4, lb, pp, part);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(xb + wave * 1024 + t * 64 + lane) = part[t];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 4 * wave + j, tt = 16 * half + t;
            f32x4 s = *reinterpret_cast<const f32x4*>(xb + t * 64 + lane);
#pragma unroll
            for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(xb + w * 1024 + t * 64 + lane);
            const f32x4 b4 = ar_vec4(tv + 2048, tt, q);
            f32x4 gw[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) gw[o] = ar_vec4(tv + 3584 + 512 * o, tt, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = gelu_f(s[r] + b4[r]);
                s1 += v;
                s2 = fmaf(v, v, s2);
#pragma unroll
                for (int o = 0; o < 4; ++o) d[o] = fmaf(v, gw[o][r], d[o]);
            }
        }
    }
    float fin[6] = {s1, s2, d[0], d[1], d[2], d[3]};
#pragma unroll
    for (int k = 0; k < 6; ++k) { fin[k] += __shfl_xor(fin[k], 16); fin[k] += __shfl_xor(fin[k], 32); }
    as_rowsum<6>(red, fin, wave, l15, q);
    const float mean = fin[0] * (1.0f / 512.0f);
    const float rstd = rsqrtf(fmaxf(fin[1] * (1.0f / 512.0f) - mean * mean, 0.0f) + 1e-5f);
    float out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) out[o] = rstd * (fin[2 + o] - mean * tv[5636 + o]) + tv[5632 + o];
    // mean, clamped log_std, action = mean + exp(log_std) eps  (agent/...:224-225, 780-782, 789)
    if (wave == 0 && q == 0 && row < a.M) {
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

}  // namespace tvcnn
