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

// Train-mode acting (TRAIN instantiation): the reference's get_action runs the policy with Dropout(0.1) live (no .eval() anywhere,
// agent/multi_algorithm_agent.py:765).  The masks are the counter hash of tvc_nn_kernels.h (drop_key / drop_factor: keyed by the
// acting-call counter, the site = 300 + op index of the training net, the handle's seed, row and column), so the per-layer path,
// this kernel and oracle/sac_torch.py: DropMasks agree element for element.
struct AsDrop {
    unsigned ctr, seed, thresh, rowmix;
    float scale;
    __device__ __forceinline__ unsigned key(int op) const {
        return drop_mix(ctr ^ ((300u + (unsigned)op) * 0x9E3779B9u) ^ seed);  // (z = 0: one parameter group)
    }
    // keep / drop factor of mask element c (= column / group) of this lane's row
    __device__ __forceinline__ float f(unsigned key, unsigned c) const {
        const unsigned x = drop_mix(key ^ rowmix ^ ((c >> 1) * 0x9E3779B1u));
        const unsigned b = (c & 1u) ? (x >> 16) : (x & 0xFFFFu);
        return b >= thresh ? scale : 0.0f;
    }
    // elementwise mask on tile t of a row (features 16 t + 4 q + r)
    __device__ __forceinline__ void tile(unsigned key, int t, int q, f32x4& v) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= f(key, (unsigned)(16 * t + 4 * q + r));
    }
};

// TRAIN = false: the deterministic net with the attention and the embedding folded (W_ov, W'): stream of rows_tables().
// TRAIN = true : the net as trained -- embedding, then per layer v_proj, head-granular attention-weight dropout, out_proj,
//                dropout1, FFN with its dropout, dropout2, and the two head dropouts: stream of rows_tables_train().
template <bool TRAIN>
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

    AsDrop dr{};
    const float* __restrict__ tvec = a.tvec;
    if (TRAIN) {
        dr.ctr = (unsigned)*a.drop_ctr; dr.seed = a.drop_seed; dr.thresh = a.drop_thresh; dr.scale = a.drop_scale;
        dr.rowmix = drop_mix((unsigned)row + 0x632BE5ABu);
    }
    // observation as the first B operand: x[m][k = 4 q + r], zero beyond obs_dim (clamped address, selected after the load)
    f32x4 xin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 4 * q + r;
        const float v = a.obs[(long)rowc * a.obs_dim + min(k, a.obs_dim - 1)];
        xin[r] = k < a.obs_dim ? v : 0.0f;
    }
    f32x4 x[16];
    int t0 = 0;  // index of the next tile of the packed stream (order: rows_tables() / rows_tables_train() in tvc_sac.hip)
    if (!TRAIN) {
        // ---- layer 0, first sublayer: embedding + PE(0) + folded attention + residual as ONE obs -> 256 Linear, then norm1
        f32x4 o[4];
        ar_zero<4>(o);
        as_pass<1, 4, false>(tiles + (long)t0 * AR_TILE_F4, lb + 1024u * wave, &xin, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += ar_vec4(vec, 4 * wave + j, q);
        as_allgather16(xb, o, x, wave, lane);
        ar_layernorm<16>(x, vec + 256, vec + 512, q);
        t0 += 2;  // (the stream carries an all-zero second tile behind W', tvc_actor_rows.h)
    } else {
        // ---- x = W_e obs + b_e + PE(0)   (agent/...:196-203)
        f32x4 o[4];
        ar_zero<4>(o);
        as_pass<1, 4, false>(tiles + (long)t0 * AR_TILE_F4, lb + 1024u * wave, &xin, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += ar_vec4(tvec, 4 * wave + j, q) + ar_vec4(a.pe0, 4 * wave + j, q);
        as_allgather16(xb, o, x, wave, lane);
        t0 += 1;
    }
    for (int l = 0; l < a.n_layers; ++l) {
        const float* lv = vec + l * AR_LAYER_VEC;
        if (TRAIN) {
            // self-attention at sequence length 1 = out_proj(dropout_heads(v_proj(x))): softmax over one key is 1, so the
            // attention-weight dropout zeroes / rescales whole heads of V (32 columns each); then dropout1, residual, norm1
            const float* tl = tvec + 256 + 512 * l;
            const unsigned kv = dr.key(1 + 6 * l), ko = dr.key(2 + 6 * l);
            f32x4 o[4];
            ar_zero<4>(o);
            as_pass<16, 4, false>(tiles + (long)t0 * AR_TILE_F4, lb + 1024u * wave, x, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hm = dr.f(kv, (unsigned)((4 * wave + j) >> 1));  // head = column / 32 = tile / 2
                o[j] = (o[j] + ar_vec4(tl, 4 * wave + j, q)) * hm;
            }
            f32x4 v[16];
            as_allgather16(xb, o, v, wave, lane);
            ar_zero<4>(o);
            as_pass<16, 4, false>(tiles + (long)(t0 + 16) * AR_TILE_F4, lb + 1024u * wave, v, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] += ar_vec4(tl + 256, 4 * wave + j, q);
                dr.tile(ko, 4 * wave + j, q, o[j]);
            }
            as_allgather16(xb, o, v, wave, lane);
#pragma unroll
            for (int t = 0; t < 16; ++t) x[t] += v[t];
            ar_layernorm<16>(x, lv + 256, lv + 512, q);
            t0 += 32;
        } else if (l > 0) {  // x = norm1(x + W_ov x + b_ov): output-split, then gathered
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
            if (TRAIN) dr.tile(dr.key(4 + 6 * l), 8 * wave + t, q, h[t]);  // the FFN's dropout (hidden units 128 w + ...)
        }
        f32x4 part[16];
        ar_zero<16>(part);
        as_pass<8, 16, false>(tiles + (long)(t0 + 16 * wave + 8) * AR_TILE_F4, lb, h, part);
        f32x4 y[16];
        as_reduce16(xb, part, y, wave, lane);
        const unsigned k2 = TRAIN ? dr.key(5 + 6 * l) : 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            y[t] += ar_vec4(lv + 1280, t, q);
            if (TRAIN) dr.tile(k2, t, q, y[t]);  // dropout2
            x[t] += y[t];
        }
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
            if (TRAIN) dr.tile(dr.key(3 + 6 * a.n_layers), 8 * wave + t, q, pp[t]);  // Dropout behind policy_head.2
        }
    }
    // ---- 512 -> 512 GELU LayerNorm -> 2A outputs, input-split over the wave's own 128 features, in two halves of 256 outputs; the
    // LayerNorm and the output Linear are folded into running sums (actor_rows_kernel): of each half's total, this wave finishes
    // the 64 outputs 64 w .. 64 w + 63
    // TRAIN: Dropout behind policy_head.6 sits between the folded LayerNorm and the output Linear,
    //   out[o] = sum_n m_n (rstd (g_n - mean) gamma_n + beta_n) W8[o, n] + b8[o]
    //          = rstd (sum_n m_n g_n gW[o][n] - mean sum_n m_n gW[o][n]) + sum_n m_n bW[o][n] + b8[o],
    // so the two "input-independent" sums of pack_head become per-row sums over the kept columns (gm, em below)
    float s1 = 0.0f, s2 = 0.0f, d[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {0.f, 0.f, 0.f, 0.f}, em[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned k6 = TRAIN ? dr.key(5 + 6 * a.n_layers) : 0u;
    const float* bw = TRAIN ? tvec + 256 + 512 * a.n_layers : nullptr;  // bW[o][n] = beta6[n] W8[o][n], then b8[4]
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
            f32x4 bwv[4];
            if (TRAIN) {
#pragma unroll
                for (int o = 0; o < 4; ++o) bwv[o] = ar_vec4(bw + 512 * o, tt, q);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = gelu_f(s[r] + b4[r]);
                s1 += v;
                s2 = fmaf(v, v, s2);
                const float m = TRAIN ? dr.f(k6, (unsigned)(16 * tt + 4 * q + r)) : 1.0f;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    d[o] = fmaf(v * m, gw[o][r], d[o]);
                    if (TRAIN) { gm[o] = fmaf(m, gw[o][r], gm[o]); em[o] = fmaf(m, bwv[o][r], em[o]); }
                }
            }
        }
    }
    float out[4];
    if (!TRAIN) {
        float fin[6] = {s1, s2, d[0], d[1], d[2], d[3]};
#pragma unroll
        for (int k = 0; k < 6; ++k) { fin[k] += __shfl_xor(fin[k], 16); fin[k] += __shfl_xor(fin[k], 32); }
        as_rowsum<6>(red, fin, wave, l15, q);
        const float mean = fin[0] * (1.0f / 512.0f);
        const float rstd = rsqrtf(fmaxf(fin[1] * (1.0f / 512.0f) - mean * mean, 0.0f) + 1e-5f);
#pragma unroll
        for (int o = 0; o < 4; ++o) out[o] = rstd * (fin[2 + o] - mean * tv[5636 + o]) + tv[5632 + o];
    } else {
        float fin[14] = {s1, s2, d[0], d[1], d[2], d[3], gm[0], gm[1], gm[2], gm[3], em[0], em[1], em[2], em[3]};
#pragma unroll
        for (int k = 0; k < 14; ++k) { fin[k] += __shfl_xor(fin[k], 16); fin[k] += __shfl_xor(fin[k], 32); }
        as_rowsum<14>(red, fin, wave, l15, q);
        const float mean = fin[0] * (1.0f / 512.0f);
        const float rstd = rsqrtf(fmaxf(fin[1] * (1.0f / 512.0f) - mean * mean, 0.0f) + 1e-5f);
#pragma unroll
        for (int o = 0; o < 4; ++o) out[o] = rstd * (fin[2 + o] - mean * fin[6 + o]) + fin[10 + o] + bw[2048 + o];
    }
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
