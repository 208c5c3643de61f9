// Device kernels of the SAC learner (fp32, MFMA f32 16x16x4 for every GEMM-shaped contraction).
//
// They replace the PyTorch eager op sequences of the reference's agent/multi_algorithm_agent.py:
//   nn.Linear (+bias, GELU / ReLU, residual)            -> gemm_kernel (64x64 tiles, fused epilogues; acting, M = envs)
//   nn.Linear + nn.LayerNorm (+ output head)            -> gemm_rowln_kernel (32 complete rows per workgroup; acting)
//   nn.Linear at batch size, forward / dX / dW          -> gemm_skinny_kernel, gemm_skinny_bwd_kernel (split-K, no LDS staging)
//   nn.Linear with <= 16 inputs (obs / [s | a] layers)  -> thin_fwd / thin_fwd_ln / thin_wgrad / thin_dgrad_kernel (plain FMA)
//   nn.LayerNorm forward / backward                      -> layernorm_fwd_kernel / layernorm_bwd_kernel
//   nn.Dropout of the train-mode update                  -> DropArgs masks inside the epilogues above
//   Linear(hidden -> 2A) / Linear(hidden -> 1) heads     -> head_fwd_kernel / head_bwd_kernel (dot-reductions, no MFMA)
//   torch.optim.Adam.step, Polyak soft update            -> adam_dev_kernel (tvc_sac.hip) / polyak_kernel
// Numerics: v_mfma_f32_16x16x4_f32 is an exact-f32 k-ordered fma chain (no TF32 on gfx950), so results
// match torch fp32 to summation-order rounding.
#pragma once
#include <hip/hip_runtime.h>

// The update's ~100 short kernels run BESIDE the acting kernel (one acting wave per SIMD issuing MFMAs back to back): raised wave
// priority lets their few instructions issue ahead of that stream instead of alternating with it (tools/step_events.py).
#ifndef TVC_LEARNER_PRIORITY
#define TVC_LEARNER_PRIORITY 3
#endif
#define TVC_LEARNER_PRIO() __builtin_amdgcn_s_setprio(TVC_LEARNER_PRIORITY)

namespace tvcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };  // sigmoid: forward only (SE gate)

// erf as one branch-free rational x P(x^2) / Q(x^2) on [-4, 4] (the float kernel of Eigen / XLA): max abs error 4.5e-7,
// i.e. GELU within 1e-6 of the erff-based one, at ~20 VALU instructions instead of a three-range libm routine whose
// divergent ranges all execute -- it matters in the epilogues that apply it to 256 values per lane
__device__ __forceinline__ float fast_erff(float x) {
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return x * p / q;  // (v_rcp_f32 instead of the IEEE division: the acting kernel's register allocation tips over, +14 %)
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + fast_erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + fast_erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float act_f(float x, int act) {
    return act == ACT_GELU ? gelu_f(x) : (act == ACT_RELU ? fmaxf(x, 0.0f) : (act == ACT_SIGMOID ? 1.0f / (1.0f + expf(-x)) : x));
}
__device__ __forceinline__ float act_grad(float z, int act) {
    return act == ACT_GELU ? gelu_grad(z) : (act == ACT_RELU ? (z > 0.0f ? 1.0f : 0.0f) : 1.0f);
}

// ------------------------------------------------------------------ kernel arguments: one miss, not five
// A kernel starts with a cold scalar cache, and hipcc fetches by-value kernel arguments lazily, basic block by basic block: the
// split-K GEMM's 420-byte argument struct arrived in FIVE dependent s_load rounds (~900 cycles each, tools/gemm_stamps.py) before
// its first operand load could even be addressed.  This touches every 64-byte line of the argument segment at once at kernel
// entry (one dword per line, results unused): the compiler's own loads then hit in the scalar cache.
#define TVC_KA4(base) "s_load_dword %0, %4, " #base "+0x0\n\ts_load_dword %1, %4, " #base "+0x40\n\t" \
                      "s_load_dword %2, %4, " #base "+0x80\n\ts_load_dword %3, %4, " #base "+0xc0\n\t"
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LINES = (BYTES + 63) / 64 + 1;  // + the first line of the implicit arguments (grid size) behind the explicit ones
    static_assert(LINES <= 16, "argument struct larger than expected");
    unsigned d0, d1, d2, d3;
    // ONE asm statement (loads + the wait): the destinations are scratch, reused by every group, and must not be handed to other
    // values while a load is still in flight
    if (LINES > 12)
        asm volatile(TVC_KA4(0x300) TVC_KA4(0x200) TVC_KA4(0x100) TVC_KA4(0x0) "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka));
    else if (LINES > 8)
        asm volatile(TVC_KA4(0x200) TVC_KA4(0x100) TVC_KA4(0x0) "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka));
    else if (LINES > 4)
        asm volatile(TVC_KA4(0x100) TVC_KA4(0x0) "s_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka));
    else
        asm volatile(TVC_KA4(0x0) "s_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka));
}

// ------------------------------------------------------------------ device-resident clocks
// The Adam step counter t and the running powers b1^t, b2^t (double) live on the device, so a captured update keeps counting.
// adam_dev_kernel only READS the clock (every workgroup derives the step's bias corrections from it); the clock advances in a
// single thread of a LATER launch of the same chain (the sampling kernel of the actor phase for the critics' clock, the weight
// packing for the actor's), so an optimiser step costs one launch, not two.
struct AdamClock { double b1t, b2t; int step; int pad; };
__device__ __forceinline__ void adam_clock_advance(AdamClock* clk, float b1, float b2) {
    clk->b1t = clk->b1t * (double)b1; clk->b2t = clk->b2t * (double)b2; clk->step += 1;
}
struct Ticks { AdamClock* clk; float b1, b2; int* ctr; };  // optional riders of a launch: advance an Adam clock / a call counter
__device__ __forceinline__ void run_ticks(const Ticks& tk) {
    if (tk.clk) adam_clock_advance(tk.clk, tk.b1, tk.b2);
    if (tk.ctr) *tk.ctr += 1;
}

// ------------------------------------------------------------------ dropout (update path only)
// nn.Dropout(p) of the reference's train-mode nets (agent/multi_algorithm_agent.py:141,159,163,600,604 and the encoder
// layers' dropout / dropout1 / dropout2 / attention-weight dropout).  torch's Philox stream cannot be reproduced, so the
// masks are a hash of (update counter, site, group, row, column): statistically equivalent, regenerated in the backward
// instead of stored.  p = thresh / 65536, kept values are scaled by 1 / (1 - p).
struct DropArgs {
    const int* ctr;   // device-resident update counter (the actor's Adam step count); nullptr = no dropout
    unsigned site;    // identifies the masked tensor and the forward call it belongs to
    unsigned thresh;  // an element is dropped when its 16 hash bits are < thresh
    float scale;
    int group;        // columns sharing one mask element: 1 = elementwise, d_model / nhead = one per attention head
    unsigned seed;    // per-handle seed (tvc_sac_cfg.dropout_seed): different learners / ranks draw different mask sequences
    unsigned zsplit;  // > 0: two forward calls batched in ONE launch -- groups z >= zsplit belong to the call whose site is site2 and are
    unsigned site2;   //      its groups z - zsplit (the critics' loss: online nets = groups 0, 1, target nets = groups 2, 3); 0 = off
};
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_key(const DropArgs& d, unsigned z) {
    const bool second = d.zsplit != 0u && z >= d.zsplit;
    const unsigned site = second ? d.site2 : d.site, zz = second ? z - d.zsplit : z;
    return drop_mix((unsigned)(*d.ctr) ^ (site * 0x9E3779B9u) ^ (zz * 0x7F4A7C15u) ^ d.seed);
}
__device__ __forceinline__ float drop_factor(const DropArgs& d, unsigned key, int row, int col) {
    const unsigned c = (unsigned)col / (unsigned)d.group;
    const unsigned x = drop_mix(key ^ drop_mix((unsigned)row + 0x632BE5ABu) ^ ((c >> 1) * 0x9E3779B1u));
    const unsigned b = (c & 1u) ? (x >> 16) : (x & 0xFFFFu);
    return b >= d.thresh ? d.scale : 0.0f;
}

// LayerNorm folded into the A-operand load of the Linear that consumes it (update path, gemm_skinny_ln_kernel): the GEMM reads the
// norm's INPUT rows, computes their statistics in registers (exact two-pass, like layernorm_fwd_kernel), normalises, applies the
// norm's output dropout, and multiplies.  The workgroups of column tile 0 also write what the backward pass and later readers
// of the norm's output need: mean / rstd per row and the materialised output Y (after dropout).  gamma2 != nullptr: two norms
// back to back (norm2 -> feature_norm of the policy): Y1 / mean / rstd belong to the first, Y / mean2 / rstd2 to the second.
struct LnA {
    const float* gamma; const float* beta; const float* gamma2; const float* beta2;  // [K], group stride gP
    float* mean; float* rstd; float* mean2; float* rstd2;                            // [M] or nullptr, group stride gS
    float* Y; float* Y1;                                                             // [M, K] or nullptr, group stride gY
    long gP, gS, gY;
};

// ------------------------------------------------------------------ GEMM  C[M,N] = epi(sum_k A(m,k) * B(n,k))
struct GemmArgs {
    const float* A;   // A_KC: A[m*lda + k]   else A[k*lda + m]
    const float* A2;  // optional second source for k >= K1 (concatenated input [s | a]); A_KC only
    const float* B;   // B_KC: B[n*ldb + k]   else B[k*ldb + n]
    float* C;         // C[m*ldc + n]
    int M, N, K, K1;
    long lda, lda2, ldb, ldc;
    const float* bias;      // [N]
    const float* rowtab;    // [rowtab_rows, N], row (m mod rows) added (positional-encoding table)
    int rowtab_rows;
    float* Zout;            // pre-activation copy (for backward), ldc
    int act;                // applied after bias/rowtab
    const float* Radd;      // added after act, ldc (residual / second gradient contribution)
    const float* dactZ;     // multiply by act'(dactZ[m,n]) (dact) after Radd: dgrad epilogue
    int dact;
    float* colsum;          // atomic column sums of the final output (bias gradient)
    // per-group element strides (blockIdx.z)
    long gA, gA2, gB, gC, gBias, gZ, gR, gDZ, gCol;
    DropArgs drop;          // forward: dropout on act(.) before the residual add   (split-K kernel only)
    DropArgs dmask;         // dgrad: the producer's dropout mask, applied before its act' (split-K kernel only)
    LnA lnA;                // lnA.gamma != nullptr: A = LayerNorm(A rows) computed in the operand load (gemm_skinny_ln_kernel)
    DropArgs lnDrop;        // ... followed by that norm's output dropout
#ifdef TVC_GEMM_STAMPS
    unsigned long long* stamps;  // timing build only (tools/gemm_stamps.py): 8 x s_memtime per workgroup
#endif
};
#ifdef TVC_GEMM_STAMPS
// timing build: stamps are kept in scalar registers and written once, at the very end (a store per stamp would sit in the memory
// pipeline in front of the accesses being timed)
#define TVC_STAMP_DECL unsigned long long tvc_st[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TVC_STAMP(g, slot) tvc_st[slot] = __builtin_amdgcn_s_memtime()
#define TVC_STAMP_FLUSH(g)                                                                                          \
    do {                                                                                                             \
        if ((g).stamps && threadIdx.x == 0) {                                                                        \
            unsigned long long* o_ = (g).stamps + 8 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
            for (int i_ = 0; i_ < 8; ++i_) o_[i_] = tvc_st[i_];                                                      \
        }                                                                                                            \
    } while (0)
#else
#define TVC_STAMP_DECL do {} while (0)
#define TVC_STAMP(g, slot) do {} while (0)
#define TVC_STAMP_FLUSH(g) do {} while (0)
#endif

constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = GBK + 4;  // LDS rows padded to 20 floats (16-B aligned)
constexpr int THIN_K = 16, THIN_ROWS = 8;  // widest input handled by the plain-FMA input-layer kernels; rows per workgroup

template <bool KC>
__device__ __forceinline__ void gemm_load_tile(const float* __restrict__ P, const float* __restrict__ P2, int K1, long ld,
                                               long ld2, int row0, int n_rows, int k0, int K, float (&r)[4], int tid,
                                               bool fast) {
    // tile = 64 rows x 16 k.  KC: thread -> row tid/4, k (tid%4)*4..+3 ; !KC: thread -> k tid/16, rows (tid%16)*4..+3
    if (KC) {
        const int row = row0 + (tid >> 2), kk = k0 + (tid & 3) * 4;
        if (fast) {
            const float4 v = *reinterpret_cast<const float4*>(P + (long)row * ld + kk);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kk + j;
                float v = 0.0f;
                if (row < n_rows && k < K) v = (P2 != nullptr && k >= K1) ? P2[(long)row * ld2 + (k - K1)] : P[(long)row * ld + k];
                r[j] = v;
            }
        }
    } else {
        const int k = k0 + (tid >> 4), row = row0 + (tid & 15) * 4;
        if (fast) {
            const float4 v = *reinterpret_cast<const float4*>(P + (long)k * ld + row);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (k < K && row + j < n_rows) ? P[(long)k * ld + row + j] : 0.0f;
        }
    }
}
template <bool KC>
__device__ __forceinline__ void gemm_store_tile(float (*S)[GLD], const float (&r)[4], int tid) {
    if (KC) {
        *reinterpret_cast<float4*>(&S[tid >> 2][(tid & 3) * 4]) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        const int k = tid >> 4, row = (tid & 15) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) S[row + j][k] = r[j];
    }
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
    // one LDS block: the A/B k-tiles during the main loop, then the 64x64 output tile for the wide-store epilogue
    constexpr int CLD = GBN + 4;
    __shared__ __attribute__((aligned(16))) float smem[GBM * CLD];
    float (*As)[GLD] = reinterpret_cast<float (*)[GLD]>(smem);
    float (*Bs)[GLD] = reinterpret_cast<float (*)[GLD]>(smem + GBM * GLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order (guide T1): workgroups are dealt round-robin over the 8 XCDs, each with its own L2, so the
    // gridDim.x column tiles that re-read one 64-row slab of A are remapped onto ONE XCD instead of 4-8 different ones
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int gx = gridDim.x, nwg = gx * gridDim.y;
        if ((nwg & 7) == 0) {
            const int lin = blockIdx.x + gx * blockIdx.y;
            const int swz = (lin & 7) * (nwg >> 3) + (lin >> 3);
            bx = swz % gx;
            by = swz / gx;
        }
    }
    const int m0 = by * GBM, n0 = bx * GBN;
    const long z = blockIdx.z;
    const float* A = g.A + z * g.gA;
    const float* A2 = g.A2 ? g.A2 + z * g.gA2 : nullptr;
    const float* B = g.B + z * g.gB;
    // fast (16-byte) global loads only when the whole tile is in range and the leading dims / bases are aligned
    const bool a_al = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && A2 == nullptr;
    const bool b_al = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const bool a_rows_full = m0 + GBM <= g.M, b_rows_full = n0 + GBN <= g.N;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float ra[4], rb[4];
    const int nk = (g.K + GBK - 1) / GBK;
    {
        const bool kfull = GBK <= g.K;
        gemm_load_tile<A_KC>(A, A2, g.K1, g.lda, g.lda2, m0, g.M, 0, g.K, ra, tid, a_al && a_rows_full && kfull);
        gemm_load_tile<B_KC>(B, nullptr, 0, g.ldb, 0, n0, g.N, 0, g.K, rb, tid, b_al && b_rows_full && kfull);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous tile's fragment reads are done
        gemm_store_tile<A_KC>(As, ra, tid);
        gemm_store_tile<B_KC>(Bs, rb, tid);
        __syncthreads();
        if (kt + 1 < nk) {  // prefetch the next tile into registers while this one is multiplied
            const int k0 = (kt + 1) * GBK;
            const bool kfull = k0 + GBK <= g.K;
            gemm_load_tile<A_KC>(A, A2, g.K1, g.lda, g.lda2, m0, g.M, k0, g.K, ra, tid, a_al && a_rows_full && kfull);
            gemm_load_tile<B_KC>(B, nullptr, 0, g.ldb, 0, n0, g.N, k0, g.K, rb, tid, b_al && b_rows_full && kfull);
        }
        // lane l holds k-slots 4*(l>>4)..+3 of row/col (l&15): one 16-byte LDS read feeds four MFMAs
        float4 a4[2], b4[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a4[i] = *reinterpret_cast<const float4*>(&As[wr * 32 + i * 16 + (lane & 15)][(lane >> 4) * 4]);
            b4[i] = *reinterpret_cast<const float4*>(&Bs[wc * 32 + i * 16 + (lane & 15)][(lane >> 4) * 4]);
        }
        // k-component outermost: consecutive MFMAs go to different accumulators (16x16x4 f32 issues every 32 cycles
        // but needs 40 between dependent ones)
        const float av[2][4] = {{a4[0].x, a4[0].y, a4[0].z, a4[0].w}, {a4[1].x, a4[1].y, a4[1].z, a4[1].w}};
        const float bv[2][4] = {{b4[0].x, b4[0].y, b4[0].z, b4[0].w}, {b4[1].x, b4[1].y, b4[1].z, b4[1].w}};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][c], bv[j][c], acc[i][j], 0, 0, 0);
    }
    // epilogue.  C/D map of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* C = g.C + z * g.gC;
    const float* bias = g.bias ? g.bias + z * g.gBias : nullptr;
    float* Zout = g.Zout ? g.Zout + z * g.gZ : nullptr;
    const float* Radd = g.Radd ? g.Radd + z * g.gR : nullptr;
    const float* dZ = g.dactZ ? g.dactZ + z * g.gDZ : nullptr;
    float* colsum = g.colsum ? g.colsum + z * g.gCol : nullptr;
    // wide path (full, aligned tile): the accumulator layout gives each lane 4-byte stores 64 B apart; staging the tile
    // through LDS turns them into 16-byte stores of 256-byte row segments (guide T21)
    const bool wide = a_rows_full && b_rows_full && colsum == nullptr && (g.ldc & 3) == 0 && (g.N & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(Zout) |
                        reinterpret_cast<uintptr_t>(Radd) | reinterpret_cast<uintptr_t>(dZ) |
                        reinterpret_cast<uintptr_t>(g.rowtab)) & 15) == 0;
    if (wide) {
        float (*Cs)[CLD] = reinterpret_cast<float (*)[CLD]>(smem);
        __syncthreads();  // every wave is done reading the last k-tile
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[wr * 32 + i * 16 + (lane >> 4) * 4 + r][wc * 32 + j * 16 + (lane & 15)] = acc[i][j][r];
        __syncthreads();
        const int rl = tid >> 2, cq = (tid & 3) * 16;
        const int row = m0 + rl;
        const long o0 = (long)row * g.ldc + n0 + cq;
        const float* rt = g.rowtab ? g.rowtab + (long)(row % g.rowtab_rows) * g.N + n0 + cq : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = *reinterpret_cast<const float4*>(&Cs[rl][cq + 4 * q]);
            if (bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + n0 + cq + 4 * q);
                v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            if (rt) {
                const float4 t4 = *reinterpret_cast<const float4*>(rt + 4 * q);
                v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
            }
            const long o = o0 + 4 * q;
            if (Zout) *reinterpret_cast<float4*>(Zout + o) = v;
            v.x = act_f(v.x, g.act); v.y = act_f(v.y, g.act); v.z = act_f(v.z, g.act); v.w = act_f(v.w, g.act);
            if (Radd) {
                const float4 r4 = *reinterpret_cast<const float4*>(Radd + o);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            if (dZ) {
                const float4 z4 = *reinterpret_cast<const float4*>(dZ + o);
                v.x *= act_grad(z4.x, g.dact); v.y *= act_grad(z4.y, g.dact);
                v.z *= act_grad(z4.z, g.dact); v.w *= act_grad(z4.w, g.dact);
            }
            *reinterpret_cast<float4*>(C + o) = v;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 32 + j * 16 + (lane & 15);
        const bool cok = col < g.N;
        const float bv = (bias && cok) ? bias[col] : 0.0f;
        float csum = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (cok && row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (g.rowtab) v += g.rowtab[(long)(row % g.rowtab_rows) * g.N + col];
                    const long o = (long)row * g.ldc + col;
                    if (Zout) Zout[o] = v;
                    v = act_f(v, g.act);
                    if (Radd) v += Radd[o];
                    if (dZ) v *= act_grad(dZ[o], g.dact);
                    C[o] = v;
                    csum += v;
                }
            }
        }
        if (colsum) {
            csum += __shfl_xor(csum, 16);
            csum += __shfl_xor(csum, 32);
            if ((lane >> 4) == 0 && cok) atomicAdd(&colsum[col], csum);
        }
    }
}

// ---- shared epilogue: bias, row table, pre-activation copy, activation, residual, act' multiply
struct GemmEpi {
    float* C; const float* bias; float* Zout; const float* Radd; const float* dZ; float* colsum;
    unsigned kdrop, kmask;  // per-launch dropout hash keys
};
__device__ __forceinline__ GemmEpi gemm_epi_ptrs(const GemmArgs& g, long z) {
    GemmEpi e;
    e.C = g.C + z * g.gC;
    e.bias = g.bias ? g.bias + z * g.gBias : nullptr;
    e.Zout = g.Zout ? g.Zout + z * g.gZ : nullptr;
    e.Radd = g.Radd ? g.Radd + z * g.gR : nullptr;
    e.dZ = g.dactZ ? g.dactZ + z * g.gDZ : nullptr;
    e.colsum = g.colsum ? g.colsum + z * g.gCol : nullptr;
    e.kdrop = g.drop.ctr ? drop_key(g.drop, (unsigned)z) : 0u;
    e.kmask = g.dmask.ctr ? drop_key(g.dmask, (unsigned)z) : 0u;
    return e;
}
__device__ __forceinline__ float gemm_epi_value(const GemmArgs& g, const GemmEpi& e, float acc, int row, int col) {
    float v = acc + (e.bias ? e.bias[col] : 0.0f);
    if (g.rowtab) v += g.rowtab[(long)(row % g.rowtab_rows) * g.N + col];
    const long o = (long)row * g.ldc + col;
    if (e.Zout) e.Zout[o] = v;
    v = act_f(v, g.act);
    if (g.drop.ctr) v *= drop_factor(g.drop, e.kdrop, row, col);
    if (e.Radd) v += e.Radd[o];
    if (g.dmask.ctr) v *= drop_factor(g.dmask, e.kmask, row, col);
    if (e.dZ) v *= act_grad(e.dZ[o], g.dact);
    return v;
}

// ------------------------------------------------------------------ acting-path fused Linear (+act, +residual) + LayerNorm
// One workgroup owns 32 complete output rows (N = 256 or 512 columns, 4 waves x 64*JT/4 ... see JT below), so the
// LayerNorm that follows the Linear in every block of the actor (norm1 / norm2 / policy_head LN, optionally
// feature_norm right behind norm2) is applied in the epilogue: the [M,N] pre-norm activation never goes to HBM.
// Inference only (nothing is saved for backward).  A, B k-contiguous and 16-byte aligned, K % 16 == 0.
struct RowLnArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    const float* bias; int act;
    const float* Radd;                    // residual added before the norm (ldc)
    const float* gamma; const float* beta;
    const float* gamma2; const float* beta2;  // optional second LayerNorm
    // optional output head behind the norm (Linear(N -> headN <= 4), the policy's mean / log_std layer): when set, the
    // normalised rows are NOT stored; only headOut[M, headN] = rows . headW^T + headB is
    const float* headW; const float* headB; float* headOut; int headN;
    // optional generated A operand (GEN instantiations): A[m, k] = genAct(genB[k] + sum_j genX[m, j] genW[k, j]), the thin
    // input layer (genK <= 16 inputs) evaluated while the k-tiles are staged, so its activation never exists in HBM
    const float* genX; const float* genW; const float* genB; int genK, genAct, genLd;
};

template <int JT>  // 16-column MFMA tiles per wave; N = 4 waves * JT * 16  (JT = 4 -> 256, JT = 8 -> 512)
__device__ __forceinline__ void rowln_normalize(f32x4 (&u)[2][JT], float* red, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int wave, int lane, int N) {
    // u[i][j][r]: row i*16 + (lane>>4)*4 + r, column wave*16*JT + j*16 + (lane&15).  Two-pass mean / variance like torch.
    const int q = lane >> 4, l15 = lane & 15;
    float mean[2][4], rstd[2][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const float d = pass == 0 ? u[i][j][r] : u[i][j][r] - mean[i][r];
                    s += pass == 0 ? d : d * d;
                }
                s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
                if (l15 == 0) red[wave * 32 + i * 16 + q * 4 + r] = s;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = i * 16 + q * 4 + r;
                const float t = (red[rl] + red[32 + rl] + red[64 + rl] + red[96 + rl]) / (float)N;
                if (pass == 0) mean[i][r] = t;
                else rstd[i][r] = rsqrtf(t + 1e-5f);
            }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int col = wave * 16 * JT + j * 16 + l15;
        const float gm = gamma[col], bt = beta[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[i][j][r] = (u[i][j][r] - mean[i][r]) * rstd[i][r] * gm + bt;
    }
}

constexpr int GEN_LD = THIN_K + 4;  // LDS row of the generated operand's layer: genK weights, the bias, zero padding (16-byte rows)
__device__ __forceinline__ f32x4 rowln_gen_a(const float (&xr)[GEN_LD], const float* gw, int act, int k0) {
    // four consecutive k of the generated operand for this thread's row; xr[genK] = 1 picks up the bias slot
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4* w = reinterpret_cast<const f32x4*>(gw + (long)(k0 + c) * GEN_LD);
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < GEN_LD / 4; ++q) {
            const f32x4 w4 = w[q];
            v = fmaf(xr[4 * q], w4[0], v); v = fmaf(xr[4 * q + 1], w4[1], v);
            v = fmaf(xr[4 * q + 2], w4[2], v); v = fmaf(xr[4 * q + 3], w4[3], v);
        }
        r[c] = act_f(v, act);
    }
    return r;
}
template <int JT, int KS, bool GEN = false>  // KS: k-tiles (K / 16) when known at compile time (16 or 32), 0 = runtime
__global__ void __launch_bounds__(256) gemm_rowln_kernel(RowLnArgs g) {
    constexpr int N = 64 * JT;  // 4 waves x JT tiles x 16 columns
    __shared__ __attribute__((aligned(16))) float As[32][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[N][GLD];
    __shared__ float red[128];
    __shared__ __attribute__((aligned(16))) float gw[GEN ? 256 * GEN_LD : 4];  // generated operand (K <= 256): [K][GEN_LD] rows
    typedef const f32x4* cv4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 32;
    const int lr = tid >> 2, lk = (tid & 3) * 4;  // loader: row lr (+64p for B), k offset lk
    const float* ap = GEN ? nullptr : g.A + (long)min(m0 + (lr & 31), g.M - 1) * g.lda + lk;
    const float* bp = g.B + (long)lr * g.ldb + lk;
    float xr[GEN_LD];
    if (GEN) {
        for (int e = tid; e < g.K * GEN_LD; e += 256) {
            const int k = e / GEN_LD, j = e - k * GEN_LD;
            gw[e] = j < g.genK ? g.genW[(long)k * g.genK + j] : (j == g.genK ? g.genB[k] : 0.0f);
        }
        const float* xp = g.genX + (long)min(m0 + (lr & 31), g.M - 1) * g.genLd;
#pragma unroll
        for (int j = 0; j < GEN_LD; ++j) {
            const float t = xp[min(j, g.genK - 1)];
            xr[j] = j < g.genK ? t : (j == g.genK ? 1.0f : 0.0f);
        }
        __syncthreads();
    }
    f32x4 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra, rb[JT];
    ra = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (GEN) { if (tid < 128) ra = rowln_gen_a(xr, gw, g.genAct, lk); }  // waves 0-1 stage the A tile
    else ra = *(cv4)ap;
#pragma unroll
    for (int p = 0; p < JT; ++p) rb[p] = *(cv4)(bp + (long)(64 * p) * g.ldb);
    const int nk = KS > 0 ? KS : g.K / GBK;
    const int l15 = lane & 15, fk = (lane >> 4) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        if (tid < 128) *(f32x4*)&As[lr][lk] = ra;
#pragma unroll
        for (int p = 0; p < JT; ++p) *(f32x4*)&Bs[lr + 64 * p][lk] = rb[p];
        __syncthreads();
        if (kt + 1 < nk) {
            const int ko = (kt + 1) * GBK;
            if (GEN) { if (tid < 128) ra = rowln_gen_a(xr, gw, g.genAct, ko + lk); }
            else ra = *(cv4)(ap + ko);
#pragma unroll
            for (int p = 0; p < JT; ++p) rb[p] = *(cv4)(bp + (long)(64 * p) * g.ldb + ko);
        }
        f32x4 a4[2], b4[JT];
#pragma unroll
        for (int i = 0; i < 2; ++i) a4[i] = *(cv4)&As[i * 16 + l15][fk];
#pragma unroll
        for (int j = 0; j < JT; ++j) b4[j] = *(cv4)&Bs[wave * 16 * JT + j * 16 + l15][fk];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][c], b4[j][c], acc[i][j], 0, 0, 0);
    }
    // epilogue: bias, activation, residual -> LayerNorm (-> second LayerNorm) -> store
    const int q = lane >> 4;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int col = wave * 16 * JT + j * 16 + l15;
        const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = act_f(acc[i][j][r] + bv, g.act);
                if (g.Radd) v += g.Radd[(long)min(m0 + i * 16 + q * 4 + r, g.M - 1) * g.ldc + col];
                acc[i][j][r] = v;
            }
    }
    if (g.gamma) rowln_normalize<JT>(acc, red, g.gamma, g.beta, wave, lane, N);  // no norm: Linear (+act) + output head only
    if (g.gamma2) rowln_normalize<JT>(acc, red, g.gamma2, g.beta2, wave, lane, N);
    if (g.headW) {  // dot every complete row with the head's weight rows: per-wave partials meet through LDS
        __shared__ float hred[4][32][4];
        for (int o = 0; o < g.headN; ++o) {
            float part[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[i][r] = 0.0f;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const float w = g.headW[(long)o * N + wave * 16 * JT + j * 16 + l15];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[i][r] = fmaf(acc[i][j][r], w, part[i][r]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = part[i][r];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    if (l15 == 0) hred[wave][i * 16 + q * 4 + r][o] = v;
                }
        }
        __syncthreads();
        if (tid < 32 * g.headN) {
            const int rl = tid / g.headN, o = tid - rl * g.headN;
            const float v = hred[0][rl][o] + hred[1][rl][o] + hred[2][rl][o] + hred[3][rl][o] + g.headB[o];
            if (m0 + rl < g.M) g.headOut[(long)(m0 + rl) * g.headN + o] = v;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int col = wave * 16 * JT + j * 16 + l15;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + i * 16 + q * 4 + r;
                if (row < g.M) g.C[(long)row * g.ldc + col] = acc[i][j][r];
            }
    }
}

// ------------------------------------------------------------------ skinny GEMM (update path, M = batch = a few hundred)
// 32x32 output tile per workgroup; the four waves split K and stream their MFMA fragments straight from global /
// L2 into registers (no LDS staging, no barrier in the k-loop: with so few rows there is no reuse to stage for,
// and the 64x64 LDS-tiled kernel above is latency-bound at ~1 us per k-tile on these shapes); partial tiles are
// summed through LDS once at the end.
// FAST: every fragment load is in range and 16-byte aligned (M, N multiples of 32, K a multiple of 16, k-contiguous
// operands with ld % 4 == 0): loads are unconditional.  Otherwise addresses are clamped into range and the value is
// zero-selected AFTER the load -- never a branch around a load (hipcc would wait vmcnt(0) per element, guide 5 item 4c).
// What a batch-256 GEMM launch costs is latency, not arithmetic (in-kernel stamps, profiles/r03_b_gemm_stamps.md: ~2.3 us until the
// first operands land in eight cold L2s, 1.0 us of MFMAs, and 1.5 us for ONE more dependent miss when the epilogue only then asks
// for its bias / residual / act' operands).  Two remedies:
//  * xcd_tile: workgroups are dealt round-robin over the 8 XCDs, each with its own (just invalidated) L2; the tiles of one XCD are
//    made a compact r x c block of the output, so that an XCD fetches M/r rows of A and N/c rows of B instead of nearly all of both;
//  * SkinnyPre: every epilogue operand of this thread's four output columns is requested at kernel entry, beside the MFMA operands.
__device__ __forceinline__ void xcd_tile(int lin, int gx, int gy, int& bx, int& by) {
    bx = lin % gx; by = lin / gx;
    const int total = gx * gy;
    if ((total & 7) != 0) return;
    int rg = 0, best = 1 << 30;
#pragma unroll
    for (int r = 8; r >= 1; r >>= 1) {
        const int c = 8 / r;
        if ((gy % r) == 0 && (gx % c) == 0) {
            const int cost = gy / r + gx / c;
            if (cost < best) { best = cost; rg = r; }
        }
    }
    if (rg == 0) return;
    const int cg = 8 / rg, tw = gx / cg, th = gy / rg;
    const int xcd = lin & 7, idx = lin >> 3;
    by = (xcd / cg) * th + idx / tw;
    bx = (xcd % cg) * tw + idx % tw;
}
struct SkinnyPre { float bias[4], rt[4], radd[4], dz[4]; };
__device__ __forceinline__ SkinnyPre skinny_prefetch(const GemmArgs& g, long z, int m0, int n0) {
    SkinnyPre p;
    const int tid = threadIdx.x, rl = tid >> 3, cl = (tid & 7) * 4;
    const int row = min(m0 + rl, g.M - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = min(n0 + cl + c, g.N - 1);
        const long o = (long)row * g.ldc + col;
        p.bias[c] = g.bias ? g.bias[z * g.gBias + col] : 0.0f;
        p.rt[c] = g.rowtab ? g.rowtab[(long)(row % g.rowtab_rows) * g.N + col] : 0.0f;
        p.radd[c] = g.Radd ? g.Radd[z * g.gR + o] : 0.0f;
        p.dz[c] = g.dactZ ? g.dactZ[z * g.gDZ + o] : 0.0f;
    }
    return p;
}
// FAST launches (every tile full, K a multiple of 256, every pointer 16-byte aligned): the thread's four output columns are one
// float4 everywhere.  Loads are UNCONDITIONAL (an absent operand reads a valid dummy address and is zero-selected afterwards):
// hipcc answers a load inside a branch, a register zeroed behind a load, or a load behind a possibly aliasing store with a full
// s_waitcnt vmcnt(0) -- the first version of this kernel spent 8 serial round trips on its 16 operand loads and 4 more on its 4
// dword stores (6 000 + 2 700 of its 12 000 cycles, tools/gemm_stamps.py).
struct SkinnyPre4 { float4 bias, rt, radd, dz; unsigned kdrop, kmask; };
__device__ __forceinline__ float4 ld4_or_zero(const float* p, const float* dummy, long off) {
    const float4 v = *reinterpret_cast<const float4*>(p ? p + off : dummy);
    return p ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ SkinnyPre4 skinny_prefetch4(const GemmArgs& g, long z, int m0, int n0) {
    SkinnyPre4 p;
    const int tid = threadIdx.x, row = m0 + (tid >> 3), col = n0 + (tid & 7) * 4;
    const long o = (long)row * g.ldc + col;
    const float* dummy = g.B;  // valid, 16-byte aligned
    p.bias = ld4_or_zero(g.bias, dummy, z * g.gBias + col);
    p.rt = ld4_or_zero(g.rowtab, dummy, g.rowtab ? (long)(row % g.rowtab_rows) * g.N + col : 0);
    p.radd = ld4_or_zero(g.Radd, dummy, z * g.gR + o);
    p.dz = ld4_or_zero(g.dactZ, dummy, z * g.gDZ + o);
    const int* ic = reinterpret_cast<const int*>(dummy);
    const unsigned c1 = (unsigned)*(g.drop.ctr ? g.drop.ctr : ic), c2 = (unsigned)*(g.dmask.ctr ? g.dmask.ctr : ic);
    p.kdrop = drop_mix(c1 ^ (g.drop.site * 0x9E3779B9u) ^ ((unsigned)z * 0x7F4A7C15u) ^ g.drop.seed);
    p.kmask = drop_mix(c2 ^ (g.dmask.site * 0x9E3779B9u) ^ ((unsigned)z * 0x7F4A7C15u) ^ g.dmask.seed);
    return p;
}
template <bool COH = false>  // COH: C is published for other workgroups of THIS launch (ln_tail): device-scope write-through stores
__device__ __forceinline__ void skinny_epilogue4(const GemmArgs& g, const f32x4 (&acc)[2][2], float (*red)[32 * 33], int m0, int n0, long z,
                                                 const SkinnyPre4& pre
#ifdef TVC_GEMM_STAMPS
                                                 , unsigned long long (&tvc_st)[8]
#endif
                                                 ) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(i * 16 + (lane >> 4) * 4 + r) * 33 + j * 16 + l15] = acc[i][j][r];
    TVC_STAMP(g, 3);  // (wave 0's MFMAs are done: their results were needed for the LDS writes)
    __syncthreads();
    TVC_STAMP(g, 4);
    const int rl = tid >> 3, cl = (tid & 7) * 4;  // 32 rows x 8 column quads
    const int row = m0 + rl, col0 = n0 + cl;
    const float bias[4] = {pre.bias.x, pre.bias.y, pre.bias.z, pre.bias.w}, rt[4] = {pre.rt.x, pre.rt.y, pre.rt.z, pre.rt.w};
    const float radd[4] = {pre.radd.x, pre.radd.y, pre.radd.z, pre.radd.w}, dz[4] = {pre.dz.x, pre.dz.y, pre.dz.z, pre.dz.w};
    float v[4], zv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int o = rl * 33 + cl + c;
        float t = red[0][o] + red[1][o] + red[2][o] + red[3][o] + bias[c] + rt[c];
        zv[c] = t;
        t = act_f(t, g.act);
        if (g.drop.ctr) t *= drop_factor(g.drop, pre.kdrop, row, col0 + c);
        t += radd[c];
        if (g.dmask.ctr) t *= drop_factor(g.dmask, pre.kmask, row, col0 + c);
        if (g.dactZ) t *= act_grad(dz[c], g.dact);
        v[c] = t;
    }
    TVC_STAMP(g, 5);
    // stores last, back to back
    const long oo = (long)row * g.ldc + col0;
    if (g.Zout) *reinterpret_cast<float4*>(g.Zout + z * g.gZ + oo) = make_float4(zv[0], zv[1], zv[2], zv[3]);
    if (COH) {
#pragma unroll
        for (int c = 0; c < 4; ++c) __hip_atomic_store(g.C + z * g.gC + oo + c, v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<float4*>(g.C + z * g.gC + oo) = make_float4(v[0], v[1], v[2], v[3]);
    }
    TVC_STAMP(g, 6);
#ifdef TVC_GEMM_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0070);  // stores acknowledged
    TVC_STAMP(g, 7);
#endif
    if (g.colsum) {  // column sums of the final tile (bias gradient): reduce the 32 rows through LDS
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[0][rl * 33 + cl + c] = v[c];
        __syncthreads();
        if (tid < 32) {
            float sum = 0.0f;
            for (int r = 0; r < 32; ++r) sum += red[0][r * 33 + tid];
            atomicAdd(g.colsum + z * g.gCol + n0 + tid, sum);
        }
    }
}
// operand fragments of one chunk (4 k-steps of 16 from k-step s0 on), every load unconditional
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void skinny_load_chunk(const GemmArgs& g, const float* __restrict__ A, const float* __restrict__ B, int m0,
                                                  int n0, int s0, int l15, int kq, float (&a)[4][2][4], float (&b)[4][2][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k0 = (s0 + u) * 16 + kq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + i * 16 + l15, col = n0 + i * 16 + l15;
            if (A_KC) {
                const float4 v = *reinterpret_cast<const float4*>(A + (long)row * g.lda + k0);
                a[u][i][0] = v.x; a[u][i][1] = v.y; a[u][i][2] = v.z; a[u][i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][i][j] = A[(long)(k0 + j) * g.lda + row];
            }
            if (B_KC) {
                const float4 v = *reinterpret_cast<const float4*>(B + (long)col * g.ldb + k0);
                b[u][i][0] = v.x; b[u][i][1] = v.y; b[u][i][2] = v.z; b[u][i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[u][i][j] = B[(long)(k0 + j) * g.ldb + col];
            }
        }
    }
}
__device__ __forceinline__ void skinny_mfma_chunk(const float (&a)[4][2][4], const float (&b)[4][2][4], f32x4 (&acc)[2][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i][j4], b[u][j][j4], acc[i][j], 0, 0, 0);
}
template <bool A_KC, bool B_KC, bool COH = false>
__device__ __forceinline__ void skinny_body_fast(const GemmArgs& g, int bx, int by, long z, float (*red)[32 * 33]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = by * 32, n0 = bx * 32;
    const float* __restrict__ A = g.A + z * g.gA;
    const float* __restrict__ B = g.B + z * g.gB;
    const int nch = g.K >> 8;          // chunks of 4 k-steps per wave: wave w owns k-steps [4 nch w, 4 nch (w + 1))
    const int s_begin = wave * 4 * nch;
    const int l15 = lane & 15, kq = (lane >> 4) * 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    TVC_STAMP_DECL;
    TVC_STAMP(g, 0);
    const SkinnyPre4 pre = skinny_prefetch4(g, z, m0, n0);
#ifdef TVC_GEMM_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0070);  // (timing build: the prefetch loads alone)
    TVC_STAMP(g, 1);
#endif
    float a0[4][2][4], b0[4][2][4];
    skinny_load_chunk<A_KC, B_KC>(g, A, B, m0, n0, s_begin, l15, kq, a0, b0);
    if (nch == 2) {  // K = 512: both chunks in flight before the first MFMA
        float a1[4][2][4], b1[4][2][4];
        skinny_load_chunk<A_KC, B_KC>(g, A, B, m0, n0, s_begin + 4, l15, kq, a1, b1);
        skinny_mfma_chunk(a0, b0, acc);
        skinny_mfma_chunk(a1, b1, acc);
    } else {
#ifdef TVC_GEMM_STAMPS
        __builtin_amdgcn_s_waitcnt(0x0070);
        TVC_STAMP(g, 2);
#endif
        skinny_mfma_chunk(a0, b0, acc);
        for (int c = 1; c < nch; ++c) {
            skinny_load_chunk<A_KC, B_KC>(g, A, B, m0, n0, s_begin + 4 * c, l15, kq, a0, b0);
            skinny_mfma_chunk(a0, b0, acc);
        }
    }
#ifdef TVC_GEMM_STAMPS
    skinny_epilogue4<COH>(g, acc, red, m0, n0, z, pre, tvc_st);
    TVC_STAMP_FLUSH(g);
#else
    skinny_epilogue4<COH>(g, acc, red, m0, n0, z, pre);
#endif
}
// partial tiles of the four waves summed through LDS, then the shared epilogue and (optionally) bias-gradient column sums
__device__ __forceinline__ void skinny_epilogue(const GemmArgs& g, const f32x4 (&acc)[2][2], float (*red)[32 * 33], int m0, int n0, long z,
                                                const SkinnyPre& pre) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(i * 16 + (lane >> 4) * 4 + r) * 33 + j * 16 + l15] = acc[i][j][r];
    __syncthreads();
    float* C = g.C + z * g.gC;
    float* Zout = g.Zout ? g.Zout + z * g.gZ : nullptr;
    float* colsum = g.colsum ? g.colsum + z * g.gCol : nullptr;
    const unsigned kdrop = g.drop.ctr ? drop_key(g.drop, (unsigned)z) : 0u;
    const unsigned kmask = g.dmask.ctr ? drop_key(g.dmask, (unsigned)z) : 0u;
    const int rl = tid >> 3, cl = (tid & 7) * 4;  // 32 rows x 8 column quads
    const int row = m0 + rl;
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = n0 + cl + c;
        const int o = rl * 33 + cl + c;
        float v = red[0][o] + red[1][o] + red[2][o] + red[3][o];
        out[c] = 0.0f;
        if (row < g.M && col < g.N) {
            const long oo = (long)row * g.ldc + col;
            v += pre.bias[c] + pre.rt[c];
            if (Zout) Zout[oo] = v;
            v = act_f(v, g.act);
            if (g.drop.ctr) v *= drop_factor(g.drop, kdrop, row, col);
            v += pre.radd[c];
            if (g.dmask.ctr) v *= drop_factor(g.dmask, kmask, row, col);
            if (g.dactZ) v *= act_grad(pre.dz[c], g.dact);
            C[oo] = v;
            out[c] = v;
        }
    }
    if (colsum) {  // column sums of the final tile (bias gradient): reduce the 32 rows through LDS
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[0][rl * 33 + cl + c] = out[c];
        __syncthreads();
        if (tid < 32 && n0 + tid < g.N) {
            float sum = 0.0f;
            for (int r = 0; r < 32; ++r) sum += red[0][r * 33 + tid];
            atomicAdd(&colsum[n0 + tid], sum);
        }
    }
}
template <bool A_KC, bool B_KC, bool FAST>
__device__ __forceinline__ void skinny_body(const GemmArgs& g, int bx, int by, long z, float (*red)[32 * 33]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = by * 32, n0 = bx * 32;
    const float* A = g.A + z * g.gA;
    const float* B = g.B + z * g.gB;
    const int ksteps = (g.K + 15) / 16;
    const int spw = (ksteps + 3) / 4;
    const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, kq = (lane >> 4) * 4;
    constexpr int CH = 4;  // k-steps per chunk: the fragment loads of a chunk are all in flight before its 64 MFMAs
    const SkinnyPre pre = skinny_prefetch(g, z, m0, n0);
    for (int sc = s_begin; sc < s_end; sc += CH) {
        float a[CH][2][4], b[CH][2][4];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const bool live = sc + u < s_end;
            const int k0 = (live ? sc + u : s_begin) * 16 + kq;  // dead steps re-read a valid step and are zeroed
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + i * 16 + l15, col = n0 + i * 16 + l15;
                const int rowc = FAST ? row : min(row, g.M - 1), colc = FAST ? col : min(col, g.N - 1);
                if (A_KC && FAST) {
                    const float4 v = *reinterpret_cast<const float4*>(A + (long)rowc * g.lda + k0);
                    a[u][i][0] = v.x; a[u][i][1] = v.y; a[u][i][2] = v.z; a[u][i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kc = FAST ? k0 + j : min(k0 + j, g.K - 1);
                        const float v = A_KC ? A[(long)rowc * g.lda + kc] : A[(long)kc * g.lda + rowc];
                        a[u][i][j] = (FAST || (row < g.M && k0 + j < g.K)) ? v : 0.0f;
                    }
                }
                if (B_KC && FAST) {
                    const float4 v = *reinterpret_cast<const float4*>(B + (long)colc * g.ldb + k0);
                    b[u][i][0] = v.x; b[u][i][1] = v.y; b[u][i][2] = v.z; b[u][i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kc = FAST ? k0 + j : min(k0 + j, g.K - 1);
                        const float v = B_KC ? B[(long)colc * g.ldb + kc] : B[(long)kc * g.ldb + colc];
                        b[u][i][j] = (FAST || (col < g.N && k0 + j < g.K)) ? v : 0.0f;
                    }
                }
                if (!live) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[u][i][j] = 0.0f; b[u][i][j] = 0.0f; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i][j4], b[u][j][j4], acc[i][j], 0, 0, 0);
    }
    skinny_epilogue(g, acc, red, m0, n0, z, pre);
}
#ifdef TVC_SKINNY_LB4
#define TVC_SKINNY_BOUNDS __launch_bounds__(256, 4)  // <= 128 VGPRs: two workgroups fit into the half of a CU the acting kernel leaves
#else
#define TVC_SKINNY_BOUNDS __launch_bounds__(256)
#endif
template <bool A_KC, bool B_KC, bool FAST>
__global__ void TVC_SKINNY_BOUNDS gemm_skinny_kernel(GemmArgs g) {
    warm_kernargs<sizeof(GemmArgs)>();
    TVC_LEARNER_PRIO();
    __shared__ float red[4][32 * 33];
    int bx, by;
    xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, bx, by);
    if (FAST) skinny_body_fast<A_KC, B_KC>(g, bx, by, blockIdx.z, red);
    else skinny_body<A_KC, B_KC, false>(g, bx, by, blockIdx.z, red);
}
// The same 32x32 split-K tile with A = LayerNorm(X) (optionally two norms, optionally dropout) computed in the operand load:
// K is the norm's width (256 or 512), so the four waves together stream the COMPLETE rows of their 32-row slab -- each wave holds
// its quarter of every row (KS k-steps) in registers, row sums meet through shuffles + a 2 KB LDS exchange, and the normalised
// fragments feed the MFMAs directly.  Saves the norm's launch and its round trip through memory: an update at batch 256 is bound
// by the number of dependent launches (~6 us each), not by arithmetic.  Requires the FAST conditions of gemm_skinny_kernel.
template <int KS>  // k-steps of 16 per wave: K = 64 KS (4 -> 256, 8 -> 512)
__global__ void __launch_bounds__(256) gemm_skinny_ln_kernel(GemmArgs g) {
    warm_kernargs<sizeof(GemmArgs)>();
    TVC_LEARNER_PRIO();
    __shared__ float red[4][32 * 33];
    __shared__ float lnstat[4][4][32];  // [pass][wave][row of the slab]
    constexpr int K = 64 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, bx, by);
    const long z = blockIdx.z;
    const int m0 = by * 32, n0 = bx * 32;
    const float* A = g.A + z * g.gA;
    const float* B = g.B + z * g.gB;
    const int l15 = lane & 15, q = lane >> 4, kq = q * 4;
    const int s_begin = wave * KS;
    const SkinnyPre4 pre = skinny_prefetch4(g, z, m0, n0);
    // all of this wave's A fragments: a[u][i][j] = X[m0 + 16 i + l15][16 (s_begin + u) + kq + j]
    float a[KS][2][4];
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(A + (long)(m0 + i * 16 + l15) * g.lda + (s_begin + u) * 16 + kq);
            a[u][i][0] = v.x; a[u][i][1] = v.y; a[u][i][2] = v.z; a[u][i][3] = v.w;
        }
    // every other operand is requested NOW, beside the rows: the weight fragments and the norms' gamma / beta do not depend on the
    // statistics, and a load issued after them would be one more full miss on the critical path
    float b[KS][2][4];
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(B + (long)(n0 + i * 16 + l15) * g.ldb + (s_begin + u) * 16 + kq);
            b[u][i][0] = v.x; b[u][i][1] = v.y; b[u][i][2] = v.z; b[u][i][3] = v.w;
        }
    const LnA& ln = g.lnA;
    const bool two = ln.gamma2 != nullptr;
    float4 gq[2][KS], bq[2][KS];  // [norm][k-step]: gamma / beta of this lane's four k
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float* gam = ((pass == 0 || !two) ? ln.gamma : ln.gamma2) + z * ln.gP;
        const float* bet = ((pass == 0 || !two) ? ln.beta : ln.beta2) + z * ln.gP;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            gq[pass][u] = *reinterpret_cast<const float4*>(gam + (s_begin + u) * 16 + kq);
            bq[pass][u] = *reinterpret_cast<const float4*>(bet + (s_begin + u) * 16 + kq);
        }
    }
    const bool writer = bx == 0;  // column tile 0 writes the rows' statistics
    // the materialised output is written by ALL column tiles, each one its share of the k-steps (k-step s by tile s mod gridDim.x):
    // one tile writing its 32 complete rows alone was 32 KB of stores through one CU, the slowest workgroup of the launch
    const int ngx = gridDim.x;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !two) break;
        float mean[2], rstd[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {  // two-pass statistics like torch: mean, then the centred second moment
            float sm[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sm[i] = 0.0f;
#pragma unroll
                for (int u = 0; u < KS; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = st == 0 ? a[u][i][j] : a[u][i][j] - mean[i];
                        sm[i] += st == 0 ? d : d * d;
                    }
                sm[i] += __shfl_xor(sm[i], 16);
                sm[i] += __shfl_xor(sm[i], 32);
                if (q == 0) lnstat[2 * pass + st][wave][i * 16 + l15] = sm[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = i * 16 + l15;
                const float t = (lnstat[2 * pass + st][0][r] + lnstat[2 * pass + st][1][r] + lnstat[2 * pass + st][2][r] +
                                 lnstat[2 * pass + st][3][r]) * (1.0f / (float)K);
                if (st == 0) mean[i] = t;
                else rstd[i] = rsqrtf(t + 1e-5f);
            }
        }
        float* mo = pass == 0 ? ln.mean : ln.mean2;
        float* ro = pass == 0 ? ln.rstd : ln.rstd2;
        if (writer && wave == 0 && q == 0 && mo) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                mo[z * ln.gS + m0 + i * 16 + l15] = mean[i];
                ro[z * ln.gS + m0 + i * 16 + l15] = rstd[i];
            }
        }
        const bool last = pass == 1 || !two;
        const unsigned key = (last && g.lnDrop.ctr) ? drop_key(g.lnDrop, (unsigned)z) : 0u;
        float* yo = last ? ln.Y : ln.Y1;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int k0 = (s_begin + u) * 16 + kq;
            const float4 g4 = gq[pass][u], b4 = bq[pass][u];
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + i * 16 + l15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = (a[u][i][j] - mean[i]) * rstd[i] * gv[j] + bv[j];
                    if (last && g.lnDrop.ctr) v *= drop_factor(g.lnDrop, key, row, k0 + j);
                    a[u][i][j] = v;
                }
                if (yo && ((s_begin + u) % ngx) == bx % ngx)
                    *reinterpret_cast<float4*>(yo + z * ln.gY + (long)row * K + k0) = make_float4(a[u][i][0], a[u][i][1], a[u][i][2], a[u][i][3]);
            }
        }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i][j4], b[u][j][j4], acc[i][j], 0, 0, 0);
#ifdef TVC_GEMM_STAMPS
    unsigned long long tvc_st[8];
    skinny_epilogue4(g, acc, red, m0, n0, z, pre, tvc_st);
#else
    skinny_epilogue4(g, acc, red, m0, n0, z, pre);
#endif
}

// Backward of one Linear at batch size in ONE launch: workgroups [0, nw) compute dW = dZ^T X (both operands
// m-contiguous), the others dX = dZ W (+ residual gradient, act', bias column sums).  The two products share dZ and
// nothing else, so running them side by side halves the launch count of the backward chain (latency-bound, H5).
struct GemmPair {
    GemmArgs w, x;
    int nw, w_tiles_x, x_tiles_x;
};
template <bool FAST>
__global__ void TVC_SKINNY_BOUNDS gemm_skinny_bwd_kernel(GemmPair p) {
    warm_kernargs<sizeof(GemmPair)>();
    TVC_LEARNER_PRIO();
    __shared__ float red[4][32 * 33];
    int b = blockIdx.x, bx, by;
    const bool al = (p.nw & 7) == 0 && ((gridDim.x - p.nw) & 7) == 0;  // both regions start on an XCD-0 workgroup
    if (b < p.nw) {
        if (al) xcd_tile(b, p.w_tiles_x, p.nw / p.w_tiles_x, bx, by);
        else { bx = b % p.w_tiles_x; by = b / p.w_tiles_x; }
        if (FAST) skinny_body_fast<false, false>(p.w, bx, by, blockIdx.z, red);
        else skinny_body<false, false, false>(p.w, bx, by, blockIdx.z, red);
    } else {
        b -= p.nw;
        if (al) xcd_tile(b, p.x_tiles_x, (gridDim.x - p.nw) / p.x_tiles_x, bx, by);
        else { bx = b % p.x_tiles_x; by = b / p.x_tiles_x; }
        if (FAST) skinny_body_fast<true, false>(p.x, bx, by, blockIdx.z, red);
        else skinny_body<true, false, false>(p.x, bx, by, blockIdx.z, red);
    }
}

// ------------------------------------------------------------------ thin Linear: in_dim <= 16 (observation / [s | a] input layers)
// K = 10 or 12 is neither a multiple of the MFMA k-step nor 16-byte aligned, so the MFMA kernels fall on their
// bounds-checked slow path (11-23 us at batch 256, 64 us at 65 536 rows).  These are plain FMA kernels: one thread
// per output column (forward), per weight row (dW), one wave per row (dX); the optional second source X2 reads the
// critic input [s | a] in place (no concat kernel).
struct ThinArgs {
    const float* X; const float* X2;   // columns [0, K1) from X (row stride ldx), [K1, K) from X2 (row stride ldx2)
    int ldx, ldx2;
    const float* W; const float* bias; // W[N, K], bias[N]
    const float* rowtab; int rowtab_rows;
    float* Y; float* Z;                // forward: Y = act(Z) (* Mul), Z optional (pre-activation for backward)
    const float* Mul;                  // optional [M, N] gate input multiplied in (SqueezeExcitation: x * sigmoid(fc2(.)))
    const float* dZ; float* dW; float* dX;
    int M, N, K, K1, act;
    long gX, gX2, gW, gB, gY, gDW, gDX;  // group strides (blockIdx.z)
    int gdiv;                            // > 1: the INPUT strides gX / gX2 apply to z / gdiv (two nets per input: thin_fwd_kernel only)
};
__device__ __forceinline__ float thin_x(const ThinArgs& a, const float* X, const float* X2, int row, int k) {
    // clamped address, zero-selected after the load (never a branch around a load)
    const int kc = min(k, a.K - 1), rc = min(row, a.M - 1);
    const bool second = X2 != nullptr && kc >= a.K1;
    const float* p = second ? X2 + (long)rc * a.ldx2 + (kc - a.K1) : X + (long)rc * a.ldx + min(kc, a.K1 - 1);
    const float v = *p;
    return (k < a.K && row < a.M) ? v : 0.0f;
}
__global__ void __launch_bounds__(256) thin_fwd_kernel(ThinArgs a) {
    TVC_LEARNER_PRIO();
    __shared__ float xs[THIN_ROWS][THIN_K];
    const int tid = threadIdx.x, n = blockIdx.y * 256 + tid, row0 = blockIdx.x * THIN_ROWS;
    const long z = blockIdx.z, zi = a.gdiv > 1 ? z / a.gdiv : z;
    const float* X = a.X + zi * a.gX;
    const float* X2 = a.X2 ? a.X2 + zi * a.gX2 : nullptr;
    if (tid < THIN_ROWS * THIN_K) xs[tid / THIN_K][tid % THIN_K] = thin_x(a, X, X2, row0 + tid / THIN_K, tid % THIN_K);
    const int nc = min(n, a.N - 1);
    const float* W = a.W + z * a.gW + (long)nc * a.K;
    float w[THIN_K];
#pragma unroll
    for (int k = 0; k < THIN_K; ++k) {
        const float v = W[min(k, a.K - 1)];
        w[k] = k < a.K ? v : 0.0f;
    }
    const float b = a.bias ? a.bias[z * a.gB + nc] : 0.0f;
    __syncthreads();
    if (n >= a.N) return;
    float* Y = a.Y + z * a.gY;
    float* Z = a.Z ? a.Z + z * a.gY : nullptr;
#pragma unroll
    for (int r = 0; r < THIN_ROWS; ++r) {
        const int row = row0 + r;
        if (row >= a.M) break;
        float v = b;
#pragma unroll
        for (int k = 0; k < THIN_K; ++k) v = fmaf(xs[r][k], w[k], v);
        if (a.rowtab) v += a.rowtab[(long)(row % a.rowtab_rows) * a.N + n];
        const long o = (long)row * a.N + n;
        if (Z) Z[o] = v;
        v = act_f(v, a.act);
        if (a.Mul) v *= a.Mul[z * a.gY + o];
        Y[o] = v;
    }
}
// thin Linear + LayerNorm over exactly 256 output columns (one thread per column, the norm is a workgroup reduction per
// row): the acting net's first block once the embedding and the first attention sublayer are folded into one
// obs -> d_model Linear (tvc_sac.hip: derive_infer)
__global__ void __launch_bounds__(256) thin_fwd_ln_kernel(ThinArgs a, const float* __restrict__ gamma, const float* __restrict__ beta) {
    __shared__ float xs[THIN_ROWS][THIN_K];
    __shared__ float part[2][4][THIN_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row0 = blockIdx.x * THIN_ROWS;
    if (tid < THIN_ROWS * THIN_K) xs[tid / THIN_K][tid % THIN_K] = thin_x(a, a.X, a.X2, row0 + tid / THIN_K, tid % THIN_K);
    const float* W = a.W + (long)tid * a.K;
    float w[THIN_K];
#pragma unroll
    for (int k = 0; k < THIN_K; ++k) {
        const float t = W[min(k, a.K - 1)];
        w[k] = k < a.K ? t : 0.0f;
    }
    const float b = a.bias ? a.bias[tid] : 0.0f;
    const float gm = gamma[tid], bt = beta[tid];
    __syncthreads();
    float v[THIN_ROWS];
#pragma unroll
    for (int r = 0; r < THIN_ROWS; ++r) {
        float t = b;
#pragma unroll
        for (int k = 0; k < THIN_K; ++k) t = fmaf(xs[r][k], w[k], t);
        v[r] = act_f(t, a.act);
        float s = v[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) part[0][wave][r] = s;
    }
    __syncthreads();
    float mean[THIN_ROWS];
#pragma unroll
    for (int r = 0; r < THIN_ROWS; ++r) {
        mean[r] = (part[0][0][r] + part[0][1][r] + part[0][2][r] + part[0][3][r]) * (1.0f / 256.0f);
        const float d = v[r] - mean[r];
        float s = d * d;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) part[1][wave][r] = s;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < THIN_ROWS; ++r) {
        const float rstd = rsqrtf((part[1][0][r] + part[1][1][r] + part[1][2][r] + part[1][3][r]) * (1.0f / 256.0f) + 1e-5f);
        const int row = row0 + r;
        if (row < a.M) a.Y[(long)row * 256 + tid] = (v[r] - mean[r]) * rstd * gm + bt;
    }
}
// Derived weights of the acting net that are cheap matrix-vector work, one launch over (output row o, encoder layer l):
//   every layer:  b_ov[o] = sum_j W_o[o, j] b_v[j] + b_o[o]                                  (bias of the folded attention W_o W_v)
//   layer 0 only: W'[o, k] = W_e[o, k] + sum_j W_ov[o, j] W_e[j, k];  b'[o] = be[o] + sum_j W_ov[o, j] be[j] + b_ov[o], be = b_e + pe0:
//                 the embedding (+ PE(0)) and the first folded attention sublayer x + W_ov x + b_ov as ONE obs -> d Linear
// (W_ov itself comes from the grouped GEMM launched before).  OV layout: layer l at l * ostride = [W_ov (d x d) | b_ov (d)].
struct FoldArgs {
    const float* P; float* OV; const float* pe0;
    long o_w, o_b, v_b, layer_stride, ostride;   // offsets of layer 0's out_proj weight / bias, v_proj bias in P; strides per layer
    int embed; long e_w, e_b, e_off;             // embedding fold on: offsets of W_e, b_e in P and of [W' | b'] in OV
    int d, obs;
};
__global__ void __launch_bounds__(256) fold_kernel(FoldArgs a, Ticks tk) {
    TVC_LEARNER_PRIO();
    // thread j carries W_o[o, j] b_v[j] and, on layer 0, W_ov[o, j] times row j of [W_e | be] (obs + 1 <= 17 values); then a
    // workgroup reduction per value
    __shared__ float part[4][THIN_K + 2];
    const int o = blockIdx.x, l = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, obs = a.obs;
    if (o == 0 && l == 0 && tid == 0) run_ticks(tk);
    const float* Wo = a.P + a.o_w + l * a.layer_stride;
    const float* bv = a.P + a.v_b + l * a.layer_stride;
    const float* bo = a.P + a.o_b + l * a.layer_stride;
    float* ov = a.OV + l * a.ostride;
    const bool emb = a.embed && l == 0;
    const float* We = a.P + a.e_w;
    const float* be_ = a.P + a.e_b;
    float acc[THIN_K + 2];
#pragma unroll
    for (int k = 0; k < THIN_K + 2; ++k) acc[k] = 0.0f;
    for (int j = tid; j < d; j += 256) {
        acc[THIN_K + 1] = fmaf(Wo[(long)o * d + j], bv[j], acc[THIN_K + 1]);
        if (emb) {
            const float w = ov[(long)o * d + j];
#pragma unroll
            for (int k = 0; k < THIN_K; ++k) {
                const float e = We[(long)j * obs + min(k, obs - 1)];
                acc[k] = fmaf(w, k < obs ? e : 0.0f, acc[k]);
            }
            acc[THIN_K] = fmaf(w, be_[j] + (a.pe0 ? a.pe0[j] : 0.0f), acc[THIN_K]);
        }
    }
#pragma unroll
    for (int k = 0; k < THIN_K + 2; ++k) {
        float v = acc[k];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) part[wave][k] = v;
    }
    __syncthreads();
    if (tid <= THIN_K + 1) {
        const float v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        const float bov = part[0][THIN_K + 1] + part[1][THIN_K + 1] + part[2][THIN_K + 1] + part[3][THIN_K + 1] + bo[o];
        if (tid == THIN_K + 1) ov[(long)d * d + o] = bov;
        if (emb) {
            float* Wout = a.OV + a.e_off;
            float* bout = Wout + (long)d * obs;
            if (tid < obs) Wout[(long)o * obs + tid] = We[(long)o * obs + tid] + v;
            if (tid == THIN_K) bout[o] = be_[o] + (a.pe0 ? a.pe0[o] : 0.0f) + v + bov;
        }
    }
}
// dW[n, k] = sum_m dZ[m, n] X[m, k]: 32 weight rows per workgroup, the 8 half-waves split the batch rows; gridDim.y > 1: the batch
// rows in gridDim.y slices of whole 64-row steps, summed with float atomics into a ZEROED dW (a 256-row batch was four dependent
// load rounds on 8 - 16 workgroups: 8 - 9 us for 50 kFLOP)
__global__ void __launch_bounds__(256) thin_wgrad_kernel(ThinArgs a) {
    TVC_LEARNER_PRIO();
    __shared__ float xs[64][THIN_K];
    __shared__ float red[8][32][THIN_K + 1];
    const int tid = threadIdx.x, l = tid & 31, rg = tid >> 5;
    const int n = blockIdx.x * 32 + l, nc = min(n, a.N - 1);
    const long z = blockIdx.z;
    const float* X = a.X + z * a.gX;
    const float* X2 = a.X2 ? a.X2 + z * a.gX2 : nullptr;
    const float* dZ = a.dZ + z * a.gY;
    float acc[THIN_K];
#pragma unroll
    for (int k = 0; k < THIN_K; ++k) acc[k] = 0.0f;
    const int steps = (a.M + 63) / 64, per = (steps + (int)gridDim.y - 1) / (int)gridDim.y;
    const int m_lo = (int)blockIdx.y * per * 64, m_hi = min(a.M, m_lo + per * 64);
    for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            xs[e / THIN_K][e % THIN_K] = thin_x(a, X, X2, m0 + e / THIN_K, e % THIN_K);
        }
        __syncthreads();
        float dz[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + rg + i * 8;
            const float v = dZ[(long)min(m, a.M - 1) * a.N + nc];
            dz[i] = m < a.M ? v : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < THIN_K; ++k) acc[k] = fmaf(dz[i], xs[rg + i * 8][k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < THIN_K; ++k) red[rg][l][k] = acc[k];
    __syncthreads();
    float* dW = a.dW + z * a.gDW;
    for (int e = tid; e < 32 * a.K; e += 256) {
        const int nl = e / a.K, k = e - nl * a.K;
        if (blockIdx.x * 32 + nl >= a.N) break;
        float s = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][nl][k];
        if (gridDim.y > 1) atomicAdd(dW + (long)(blockIdx.x * 32 + nl) * a.K + k, s);
        else dW[(long)(blockIdx.x * 32 + nl) * a.K + k] = s;
    }
}
// dX[m, k] = sum_n dZ[m, n] W[n, k]: one wave per row
__global__ void __launch_bounds__(256) thin_dgrad_kernel(ThinArgs a) {
    TVC_LEARNER_PRIO();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const long z = blockIdx.z;
    const float* dZ = a.dZ + z * a.gY + (long)row * a.N;
    const float* W = a.W + z * a.gW;
    float acc[THIN_K];
#pragma unroll
    for (int k = 0; k < THIN_K; ++k) acc[k] = 0.0f;
    const bool vec = (a.K & 3) == 0;  // weight rows are 16-byte aligned: K / 4 float4 loads per lane
    const int kq = a.K >> 2;
    for (int n0 = 0; n0 < a.N; n0 += 64) {
        const int n = n0 + lane, nc = min(n, a.N - 1);
        const float dv = dZ[nc];
        const float dz = n < a.N ? dv : 0.0f;
        if (vec) {
            const float4* wr = reinterpret_cast<const float4*>(W + (long)nc * a.K);
#pragma unroll
            for (int q = 0; q < THIN_K / 4; ++q) {
                const float4 w = wr[min(q, kq - 1)];
                acc[q * 4] = fmaf(dz, w.x, acc[q * 4]); acc[q * 4 + 1] = fmaf(dz, w.y, acc[q * 4 + 1]);
                acc[q * 4 + 2] = fmaf(dz, w.z, acc[q * 4 + 2]); acc[q * 4 + 3] = fmaf(dz, w.w, acc[q * 4 + 3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < THIN_K; ++k) acc[k] = fmaf(dz, W[(long)nc * a.K + min(k, a.K - 1)], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < THIN_K; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    float* dX = a.dX + z * a.gDX + (long)row * a.K;
    float mine = 0.0f;
#pragma unroll
    for (int k = 0; k < THIN_K; ++k) mine = lane == k ? acc[k] : mine;
    if (lane < a.K) dX[lane] = mine;
}

// ------------------------------------------------------------------ LayerNorm (eps 1e-5, biased variance): one wave per row
struct LnArgs {
    const float* X; float* Y; const float* gamma; const float* beta;
    float* mean; float* rstd;  // [M] saved for backward (may be null)
    int M, N;                  // N in {256, 512} (multiple of 256 handled as N/64 floats per lane, <= 8)
    long gX, gY, gP, gS;       // group strides: activations, params, stats
    DropArgs drop;             // dropout on the normalised output (policy head / critics: Linear-GELU-LN-Dropout)
    // optional output head behind the norm: headOut[M, headN] = Y . headW^T + headB (headN <= 4), saving its launch
    const float* headW; const float* headB; float* headOut; int headN;
    long gHW, gHO;             // group strides of the head's parameters / output
};
// one row by one wave (lane = 0..63); x = the row's input
template <int VPL, bool COH = false>  // values per lane = N / 64; COH: X was written by other workgroups of this launch (ln_tail)
__device__ __forceinline__ void ln_fwd_row(const LnArgs& a, int row, long z, int lane) {
    const float* x = a.X + z * a.gX + (long)row * a.N;
    float v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i += 4) {
        if (COH) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[i + j] = __hip_atomic_load(x + (i / 4) * 256 + lane * 4 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float4 t = *reinterpret_cast<const float4*>(x + (i / 4) * 256 + lane * 4);
            v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)a.N;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)a.N + 1e-5f);
    const float* gm = a.gamma + z * a.gP;
    const float* bt = a.beta + z * a.gP;
    float* y = a.Y + z * a.gY + (long)row * a.N;
    float hacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; i += 4) {
        const int c = (i / 4) * 256 + lane * 4;
        const float4 gg = *reinterpret_cast<const float4*>(gm + c), bb = *reinterpret_cast<const float4*>(bt + c);
        float4 o;
        o.x = (v[i] - mean) * rstd * gg.x + bb.x;
        o.y = (v[i + 1] - mean) * rstd * gg.y + bb.y;
        o.z = (v[i + 2] - mean) * rstd * gg.z + bb.z;
        o.w = (v[i + 3] - mean) * rstd * gg.w + bb.w;
        if (a.drop.ctr) {
            const unsigned key = drop_key(a.drop, (unsigned)z);
            o.x *= drop_factor(a.drop, key, row, c); o.y *= drop_factor(a.drop, key, row, c + 1);
            o.z *= drop_factor(a.drop, key, row, c + 2); o.w *= drop_factor(a.drop, key, row, c + 3);
        }
        *reinterpret_cast<float4*>(y + c) = o;
        if (a.headW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < a.headN) {
                    const float4 w = *reinterpret_cast<const float4*>(a.headW + z * a.gHW + (long)j * a.N + c);
                    hacc[j] += o.x * w.x + o.y * w.y + o.z * w.z + o.w * w.w;
                }
            }
        }
    }
    if (a.headW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) hacc[j] += __shfl_xor(hacc[j], s);
        }
        if (lane == 0)
            for (int j = 0; j < a.headN; ++j) a.headOut[z * a.gHO + (long)row * a.headN + j] = hacc[j] + a.headB[z * a.gHW + j];
    }
    if (lane == 0 && a.mean) {
        a.mean[z * a.gS + row] = mean;
        a.rstd[z * a.gS + row] = rstd;
    }
}
template <int VPL>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(LnArgs a) {
    TVC_LEARNER_PRIO();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    ln_fwd_row<VPL>(a, row, blockIdx.y, lane);
}

// LayerNorm(s) behind a Linear WITHOUT a launch of their own (update path): the Linear's output row is spread over gridDim.x
// column-tile workgroups, each of which publishes its tile (release fence), then takes a ticket on the row tile's counter; the
// workgroup that arrives LAST sees every tile (acquire fence) and normalises the 32 rows (8 per wave).  A second norm directly
// behind the first (norm2 -> feature_norm) and the output head behind the last norm ride along.  The counter resets itself.
struct LnTail {
    unsigned* cnt;   // [groups x row tiles], zero between launches; nullptr: no tail
    LnArgs n1, n2;   // n2.X == nullptr: one norm
};
template <int VPL>
__device__ __forceinline__ void ln_tail(const LnTail& t, int m0, int rows, int tiles_x, unsigned cnt_index, long z) {
    // The tile was stored with device-scope write-through stores (skinny_epilogue4<COH>) and is read back with device-scope loads:
    // no cache maintenance.  (Plain stores + a device-scope release fence = buffer_wbl2, a write-back of the whole L2 by every
    // workgroup: measured +15 us per launch, 752 instead of 536 us per update.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): this thread's tile stores have been performed ...
    __syncthreads();                                         // ... and so have those of every thread of the workgroup
    __shared__ unsigned tail_last;
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(t.cnt + cnt_index, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tail_last = prev == (unsigned)tiles_x - 1u ? 1u : 0u;
        if (tail_last) t.cnt[cnt_index] = 0u;  // ready for the next launch (stream order)
    }
    __syncthreads();
    if (!tail_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, per = rows / 4;
#pragma unroll 4
    for (int r = 0; r < per; ++r) ln_fwd_row<VPL, true>(t.n1, m0 + wave * per + r, z, lane);
    if (t.n2.X) {  // (each lane re-reads exactly the columns it wrote)
#pragma unroll 4
        for (int r = 0; r < per; ++r) ln_fwd_row<VPL>(t.n2, m0 + wave * per + r, z, lane);
    }
}

// split-K Linear (fast path) + the LayerNorm(s) / output head behind it, one launch (ln_tail)
template <int VPL>
__global__ void TVC_SKINNY_BOUNDS gemm_skinny_lnt_kernel(GemmArgs g, LnTail t) {
    warm_kernargs<sizeof(GemmArgs) + sizeof(LnTail)>();
    TVC_LEARNER_PRIO();
    __shared__ float red[4][32 * 33];
    int bx, by;
    xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, bx, by);
    skinny_body_fast<true, true, true>(g, bx, by, blockIdx.z, red);
    ln_tail<VPL>(t, by * 32, 32, gridDim.x, blockIdx.z * gridDim.y + by, blockIdx.z);
}

// any width up to 1024 (inference helpers: the 128-wide norm of the hierarchical goal policy): strided columns per lane
__global__ void __launch_bounds__(256) layernorm_fwd_any_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const long z = blockIdx.y;
    const float* x = a.X + z * a.gX + (long)row * a.N;
    float v[16];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        const float t = x[min(c, a.N - 1)];
        v[i] = c < a.N ? t : 0.0f;
        s += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)a.N;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float d = (lane + 64 * i) < a.N ? v[i] - mean : 0.0f;
        q += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)a.N + 1e-5f);
    float* y = a.Y + z * a.gY + (long)row * a.N;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i, cc = min(c, a.N - 1);
        const float o = (v[i] - mean) * rstd * a.gamma[z * a.gP + cc] + a.beta[z * a.gP + cc];
        if (c < a.N) y[c] = o;
    }
    if (lane == 0 && a.mean) {
        a.mean[z * a.gS + row] = mean;
        a.rstd[z * a.gS + row] = rstd;
    }
}

// backward of Y = LN(X): dX = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dY * gamma.
// Optional fusions: dX *= act'(Zp) (the LN input was act(Zp)); column sums of the result (bias grad of the
// producing Linear); dgamma / dbeta accumulated with float atomics (8 rows per block).
struct LnBwdArgs {
    const float* dY; const float* X; const float* gamma; const float* mean; const float* rstd;
    float* dX; float* dgamma; float* dbeta;
    const float* Zp; int dact;  // optional
    float* colsum;              // optional [N]
    int M, N;
    long gA, gP, gS;            // strides: activations (dY, X, dX, Zp), params (gamma, dgamma, dbeta, colsum), stats
    // dY given implicitly by an output head behind this norm: dY[m, c] = sum_j hdOut[m, j] hW[j, c] (hN <= 4), saving the
    // head's dX launch; dY is then unused
    const float* hdOut; const float* hW; int hN; long gHD, gHW;
    DropArgs dmask;             // this LayerNorm's output dropout: the incoming dY is masked first
    float* dXm; DropArgs omask; // optional second output dX * mask: the dZ of a producing Linear whose (dropped) output was
                                // added to a residual (dX itself stays unmasked for the residual path)
};
#ifndef LN_BWD_RPW
#define LN_BWD_RPW 1   // rows per wave (4 waves per workgroup): 1 = 64 workgroups at batch 256 (update 517 -> 498 us; 2: 517, 4: 567)
#endif
template <int VPL>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(LnBwdArgs a) {
    TVC_LEARNER_PRIO();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long z = blockIdx.y;
    const float* gm = a.gamma + z * a.gP;
    float gam[VPL], dg[VPL], db[VPL], cs[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(gm + (i / 4) * 256 + lane * 4);
        gam[i] = t.x; gam[i + 1] = t.y; gam[i + 2] = t.z; gam[i + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) { dg[i] = 0.f; db[i] = 0.f; cs[i] = 0.f; }
    const int row_base = blockIdx.x * (4 * LN_BWD_RPW) + wave * LN_BWD_RPW;  // one row per wave: the shortest dependency chain, 64+ workgroups at batch 256
    for (int rr = 0; rr < LN_BWD_RPW; ++rr) {
        const int row = row_base + rr;
        if (row >= a.M) break;
        const long off = z * a.gA + (long)row * a.N;
        const float mean = a.mean[z * a.gS + row], rstd = a.rstd[z * a.gS + row];
        float xh[VPL], gy[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i += 4) {
            const int c = (i / 4) * 256 + lane * 4;
            const float4 xv = *reinterpret_cast<const float4*>(a.X + off + c);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float ds[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.hW) {
                for (int j = 0; j < a.hN; ++j) {
                    const float dj = a.hdOut[z * a.gHD + (long)row * a.hN + j];
                    const float4 w = *reinterpret_cast<const float4*>(a.hW + z * a.gHW + (long)j * a.N + c);
                    ds[0] = fmaf(dj, w.x, ds[0]); ds[1] = fmaf(dj, w.y, ds[1]);
                    ds[2] = fmaf(dj, w.z, ds[2]); ds[3] = fmaf(dj, w.w, ds[3]);
                }
            } else {
                const float4 dv = *reinterpret_cast<const float4*>(a.dY + off + c);
                ds[0] = dv.x; ds[1] = dv.y; ds[2] = dv.z; ds[3] = dv.w;
            }
            if (a.dmask.ctr) {
                const unsigned key = drop_key(a.dmask, (unsigned)z);
#pragma unroll
                for (int j = 0; j < 4; ++j) ds[j] *= drop_factor(a.dmask, key, row, c + j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[i + j] = (xs[j] - mean) * rstd;
                dg[i + j] += ds[j] * xh[i + j];
                db[i + j] += ds[j];
                gy[i + j] = ds[j] * gam[i + j];
                s1 += gy[i + j];
                s2 += gy[i + j] * xh[i + j];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float m1 = s1 / (float)a.N, m2 = s2 / (float)a.N;
#pragma unroll
        for (int i = 0; i < VPL; i += 4) {
            const int c = (i / 4) * 256 + lane * 4;
            float o4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = rstd * (gy[i + j] - m1 - xh[i + j] * m2);
            if (a.Zp) {
                const float4 zv = *reinterpret_cast<const float4*>(a.Zp + off + c);
                o4[0] *= act_grad(zv.x, a.dact); o4[1] *= act_grad(zv.y, a.dact);
                o4[2] *= act_grad(zv.z, a.dact); o4[3] *= act_grad(zv.w, a.dact);
            }
            *reinterpret_cast<float4*>(a.dX + off + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            if (a.dXm) {  // the producing Linear's bias sits inside its dropout: its gradient sums the MASKED values
                const unsigned key = drop_key(a.omask, (unsigned)z);
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] *= drop_factor(a.omask, key, row, c + j);
                *reinterpret_cast<float4*>(a.dXm + off + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[i + j] += o4[j];
        }
    }
    // column sums (dgamma, dbeta, bias gradient of the producing Linear): the four waves are reduced through LDS
    // first, so a block issues one float atomic per column and sum instead of four (same-address atomics from
    // 32+ blocks serialise in L2 and were 2/3 of this kernel's time)
    __shared__ float red[3][4][VPL * 64];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i / 4) * 256 + lane * 4 + (i & 3);
        red[0][wave][c] = dg[i]; red[1][wave][c] = db[i]; red[2][wave][c] = cs[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < VPL * 64; c += 256) {
        if (a.dgamma) {
            atomicAdd(a.dgamma + z * a.gP + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            atomicAdd(a.dbeta + z * a.gP + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
        if (a.colsum) atomicAdd(a.colsum + z * a.gP + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
    }
}

// ------------------------------------------------------------------ small heads: out[m, j] = X[m,:] . W[j,:] + b[j], j < NO (<= 4)
struct HeadArgs {
    const float* X; const float* W; const float* b; float* out;
    int M, K, NO;
    long gX, gW, gB, gO;
};
__global__ void __launch_bounds__(256) head_fwd_kernel(HeadArgs a) {
    TVC_LEARNER_PRIO();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const long z = blockIdx.y;
    const float* x = a.X + z * a.gX + (long)row * a.K;
    const float* W = a.W + z * a.gW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < a.K; k += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        for (int j = 0; j < a.NO; ++j) {
            const float4 wv = *reinterpret_cast<const float4*>(W + (long)j * a.K + k);
            acc[j] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
    for (int j = 0; j < a.NO; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    }
    if (lane == 0)
        for (int j = 0; j < a.NO; ++j) a.out[z * a.gO + (long)row * a.NO + j] = acc[j] + a.b[z * a.gB + j];
}
// backward of the head: dX[m,k] = sum_j dOut[m,j] W[j,k];  dW[j,k] += sum_m dOut[m,j] X[m,k];  db[j] += sum_m dOut[m,j]
struct HeadBwdArgs {
    const float* dOut; const float* X; const float* W;
    float* dX; float* dW; float* db;  // dW/db may be null (dgrad-only pass)
    int M, K, NO;
    long gD, gX, gW, gB;
};
__global__ void __launch_bounds__(256) head_bwd_dx_kernel(HeadBwdArgs a) {
    TVC_LEARNER_PRIO();
    const long z = blockIdx.y;
    const int row = blockIdx.x;
    const float* dO = a.dOut + z * a.gD + (long)row * a.NO;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.NO; ++j) d[j] = dO[j];
    for (int k = threadIdx.x; k < a.K; k += blockDim.x) {
        float v = 0.f;
        for (int j = 0; j < a.NO; ++j) v += d[j] * a.W[z * a.gW + (long)j * a.K + k];
        a.dX[z * a.gX + (long)row * a.K + k] = v;
    }
}
__global__ void __launch_bounds__(256) head_bwd_dw_kernel(HeadBwdArgs a) {
    TVC_LEARNER_PRIO();
    // thread per (j, k), blockIdx.z = chunk of 32 rows; partial sums meet through float atomics
    const long z = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.NO * a.K) return;
    const int j = idx / a.K, k = idx - j * a.K;
    const int m_begin = blockIdx.z * 32, m_end = min(a.M, m_begin + 32);
    float s = 0.f, sb = 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += 8) {  // 8 row loads in flight (clamped, zero-selected after the load)
        float dv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = min(m0 + u, a.M - 1);
            dv[u] = a.dOut[z * a.gD + (long)m * a.NO + j];
            xv[u] = a.X[z * a.gX + (long)m * a.K + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float d = m0 + u < m_end ? dv[u] : 0.0f;
            s = fmaf(d, xv[u], s);
            sb += d;
        }
    }
    atomicAdd(&a.dW[z * a.gW + idx], s);
    if (k == 0) atomicAdd(&a.db[z * a.gB + j], sb);
}

// ------------------------------------------------------------------ optimiser
// target = tau * online + (1 - tau) * target   (agent/...:1005-1010)
__global__ void polyak_kernel(float* __restrict__ tgt, const float* __restrict__ src, long n, float tau) {
    TVC_LEARNER_PRIO();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        tgt[i] = tau * src[i] + (1.0f - tau) * tgt[i];
}

}  // namespace tvcnn
