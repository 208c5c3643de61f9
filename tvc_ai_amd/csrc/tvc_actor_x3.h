// Acting pass of the reference-shape policy as ONE launch on the BF16 matrix pipe, fp32-exact: the "split-operand" row-owner kernel.
//
// Same network, same row-owner design and the same epilogues as actor_rows_kernel (tvc_actor_rows.h; reference:
// agent/multi_algorithm_agent.py:192-227, 765-789), but every Linear runs on v_mfma_f32_16x16x32_bf16 (16x the rate of the f32-input
// MFMA on gfx950) with BOTH operands written as a sum of three bf16 terms,
//     x = x_h + x_m + x_l        (round-to-nearest at each level, residuals exact in fp32: 8 + 8 + 8 = the 24 mantissa bits of an fp32)
// and the six products whose weight is >= 2^-16 of the leading one -- hh, hm, mh, hl, lh, mm -- accumulated in fp32 by the matrix pipe.
// Every bf16 x bf16 product is exact in fp32; what is dropped (ml, lm, ll) is <= 2^-24 of |x||w| per term, the rounding an fp32 FMA
// chain makes anyway.  Measured against an fp64 sum (tools/micro/bf16x3.hip, profiles/r03_i_bf16x3_microbench.txt): the six-product
// form is as close as the f32-input MFMA (max error / sum|x w| 1.7e-7 vs 1.6e-7 on N(0,1) x U(-1/16,1/16), K = 256; nine products
// change nothing; three products -- a 16-bit operand -- are 10x worse and are NOT used).  6 / 16 of the f32 MFMA cycles per Linear.
//
// Operand chain: out^T = W x^T as before.  A operand (16 n x 32 k): lane l holds W[n = l % 16][8 k-slots of q = l / 16];
// B operand (32 k x 16 m): lane l holds x[m = l % 16][the same 8 k-slots]; accumulator lane l holds out[m = l % 16][16 t + 4 q + r].
// The 8 k-slots of lane-quarter q in k-block kb are the accumulator registers (tile 2 kb, r = 0..3) and (tile 2 kb + 1, r = 0..3) of the
// producing Linear, i.e. input feature 32 kb + 16 (c / 4) + 4 q + (c % 4) for slot c: the weight stream is packed in that order
// (pack_x3_tile), so activations still go from accumulator registers to operand registers without a shuffle -- through the split
// (11 VALU instructions per two values: 3 v_cvt_pk_bf16_f32, 4 shift / mask, 4 subtract).
//
// Weight stream: "triples" of 3 KB = the hi, mid, lo fragments (64 lanes x 16 B each) of one (n-tile, k-block); 8 triples = one 24 KB
// tile, copied by global_load_lds_dwordx4 one tile ahead into a two-buffer LDS ring (48 KB), one barrier per tile; a pass walks
// k-blocks outer, n-tiles inner, so one split of the activation block feeds 6 MFMAs on each of the pass's n-tiles.
#pragma once
#include "tvc_actor_rows.h"
#include "tvc_actor_split.h"   // AsDrop: the counter-hash dropout masks of train-mode acting

namespace tvcnn {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#ifndef X3_PARK
#define X3_PARK 8   // activation tiles of the head parked in LDS (see actor_x3_kernel)
#endif

#ifndef X3_HQ
#define X3_HQ 4   // output slices of policy_head.4 (2 = halves, 4 = quarters); the stream is packed to match (rows_tables_x3)
#endif
#ifndef X3_NOCSE
#define X3_NOCSE 1   // the split of x is recomputed per FFN quarter instead of kept in 96 registers across them (see the layer loop)
#endif
#ifndef X3_NW
#define X3_NW 4   // waves (16 rows each) per workgroup sharing one tile stream
#endif
#ifndef X3_KT
#define X3_KT 8     // triples per ring slot (a slot = X3_KT x 3 KB); the packed stream's 24 KB tiles are X3_TRI / X3_KT consecutive slots
#endif
#ifndef X3_NBUF
#define X3_NBUF 2   // ring slots: the copy of slot s + X3_NBUF - 1 is issued at the barrier that opens slot s
#endif
#ifndef X3_ABL
#define X3_ABL 0   // timing-only ablations (wrong results): 1 = stream wraps after 768 KB (L2-resident), 2 = no copies after the first
#endif             // slots, 3 = no workgroup barrier (waits kept)
#ifndef X3_DEPTH
#define X3_DEPTH 2  // fragment register sets: the reads of triple g + X3_DEPTH - 1 are issued before the MFMAs of triple g
#endif
constexpr int X3_SLOT_BYTES = X3_KT * 3072;
constexpr int X3_CPW = X3_SLOT_BYTES / 1024 / X3_NW;  // copy instructions (1 KB each) per wave and slot
static_assert(X3_CPW * 1024 * X3_NW == X3_SLOT_BYTES && X3_CPW <= 8, "slot = waves x (<= 8 pieces of 1 KB: two address pairs + immediate offsets)");
struct X3Pipe {
    const char* tiles; char* Bs;
    int ti, n_tiles;   // in slots
    int cp_ti;         // slot whose copy the last x3_next started (>= n_tiles: none)
    int wave;
    unsigned goff;     // (wave * X3_CPW * 64 + lane) * 16: this lane's byte offset inside a slot
#ifdef X3_WAITS
    unsigned long long w_vm, w_lgkm, w_bar, w_issue;   // cycles of wave 0 in the copy wait / LDS wait / barrier / copy issue of x3_next
#endif
};
// piece c (1 KB per wave) of slot ti; the immediate offset (< 4096) advances both the global and the LDS address
template <int C>
__device__ __forceinline__ void x3_issue_piece(const X3Pipe& p, int ti) {
    // scalar slot base + one 32-bit lane offset (the saddr form); the product is made in SGPRs so that hipcc does not fold the
    // loop-invariant (stream base + lane offset) into a per-lane 64-bit pointer that it then spills and reloads once per slot
#if X3_ABL == 1
    const size_t toff = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((ti & 31) * X3_SLOT_BYTES + (C >> 2) * 4096);
#else
    const size_t toff = (size_t)(unsigned)__builtin_amdgcn_readfirstlane(ti * X3_SLOT_BYTES + (C >> 2) * 4096);
#endif
    const char* src = p.tiles + toff + p.goff;
    char* dst = p.Bs + (ti & (X3_NBUF - 1)) * X3_SLOT_BYTES + p.wave * (X3_SLOT_BYTES / X3_NW) + (C >> 2) * 4096;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, (C & 3) * 1024, 0);
}
__device__ __forceinline__ void x3_issue_tile(const X3Pipe& p, int ti) {
    x3_issue_piece<0>(p, ti);
    if (X3_CPW > 1) x3_issue_piece<1>(p, ti);
    if (X3_CPW > 2) x3_issue_piece<2>(p, ti);
    if (X3_CPW > 3) x3_issue_piece<3>(p, ti);
    if (X3_CPW > 4) x3_issue_piece<4>(p, ti);
    if (X3_CPW > 5) x3_issue_piece<5>(p, ti);
    if (X3_CPW > 6) x3_issue_piece<6>(p, ti);
    if (X3_CPW > 7) x3_issue_piece<7>(p, ti);
}
// Make slot p.ti readable, start the copy of slot p.ti + X3_NBUF - 1 into the slot that p.ti - 1 just vacated, return the readable slot.
// Before the barrier: my share of slot p.ti has landed and my fragment reads of slot p.ti - 1 are back.
// Not kept (measured, profiles/r03_i_x3_variants.txt): the pieces of the next slot issued one per triple instead of in a burst (2 001 vs
// 1 205 us at 65 536 rows: an LDS-DMA between fragment reads costs more than the burst's queueing); the slot through registers
// (global_load_dwordx4 + ds_write_b128 in two batches of three: 1 316 vs 1 191 us, 12 more live registers).
__device__ __forceinline__ const char* x3_next(X3Pipe& p) {
#ifdef X3_WAITS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#elif X3_ABL == 3
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_sched_barrier(0);
#elif X3_NBUF == 2
    __syncthreads();   // = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier, and hipcc's waitcnt pass knows the counters are zero behind it
#else
    constexpr int INFL = (X3_NBUF - 2) * X3_CPW;   // copy instructions of newer slots allowed to stay in flight
    static_assert(INFL <= 15, "vmcnt field");
    if (p.ti + X3_NBUF - 2 < p.n_tiles) __builtin_amdgcn_s_waitcnt(0x0070 | INFL);   // vmcnt(INFL) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
#endif
    const char* cur = p.Bs + (p.ti & (X3_NBUF - 1)) * X3_SLOT_BYTES;
    p.cp_ti = p.ti + X3_NBUF - 1;
#if X3_ABL == 2
    if (p.cp_ti >= X3_NBUF) p.cp_ti = p.n_tiles;
#endif
    if (p.cp_ti < p.n_tiles) x3_issue_tile(p, p.cp_ti);
#ifdef X3_WAITS
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    p.w_vm += t1 - t0; p.w_lgkm += t2 - t1; p.w_bar += t3 - t2; p.w_issue += t4 - t3;
#endif
    p.ti += 1;
    return cur;
}

// GELU of this kernel: the same rational erf as fast_erff (tvc_nn_kernels.h) with the quotient as p x rcp(q) (v_rcp_f32: 1 ulp) instead of
// the IEEE division sequence (10 instructions): the SIMD's instruction issue, not the matrix pipe, bounds this kernel
#ifndef X3_GELU_RCP
#define X3_GELU_RCP 1
#endif
__device__ __forceinline__ float x3_gelu(float x) {
#if X3_GELU_RCP
    const float z = fminf(fmaxf(x * 0.7071067811865476f, -4.0f), 4.0f);
    const float x2 = z * z;
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
    const float e = z * p * __builtin_amdgcn_rcpf(q);
    const float hx = 0.5f * x;
    return fmaf(hx, e, hx);
#else
    return gelu_f(x);
#endif
}
// ... and four values at a time with packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per issue slot;
// the library is built with -fno-slp-vectorize, so only explicit vector arithmetic is packed): 10 instead of 19 instructions per value
#ifndef X3_PACKED
#define X3_PACKED 3    // bit 0: GELU of the FFN, bit 1: LayerNorms, bit 2: GELU of policy_head.0, bit 3: the folded output head
#endif
__device__ __forceinline__ f32x4 x3_splat(float c) { return (f32x4){c, c, c, c}; }
template <bool PK>
__device__ __forceinline__ f32x4 x3_gelu4(const f32x4 x) {
    if (PK) {
    f32x4 z = x * 0.7071067811865476f;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = fminf(fmaxf(z[r], -4.0f), 4.0f);
    const f32x4 x2 = z * z;
    f32x4 p = x3_splat(-2.72614225801306e-10f);
    p = __builtin_elementwise_fma(p, x2, x3_splat(2.77068142495902e-08f));
    p = __builtin_elementwise_fma(p, x2, x3_splat(-2.10102402082508e-06f));
    p = __builtin_elementwise_fma(p, x2, x3_splat(-5.69250639462346e-05f));
    p = __builtin_elementwise_fma(p, x2, x3_splat(-7.34990630326855e-04f));
    p = __builtin_elementwise_fma(p, x2, x3_splat(-2.95459980854025e-03f));
    p = __builtin_elementwise_fma(p, x2, x3_splat(-1.60960333262415e-02f));
    f32x4 q = x3_splat(-1.45660718464996e-05f);
    q = __builtin_elementwise_fma(q, x2, x3_splat(-2.13374055278905e-04f));
    q = __builtin_elementwise_fma(q, x2, x3_splat(-1.68282697438203e-03f));
    q = __builtin_elementwise_fma(q, x2, x3_splat(-7.37332916720468e-03f));
    q = __builtin_elementwise_fma(q, x2, x3_splat(-1.42647390514189e-02f));
    f32x4 rq;
#pragma unroll
    for (int r = 0; r < 4; ++r) rq[r] = __builtin_amdgcn_rcpf(q[r]);
    const f32x4 e = z * p * rq;
    const f32x4 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
    }
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = x3_gelu(x[r]);
    return o;
}
// nn.LayerNorm over the 16 * NT features of each row like ar_layernorm (two-pass), packed arithmetic
template <int NT>
__device__ __forceinline__ void x3_layernorm(f32x4* __restrict__ u, const float* __restrict__ gamma, const float* __restrict__ beta, int q) {
#if X3_PACKED & 2
    f32x4 s4 = u[0];
#pragma unroll
    for (int t = 1; t < NT; ++t) s4 += u[t];
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (16.0f * NT));
    const f32x4 m4 = x3_splat(mean);
    f32x4 v4 = x3_splat(0.0f);
#pragma unroll
    for (int t = 0; t < NT; ++t) { const f32x4 d = u[t] - m4; v4 = __builtin_elementwise_fma(d, d, v4); }
    float v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    const float rstd = rsqrtf(v * (1.0f / (16.0f * NT)) + 1e-5f);
    const f32x4 r4 = x3_splat(rstd);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 g4 = ar_vec4(gamma, t, q), b4 = ar_vec4(beta, t, q);
        u[t] = __builtin_elementwise_fma((u[t] - m4) * r4, g4, b4);
        if (NT > 16 && (t & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (see ar_layernorm)
    }
#else
    ar_layernorm<NT>(u, gamma, beta, q);
#endif
}
// two fp32 -> the bf16 pair of their three terms (low half = first value)
__device__ __forceinline__ unsigned x3_cvt_pk(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& H, unsigned& M, unsigned& L) {
    H = x3_cvt_pk(a, b);
    const float ra = a - __uint_as_float(H << 16), rb = b - __uint_as_float(H & 0xFFFF0000u);
    M = x3_cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(M << 16), sb = rb - __uint_as_float(M & 0xFFFF0000u);
    L = x3_cvt_pk(sa, sb);
}
struct X3Op { u32x4_t h, m, l; };
// B operand of one k-block from two accumulator tiles (k-slots c = 0..3 <- a, 4..7 <- b)
__device__ __forceinline__ X3Op x3_split(const f32x4 a, const f32x4 b) {
    unsigned h[4], m[4], l[4];
    x3_split2(a[0], a[1], h[0], m[0], l[0]);
    x3_split2(a[2], a[3], h[1], m[1], l[1]);
    x3_split2(b[0], b[1], h[2], m[2], l[2]);
    x3_split2(b[2], b[3], h[3], m[3], l[3]);
    X3Op o;
    o.h = (u32x4_t){h[0], h[1], h[2], h[3]}; o.m = (u32x4_t){m[0], m[1], m[2], m[3]}; o.l = (u32x4_t){l[0], l[1], l[2], l[3]};
    return o;
}
__device__ __forceinline__ void x3_frag(u32x4_t (&w)[3], const char* base, int j) {
#pragma unroll
    for (int t = 0; t < 3; ++t) w[t] = *reinterpret_cast<const u32x4_t*>(base + (3 * j + t) * 1024);
}
#define X3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, B), C, 0, 0, 0)
// acc += W x^T over the six kept products, small terms first
__device__ __forceinline__ f32x4 x3_mfma6(const u32x4_t (&w)[3], const X3Op& x, f32x4 acc) {
    acc = X3_MFMA(w[2], x.h, acc);
    acc = X3_MFMA(w[0], x.l, acc);
    acc = X3_MFMA(w[1], x.m, acc);
    acc = X3_MFMA(w[1], x.h, acc);
    acc = X3_MFMA(w[0], x.m, acc);
    acc = X3_MFMA(w[0], x.h, acc);
    return acc;
}
// One Linear (or a slice of one): acc[t] (t < NT) += W[16 t .. 16 t + 15][32 KB inputs] . x^T, x = 2 KB accumulator tiles of the producer.
// Stream order: k-block outer, n-tile inner; NT * KB is a multiple of 8 (whole tiles).  The fragments of triple g + 1 are read while
// the six MFMAs of triple g run (two register sets); the tile barrier sits in front of the first read of a tile.
// PIPE: the split of k-block kb + 1 is made in the shadow of block kb's MFMAs (12 more registers); otherwise at the block's start.
// PARK0 >= 0: input tiles PARK0 .. are not in registers but parked in LDS (park[(i - PARK0) * (64 * X3_NW)], this thread's slot), see the head.
template <int NT, int KB, bool PIPE = true, int PARK0 = -1>
__device__ __forceinline__ void x3_pass(X3Pipe& p, const f32x4* __restrict__ x, f32x4* __restrict__ acc, unsigned lane16,
                                        const f32x4* park = nullptr) {
    static_assert((NT * KB) % X3_KT == 0, "a pass is a whole number of ring slots");
    constexpr int VPT = (44 + NT - 1) / NT;  // VALU instructions of the NEXT block's split placed behind each triple of this block
    auto src = [&](int i) -> f32x4 { return (PARK0 >= 0 && i >= PARK0) ? park[(i - PARK0) * (64 * X3_NW)] : x[i]; };
    // fragment ring: X3_DEPTH register sets; the reads of triple g + X3_DEPTH - 1 go out before the MFMAs of triple g.  A read of the
    // first triple of a tile needs that tile's barrier first: x3_next is called when the read-ahead index crosses a tile boundary.
    constexpr int D = X3_DEPTH, NTRI = NT * KB;
    u32x4_t w[D][3];
    const char* base = x3_next(p) + lane16;
#pragma unroll
    for (int a = 0; a < D - 1 && a < NTRI; ++a) x3_frag(w[a], base, a);
    static_assert(D - 1 <= X3_KT, "the read-ahead stays inside the first slot of a pass");
    X3Op xo = x3_split(src(0), src(1));
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        // without this fence hipcc hoists the splits of ALL k-blocks (12 registers each) in front of the pass and spills around them
        __builtin_amdgcn_sched_barrier(0);
        if (!PIPE && kb > 0) {
            xo = x3_split(src(2 * kb), src(2 * kb + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        X3Op xn = xo;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int g = kb * NT + t, ga = g + D - 1;
            if (ga < NTRI) {
                if (ga % X3_KT == 0) base = x3_next(p) + lane16;
                x3_frag(w[ga % D], base, ga % X3_KT);
            }
            acc[t] = x3_mfma6(w[g % D], xo, acc[t]);
            if (PIPE && t == 0 && kb + 1 < KB) xn = x3_split(src(2 * kb + 2), src(2 * kb + 3));
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);    // the three reads of the triple ahead first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    // ... then this triple's six MFMAs ...
            if (PIPE && kb + 1 < KB) __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);  // ... with a slice of the next block's split in their shadow
        }
        if (PIPE) xo = xn;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// TRAIN = false: the deterministic net with the attention and the embedding folded (stream of rows_tables_x3()).
// TRAIN = true : the net as trained, Dropout live at every site like the reference's get_action (agent/...:765; stream of
//                rows_tables_x3_train(), masks = AsDrop's counter hash, element for element those of the per-layer path and of DropMasks)
template <bool TRAIN>
#if X3_NW == 4
__global__ void __launch_bounds__(256, 2) actor_x3_kernel(ActRowsArgs a) {
#else
__global__ void __launch_bounds__(64 * X3_NW) actor_x3_kernel(ActRowsArgs a) {
#endif
    __shared__ __attribute__((aligned(16))) char Bs[X3_NBUF * X3_SLOT_BYTES];  // the weight ring (2 x 24 KB)
    // ... and 32 KB where the head parks the last eight tiles of its 512-wide activation (32 registers per lane) while it is the
    // B operand of policy_head.4: 128 (activation) + 64 (accumulators) + fragments + split do not fit 256 registers, and what hipcc
    // spills goes to scratch = the Infinity Cache and back (two workgroups = exactly the CU's 160 KB; 6 tiles: 1 227 us, 8: 1 172)
    __shared__ __attribute__((aligned(16))) f32x4 Park[X3_PARK * 64 * X3_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * (16 * X3_NW) + wave * 16 + l15;
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
    int tr = 6; (void)tr;  // AR_TRACE: stamps 6.. = s_memtime of wave 0 after every pass / epilogue
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    X3Pipe p{reinterpret_cast<const char*>(a.tiles), Bs, 0, a.n_tiles * (X3_TRI / X3_KT), 0, wv, (unsigned)((wv * X3_CPW * 64 + lane) * 16)};
#pragma unroll
    for (int s0 = 0; s0 < X3_NBUF - 1; ++s0) x3_issue_tile(p, s0);
    const unsigned lane16 = lane * 16;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // observation as the first B operand: k-slots 0..3 of lane-quarter q = obs[4 q + r] (zero beyond obs_dim), slots 4..7 zero
    f32x4 xin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 4 * q + r;
        const float v = a.obs[(long)rowc * a.obs_dim + min(k, a.obs_dim - 1)];
        xin[r] = k < a.obs_dim ? v : 0.0f;
    }
    const float* vec = a.vec;
    const float* __restrict__ tvec = a.tvec;
    AsDrop dr{};
    if (TRAIN) {
        dr.ctr = (unsigned)*a.drop_ctr; dr.seed = a.drop_seed; dr.thresh = a.drop_thresh; dr.scale = a.drop_scale;
        dr.rowmix = drop_mix((unsigned)row + 0x632BE5ABu);
    }
    f32x4 x[16];
    if (!TRAIN) {   // layer 0, first sublayer: embedding + PE(0) + folded attention + residual as ONE obs -> 256 Linear (two tiles), then norm1
        ar_zero<16>(x);
        const f32x4 xin2[2] = {xin, zero4};
        x3_pass<16, 1>(p, xin2, x, lane16); AR_T();
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] += ar_vec4(vec, t, q);
        x3_layernorm<16>(x, vec + 256, vec + 512, q); AR_T();
    } else {        // x = W_e obs + b_e + PE(0)   (agent/...:196-203)
        ar_zero<16>(x);
        const f32x4 xin2[2] = {xin, zero4};
        x3_pass<16, 1>(p, xin2, x, lane16); AR_T();
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] += ar_vec4(tvec, t, q) + ar_vec4(a.pe0, t, q);
    }
    for (int l = 0; l < a.n_layers; ++l) {
        const float* lv = vec + l * AR_LAYER_VEC;
        // (the masks' column terms (c >> 1) * K depend on the lane quarter only: hipcc hoists all of them out of the layer loop, ~60
        // registers it then spills in the prologue and reloads behind tile copies; an opaque copy of q per layer keeps them local)
        int qd = q;
        if (TRAIN) asm volatile("" : "+v"(qd));
        if (TRAIN) {
            // self-attention at sequence length 1 = out_proj(dropout_heads(v_proj(x))): the attention-weight dropout zeroes / rescales whole
            // heads of V (32 columns = two tiles each); then dropout1, residual, norm1
            const float* tl = tvec + 256 + 512 * l;
            const unsigned kv = dr.key(1 + 6 * l), ko = dr.key(2 + 6 * l);
            f32x4 v[16];
            ar_zero<16>(v);
            x3_pass<16, 8>(p, x, v, lane16); AR_T();
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = (v[t] + ar_vec4(tl, t, q)) * dr.f(kv, (unsigned)(t >> 1));
            // x (64 registers), V (64) and out_proj's accumulators (64) would be live together: the upper half of x waits in the head's
            // LDS parking area (idle until the head) while out_proj runs
            f32x4* xpark = Park + tid;
#pragma unroll
            for (int t = 0; t < X3_PARK; ++t) xpark[t * (64 * X3_NW)] = x[16 - X3_PARK + t];
            f32x4 o[16];
            ar_zero<16>(o);
            x3_pass<16, 8, false>(p, v, o, lane16); AR_T();
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                o[t] += ar_vec4(tl + 256, t, q);
                dr.tile(ko, t, qd, o[t]);
                if (t >= 16 - X3_PARK) x[t] = xpark[(t - (16 - X3_PARK)) * (64 * X3_NW)] + o[t];
                else x[t] += o[t];
            }
            x3_layernorm<16>(x, lv + 256, lv + 512, q); AR_T();
        } else if (l > 0) {  // x = norm1(x + W_ov x + b_ov)
            f32x4 acc[16];
            ar_zero<16>(acc);
            x3_pass<16, 8>(p, x, acc, lane16); AR_T();
#pragma unroll
            for (int t = 0; t < 16; ++t) x[t] += acc[t] + ar_vec4(lv, t, q);
            x3_layernorm<16>(x, lv + 256, lv + 512, q); AR_T();
        }
        // x = norm2(x + W2 gelu(W1 x + b1) + b2), the 512 hidden units in four quarters of 128 (32 registers of hidden activation)
        f32x4 acc2[16];
        ar_zero<16>(acc2);
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            f32x4 h[8];
            ar_zero<8>(h);
#if X3_NOCSE
            // hipcc otherwise keeps the split of x (8 k-blocks x 12 registers) alive across the four quarters to save 3 x 352 VALU
            // instructions, and spills around it: the VALU count does not bound this kernel, the scratch traffic does
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("" : "+v"(x[t]));
#endif
            x3_pass<8, 8>(p, x, h, lane16); AR_T();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 b4 = ar_vec4(lv + 768 + 128 * quarter, t, q);
#if X3_PACKED & 1
                h[t] = x3_gelu4<true>(h[t] + b4);
#else
#pragma unroll
                for (int r = 0; r < 4; ++r) h[t][r] = x3_gelu(h[t][r] + b4[r]);
#endif
                if (TRAIN) dr.tile(dr.key(4 + 6 * l), 8 * quarter + t, qd, h[t]);  // the FFN's dropout (hidden units 128 quarter + ...)
            }
            AR_T();
            x3_pass<16, 4>(p, h, acc2, lane16); AR_T();
        }
        const unsigned k2 = TRAIN ? dr.key(5 + 6 * l) : 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            f32x4 y = acc2[t] + ar_vec4(lv + 1280, t, q);
            if (TRAIN) dr.tile(k2, t, qd, y);  // dropout2
            x[t] += y;
        }
        x3_layernorm<16>(x, lv + 1536, lv + 1792, q); AR_T();
    }
    const float* tv = vec + a.n_layers * AR_LAYER_VEC;
    x3_layernorm<16>(x, tv, tv + 256, q); AR_T();  // feature_norm
    if (!TRAIN && a.use_se) {  // x *= sigmoid(fc2(relu(fc1(x))))
        const float* sv = tv + AR_TAIL_VEC;
        f32x4 s4[1] = {zero4};
        x3_pass<1, 8>(p, x, s4, lane16); AR_T();   // fc1 (256 -> 16): one tile
#pragma unroll
        for (int z = 0; z < X3_TRI / X3_KT; ++z) (void)x3_next(p);   // + the all-zero tile behind it in the stream
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sv + 4 * q);
        f32x4 yy[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) yy[0][r] = fmaxf(s4[0][r] + b1[r], 0.0f);
        yy[1] = zero4;
        f32x4 g[16];
        ar_zero<16>(g);
        x3_pass<16, 1>(p, yy, g, lane16); AR_T();  // fc2 (16 -> 256)
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
    x3_pass<16, 8, false>(p, x, pp, lane16); AR_T();
    x3_pass<16, 8, false>(p, x, pp + 16, lane16); AR_T();
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const f32x4 b4 = ar_vec4(tv + 512, t, q);
#if X3_PACKED & 4
        pp[t] = x3_gelu4<true>(pp[t] + b4);
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[t][r] = x3_gelu(pp[t][r] + b4[r]);
#endif
        if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    x3_layernorm<32>(pp, tv + 1024, tv + 1536, q); AR_T();
    if (TRAIN) {  // Dropout behind policy_head.2
        const unsigned k3 = dr.key(3 + 6 * a.n_layers);
        int qd = q;
        asm volatile("" : "+v"(qd));
#pragma unroll
        for (int t = 0; t < 32; ++t) dr.tile(k3, t, qd, pp[t]);
    }
    f32x4* park = Park + tid;
#pragma unroll
    for (int i = 0; i < X3_PARK; ++i) park[i * (64 * X3_NW)] = pp[32 - X3_PARK + i];
    // ---- 512 -> 512 GELU LayerNorm -> 2A outputs, the LayerNorm + output Linear folded into running sums (see actor_rows_kernel)
    // TRAIN: Dropout behind policy_head.6 sits between the folded LayerNorm and the output Linear,
    //   out[o] = rstd (sum_n m_n g_n gW[o][n] - mean sum_n m_n gW[o][n]) + sum_n m_n bW[o][n] + b8[o]      (see actor_split_kernel)
    float s1 = 0.0f, s2 = 0.0f, d[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {0.f, 0.f, 0.f, 0.f}, em[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned k6 = TRAIN ? dr.key(5 + 6 * a.n_layers) : 0u;
    const float* bw = TRAIN ? tvec + 256 + 512 * a.n_layers : nullptr;  // bW[o][n] = beta6[n] W8[o][n], then b8[4]
    // policy_head.4 in X3_HQ output slices of 16 * HN features (4 quarters: 32 accumulators instead of 64 beside the 96 + 32 parked
    // registers of the activation -- with halves hipcc kept 24 activation tiles in scratch and reloaded each behind a fresh tile copy)
    constexpr int HN = 32 / X3_HQ;
#pragma unroll
    for (int part = 0; part < X3_HQ; ++part) {
        f32x4 a2[HN];
        ar_zero<HN>(a2);
        x3_pass<HN, 16, false, 32 - X3_PARK>(p, pp, a2, lane16, park); AR_T();
#pragma unroll
        for (int t = 0; t < HN; ++t) {
            const int tt = HN * part + t;
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
                const float v = x3_gelu(a2[t][r] + b4[r]);
                s1 += v;
                s2 = fmaf(v, v, s2);
                const float m = TRAIN ? dr.f(k6, (unsigned)(16 * tt + 4 * q + r)) : 1.0f;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    d[o] = fmaf(v * m, gw[o][r], d[o]);
                    if (TRAIN) { gm[o] = fmaf(m, gw[o][r], gm[o]); em[o] = fmaf(m, bwv[o][r], em[o]); }
                }
            }
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (keeps the vector loads of four tiles, not sixteen, in flight)
        }
        // a slice's epilogue must not sink into the next slice's pass (hipcc moves it towards the use of its sums, behind that
        // pass: its accumulators live beside the next ones, the hoisted vector loads spilled, every reload a vmcnt(0) behind a fresh copy)
        asm volatile("" : "+v"(s1), "+v"(s2), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
        if (TRAIN) asm volatile("" : "+v"(gm[0]), "+v"(gm[1]), "+v"(gm[2]), "+v"(gm[3]), "+v"(em[0]), "+v"(em[1]), "+v"(em[2]), "+v"(em[3]));
        __builtin_amdgcn_sched_barrier(0);
    }
#define X3_RED(v) v += __shfl_xor(v, 16); v += __shfl_xor(v, 32)
    X3_RED(s1); X3_RED(s2);
#pragma unroll
    for (int o = 0; o < 4; ++o) { X3_RED(d[o]); if (TRAIN) { X3_RED(gm[o]); X3_RED(em[o]); } }
#undef X3_RED
    const float mean = s1 * (1.0f / 512.0f);
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / 512.0f) - mean * mean, 0.0f) + 1e-5f);
    float out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
        out[o] = TRAIN ? rstd * (d[o] - mean * gm[o]) + em[o] + bw[2048 + o] : rstd * (d[o] - mean * tv[5636 + o]) + tv[5632 + o];
    if (a.stamps && tid == 0) {
        a.stamps[AR_STAMPS * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        a.stamps[AR_STAMPS * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
#ifdef X3_WAITS
        static_assert(AR_STAMPS >= 96, "X3_WAITS needs -DAR_TRACE (96 stamp slots per workgroup)");
        a.stamps[AR_STAMPS * blockIdx.x + 90] = p.w_vm; a.stamps[AR_STAMPS * blockIdx.x + 91] = p.w_lgkm;
        a.stamps[AR_STAMPS * blockIdx.x + 92] = p.w_bar; a.stamps[AR_STAMPS * blockIdx.x + 93] = p.w_issue;
#endif
    }
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

}  // namespace tvcnn
