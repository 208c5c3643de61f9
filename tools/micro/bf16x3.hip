// Microbenchmark for the split-operand acting kernel (tvc_actor_x3.h): an fp32 Linear computed on the bf16 matrix pipe by writing
// every operand as hi + mid + lo (three bf16 terms, 8 + 8 + 8 mantissa bits = the 24 of an fp32) and summing the SIX products
// whose weight is >= 2^-16 (hh, hm, mh, hl, lh, mm) with v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
//   part 1 (numerics): Y[16 rows][256] = X W^T, K = 256, four ways -- exact f32 MFMA (v_mfma_f32_16x16x4_f32), 6 products, 9 products,
//     3 products (hh, hm, mh: a 16-bit operand) -- against an fp64 host reference; two operand distributions.
//   part 2 (throughput): the inner loop of one (n-tile, k-block) triple = 3 ds_read_b128 + 6 MFMAs, two waves per SIMD, with F
//     filler VALU instructions per triple (the split of the next activation block / epilogue arithmetic).
// build: hipcc -O3 --offload-arch=gfx950 bf16x3.hip -o bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- the split: x = h + m + l, every term a bf16 (round to nearest even at each level; the residuals are exact in fp32)
__host__ __device__ inline unsigned bf16_rne_bits(float x) {  // upper 16 bits of the rounded value, as the upper half of a word
    unsigned u; memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u & 0xFFFF0000u;
}
__host__ __device__ inline float bits_f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ inline void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne_bits(x);
    const float r = x - bits_f(h);
    m = bf16_rne_bits(r);
    const float r2 = r - bits_f(m);
    l = bf16_rne_bits(r2);
}
__device__ __forceinline__ unsigned pack2(unsigned lo_word, unsigned hi_word) { return (lo_word >> 16) | hi_word; }

// W as three fragment images: frag[(t * KB + kb) * 3 + term][lane] = 8 bf16 of W[16 t + lane % 16][feat(kb, lane / 16, c)], c = 0..7
// with the k-permutation of the accumulator chain: feat(kb, q, c) = 32 kb + 16 (c / 4) + 4 q + (c % 4)
__host__ __device__ inline int feat(int kb, int q, int c) { return 32 * kb + 16 * (c >> 2) + 4 * q + (c & 3); }

template <int NPROD>
__global__ void __launch_bounds__(64) lin_x3(const float* __restrict__ X, const u32x4* __restrict__ Wf, float* __restrict__ Y) {
    const int lane = threadIdx.x, l15 = lane & 15, q = lane >> 4;
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < 8; ++kb) {
        u32x4 xh, xm, xl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3(X[l15 * 256 + feat(kb, q, 2 * p)], h0, m0, l0);
            split3(X[l15 * 256 + feat(kb, q, 2 * p + 1)], h1, m1, l1);
            xh[p] = pack2(h0, h1); xm[p] = pack2(m0, m1); xl[p] = pack2(l0, l1);
        }
        const bf16x8 bh = __builtin_bit_cast(bf16x8, xh), bm = __builtin_bit_cast(bf16x8, xm), bl = __builtin_bit_cast(bf16x8, xl);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, Wf[((t * 8 + kb) * 3 + 0) * 64 + lane]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, Wf[((t * 8 + kb) * 3 + 1) * 64 + lane]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, Wf[((t * 8 + kb) * 3 + 2) * 64 + lane]);
            f32x4 a = acc[t];
            // small terms first
            if (NPROD >= 9) {
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bl, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, bl, a, 0, 0, 0);
            }
            if (NPROD >= 6) {
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bl, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, bm, a, 0, 0, 0);
            }
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, bh, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bm, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bh, a, 0, 0, 0);
            acc[t] = a;
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[l15 * 256 + 16 * t + 4 * q + r] = acc[t][r];
}
__global__ void __launch_bounds__(64) lin_f32(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y) {
    const int lane = threadIdx.x, l15 = lane & 15, q = lane >> 4;
    for (int t = 0; t < 16; ++t) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 64; ++k)
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(16 * t + l15) * 256 + 4 * k + q], X[l15 * 256 + 4 * k + q], a, 0, 0, 0);
        for (int r = 0; r < 4; ++r) Y[l15 * 256 + 16 * t + 4 * q + r] = a[r];
    }
}

// ---- throughput of the inner loop
template <int FILL>
__global__ void __launch_bounds__(256, 2) loop_x3(const u32x4* __restrict__ img, float* out, unsigned long long* cyc, int n_iter) {
    __shared__ __attribute__((aligned(16))) u32x4 S[24 * 64];  // 24 KB: 8 triples
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 24 * 64; i += 256) S[i] = img[i];
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 xh = img[lane], xm = img[64 + lane], xl = img[128 + lane];
    float fill[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fill[i] = 1.0f + lane + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {  // 16 triples per iteration (two sweeps of the 8 in LDS)
            const u32x4 wh = S[((j & 7) * 3 + 0) * 64 + lane], wm = S[((j & 7) * 3 + 1) * 64 + lane], wl = S[((j & 7) * 3 + 2) * 64 + lane];
            const bf16x8 bh = __builtin_bit_cast(bf16x8, xh), bm = __builtin_bit_cast(bf16x8, xm), bl = __builtin_bit_cast(bf16x8, xl);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), am = __builtin_bit_cast(bf16x8, wm), al = __builtin_bit_cast(bf16x8, wl);
            f32x4 a = acc[j];
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, a, 0, 0, 0);
            acc[j] = a;
#pragma unroll
            for (int f = 0; f < FILL; ++f) fill[f & 7] = fmaf(fill[f & 7], 1.0001f, 0.5f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fill[i];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 16777216.0f; }
static float gauss(unsigned& s) { float u1 = rnd(s) + 1e-7f, u2 = rnd(s); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

template <int FILL>
static int run_loop(const u32x4* dimg, float* dout, unsigned long long* dcyc, int blocks, int n_iter) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(loop_x3<FILL>, dim3(blocks), dim3(256), 0, 0, dimg, dout, dcyc, n_iter);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(loop_x3<FILL>, dim3(blocks), dim3(256), 0, 0, dimg, dout, dcyc, n_iter);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(blocks);
    CK(hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double triples = 16.0 * n_iter;
    // s_memtime ticks at 100 MHz: cycles = ticks x (shader clock / 100 MHz) is unknown here, so report time per triple from the event
    const double us = ms * 1e3;
    const double flops = (double)blocks * 4 * triples * 6 * 16 * 16 * 32 * 2;
    printf("fill %2d: %8.1f us, %.1f ns per triple and wave pair slot, executed bf16 MFMA rate %.0f TFLOP/s (= %.0f 'fp32' TFLOP/s), memtime ticks %.0f\n",
           FILL, us, us * 1e3 / triples, flops / us * 1e-6, flops / 6 / us * 1e-6, mean);
    return 0;
}

int main() {
    // ---------------- part 1
    for (int dist = 0; dist < 3; ++dist) {
        std::vector<float> X(16 * 256), W(256 * 256);
        unsigned s = 12345u + dist;
        for (auto& v : X) v = dist == 0 ? gauss(s) : (dist == 1 ? gauss(s) * expf(4.f * gauss(s)) : 1.0f + 0.001f * gauss(s));
        for (auto& v : W) v = dist == 2 ? 0.0625f + 0.0001f * gauss(s) : (rnd(s) * 2.f - 1.f) / 16.f;
        std::vector<unsigned> Wf((size_t)16 * 8 * 3 * 64 * 4);
        for (int t = 0; t < 16; ++t) for (int kb = 0; kb < 8; ++kb) for (int lane = 0; lane < 64; ++lane) {
            const int n = 16 * t + (lane & 15), q = lane >> 4;
            unsigned h[8], m[8], l[8];
            for (int c = 0; c < 8; ++c) split3(W[n * 256 + feat(kb, q, c)], h[c], m[c], l[c]);
            for (int p = 0; p < 4; ++p) {
                Wf[((((size_t)t * 8 + kb) * 3 + 0) * 64 + lane) * 4 + p] = (h[2 * p] >> 16) | h[2 * p + 1];
                Wf[((((size_t)t * 8 + kb) * 3 + 1) * 64 + lane) * 4 + p] = (m[2 * p] >> 16) | m[2 * p + 1];
                Wf[((((size_t)t * 8 + kb) * 3 + 2) * 64 + lane) * 4 + p] = (l[2 * p] >> 16) | l[2 * p + 1];
            }
        }
        float *dX, *dW, *dY; u32x4* dWf;
        CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dY, 16 * 256 * 4)); CK(hipMalloc(&dWf, Wf.size() * 4));
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dWf, Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> ref(16 * 256), mag(16 * 256);
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 256; ++n) {
            double a = 0, b = 0;
            for (int k = 0; k < 256; ++k) { a += (double)X[m * 256 + k] * W[n * 256 + k]; b += fabs((double)X[m * 256 + k] * W[n * 256 + k]); }
            ref[m * 256 + n] = a; mag[m * 256 + n] = b;
        }
        std::vector<float> Y(16 * 256);
        auto report = [&](const char* name) {
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, sum = 0, mxabs = 0;
            for (int i = 0; i < 16 * 256; ++i) {
                const double e = fabs(Y[i] - ref[i]);
                mx = fmax(mx, e / mag[i]); sum += e / mag[i]; mxabs = fmax(mxabs, e);
            }
            printf("dist %d  %-12s max |err| / sum|x w| = %.3e   mean = %.3e   max abs err = %.3e\n", dist, name, mx, sum / (16 * 256), mxabs);
        };
        hipLaunchKernelGGL(lin_f32, dim3(1), dim3(64), 0, 0, dX, dW, dY); report("f32 mfma");
        hipLaunchKernelGGL(lin_x3<9>, dim3(1), dim3(64), 0, 0, dX, dWf, dY); report("bf16 x9");
        hipLaunchKernelGGL(lin_x3<6>, dim3(1), dim3(64), 0, 0, dX, dWf, dY); report("bf16 x6");
        hipLaunchKernelGGL(lin_x3<3>, dim3(1), dim3(64), 0, 0, dX, dWf, dY); report("bf16 x3");
        // host fp32 fmaf chain (what torch's CPU fp32 Linear is at best)
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 256; ++n) {
            float a = 0.f;
            for (int k = 0; k < 256; ++k) a = fmaf(X[m * 256 + k], W[n * 256 + k], a);
            Y[m * 256 + n] = a;
        }
        {
            double mx = 0, sum = 0;
            for (int i = 0; i < 16 * 256; ++i) { const double e = fabs(Y[i] - ref[i]); mx = fmax(mx, e / mag[i]); sum += e / mag[i]; }
            printf("dist %d  %-12s max |err| / sum|x w| = %.3e   mean = %.3e\n", dist, "host fmaf", mx, sum / (16 * 256));
        }
        (void)hipFree(dX); (void)hipFree(dW); (void)hipFree(dY); (void)hipFree(dWf);
    }
    // ---------------- part 2
    const int blocks = 512, n_iter = 400;
    std::vector<unsigned> img(24 * 64 * 4);
    unsigned s = 777u;
    for (auto& v : img) { unsigned h, m, l; split3(gauss(s), h, m, l); unsigned h2, m2, l2; split3(gauss(s), h2, m2, l2); v = (h >> 16) | h2; }
    u32x4* dimg; float* dout; unsigned long long* dcyc;
    CK(hipMalloc(&dimg, img.size() * 4)); CK(hipMalloc(&dout, blocks * 256 * 4)); CK(hipMalloc(&dcyc, blocks * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    if (run_loop<0>(dimg, dout, dcyc, blocks, n_iter)) return 1;
    if (run_loop<4>(dimg, dout, dcyc, blocks, n_iter)) return 1;
    if (run_loop<8>(dimg, dout, dcyc, blocks, n_iter)) return 1;
    if (run_loop<12>(dimg, dout, dcyc, blocks, n_iter)) return 1;
    if (run_loop<16>(dimg, dout, dcyc, blocks, n_iter)) return 1;
    if (run_loop<24>(dimg, dout, dcyc, blocks, n_iter)) return 1;
    return 0;
}
