// Microbenchmark of the acting kernel's inner loop (tvc_actor_rows.h: ar_pass): per 16 KB weight tile 64 v_mfma_f32_16x16x4_f32
// fed by 16 ds_read_b128 per wave, software-pipelined one fragment group ahead.  Which ingredient costs what, one wave per SIMD:
//   mode bit 0: __syncthreads() per tile      bit 1: 4 global_load_lds_dwordx4 per wave per tile (next tile, L2-resident stream)
//   bit 2: NO fragment reads (MFMAs on stale registers)       bit 3: fragment reads 2 groups ahead instead of 1
// build: hipcc -O3 --offload-arch=gfx950 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SCHED() __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 8, 0)
__device__ __forceinline__ void mfma16(const float4 (&w)[4], const f32x4 xk, f32x4* acc) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wv = c == 0 ? w[j].x : (c == 1 ? w[j].y : (c == 2 ? w[j].z : w[j].w));
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xk[c], acc[j], 0, 0, 0);
        }
}
__device__ __forceinline__ void frag4(float4 (&w)[4], const float4* base, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = base[(4 * g + j) * 16];
}
template <int MODE>
__global__ void __launch_bounds__(256, 2) loop(const float4* __restrict__ tiles, float* out, unsigned long long* cyc, int n_tiles, int n_stream) {
    __shared__ __attribute__((aligned(16))) float4 Bs[2 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    for (int i = tid; i < 2048; i += 256) Bs[i] = tiles[i];
    __syncthreads();
    f32x4 acc[16], x[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; x[t] = (f32x4){1.f + lane, 0.5f, 0.25f, 2.f}; }
    float4 wa[4], wb[4];
    int sidx = 0;  // position in the weight stream (n_stream tiles, cyclic)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const float4* base = Bs + q * 256 + l15;
    frag4(wa, base, 0);
    if (MODE & 8) frag4(wb, base, 1);
    for (int ti = 0; ti < n_tiles; ti += 16) {
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            if (MODE & 4) {
                mfma16(wa, x[kt], acc); mfma16(wb, x[kt], acc + 4); mfma16(wa, x[kt], acc + 8);
                if (MODE & 1) __syncthreads();
                mfma16(wb, x[kt], acc + 12);
                continue;
            }
            frag4(wb, base, 1);
            mfma16(wa, x[kt], acc);
            SCHED();
            frag4(wa, base, 2);
            mfma16(wb, x[kt], acc + 4);
            SCHED();
            frag4(wb, base, 3);
            mfma16(wa, x[kt], acc + 8);
            SCHED();
            if (MODE & 1) __syncthreads();
            if (MODE & 2) {
                sidx = sidx + 1 < n_stream ? sidx + 1 : 0;
                const float4* src = tiles + (long)sidx * 1024 + wave * 256 + lane;
                float4* dst = Bs + ((ti + kt + 1) & 1) * 1024 + wave * 256;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 2048, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 3072, 0);
            }
            base = Bs + ((ti + kt + 1) & 1) * 1024 + q * 256 + l15;
            frag4(wa, base, 0);
            mfma16(wb, x[kt], acc + 12);
            SCHED();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + tid] = s + wa[0].x + wb[0].x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
// does the immediate offset of global_load_lds_dwordx4 advance BOTH addresses?  4 pieces of 1 KB per wave from one base pair
__global__ void __launch_bounds__(256) copycheck(const float4* __restrict__ src, float4* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float4 Bs[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* s = src + wave * 256 + lane;
    float4* d = Bs + wave * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 3072, 0);
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = Bs[i];
}
template <int MODE>
void run(const float4* tiles, float* out, unsigned long long* cyc, int wg_per_cu, const char* what, int n_stream = 64) {
    const int n_tiles = 4000, blocks = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(256), 0, 0, tiles, out, cyc, n_tiles, n_stream);
        hipDeviceSynchronize();
    }
    unsigned long long h[64];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 64; ++i) m += (double)h[i] / 64.0;
    printf("%d workgroup(s)/CU  %-58s %7.1f cycles per tile (64 MFMAs = 2048)\n", wg_per_cu, what, m / n_tiles);
}
int main() {
    float4* tiles; float* out; unsigned long long* cyc;
    hipMalloc(&tiles, 4000L * 16384); hipMemset(tiles, 0, 4000L * 16384);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    {
        float* hsrc = new float[4096]; float* hout = new float[4096];
        for (int i = 0; i < 4096; ++i) hsrc[i] = (float)i;
        hipMemcpy(tiles, hsrc, 16384, hipMemcpyHostToDevice);
        float4* o4; hipMalloc(&o4, 16384); hipMemset(o4, 0, 16384);
        hipLaunchKernelGGL(copycheck, dim3(1), dim3(256), 0, 0, tiles, o4);
        hipDeviceSynchronize();
        hipMemcpy(hout, o4, 16384, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 4096; ++i) bad += hout[i] != hsrc[i];
        printf("global_load_lds immediate offset applied to both addresses: %s (%d of 4096 words differ)\n", bad ? "NO" : "yes", bad);
        hipMemset(tiles, 0, 4000L * 16384);
    }
    for (int w = 1; w <= 2; ++w) {
        run<4>(tiles, out, cyc, w, "MFMAs only");
        run<5>(tiles, out, cyc, w, "MFMAs + barrier");
        run<0>(tiles, out, cyc, w, "MFMAs + fragment reads");
        run<1>(tiles, out, cyc, w, "MFMAs + fragment reads + barrier");
        run<2>(tiles, out, cyc, w, "MFMAs + fragment reads + LDS-DMA (no barrier)");
        run<3>(tiles, out, cyc, w, "MFMAs + fragment reads + barrier + LDS-DMA (the kernel)");
        run<3>(tiles, out, cyc, w, "the same over a 402-tile stream (6.4 MB, the policy)", 402);
        run<3>(tiles, out, cyc, w, "the same over a 4000-tile stream (64 MB)", 4000);
    }
    return 0;
}
