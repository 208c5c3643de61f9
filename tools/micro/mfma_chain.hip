// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 as a function of the number of independent accumulators between dependent
// ones (one wave per SIMD, no memory traffic).  build: hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ void __launch_bounds__(256) chain(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / D; ++r)
#pragma unroll
            for (int i = 0; i < D; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int D>
void run(float* out, unsigned long long* cyc, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(chain<D>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(chain<D>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("independent accumulators %2d, %d wave(s)/SIMD: %.1f cycles per MFMA per wave\n", D, waves_per_simd, (double)h[0] / (iters * 64.0));
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int w = 1; w <= 2; ++w) { run<1>(out, cyc, w); run<2>(out, cyc, w); run<4>(out, cyc, w); run<8>(out, cyc, w); run<16>(out, cyc, w); }
    return 0;
}
