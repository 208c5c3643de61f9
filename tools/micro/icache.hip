// Is a dependent launch's first pass over its code a string of instruction-cache misses?  One workgroup of one wave executes the
// SAME straight-line block of 8-byte VALU instructions twice (loop, not unrolled) and stamps both passes with s_memtime; the chain
// repeats the launch, so the first pass of launch k tells whether the instruction cache kept the code from launch k - 1.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/icache tools/micro/icache.hip ; run: tools/micro/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
#define F4 F1 F1 F1 F1
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16
#define F256 F64 F64 F64 F64
#define F1024 F256 F256 F256 F256
template <int KB>  // code of the block in KiB (128 instructions per KiB)
__global__ void straight(float* out, unsigned long long* stamps, int launch) {
    float x = threadIdx.x, c1 = 1.0001f, c2 = 0.5f;
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (KB >= 8) { F1024 }
        if (KB >= 16) { F1024 }
        if (KB >= 32) { F1024 F1024 }
        if (KB == 2) { F256 }
        if (KB == 4) { F256 F256 }
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) stamps[(launch * gridDim.x + blockIdx.x) * 2 + r] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void other(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
template <int KB>
void run(const char* name, int blocks) {
    float* out; unsigned long long* st; float* buf;
    const int L = 6;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&st, L * blocks * 2 * 8); hipMalloc(&buf, 1 << 20);
    hipMemset(st, 0, L * blocks * 2 * 8);
    for (int l = 0; l < L; ++l) {
        straight<KB><<<blocks, 64>>>(out, st, l);
        if (l == 2) other<<<1024, 256>>>(buf, 1 << 18);  // a different kernel in between
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(L * blocks * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%-8s %4d blocks: ", name, blocks);
    for (int l = 0; l < L; ++l) {
        double a = 0, b = 0;
        for (int k = 0; k < blocks; ++k) { a += h[(l * blocks + k) * 2]; b += h[(l * blocks + k) * 2 + 1]; }
        printf(" launch%d first %.0f second %.0f |", l, a / blocks, b / blocks);
    }
    printf("\n");
    hipFree(out); hipFree(st); hipFree(buf);
}
int main() {
    run<2>("2 KiB", 1); run<4>("4 KiB", 1); run<8>("8 KiB", 1); run<16>("16 KiB", 1); run<32>("32 KiB", 1);
    run<8>("8 KiB", 128); run<16>("16 KiB", 256);
    return 0;
}
