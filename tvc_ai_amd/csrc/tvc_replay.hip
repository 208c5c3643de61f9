// libtvc_hip.so -- device-resident uniform replay buffer (K8).
// The shipped reference has no replay buffer; capacity / batch follow BASELINE.json (1M, 256) and the
// legacy surface store_transition / len(replay_buffer) (tests/test_agent.py:99-108).
// Layout: array of rows {s[obs], a[A], r, s2[obs], d}: a sampled row is one contiguous 96-byte segment, so a
// batch gather is 256 short contiguous reads; insert streams whole rows.
#include <algorithm>
#include <new>

#include "tvc_common.h"

namespace {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
    constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// st = {head, size, sample_counter, arrivals of the sample kernel, arrivals of the insert kernel}: device-resident so that a
// captured hipGraph of train steps keeps advancing.  The counters advance inside the kernel that used them: every workgroup
// reads them first, and the workgroup that ARRIVES LAST (one returning atomic per workgroup, at most 256 of them) writes the new
// values -- no other workgroup can still be about to read the old ones -- so insert and sample are one launch each, not two.
__device__ __forceinline__ bool last_arrival(unsigned* cnt) {
    __syncthreads();  // every thread of this workgroup has read the counters (and issued its stores)
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(cnt, 1u);
        last = prev == gridDim.x - 1 ? 1u : 0u;
        if (last) *cnt = 0u;  // ready for the next launch (stream order)
    }
    __syncthreads();
    return last != 0u;
}

__global__ void replay_insert_kernel(float* __restrict__ buf, long cap, long* __restrict__ st, const float* __restrict__ s,
                                     const float* __restrict__ a, const float* __restrict__ r, const float* __restrict__ s2,
                                     const unsigned char* __restrict__ term, const unsigned char* __restrict__ trunc, int n,
                                     int no, int na, int fused_advance) {
    const int W = 2 * no + na + 2;
    const long head = st[0], size = st[1];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * W; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / W), c = (int)(i - (long)row * W);
        float v;
        if (c < no) v = s[(long)row * no + c];
        else if (c < no + na) v = a[(long)row * na + (c - no)];
        else if (c == no + na) v = r[row];
        else if (c < 2 * no + na + 1) v = s2[(long)row * no + (c - no - na - 1)];
        else v = (term[row] | (trunc ? trunc[row] : 0)) ? 1.0f : 0.0f;  // done = terminated or truncated (scripts/train.py:582)
        buf[((head + row) % cap) * W + c] = v;
    }
    if (fused_advance && last_arrival(reinterpret_cast<unsigned*>(st + 4)) && threadIdx.x == 0) {
        st[0] = (head + n) % cap;
        st[1] = size + n > cap ? cap : size + n;
    }
}
__global__ void replay_advance_kernel(long* st, long cap, int n) {
    st[0] = (st[0] + n) % cap;
    st[1] = st[1] + n > cap ? cap : st[1] + n;
}

__global__ void replay_sample_kernel(const float* __restrict__ buf, long* __restrict__ st, unsigned seed_lo,
                                     unsigned seed_hi, unsigned ctr_lo, unsigned ctr_hi, int dev_counter,
                                     float* __restrict__ s, float* __restrict__ a,
                                     float* __restrict__ r, float* __restrict__ s2, float* __restrict__ d, int batch, int no,
                                     int na) {
    const int W = 2 * no + na + 2;
    const long size = st[1], counter = st[2];
    if (dev_counter) { ctr_lo = (unsigned)(counter & 0xFFFFFFFF); ctr_hi = (unsigned)(counter >> 32); }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < batch * W; i += gridDim.x * blockDim.x) {
        const int b = i / W, c = i - b * W;
        unsigned u[4];
        philox4x32_10((unsigned)b, 0u, ctr_lo, ctr_hi, seed_lo, seed_hi, u);
        const unsigned long long rnd = ((unsigned long long)u[0] << 32) | u[1];
        const long idx = size > 0 ? (long)(rnd % (unsigned long long)size) : 0;
        const float v = buf[idx * W + c];
        if (c < no) s[(long)b * no + c] = v;
        else if (c < no + na) a[(long)b * na + (c - no)] = v;
        else if (c == no + na) r[b] = v;
        else if (c < 2 * no + na + 1) s2[(long)b * no + (c - no - na - 1)] = v;
        else d[b] = v;
    }
    if (dev_counter && last_arrival(reinterpret_cast<unsigned*>(st + 3)) && threadIdx.x == 0) st[2] = counter + 1;
}

}  // namespace

struct tvc_replay {
    float* buf;
    long* st;  // device: head, size, sample counter
    long cap;
    int no, na, device;
};

extern "C" {

int tvc_replay_create(int64_t capacity, int32_t obs_dim, int32_t act_dim, int32_t device, tvc_replay** out) {
    if (!out) return tvc::set_error(TVC_EINVAL, "out is NULL");
    *out = nullptr;
    if (capacity < 1 || obs_dim < 1 || act_dim < 1) return tvc::set_error(TVC_EINVAL, "bad capacity / dims");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tvc::set_error(TVC_ENODEV, "no HIP device visible: libtvc_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return tvc::set_error(TVC_EINVAL, "device out of range");
    TVC_HIP_CHECK(hipSetDevice(device));
    tvc_replay* rb = new (std::nothrow) tvc_replay();
    if (!rb) return tvc::set_error(TVC_ENOMEM, "host allocation failed");
    rb->cap = capacity; rb->no = obs_dim; rb->na = act_dim; rb->device = device;
    const long bytes = capacity * (2 * obs_dim + act_dim + 2) * 4;
    hipError_t he = hipMalloc((void**)&rb->buf, bytes + 64);
    if (he != hipSuccess) {
        delete rb;
        return tvc::set_error(TVC_ENOMEM, "hipMalloc(%ld) failed: %s", bytes, hipGetErrorString(he));
    }
    rb->st = (long*)((char*)rb->buf + bytes);
    // rows AND counters start at zero: a sample drawn before the first insert gathers row 0 = an all-zero transition
    // (finite), never uninitialised memory
    he = hipMemset(rb->buf, 0, bytes + 64);
    if (he != hipSuccess) {
        (void)hipFree(rb->buf);
        delete rb;
        return tvc::set_error(TVC_EHIP, "hipMemset failed: %s", hipGetErrorString(he));
    }
    *out = rb;
    return 0;
}

void tvc_replay_destroy(tvc_replay* rb) {
    if (!rb) return;
    (void)hipSetDevice(rb->device);
    (void)hipFree(rb->buf);
    delete rb;
}

int64_t tvc_replay_size(const tvc_replay* rb) {
    if (!rb) return 0;
    long v = 0;
    (void)hipSetDevice(rb->device);
    if (hipMemcpy(&v, rb->st + 1, sizeof(long), hipMemcpyDeviceToHost) != hipSuccess) return -1;  // synchronises
    return v;
}

int tvc_replay_insert(tvc_replay* rb, const float* s, const float* a, const float* r, const float* s2, const uint8_t* term,
                      const uint8_t* trunc, int32_t n, void* stream) {
    if (!rb || !s || !a || !r || !s2 || !term) return tvc::set_error(TVC_EINVAL, "null argument");
    if (n < 1 || n > rb->cap) return tvc::set_error(TVC_EINVAL, "n must be in [1, capacity]");
    TVC_HIP_CHECK(hipSetDevice(rb->device));
    const int W = 2 * rb->no + rb->na + 2;
    const long total = (long)n * W;
    // up to 512 workgroups the counters advance inside this launch (last workgroup to arrive); a larger insert keeps one element
    // per thread and a second, one-thread launch: thousands of arrivals on one address would cost more than the launch they save
    const long blocks = (total + 255) / 256;
    const int fused = blocks <= 512 ? 1 : 0;
    hipLaunchKernelGGL(replay_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rb->buf, rb->cap, rb->st, s, a,
                       r, s2, term, trunc, n, rb->no, rb->na, fused);
    if (!fused) hipLaunchKernelGGL(replay_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rb->st, rb->cap, (int)n);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_replay_sample(tvc_replay* rb, int32_t batch, uint64_t seed, uint64_t counter, float* s, float* a, float* r, float* s2,
                      float* d, void* stream) {
    if (!rb || !s || !a || !r || !s2 || !d) return tvc::set_error(TVC_EINVAL, "null argument");
    if (batch < 1) return tvc::set_error(TVC_EINVAL, "batch must be >= 1");
    TVC_HIP_CHECK(hipSetDevice(rb->device));
    const int W = 2 * rb->no + rb->na + 2;
    const int dev_counter = counter == UINT64_MAX;  // auto-incrementing device counter (hipGraph-replayable)
    hipLaunchKernelGGL(replay_sample_kernel, dim3(std::min((batch * W + 255) / 256, 256)), dim3(256), 0, (hipStream_t)stream,
                       rb->buf, rb->st, (unsigned)(seed & 0xFFFFFFFFu), (unsigned)(seed >> 32), (unsigned)(counter & 0xFFFFFFFFu),
                       (unsigned)(counter >> 32), dev_counter, s, a, r, s2, d, batch, rb->no, rb->na);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// checkpoint support (synchronous): rows_dev receives / provides the first `size` rows in storage order
// ({s, a, r, s2, d} x row), meta = {head, size, sample counter}
int tvc_replay_export(tvc_replay* rb, float* rows_dev, int64_t meta[3]) {
    if (!rb || !meta) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(rb->device));
    TVC_HIP_CHECK(hipDeviceSynchronize());
    long st[3];
    TVC_HIP_CHECK(hipMemcpy(st, rb->st, sizeof(st), hipMemcpyDeviceToHost));
    meta[0] = st[0]; meta[1] = st[1]; meta[2] = st[2];
    if (rows_dev && st[1] > 0)
        TVC_HIP_CHECK(hipMemcpy(rows_dev, rb->buf, (size_t)st[1] * (2 * rb->no + rb->na + 2) * sizeof(float), hipMemcpyDeviceToDevice));
    return 0;
}
int tvc_replay_import(tvc_replay* rb, const float* rows_dev, const int64_t meta[3]) {
    if (!rb || !meta) return tvc::set_error(TVC_EINVAL, "null argument");
    if (meta[1] < 0 || meta[1] > rb->cap || meta[0] < 0 || meta[0] >= rb->cap || (meta[1] > 0 && !rows_dev))
        return tvc::set_error(TVC_EINVAL, "snapshot does not fit this buffer (capacity %ld)", rb->cap);
    TVC_HIP_CHECK(hipSetDevice(rb->device));
    TVC_HIP_CHECK(hipDeviceSynchronize());
    if (meta[1] > 0)
        TVC_HIP_CHECK(hipMemcpy(rb->buf, rows_dev, (size_t)meta[1] * (2 * rb->no + rb->na + 2) * sizeof(float), hipMemcpyDeviceToDevice));
    long st[3] = {(long)meta[0], (long)meta[1], (long)meta[2]};
    TVC_HIP_CHECK(hipMemcpy(rb->st, st, sizeof(st), hipMemcpyHostToDevice));
    return 0;
}

}  // extern "C"
