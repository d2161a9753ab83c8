// libyhbench: measurement kernels for bench.py and tools/ -- yardsticks, not part of
// the head's C ABI (libyolohead.so does not contain them and the package never loads
// this library).
//
// yhb_stream_copy: the plain streaming copy every HBM-bound kernel of the head is
// compared with.  A thread moves 16 bytes per step of a grid-stride loop, `unroll`
// loads in flight before the first store, so a wavefront instruction covers eight
// whole 128-byte lines; loads and stores can be non-temporal (the copied bytes are
// not read again, as with the head tensor of a training step).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void stream_copy_kernel(const v4f* __restrict__ src,
                                                          v4f* __restrict__ dst, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            v[u] = NT_LOAD ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT_STORE) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) {
        const v4f v = NT_LOAD ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT_STORE) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

// reads only: what the memory system delivers when nothing is written
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read_kernel(const v4f* __restrict__ src,
                                                          float* __restrict__ sink, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += src[i];
    // never true for finite data; keeps the loads alive
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) *sink = acc.x;
}

// reads n_r chunks of 16 bytes and writes the first n_w of them back out (n_w <= n_r): the
// traffic shape of a kernel that reads more than it writes, as a plain streaming kernel
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_rw_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst,
                                                        float* __restrict__ sink, long long n_r,
                                                        long long n_w) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n_r; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (i + u * stride < n_w) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else acc += v[u];
        }
    }
    for (; i < n_r; i += stride) {
        const v4f v = __builtin_nontemporal_load(src + i);
        if (i < n_w) __builtin_nontemporal_store(v, dst + i);
        else acc += v;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) *sink = acc.x;   // never true for finite data
}

// The head kernels' own read pattern without their arithmetic: a wavefront per image, a lane
// per cell (49 of 64 lanes), the cell's 30 channels as 30 four-byte loads 196 bytes apart,
// four images per workgroup.  Against stream_read_kernel on the same bytes it prices the
// access pattern itself.
__global__ __launch_bounds__(256) void read_cells_kernel(const float* __restrict__ pred,
                                                         float* __restrict__ sink, long long batch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long img = (long long)blockIdx.x * 4 + wave;
    if (img >= batch || lane >= 49) return;
    const float* p = pred + img * 1470 + lane;
    float v[30];
#pragma unroll
    for (int ch = 0; ch < 30; ++ch) v[ch] = p[ch * 49];
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < 30; ++ch) acc += v[ch];
    if (acc == 1.2345e38f) *sink = acc;
}

// the same bytes brought in as the workgroup's contiguous block (4 images, 23,520 bytes) in
// 16-byte loads, whole lines per wavefront instruction
__global__ __launch_bounds__(256) void read_blocks_kernel(const float* __restrict__ pred,
                                                          float* __restrict__ sink, long long batch) {
    const long long img0 = (long long)blockIdx.x * 4;
    const long long left = batch - img0;
    const int n4 = (int)(left < 4 ? left : 4) * 1470 / 4;      // the block starts 16-byte aligned
    const v4f* p = reinterpret_cast<const v4f*>(pred + img0 * 1470);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    v4f v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int i = threadIdx.x + 256 * u;
        v[u] = i < n4 ? p[i] : acc;
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) acc += v[u];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) *sink = acc.x;
}

}  // namespace

extern "C" {

// pattern 0: read_cells_kernel, pattern 1: read_blocks_kernel, on a (batch, 30, 7, 7) tensor
int yhb_read_head(const float* pred, float* sink, long long batch, int pattern, void* stream) {
    if (!pred || !sink || batch < 1 || ((uintptr_t)pred & 15)) return -1;
    const unsigned grid = (unsigned)((batch + 3) / 4);
    if (pattern == 0)
        hipLaunchKernelGGL(read_cells_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, sink, batch);
    else
        hipLaunchKernelGGL(read_blocks_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, sink, batch);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// reads src[0 .. rbytes) and writes dst[0 .. wbytes) = src[0 .. wbytes), wbytes <= rbytes, both
// multiples of 16 and the pointers 16-byte aligned: a streaming kernel that moves rbytes + wbytes,
// the yardstick for a kernel with that traffic.  blocks: workgroups of 256 threads (0: 4096).
int yhb_stream_rw(const void* src, size_t rbytes, void* dst, size_t wbytes, float* sink, int blocks,
                  void* stream) {
    if (!src || !dst || !sink || (rbytes & 15) || (wbytes & 15) || wbytes > rbytes ||
        ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return -1;
    if (blocks <= 0) blocks = 4096;
    hipLaunchKernelGGL((stream_rw_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const v4f*)src, (v4f*)dst, sink, (long long)(rbytes / 16), (long long)(wbytes / 16));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// dst[0 .. nbytes) = src[0 .. nbytes); both 16-byte aligned, nbytes a multiple of 16.
// blocks: workgroups of 256 threads (0: 16 per CU of a 256-CU part); unroll: 1, 2, 4 or 8;
// flags: bit 0 non-temporal loads, bit 1 non-temporal stores.
// Returns 0, -1 for a bad argument, or the hipError_t of the launch.
int yhb_stream_copy(void* dst, const void* src, size_t nbytes, int blocks, int unroll, int flags,
                    void* stream) {
    if (!dst || !src || (nbytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return -1;
    if (blocks <= 0) blocks = 256 * 16;
    const long long n16 = (long long)(nbytes / 16);
    hipStream_t st = (hipStream_t)stream;
#define YHB_CASE(U, L, S)                                                                     \
    hipLaunchKernelGGL((stream_copy_kernel<U, L, S>), dim3((unsigned)blocks), dim3(256), 0, st, \
                       (const v4f*)src, (v4f*)dst, n16)
#define YHB_FLAGS(U)                                                                          \
    switch (flags & 3) {                                                                      \
        case 0: YHB_CASE(U, false, false); break;                                             \
        case 1: YHB_CASE(U, true, false); break;                                              \
        case 2: YHB_CASE(U, false, true); break;                                              \
        default: YHB_CASE(U, true, true); break;                                              \
    }
    switch (unroll) {
        case 1: YHB_FLAGS(1) break;
        case 2: YHB_FLAGS(2) break;
        case 4: YHB_FLAGS(4) break;
        case 8: YHB_FLAGS(8) break;
        default: return -1;
    }
#undef YHB_FLAGS
#undef YHB_CASE
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// reads src[0 .. nbytes) and writes nothing (sink: one float the kernel never stores to
// for finite data)
int yhb_stream_read(const void* src, float* sink, size_t nbytes, int blocks, void* stream) {
    if (!src || !sink || (nbytes & 15) || ((uintptr_t)src & 15)) return -1;
    if (blocks <= 0) blocks = 256 * 16;
    hipLaunchKernelGGL((stream_read_kernel<4>), dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (const v4f*)src, sink, (long long)(nbytes / 16));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
