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

}  // namespace

extern "C" {

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
