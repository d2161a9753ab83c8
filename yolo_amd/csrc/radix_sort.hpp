// Segmented, stable least-significant-digit radix sort of 64-bit keys in global memory, on a
// bit range [bit_lo, bit_hi) of the keys, eight bits per pass.  `nseg` segments of `n` keys
// each lie back to back; every segment is sorted by itself.  Used where a sort does not fit a
// CU's LDS: the candidates of an image above 2,048 (nms_large_kernels.hpp; one segment per
// image) and the ranking of a dataset's predictions for the mAP (eval.py:121; one segment).
// Keys that are equal on the bit range keep their input order, so a payload in the low bits
// (an index that rises with the position) comes out in order without being looked at.
//
// A pass is three launches over tiles of RADIX_TILE keys:
//   1  radix_hist_kernel     digit counts of every tile -> hist[seg][digit][tile]
//   2  radix_scan_kernel     exclusive prefix sum over (digit, tile) in that order, per segment:
//                            where each tile's keys of each digit go
//   3  radix_scatter_kernel  every wavefront owns a contiguous eighth of its tile and walks it
//                            64 keys at a time in order; the lanes that hold the same digit find
//                            each other with eight ballots (one per bit), rank themselves by
//                            popcount, and the running count of the digit (per wavefront, in
//                            LDS) moves on -- stable, no atomics on the way to memory.
// Nothing is allocated and nothing is asked of the device on the host: every launch is legal
// under graph capture.  The caller provides two key buffers (ping, pong) and the histogram.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yh {

constexpr int RADIX_THREADS = 256;
constexpr int RADIX_WAVES = RADIX_THREADS / 64;
constexpr int RADIX_PER_WAVE = 512;                          // keys a wavefront owns: 8 rounds of 64
constexpr int RADIX_TILE = RADIX_WAVES * RADIX_PER_WAVE;     // 2,048 keys per workgroup

inline long long radix_tiles(long long n) { return (n + RADIX_TILE - 1) / RADIX_TILE; }
// unsigned ints of the histogram for nseg segments of n keys
inline size_t radix_hist_words(long long nseg, long long n) {
    return (size_t)nseg * 256 * (size_t)radix_tiles(n);
}

// lanes of the wavefront whose `digit` equals mine (all 64 lanes must call; a lane that holds no
// key passes active = false and is in nobody's set)
__device__ __forceinline__ unsigned long long radix_peers(unsigned int digit, bool active) {
    unsigned long long peers = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return active ? peers : 0ull;
}

static __global__ __launch_bounds__(RADIX_THREADS) void radix_hist_kernel(
    const unsigned long long* __restrict__ keys, unsigned int* __restrict__ hist, long long n,
    int tiles, int shift) {
    __shared__ unsigned int h[256];
    const long long seg = blockIdx.x / (unsigned)tiles;      // one-dimensional grid: seg * tiles + tile
    const int tile = (int)(blockIdx.x - seg * (unsigned)tiles);
    h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long* K = keys + seg * n;
    const long long base = (long long)tile * RADIX_TILE;
#pragma unroll
    for (int u = 0; u < RADIX_TILE / RADIX_THREADS; ++u) {
        const long long i = base + u * RADIX_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned int)(K[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[((size_t)seg * 256 + threadIdx.x) * tiles + tile] = h[threadIdx.x];
}

// One workgroup of 1,024 threads per segment: exclusive prefix sum, in place, over the
// segment's 256 * tiles counts in (digit, tile) order.
static __global__ __launch_bounds__(1024) void radix_scan_kernel(unsigned int* __restrict__ hist, int tiles) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry_s;
    unsigned int* H = hist + (size_t)blockIdx.x * 256 * tiles;
    const long long total = 256LL * tiles;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    // 4 consecutive entries per thread and round
    for (long long base = 0; base < total; base += 4096) {
        const long long i0 = base + 4LL * threadIdx.x;
        unsigned int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u < total ? H[i0 + u] : 0u;
        const unsigned int mine = (v[0] + v[1]) + (v[2] + v[3]);
        // inclusive scan of `mine` over the wavefront, then over the 16 wavefronts
        unsigned int x = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        unsigned int before = carry_s;
        for (int q = 0; q < w; ++q) before += wsum[q];
        unsigned int run = before + x - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u < total) H[i0 + u] = run;
            run += v[u];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;    // the last thread's running total: everything so far
        __syncthreads();
    }
}

static __global__ __launch_bounds__(RADIX_THREADS) void radix_scatter_kernel(
    const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
    const unsigned int* __restrict__ hist, long long n, int tiles, int shift) {
    __shared__ unsigned int cnt[RADIX_WAVES][256];      // per wavefront: keys of each digit so far
    const long long seg = blockIdx.x / (unsigned)tiles;      // one-dimensional grid: seg * tiles + tile
    const int tile = (int)(blockIdx.x - seg * (unsigned)tiles);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long* K = in + seg * n;
    unsigned long long* O = out + seg * n;
    const long long base = (long long)tile * RADIX_TILE + (long long)w * RADIX_PER_WAVE;
    unsigned long long key[RADIX_PER_WAVE / 64];
#pragma unroll
    for (int r = 0; r < RADIX_PER_WAVE / 64; ++r) {
        const long long i = base + 64 * r + lane;
        key[r] = i < n ? K[i] : 0ull;
    }
#pragma unroll
    for (int d = lane; d < 256; d += 64) cnt[w][d] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // first walk: this wavefront's count of every digit
#pragma unroll
    for (int r = 0; r < RADIX_PER_WAVE / 64; ++r) {
        const bool active = base + 64 * r + lane < n;
        const unsigned int d = (unsigned int)(key[r] >> shift) & 255u;
        const unsigned long long peers = radix_peers(d, active);
        if (active && (peers & ((1ull << lane) - 1ull)) == 0ull) cnt[w][d] += (unsigned int)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // where this wavefront's keys of digit d start: the tile's place for d plus the counts of the
    // wavefronts before this one; the wavefront's own counter then restarts from there
    unsigned int start[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int d = lane + 64 * u;
        unsigned int s = hist[((size_t)seg * 256 + d) * tiles + tile];
        for (int q = 0; q < w; ++q) s += cnt[q][d];
        start[u] = s;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) cnt[w][lane + 64 * u] = start[u];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // second walk, in order: rank among the lanes with the same digit, then move the counter on
#pragma unroll
    for (int r = 0; r < RADIX_PER_WAVE / 64; ++r) {
        const bool active = base + 64 * r + lane < n;
        const unsigned int d = (unsigned int)(key[r] >> shift) & 255u;
        const unsigned long long peers = radix_peers(d, active);
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        unsigned int first = 0u;
        if (active) first = cnt[w][d];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (active) {
            O[first + (unsigned int)__popcll(below)] = key[r];
            if (below == 0ull) cnt[w][d] = first + (unsigned int)__popcll(peers);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Sorts the nseg segments of n keys in `a` on bits [bit_lo, bit_hi) ((bit_hi - bit_lo) a multiple
// of 8).  `b` is a second buffer of the same size, `hist` radix_hist_words(nseg, n) unsigned
// ints.  Returns the buffer that holds the result (a after an even number of passes, else b) or
// nullptr when a launch failed (*err set to the hipError_t).
inline unsigned long long* radix_sort_u64(unsigned long long* a, unsigned long long* b, unsigned int* hist,
                                          long long nseg, long long n, int bit_lo, int bit_hi,
                                          hipStream_t st, int* err) {
    *err = 0;
    if (nseg < 1 || n < 1) return a;
    const int tiles = (int)radix_tiles(n);
    const dim3 grid((unsigned)(nseg * tiles));
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, grid, dim3(RADIX_THREADS), 0, st, a, hist, n, tiles, shift);
        hipLaunchKernelGGL(radix_scan_kernel, dim3((unsigned)nseg), dim3(1024), 0, st, hist, tiles);
        hipLaunchKernelGGL(radix_scatter_kernel, grid, dim3(RADIX_THREADS), 0, st, a, b, hist, n, tiles, shift);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { *err = (int)e; return nullptr; }
        unsigned long long* t = a; a = b; b = t;
    }
    return a;
}

}  // namespace yh
