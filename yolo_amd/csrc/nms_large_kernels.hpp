// Decode + per-class NMS for more candidates per image than fit a CU's LDS
// (above YH_MAX_CANDIDATES, up to 65,536).  Three steps, all in global memory:
//   1  one lane per cell (or per supplied candidate): the candidate's box to a
//      box array and its 64-bit sort key (class | inverted score | flat index, the
//      key nms_kernel ranks by; all ones for a candidate that is not valid)
//   2  segmented radix sort of the keys, one segment per image (radix_sort.hpp)
//   3  one workgroup per image walks the sorted keys in chunks of up to 64
//      candidates of one class: a chunk is first tested against the rows already
//      kept for that class -- they are the tail of the output itself, read back
//      256 at a time through LDS, the four wavefronts sharing the rows -- then
//      resolved inside the chunk in order, and the survivors are appended.
// Same keep rule, order and bits as nms_kernel.
#pragma once
#include "nms_kernels.hpp"

namespace yh {

__device__ __forceinline__ unsigned long long nms_key(int cls, float score, int flat) {
    const unsigned int sk = ~float_to_ordered(score + 0.0f);   // descending score
    return ((unsigned long long)(unsigned int)cls << 48) | ((unsigned long long)sk << 16) |
           (unsigned int)flat;
}
__device__ __forceinline__ float nms_key_score(unsigned long long key) {
    const unsigned int o = ~(unsigned int)(key >> 16);
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// candidates of the head tensor: boxes (batch, S*S*nb) and keys (batch, S*S*nb)
template <bool BIAS>
__global__ __launch_bounds__(256) void nms_large_decode_kernel(
    const float* __restrict__ pred, const float* __restrict__ bias, float4* __restrict__ boxes,
    unsigned long long* __restrict__ keys, long long batch, int S, int nb, int nc, int mode,
    float cellf, float imgf, float conf_thr) {
    const int S2 = S * S;
    const long long total = batch * S2;
    const long long D = 5 * nb + nc;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long img = g / S2;
        const int cell = (int)(g - img * S2);
        const MemLogitsT<BIAS> logit{pred + img * D * S2 + cell, BIAS ? bias + cell : nullptr, S2};
        const int cls = cell_class_mem(logit, 5 * nb, nc);
        float cb;
        const int best = cell_best_conf(logit, nb, cb);
        for (int k = 0; k < nb; ++k) {
            const long long o = g * nb + k;
            const Box b = cell_box(logit, k, cell, S, cellf, imgf);
            boxes[o] = make_float4(b.l, b.t, b.r, b.b);
            const float score = sigmoidf_(logit(5 * k + 4));
            const bool valid = (mode == 1 || k == best) && score >= conf_thr;
            keys[o] = valid ? nms_key(cls, score, cell * nb + k) : ~0ull;
        }
    }
}

// keys of caller-supplied candidates (N per image)
__global__ __launch_bounds__(256) void nms_large_cand_keys_kernel(
    const float* __restrict__ c_score, const int32_t* __restrict__ c_cls,
    const uint8_t* __restrict__ c_present, unsigned long long* __restrict__ keys, long long batch,
    int N, float conf_thr) {
    const long long total = batch * N;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const float score = c_score[g];
        const int cls = c_cls[g];
        const bool present = c_present ? c_present[g] != 0 : true;
        const bool valid = present && score >= conf_thr && cls >= 0 && cls <= 32767;
        keys[g] = valid ? nms_key(cls, score, (int)(g % N)) : ~0ull;
    }
}

// scores: the candidates' scores as supplied (bit-exact copy into the rows), or
// null: the score is taken back out of the key.
__global__ __launch_bounds__(256) void nms_large_kernel(
    const unsigned long long* __restrict__ keys_sorted, const float4* __restrict__ boxes,
    const float* __restrict__ scores, float* dets, int32_t* __restrict__ keep_count,
    int32_t* __restrict__ keep_idx, int N, float iou_thr) {
    const long long img = blockIdx.x;
    const unsigned long long* K = keys_sorted + img * N;
    const float4* B = boxes + img * N;
    float2* dst2 = reinterpret_cast<float2*>(dets + img * (long long)N * 6);
    int32_t* kd = keep_idx ? keep_idx + img * (long long)N : nullptr;
    __shared__ float4 tile[256];
    __shared__ unsigned long long sdead[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const IouBand band = make_iou_band(iou_thr);
    int total = 0, class_start = 0;
    unsigned int prev_cls = 0xFFFFFFFFu;
    int pos = 0;
    while (pos < N) {
        const unsigned long long key = pos + lane < N ? K[pos + lane] : ~0ull;
        const unsigned long long k0 = __shfl(key, 0, 64);
        if (k0 == ~0ull) break;                          // the valid candidates come first
        const unsigned int cls0 = (unsigned int)(k0 >> 48);
        const unsigned long long same =
            __ballot(key != ~0ull && (unsigned int)(key >> 48) == cls0);
        const int len = same == ~0ull ? 64 : __ffsll((long long)~same) - 1;   // >= 1
        if (cls0 != prev_cls) {
            class_start = total;
            prev_cls = cls0;
        }
        const bool mine = lane < len;
        const int idx = (int)(key & 0xFFFFull);
        Box mybox = {0.f, 0.f, 0.f, 0.f};
        if (mine) {
            const float4 b = B[idx];
            mybox = {b.x, b.y, b.z, b.w};
        }
        // against the rows kept so far for this class
        bool dead = false;
        for (int t0 = class_start; t0 < total; t0 += 256) {
            __syncthreads();
            const int r = t0 + (int)threadIdx.x;
            if (r < total) {
                const float2 a = dst2[r * 3], b = dst2[r * 3 + 1];
                tile[threadIdx.x] = make_float4(a.x, a.y, b.x, b.y);
            }
            __syncthreads();
            const int nt = total - t0 < 256 ? total - t0 : 256;
            for (int q = w; q < nt; q += 4) {
                if (__ballot(mine && !dead) == 0ull) break;
                if (mine && !dead) {
                    const float4 p = tile[q];
                    if (iou_above_band({p.x, p.y, p.z, p.w}, mybox, band)) dead = true;
                }
            }
        }
        const unsigned long long dm = __ballot(dead);
        if (lane == 0) sdead[w] = dm;
        __syncthreads();
        const unsigned long long deadmask = sdead[0] | sdead[1] | sdead[2] | sdead[3];
        unsigned long long alive = (len == 64 ? ~0ull : (1ull << len) - 1ull) & ~deadmask;
        // inside the chunk, in order (every wavefront does the same arithmetic)
        for (int i = 0; i < len; ++i) {
            if (!((alive >> i) & 1ull)) continue;
            const Box pb = {__shfl(mybox.l, i, 64), __shfl(mybox.t, i, 64), __shfl(mybox.r, i, 64),
                            __shfl(mybox.b, i, 64)};
            const bool cand = lane > i && ((alive >> lane) & 1ull);
            bool kill = false;
            if (cand) kill = iou_above_band(pb, mybox, band);
            alive &= ~__ballot(kill);
        }
        if (w == 0 && ((alive >> lane) & 1ull)) {
            const int r = total + __popcll(alive & ((1ull << lane) - 1ull));
            const float score = scores ? scores[img * (long long)N + idx] : nms_key_score(key);
            dst2[r * 3] = make_float2(mybox.l, mybox.t);
            dst2[r * 3 + 1] = make_float2(mybox.r, mybox.b);
            dst2[r * 3 + 2] = make_float2(score, (float)cls0);
            if (kd) kd[r] = idx;
        }
        total += __popcll(alive);
        pos += len;
        __syncthreads();     // the new rows are in place (and sdead free) before the next chunk
    }
    for (int i = total * 3 + (int)threadIdx.x; i < N * 3; i += 256)
        store_stream(dst2 + i, make_float2(0.0f, 0.0f));
    if (kd)
        for (int i = total + (int)threadIdx.x; i < N; i += 256) kd[i] = -1;
    if (threadIdx.x == 0) keep_count[img] = total;
}

}  // namespace yh
