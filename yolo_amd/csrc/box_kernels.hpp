// Pairwise box geometry of eval.py:22-83 (area, intersection, IoU, GIoU): a
// workgroup owns part of one row i, lanes run over the columns j, so the (N, M)
// result is written coalesced and no index division is needed.
#pragma once
#include "head_math.hpp"

namespace yh {

constexpr int BOX_IOU = 0, BOX_GIOU = 1, BOX_INTERSECTION = 2, BOX_ENCLOSING = 3;

// grid: x covers the columns j (grid-stride), y covers the rows i (grid-stride).
// A lane keeps its column's box and area in registers and walks down the rows
// of its workgroup; row i's box is uniform across the workgroup.  Four rows per
// pass, so that four independent quotients are in flight.
template <int KIND>
__device__ __forceinline__ float pair_value(const Box& p, float ap, const Box& q, float aq) {
    if constexpr (KIND == BOX_ENCLOSING) {
        // get_smallest_enclosing_area (eval.py:68-73)
        const float ew = fmaxf(fmaxf(p.r, q.r) - fminf(p.l, q.l), 0.0f);
        const float eh = fmaxf(fmaxf(p.b, q.b) - fminf(p.t, q.t), 0.0f);
        return ew * eh;
    }
    // get_intersection_area (eval.py:46-50)
    const float iw = fmaxf(fminf(p.r, q.r) - fmaxf(p.l, q.l), 0.0f);
    const float ih = fmaxf(fminf(p.b, q.b) - fmaxf(p.t, q.t), 0.0f);
    const float inter = iw * ih;
    if constexpr (KIND == BOX_INTERSECTION) return inter;
    // get_iou (eval.py:61-65)
    const float uni = (ap + aq) - inter;
    const float iou = uni == 0.0f ? 0.0f : inter / uni;
    if constexpr (KIND == BOX_IOU) return iou;
    // get_smallest_enclosing_area and get_giou (eval.py:68-83)
    const float cw = fmaxf(fmaxf(p.r, q.r) - fminf(p.l, q.l), 0.0f);
    const float ch = fmaxf(fmaxf(p.b, q.b) - fminf(p.t, q.t), 0.0f);
    const float c = cw * ch;
    return c == 0.0f ? -1.0f : iou - ((c - uni) / c);
}

template <int KIND>
__global__ __launch_bounds__(256) void pairwise_boxes_kernel(const float* __restrict__ ltrb1,
                                                             const float* __restrict__ ltrb2,
                                                             float* __restrict__ out, long long N,
                                                             long long M) {
    const float4* rows = reinterpret_cast<const float4*>(ltrb1);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < M;
         j += (long long)gridDim.x * blockDim.x) {
        const float4 b = reinterpret_cast<const float4*>(ltrb2)[j];
        const Box q = {b.x, b.y, b.z, b.w};
        const float aq = box_area(q);
        float* col = out + j;
        long long i = blockIdx.y;
        const long long step = gridDim.y;
        for (; i + 3 * step < N; i += 4 * step) {
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 a = rows[i + u * step];
                const Box p = {a.x, a.y, a.z, a.w};
                r[u] = pair_value<KIND>(p, box_area(p), q, aq);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) col[(i + u * step) * M] = r[u];
        }
        for (; i < N; i += step) {
            const float4 a = rows[i];
            const Box p = {a.x, a.y, a.z, a.w};
            col[i * M] = pair_value<KIND>(p, box_area(p), q, aq);
        }
    }
}

__global__ __launch_bounds__(256) void box_area_kernel(const float* __restrict__ ltrb,
                                                       float* __restrict__ out, long long n) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n;
         g += (long long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(ltrb)[g];
        const Box p = {a.x, a.y, a.z, a.w};
        out[g] = box_area(p);
    }
}

}  // namespace yh
