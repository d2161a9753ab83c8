// Target encoder: padded ground-truth boxes -> gt_norm_xywh / gt_cls_prob /
// obj_mask (data.py:173-194, model.py:213-222).  One lane per (image, cell): it
// scans the image's boxes in order and keeps the last one whose centre falls in
// its cell ("last writer wins", one box per cell, data.py:32), then writes the
// cell's 16 + 4*C + 1 bytes.  No atomics, no ordering hazards, bit-exact.
#pragma once
#include "head_math.hpp"

namespace yh {

// torch's floor division for floats (a // b): fmod-based, as Python defines it
__device__ __forceinline__ float floordiv_f32(float a, float b) {
    if (b == 0.0f) return a / b;
    const float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div -= 1.0f;
    if (div != 0.0f) {
        float fl = floorf(div);
        if (div - fl > 0.5f) fl += 1.0f;
        return fl;
    }
    return copysignf(0.0f, a / b);
}

// torch.remainder for floats (a % b): result takes the sign of b
__device__ __forceinline__ float remainder_f32(float a, float b) {
    float mod = fmodf(a, b);
    if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) mod += b;
    return mod;
}

__global__ __launch_bounds__(256) void encode_targets_kernel(
    const float* __restrict__ gt_ltrb, const int32_t* __restrict__ gt_label,
    const int32_t* __restrict__ gt_count, float* __restrict__ gt_xywh,
    float* __restrict__ gt_cls, uint8_t* __restrict__ obj_mask, long long batch, int M, int S,
    int C, float cellf, float imgf, int cls_vec4) {
    const int S2 = S * S;
    const long long total = batch * S2;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long img = g / S2;
        const int cell = (int)(g - img * S2);
        int n = gt_count[img];
        n = n < 0 ? 0 : (n > M ? M : n);
        const float4* boxes = reinterpret_cast<const float4*>(gt_ltrb) + img * M;
        int win = -1;
        float wx = 0.f, wy = 0.f, ww = 0.f, wh = 0.f;
        for (int m = 0; m < n; ++m) {
            const float4 b = boxes[m];
            // ltrb_to_xywh (data.py:173-183)
            const float cx = (b.x + b.z) / 2.0f;
            const float cy = (b.y + b.w) / 2.0f;
            // get_object_mask (model.py:217-221): [x // cell_size, y // cell_size]
            int xc = (int)floordiv_f32(cx, cellf);
            int yc = (int)floordiv_f32(cy, cellf);
            xc = xc < 0 ? 0 : (xc > S - 1 ? S - 1 : xc);
            yc = yc < 0 ? 0 : (yc > S - 1 ? S - 1 : yc);
            if (xc * S + yc == cell) {
                win = m; wx = cx; wy = cy; ww = b.z - b.x; wh = b.w - b.y;
            }
        }
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        int label = -1;
        if (win >= 0) {
            // normalize_xywh (data.py:185-194)
            out.x = remainder_f32(wx, cellf) / cellf;
            out.y = remainder_f32(wy, cellf) / cellf;
            out.z = ww / imgf;
            out.w = wh / imgf;
            label = gt_label[img * M + win];
        }
        reinterpret_cast<float4*>(gt_xywh)[g] = out;
        obj_mask[g] = win >= 0 ? 1 : 0;
        float* row = gt_cls + g * C;
        if (cls_vec4) {
            for (int j = 0; j < C; j += 4)
                *reinterpret_cast<float4*>(row + j) =
                    make_float4(label == j ? 1.f : 0.f, label == j + 1 ? 1.f : 0.f,
                                label == j + 2 ? 1.f : 0.f, label == j + 3 ? 1.f : 0.f);
        } else {
            for (int j = 0; j < C; ++j) row[j] = label == j ? 1.f : 0.f;
        }
    }
}

}  // namespace yh
