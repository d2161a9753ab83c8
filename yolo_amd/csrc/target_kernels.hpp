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

// one box: ltrb_to_xywh (data.py:173-183), its cell (model.py:217-221) and
// normalize_xywh (data.py:185-194)
__device__ __forceinline__ int encode_box(const float4 b, int S, float cellf, float imgf,
                                          float4& norm) {
    const float cx = (b.x + b.z) / 2.0f;
    const float cy = (b.y + b.w) / 2.0f;
    int xc = (int)floordiv_f32(cx, cellf);
    int yc = (int)floordiv_f32(cy, cellf);
    xc = xc < 0 ? 0 : (xc > S - 1 ? S - 1 : xc);
    yc = yc < 0 ? 0 : (yc > S - 1 ? S - 1 : yc);
    norm.x = remainder_f32(cx, cellf) / cellf;
    norm.y = remainder_f32(cy, cellf) / cellf;
    norm.z = (b.z - b.x) / imgf;
    norm.w = (b.w - b.y) / imgf;
    return xc * S + yc;
}

// Tiled encoder for the usual case of few boxes per image (M <= 64): a workgroup
// owns G consecutive images.  Phase 1 encodes every box once (one lane per box)
// into LDS; phase 2 gives every cell its last matching box and writes xywh and
// the object mask; phase 3 writes the class rows of the whole tile as one
// contiguous, fully coalesced range (a lane computes the four one-hot values of
// its float4 from the winner label of that cell).
__global__ __launch_bounds__(256) void encode_targets_tiled_kernel(
    const float* __restrict__ gt_ltrb, const int32_t* __restrict__ gt_label,
    const int32_t* __restrict__ gt_count, float* __restrict__ gt_xywh,
    float* __restrict__ gt_cls, uint8_t* __restrict__ obj_mask, long long batch, int M, int S,
    int C, float cellf, float imgf, int G, int cls_vec4) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    const int S2 = S * S;
    float4* e_norm = reinterpret_cast<float4*>(tsm);                 // [G*M]
    int* e_cell = reinterpret_cast<int*>(tsm + (size_t)G * M * 16);  // [G*M]
    int* e_label = e_cell + G * M;                                   // [G*M]
    int* w_label = e_label + G * M;                                  // [G*S2]
    const long long img0 = (long long)blockIdx.x * G;
    const int nimg = (int)((batch - img0) < G ? (batch - img0) : G);

    for (int i = threadIdx.x; i < nimg * M; i += 256) {
        const int gi = i / M, m = i - gi * M;
        const long long img = img0 + gi;
        // the box and its label are requested together with the count, not after it: the
        // rows beyond the count are padding, but they exist
        int n = gt_count[img];
        const float4 box = reinterpret_cast<const float4*>(gt_ltrb)[img * M + m];
        const int label = gt_label[img * M + m];
        n = n < 0 ? 0 : (n > M ? M : n);
        int cell = -1;
        if (m < n) {
            float4 norm;
            cell = encode_box(box, S, cellf, imgf, norm);
            e_norm[i] = norm;
            e_label[i] = label;
        }
        e_cell[i] = cell;
    }
    __syncthreads();
    for (int item = threadIdx.x; item < nimg * S2; item += 256) {
        const int gi = item / S2, cell = item - gi * S2;
        int win = -1;
        for (int m = 0; m < M; ++m) win = e_cell[gi * M + m] == cell ? m : win;
        const long long g = (img0 + gi) * S2 + cell;
        reinterpret_cast<float4*>(gt_xywh)[g] =
            win >= 0 ? e_norm[gi * M + win] : make_float4(0.f, 0.f, 0.f, 0.f);
        obj_mask[g] = win >= 0 ? 1 : 0;
        w_label[item] = win >= 0 ? e_label[gi * M + win] : -1;
    }
    __syncthreads();
    float* rows = gt_cls + img0 * (long long)S2 * C;
    const int nfl = nimg * S2 * C;
    if (cls_vec4) {
        for (int q = threadIdx.x; q < nfl / 4; q += 256) {
            const int f = q * 4;
            const int item = f / C, j = f - item * C;
            const int lab = w_label[item];
            store_stream(reinterpret_cast<float4*>(rows) + q,
                         make_float4(lab == j ? 1.f : 0.f, lab == j + 1 ? 1.f : 0.f,
                                     lab == j + 2 ? 1.f : 0.f, lab == j + 3 ? 1.f : 0.f));
        }
    } else {
        for (int f = threadIdx.x; f < nfl; f += 256) {
            const int item = f / C, j = f - item * C;
            rows[f] = w_label[item] == j ? 1.f : 0.f;
        }
    }
}

// General encoder (any M): one lane per (image, cell) scans the image's boxes.
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
