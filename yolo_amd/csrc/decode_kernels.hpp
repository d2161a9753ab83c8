// Dense grid decode: every per-cell quantity the reference's inference-side
// helpers return (model.py:150-188, 242-250, 278-286, 379-394), one lane per cell.
#pragma once
#include "nms_kernels.hpp"

namespace yh {

__global__ __launch_bounds__(256) void decode_grid_kernel(
    const float* __restrict__ pred, float* __restrict__ ltrb, float* __restrict__ conf,
    float* __restrict__ cls_prob, long long* __restrict__ cls_idx,
    long long* __restrict__ best_k, float* __restrict__ best_ltrb,
    uint8_t* __restrict__ conf_mask, float conf_thresh, long long batch, int S, int nb, int nc,
    float cellf, float imgf) {
    const int S2 = S * S;
    const long long total = batch * S2;
    const long long D = 5 * nb + nc;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long img = g / S2;
        const int cell = (int)(g - img * S2);
        const float* p = pred + img * D * S2 + cell;
        const MemLogits logit{p, nullptr, S2};
        if (ltrb || conf) {
            for (int k = 0; k < nb; ++k) {
                if (ltrb) {
                    const Box b = cell_box(logit, k, cell, S, cellf, imgf);
                    reinterpret_cast<float4*>(ltrb)[g * nb + k] = make_float4(b.l, b.t, b.r, b.b);
                }
                if (conf) conf[g * nb + k] = sigmoidf_(logit(5 * k + 4));
            }
        }
        if (best_k || best_ltrb || conf_mask) {
            float cb;
            const int k = cell_best_conf(logit, nb, cb);
            if (best_k) best_k[g] = k;
            if (best_ltrb) {
                const Box b = cell_box(logit, k, cell, S, cellf, imgf);
                reinterpret_cast<float4*>(best_ltrb)[g] = make_float4(b.l, b.t, b.r, b.b);
            }
            if (conf_mask) conf_mask[g] = cb >= conf_thresh ? 1 : 0;
        }
        if (cls_idx) cls_idx[g] = cell_class_mem(logit, 5 * nb, nc);
        if (cls_prob) {
            const float* pc = p + 5 * nb * S2;
            float m = pc[0];
            for (int j = 1; j < nc; ++j) m = fmaxf(m, pc[j * S2]);
            float s = 0.f;
            for (int j = 0; j < nc; ++j) s += exp_fast(pc[j * S2] - m);
            const float rs = rcp_fast(s);
            for (int j = 0; j < nc; ++j) cls_prob[g * nc + j] = exp_fast(pc[j * S2] - m) * rs;
        }
    }
}

// get_norm_xywh (model.py:157-165), batched: out[b, cell, k, :] = sigmoid of
// channels 5k .. 5k+3 of the cell, cell = i*S + j.
__global__ __launch_bounds__(256) void norm_xywh_kernel(const float* __restrict__ pred,
                                                        float* __restrict__ out, long long batch,
                                                        int S, int nb, int nc) {
    const int S2 = S * S;
    const long long total = batch * S2;
    const long long D = 5 * nb + nc;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long img = g / S2;
        const int cell = (int)(g - img * S2);
        const float* p = pred + img * D * S2 + cell;
        for (int k = 0; k < nb; ++k)
            reinterpret_cast<float4*>(out)[g * nb + k] =
                make_float4(sigmoidf_(p[(5 * k) * S2]), sigmoidf_(p[(5 * k + 1) * S2]),
                            sigmoidf_(p[(5 * k + 2) * S2]), sigmoidf_(p[(5 * k + 3) * S2]));
    }
}

// denormalize_xywh (model.py:167-183) on (batch, S*S, nb, 4): x, y scaled by the
// cell size and moved to the cell (first grid index carries x), w, h by the image size.
__global__ __launch_bounds__(256) void denormalize_xywh_kernel(const float* __restrict__ in,
                                                               float* __restrict__ out,
                                                               long long nboxes, int S, int nb,
                                                               float cellf, float imgf) {
    const int S2 = S * S;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nboxes;
         g += (long long)gridDim.x * blockDim.x) {
        const int cell = (int)((g / nb) % S2);
        const float4 v = reinterpret_cast<const float4*>(in)[g];
        reinterpret_cast<float4*>(out)[g] =
            make_float4(v.x * cellf + cellf * (float)(cell / S),
                        v.y * cellf + cellf * (float)(cell % S), v.z * imgf, v.w * imgf);
    }
}

// xywh_to_ltrb (model.py:150-155): one-sided clips, as the reference has them
__global__ __launch_bounds__(256) void xywh_to_ltrb_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out,
                                                           long long nboxes, float imgf) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nboxes;
         g += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[g];
        const float hw = v.z / 2.0f, hh = v.w / 2.0f;
        reinterpret_cast<float4*>(out)[g] = make_float4(fmaxf(v.x - hw, 0.0f), fmaxf(v.y - hh, 0.0f),
                                                        fminf(v.x + hw, imgf), fminf(v.y + hh, imgf));
    }
}

// _get_responsibility_mask (model.py:190-211) as the loss uses it: in a cell that
// holds an object the predictor with the highest IoU against the cell's ground
// truth (first index on ties), nothing elsewhere.  Same arithmetic as the loss
// kernel, so the mask is the one the loss was computed with.
__global__ __launch_bounds__(256) void responsibility_mask_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt_xywh,
    const uint8_t* __restrict__ obj_mask, uint8_t* __restrict__ resp, long long batch, int S,
    int nb, int nc, float cellf, float imgf) {
    const int S2 = S * S;
    const long long total = batch * S2;
    const long long D = 5 * nb + nc;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long img = g / S2;
        const int cell = (int)(g - img * S2);
        int best = -1;
        if (obj_mask[g] != 0) {
            const float* p = pred + img * D * S2 + cell;
            const float4 gv = reinterpret_cast<const float4*>(gt_xywh)[g];
            const float xoff = cellf * (float)(cell / S), yoff = cellf * (float)(cell % S);
            const Box gbox = decode_box(gv.x, gv.y, gv.z, gv.w, xoff, yoff, cellf, imgf);
            float biou = 0.f;
            for (int k = 0; k < nb; ++k) {
                const Box b = decode_box(sigmoidf_(p[(5 * k) * S2]), sigmoidf_(p[(5 * k + 1) * S2]),
                                         sigmoidf_(p[(5 * k + 2) * S2]),
                                         sigmoidf_(p[(5 * k + 3) * S2]), xoff, yoff, cellf, imgf);
                const float iou = iou_pair(b, gbox);
                if (k == 0 || iou > biou) { best = k; biou = iou; }
            }
        }
        for (int k = 0; k < nb; ++k) resp[g * nb + k] = k == best ? 1 : 0;
    }
}

}  // namespace yh
