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

}  // namespace yh
