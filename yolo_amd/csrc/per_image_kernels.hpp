// Per-image loss terms (forward only): the three sums of model.py:224-294 for every image of
// the batch by itself -- what the reference's per-batch get_loss (model.py:296-317) adds up.
// One wavefront per image when the grid has at most 64 cells (lane = cell, wavefront
// reduction), else one workgroup per image; the sums are formed in a fixed order, so the
// values are the same from run to run.  The formulas are those of the fused loss kernel
// (loss_obj_cell), evaluated by one lane per cell for run-time predictor and class counts.
#pragma once
#include "head_math.hpp"

namespace yh {

template <class L>
__device__ __forceinline__ void cell_loss_terms(const L& logit, const float* __restrict__ gxywh,
                                                const float* __restrict__ gcls, bool has_obj, int cell,
                                                int S, int nb, int nc, float cellf, float imgf, float lc,
                                                float ln, float& coord, float& conf, float& cls) {
    if (!has_obj) {
        // only the no-object confidence term (model.py:270-274)
        for (int k = 0; k < nb; ++k) {
            const float c = sigmoidf_(logit(5 * k + 4));
            conf += ln * (c * c);
        }
        return;
    }
    const float4 gv = *reinterpret_cast<const float4*>(gxywh);
    const float xoff = cellf * (float)(cell / S);   // first spatial axis carries x (model.py:175-177)
    const float yoff = cellf * (float)(cell % S);
    const Box gbox = decode_box(gv.x, gv.y, gv.z, gv.w, xoff, yoff, cellf, imgf);
    // responsible predictor: highest IoU, first index on ties (model.py:210)
    int best = 0;
    float bnx = 0.f, bny = 0.f, bnw = 0.f, bnh = 0.f, biou = 0.f;
    for (int k = 0; k < nb; ++k) {
        const float nx = sigmoidf_(logit(5 * k + 0)), ny = sigmoidf_(logit(5 * k + 1));
        const float nw = sigmoidf_(logit(5 * k + 2)), nh = sigmoidf_(logit(5 * k + 3));
        const float iou = iou_pair(decode_box(nx, ny, nw, nh, xoff, yoff, cellf, imgf), gbox);
        if (k == 0 || iou > biou) { best = k; biou = iou; bnx = nx; bny = ny; bnw = nw; bnh = nh; }
    }
    {   // coordinate term (model.py:224-240)
        const float dx = bnx - gv.x, dy = bny - gv.y;
        const float dw = __builtin_amdgcn_sqrtf(bnw) - __builtin_amdgcn_sqrtf(gv.z);
        const float dh = __builtin_amdgcn_sqrtf(bnh) - __builtin_amdgcn_sqrtf(gv.w);
        coord += lc * ((dx * dx + dy * dy) + (dw * dw + dh * dh));
    }
    for (int k = 0; k < nb; ++k) {   // confidence terms (model.py:252-276)
        const float c = sigmoidf_(logit(5 * k + 4));
        if (k == best) { const float d = c - biou; conf += d * d; }
        else conf += ln * (c * c);
    }
    // class term (model.py:278-294): squared error on the softmax
    const int c0 = 5 * nb;
    float m = logit(c0);
    for (int j = 1; j < nc; ++j) m = fmaxf(m, logit(c0 + j));
    float s = 0.f;
    for (int j = 0; j < nc; ++j) s += exp_fast(logit(c0 + j) - m);
    const float rs = rcp_fast(s);
    float l2 = 0.f;
    for (int j = 0; j < nc; ++j) {
        const float d = exp_fast(logit(c0 + j) - m) * rs - gcls[j];
        l2 += d * d;
    }
    cls += l2;
}

// imgs_per_block: 4 (a wavefront per image, S*S <= 64) or 1 (the workgroup's 256 threads share it)
__global__ __launch_bounds__(256) void loss_per_image_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt_xywh, const float* __restrict__ gt_cls,
    const uint8_t* __restrict__ obj_mask, float* __restrict__ per_image, long long batch, int S, int nb,
    int nc, int imgs_per_block, float cellf, float imgf, float lc, float ln) {
    __shared__ float red[4][3];
    const int S2 = S * S;
    const long long D = 5 * nb + nc;
    const int NT = 256 / imgs_per_block;
    const int sub = threadIdx.x / NT, t = threadIdx.x % NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long img = (long long)blockIdx.x * imgs_per_block + sub;
    float coord = 0.f, conf = 0.f, cls = 0.f;
    if (img < batch) {
        for (int cell = t; cell < S2; cell += NT) {
            const long long tcell = img * S2 + cell;
            const float* p = pred + img * D * S2 + cell;
            auto logit = [&](int ch) -> float { return p[(size_t)ch * S2]; };
            cell_loss_terms(logit, gt_xywh + tcell * 4, gt_cls + tcell * nc, obj_mask[tcell] != 0, cell, S, nb,
                            nc, cellf, imgf, lc, ln, coord, conf, cls);
        }
    }
    coord = wave_sum(coord);
    conf = wave_sum(conf);
    cls = wave_sum(cls);
    if (imgs_per_block == 4) {
        if (lane == 0 && img < batch) {
            float* o = per_image + img * 4;
            o[0] = coord; o[1] = conf; o[2] = cls; o[3] = (coord + conf) + cls;
        }
        return;
    }
    if (lane == 0) { red[wave][0] = coord; red[wave][1] = conf; red[wave][2] = cls; }
    __syncthreads();
    if (threadIdx.x == 0 && img < batch) {
        float a[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
        float* o = per_image + img * 4;
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = (a[0] + a[1]) + a[2];
    }
}

}  // namespace yh
