// Fused YOLOv1 loss forward + backward (model.py:190-317 and its autograd
// backward).  The head tensor is channel-major, so with one lane per grid cell
// consecutive lanes read and write consecutive addresses of a channel.  Two
// passes per workgroup: pass 1, a lane per cell, finishes the cells without an
// object (their confidence logits in, their gradients out) and lists the cells
// that hold one; pass 2 works on the listed cells only -- a few lanes per
// workgroup in the compile-time-shape kernel, a group of lanes per cell in the
// run-time-shape kernel -- so the long object path is not executed by every
// wavefront for one or two of its lanes.  All gradients are analytic.  Loss
// partials are reduced wave -> workgroup -> one row of `partials` per workgroup
// and the rows are added in a fixed order in fp64 (by a second tiny kernel, or
// by the first workgroup of the next call), so results are reproducible bit for
// bit.
#pragma once
#include <type_traits>

#include "head_math.hpp"

namespace yh {

constexpr int LOSS_THREADS = 256;
// Images per workgroup for a grid of S2 cells (one lane per cell).  As many as fit the
// workgroup -- except at S = 7, where four (196 of 256 lanes) beat five: a batch of
// 4096 is then 1024 workgroups, four on every CU, where 820 leave a fifth of the CUs
// with one workgroup more than the rest, and the launch waits for those (12.0 -> 11.2 us
// over rotating buffers; three: 14.0, two: 14.2).
constexpr int loss_ipb_for(int S2) { return S2 == 49 ? 4 : LOSS_THREADS / S2; }
// ground-truth slots per wavefront of the compile-time-shape loss kernel (0: pass 2 fetches)
#ifndef YH_LOSS_GT_SLOTS
#define YH_LOSS_GT_SLOTS 16
#endif
constexpr int LOSS_GT_SLOTS = YH_LOSS_GT_SLOTS;

#ifdef YH_PHASE_PROBE
// probe builds only: eight time stamps per workgroup (100 MHz counter) of loss_block
__device__ unsigned long long* yh_loss_stamps = nullptr;
#define YH_LSTAMP(i)                                                                        \
    do {                                                                                    \
        if (yh_loss_stamps && threadIdx.x == 0)                                             \
            yh_loss_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define YH_LSTAMP(i) do {} while (0)
#endif


// A cell without object: only the no-object confidence term (model.py:270-274);
// every other gradient of the cell is zero.  `logit(ch)` returns channel ch of
// the cell, `g` points at the cell's gradient (channels S2 floats apart).
// loss_noobj_cell for run-time predictor and class counts: the confidence logits
// are fetched four at a time and the zero stores are written out eight per
// iteration (a rolled loop costs an address computation and a branch per store).
template <class L>
__device__ __forceinline__ void loss_noobj_cell_rt(const L& logit, float* __restrict__ g, int S2,
                                                   int nb, int nc, float ln, float inv_gb,
                                                   float& acc_conf) {
    for (int k0 = 0; k0 < nb; k0 += 4) {
        float c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = k0 + u < nb ? logit(5 * (k0 + u) + 4) : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u < nb) {
                const float conf = sigmoidf_(c[u]);
                acc_conf += ln * (conf * conf);
                float* q = g + (size_t)(5 * (k0 + u)) * S2;
                q[0] = 0.0f;
                q[S2] = 0.0f;
                q[2 * S2] = 0.0f;
                q[3 * S2] = 0.0f;
                q[4 * S2] = 2.0f * ln * conf * (conf * (1.0f - conf)) * inv_gb;
            }
        }
    }
    float* q = g + (size_t)(5 * nb) * S2;
    int j = 0;
    for (; j + 8 <= nc; j += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u * S2] = 0.0f;
        q += 8 * (size_t)S2;
    }
    for (; j < nc; ++j) {
        *q = 0.0f;
        q += S2;
    }
}

template <class L>
__device__ __forceinline__ void loss_noobj_cell(const L& logit, float* __restrict__ g, int S2,
                                                int nb, int nc, float ln, float inv_gb,
                                                float& acc_conf) {
#pragma unroll
    for (int k = 0; k < nb; ++k) {
        const float conf = sigmoidf_(logit(5 * k + 4));
        acc_conf += ln * (conf * conf);
        g[(5 * k + 0) * S2] = 0.0f;
        g[(5 * k + 1) * S2] = 0.0f;
        g[(5 * k + 2) * S2] = 0.0f;
        g[(5 * k + 3) * S2] = 0.0f;
        g[(5 * k + 4) * S2] = 2.0f * ln * conf * (conf * (1.0f - conf)) * inv_gb;
    }
#pragma unroll
    for (int j = 0; j < nc; ++j) g[(5 * nb + j) * S2] = 0.0f;
}

// A cell that holds an object: responsible predictor by IoU, coordinate,
// objectness, no-object (other predictors) and class terms with all gradients
// (model.py:190-294 and their autograd backward), by one lane, for a class count
// known at compile time.  C_CT % 4 == 0: the class targets are fetched as float4
// (gcls 16-byte aligned).  PART 1 does the box and confidence work of the cell
// only, PART 2 the class work only (the two touch different channels and
// different sums), PART 0 both.
template <int C_CT, int PART = 0, class L>
__device__ __forceinline__ void loss_obj_cell(const L& logit, float* __restrict__ g,
                                              const float* __restrict__ gxywh,
                                              const float* __restrict__ gcls, int cell, int S,
                                              int S2, int nb, int nc, float cellf, float imgf,
                                              float lc, float ln, float inv_gb, float& acc_coord,
                                              float& acc_conf, float& acc_cls) {
    static_assert(C_CT > 0, "run-time class counts take loss_obj_cell_wave");
    constexpr bool kBox = PART != 2, kCls = PART != 1;   // which half of the cell's work
    const int c0 = 5 * nb;
    float gtc[C_CT];
    if constexpr (kCls && C_CT % 4 == 0) {
#pragma unroll
        for (int j = 0; j < C_CT / 4; ++j) {
            const float4 q = reinterpret_cast<const float4*>(gcls)[j];
            gtc[4 * j] = q.x; gtc[4 * j + 1] = q.y; gtc[4 * j + 2] = q.z; gtc[4 * j + 3] = q.w;
        }
    }
    auto gtcls = [&](int j) -> float {
        if constexpr (C_CT % 4 == 0) return gtc[j];
        else return gcls[j];
    };
    if constexpr (kBox) {
    // ground truth through the same decode as the predictions (model.py:261-262)
    const float4 gv = *reinterpret_cast<const float4*>(gxywh);
    const float gnx = gv.x, gny = gv.y, gnw = gv.z, gnh = gv.w;
    const float xoff = cellf * (float)(cell / S);   // first spatial axis carries x (model.py:175-177)
    const float yoff = cellf * (float)(cell % S);
    const Box gbox = decode_box(gnx, gny, gnw, gnh, xoff, yoff, cellf, imgf);

    // responsible predictor: highest IoU, first index on ties (model.py:210)
    int best = 0;
    float bnx = 0.f, bny = 0.f, bnw = 0.f, bnh = 0.f, biou = 0.f;
#pragma unroll
    for (int k = 0; k < nb; ++k) {
        const float nx = sigmoidf_(logit(5 * k + 0));
        const float ny = sigmoidf_(logit(5 * k + 1));
        const float nw = sigmoidf_(logit(5 * k + 2));
        const float nh = sigmoidf_(logit(5 * k + 3));
        const float iou = iou_pair(decode_box(nx, ny, nw, nh, xoff, yoff, cellf, imgf), gbox);
        if (k == 0 || iou > biou) {
            best = k; biou = iou; bnx = nx; bny = ny; bnw = nw; bnh = nh;
        }
    }

    // gradients of the responsible predictor's box (model.py:224-240, 267-269)
    float r0, r1, r2, r3;
    {
        const float dx = bnx - gnx, dy = bny - gny;
        const float sw = __builtin_amdgcn_sqrtf(bnw), sh = __builtin_amdgcn_sqrtf(bnh);
        const float dw = sw - __builtin_amdgcn_sqrtf(gnw), dh = sh - __builtin_amdgcn_sqrtf(gnh);
        acc_coord += lc * ((dx * dx + dy * dy) + (dw * dw + dh * dh));
        r0 = 2.0f * lc * dx;
        r1 = 2.0f * lc * dy;
        // d/dn of lc*(sqrt(n) - sqrt(g))^2 times sigmoid' = lc*(sw - sg)*sw*(1 - n)
        r2 = lc * dw * sw * (1.0f - bnw);
        r3 = lc * dh * sh * (1.0f - bnh);
    }

#pragma unroll
    for (int k = 0; k < nb; ++k) {
        const float conf = sigmoidf_(logit(5 * k + 4));
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, gc;
        if (k == best) {
            // objectness term; IoU is part of the graph (model.py:263-269)
            const float dc = conf - biou;
            acc_conf += dc * dc;
            float ul, ut, ur, ub, gl, gt, gr, gb;
            const Box pb = decode_box(bnx, bny, bnw, bnh, xoff, yoff, cellf, imgf, ul, ut, ur, ub);
            iou_pair_bwd(pb, gbox, -2.0f * dc, gl, gt, gr, gb);
            // xywh_to_ltrb backward: a clipped edge passes nothing
            gl = ul >= 0.0f ? gl : 0.0f;
            gt = ut >= 0.0f ? gt : 0.0f;
            gr = ur <= imgf ? gr : 0.0f;
            gb = ub <= imgf ? gb : 0.0f;
            const float gx_iou = (gl + gr) * cellf;
            const float gy_iou = (gt + gb) * cellf;
            const float gw_iou = (gr - gl) * 0.5f * imgf;
            const float gh_iou = (gb - gt) * 0.5f * imgf;
            g0 = (r0 + gx_iou) * (bnx * (1.0f - bnx));
            g1 = (r1 + gy_iou) * (bny * (1.0f - bny));
            g2 = r2 + gw_iou * (bnw * (1.0f - bnw));
            g3 = r3 + gh_iou * (bnh * (1.0f - bnh));
            gc = 2.0f * dc * (conf * (1.0f - conf));
        } else {
            // the other predictors of the cell carry the no-object term (model.py:270-274)
            acc_conf += ln * (conf * conf);
            gc = 2.0f * ln * conf * (conf * (1.0f - conf));
        }
        g[(5 * k + 0) * S2] = g0 * inv_gb;
        g[(5 * k + 1) * S2] = g1 * inv_gb;
        g[(5 * k + 2) * S2] = g2 * inv_gb;
        g[(5 * k + 3) * S2] = g3 * inv_gb;
        g[(5 * k + 4) * S2] = gc * inv_gb;
    }

    }   // kBox
    if constexpr (!kCls) return;
    // class term (model.py:278-294): squared error on the softmax
    float m = logit(c0);
#pragma unroll
    for (int j = 1; j < C_CT; ++j) m = fmaxf(m, logit(c0 + j));
    float s = 0.f, dot = 0.f, l2 = 0.f;
    // exponentials kept in registers, logits re-read from their (cheap) source
    float e[C_CT];
#pragma unroll
    for (int j = 0; j < C_CT; ++j) {
        e[j] = exp_fast(logit(c0 + j) - m);
        s += e[j];
    }
    const float rs = rcp_fast(s);
#pragma unroll
    for (int j = 0; j < C_CT; ++j) {
        e[j] *= rs;
        const float d = e[j] - gtcls(j);
        l2 += d * d;
        dot += 2.0f * d * e[j];
    }
#pragma unroll
    for (int j = 0; j < C_CT; ++j)
        g[(c0 + j) * S2] = e[j] * (2.0f * (e[j] - gtcls(j)) - dot) * inv_gb;
    acc_cls += l2;
}

// The box and confidence work of loss_obj_cell (PART 1) for a cell with TWO predictors, by a
// pair of neighbouring lanes: lane parity k = the predictor the lane owns (its five logits
// `logit5(0..4)`, its IoU with the ground truth, its five gradients at g[u * S2]).  The two
// IoUs -- the long part of the path -- are formed side by side and meet through one lane
// swap; the responsible lane then carries the coordinate and objectness terms, the other the
// no-object term.  Same arithmetic per value as loss_obj_cell.  Both lanes of a pair must call.
template <class L>
__device__ __forceinline__ void loss_obj_cell_box_pair(const L& logit5, float* __restrict__ g,
                                                       const float* __restrict__ gxywh, int cell, int S,
                                                       int S2, int k, float cellf, float imgf, float lc,
                                                       float ln, float inv_gb, float& acc_coord,
                                                       float& acc_conf) {
    const float4 gv = *reinterpret_cast<const float4*>(gxywh);
    const float gnx = gv.x, gny = gv.y, gnw = gv.z, gnh = gv.w;
    const float xoff = cellf * (float)(cell / S);   // first spatial axis carries x (model.py:175-177)
    const float yoff = cellf * (float)(cell % S);
    const Box gbox = decode_box(gnx, gny, gnw, gnh, xoff, yoff, cellf, imgf);
    const float bnx = sigmoidf_(logit5(0));
    const float bny = sigmoidf_(logit5(1));
    const float bnw = sigmoidf_(logit5(2));
    const float bnh = sigmoidf_(logit5(3));
    float ul, ut, ur, ub;
    const Box pb = decode_box(bnx, bny, bnw, bnh, xoff, yoff, cellf, imgf, ul, ut, ur, ub);
    const float biou = iou_pair(pb, gbox);
    // the neighbour's IoU (lanes 2i and 2i+1 swap)
    const float other =
        __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(biou), 0xB1, 0xF, 0xF, true));
    // responsible predictor: highest IoU, first index on ties (model.py:210): predictor 1 iff
    // its IoU is greater
    const bool one_wins = k == 0 ? other > biou : biou > other;
    const bool best = (k == 1) == one_wins;
    const float conf = sigmoidf_(logit5(4));
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, gc;
    if (best) {
        // gradients of the responsible predictor's box (model.py:224-240, 267-269)
        const float dx = bnx - gnx, dy = bny - gny;
        const float sw = __builtin_amdgcn_sqrtf(bnw), sh = __builtin_amdgcn_sqrtf(bnh);
        const float dw = sw - __builtin_amdgcn_sqrtf(gnw), dh = sh - __builtin_amdgcn_sqrtf(gnh);
        acc_coord += lc * ((dx * dx + dy * dy) + (dw * dw + dh * dh));
        const float r0 = 2.0f * lc * dx;
        const float r1 = 2.0f * lc * dy;
        const float r2 = lc * dw * sw * (1.0f - bnw);
        const float r3 = lc * dh * sh * (1.0f - bnh);
        // objectness term; IoU is part of the graph (model.py:263-269)
        const float dc = conf - biou;
        acc_conf += dc * dc;
        float gl, gt, gr, gb;
        iou_pair_bwd(pb, gbox, -2.0f * dc, gl, gt, gr, gb);
        // xywh_to_ltrb backward: a clipped edge passes nothing
        gl = ul >= 0.0f ? gl : 0.0f;
        gt = ut >= 0.0f ? gt : 0.0f;
        gr = ur <= imgf ? gr : 0.0f;
        gb = ub <= imgf ? gb : 0.0f;
        const float gx_iou = (gl + gr) * cellf;
        const float gy_iou = (gt + gb) * cellf;
        const float gw_iou = (gr - gl) * 0.5f * imgf;
        const float gh_iou = (gb - gt) * 0.5f * imgf;
        g0 = (r0 + gx_iou) * (bnx * (1.0f - bnx));
        g1 = (r1 + gy_iou) * (bny * (1.0f - bny));
        g2 = r2 + gw_iou * (bnw * (1.0f - bnw));
        g3 = r3 + gh_iou * (bnh * (1.0f - bnh));
        gc = 2.0f * dc * (conf * (1.0f - conf));
    } else {
        // the other predictor of the cell carries the no-object term (model.py:270-274)
        acc_conf += ln * (conf * conf);
        gc = 2.0f * ln * conf * (conf * (1.0f - conf));
    }
    g[0 * S2] = g0 * inv_gb;
    g[1 * S2] = g1 * inv_gb;
    g[2 * S2] = g2 * inv_gb;
    g[3 * S2] = g3 * inv_gb;
    g[4 * S2] = gc * inv_gb;
}

// The same cell, worked on by a group of GW lanes (16 or 32: four or two cells per
// wavefront at a time, for up to 4 GW classes; 64: a whole wavefront), for run-time predictor
// and class counts: lane k < nb of the group owns predictor k (its five logits,
// its IoU with the ground truth, its five gradients; nb <= 16), the classes are
// spread over the GW lanes.  Every pass over the classes is one memory round
// trip (four loads per lane in flight) instead of one per class, and the cell
// costs a few hundred instructions instead of a few thousand executed for a
// single lane.  The box arithmetic is that of loss_obj_cell, value for value; the
// softmax sums are formed per lane and then across lanes (the order differs from
// the one-lane form, the result within rounding).  `p`, `b`, `g`: the cell's
// channel 0 in the head tensor, the bias and the gradient (channels S2 floats
// apart).  Every lane of the group must be active.
template <int GW>
__device__ __forceinline__ float group_maxf(float v) {
    v = row16_maxf(v);
#pragma unroll
    for (int o = 16; o < GW; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int GW>
__device__ __forceinline__ float group_sum(float v) {
    v = row16_sum(v);
#pragma unroll
    for (int o = 16; o < GW; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int GW, class L>
__device__ __forceinline__ void loss_obj_cell_wave(
    const L& logit, float* g, const float* __restrict__ gxywh, const float* __restrict__ gcls,
    int cell, int S, int S2, int nb, int nc, float cellf, float imgf, float lc, float ln,
    float inv_gb, float& acc_coord, float& acc_conf, float& acc_cls) {
    const int lane = threadIdx.x & (GW - 1);      // position in the group
    const int c0 = 5 * nb;
    // everything the cell needs from memory is requested here, in one round trip
    float t0[4], q0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = lane + GW * u;
        t0[u] = j < nc ? logit(c0 + j) : 0.0f;
        q0[u] = j < nc ? gcls[j] : 0.0f;
    }
    // ---- boxes and confidences (model.py:190-276)
    const float4 gv = *reinterpret_cast<const float4*>(gxywh);
    const float gnx = gv.x, gny = gv.y, gnw = gv.z, gnh = gv.w;
    const float xoff = cellf * (float)(cell / S);   // first spatial axis carries x (model.py:175-177)
    const float yoff = cellf * (float)(cell % S);
    const Box gbox = decode_box(gnx, gny, gnw, gnh, xoff, yoff, cellf, imgf);
    float nx = 0.f, ny = 0.f, nw = 0.f, nh = 0.f, conf = 0.f, iou = 0.f;
    if (lane < nb) {
        const float l0 = logit(5 * lane), l1 = logit(5 * lane + 1), l2 = logit(5 * lane + 2),
                    l3 = logit(5 * lane + 3), l4 = logit(5 * lane + 4);
        nx = sigmoidf_(l0);
        ny = sigmoidf_(l1);
        nw = sigmoidf_(l2);
        nh = sigmoidf_(l3);
        conf = sigmoidf_(l4);
        iou = iou_pair(decode_box(nx, ny, nw, nh, xoff, yoff, cellf, imgf), gbox);
    }
    // responsible predictor: highest IoU, first index on ties (model.py:210) -- the
    // one-lane rule, applied to the lanes' values in predictor order
    int best = 0;
    float biou = __shfl(iou, 0, GW);
    for (int k = 1; k < nb; ++k) {
        const float v = __shfl(iou, k, GW);
        if (v > biou) { best = k; biou = v; }
    }
    if (lane < nb) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, gc;
        if (lane == best) {
            // coordinate term (model.py:224-240)
            const float dx = nx - gnx, dy = ny - gny;
            const float sw = __builtin_amdgcn_sqrtf(nw), sh = __builtin_amdgcn_sqrtf(nh);
            const float dw = sw - __builtin_amdgcn_sqrtf(gnw), dh = sh - __builtin_amdgcn_sqrtf(gnh);
            acc_coord += lc * ((dx * dx + dy * dy) + (dw * dw + dh * dh));
            const float r0 = 2.0f * lc * dx;
            const float r1 = 2.0f * lc * dy;
            const float r2 = lc * dw * sw * (1.0f - nw);
            const float r3 = lc * dh * sh * (1.0f - nh);
            // objectness term; IoU is part of the graph (model.py:263-269)
            const float dc = conf - iou;
            acc_conf += dc * dc;
            float ul, ut, ur, ub, gl, gt, gr, gb;
            const Box pb = decode_box(nx, ny, nw, nh, xoff, yoff, cellf, imgf, ul, ut, ur, ub);
            iou_pair_bwd(pb, gbox, -2.0f * dc, gl, gt, gr, gb);
            gl = ul >= 0.0f ? gl : 0.0f;     // xywh_to_ltrb backward: a clipped edge passes nothing
            gt = ut >= 0.0f ? gt : 0.0f;
            gr = ur <= imgf ? gr : 0.0f;
            gb = ub <= imgf ? gb : 0.0f;
            const float gx_iou = (gl + gr) * cellf;
            const float gy_iou = (gt + gb) * cellf;
            const float gw_iou = (gr - gl) * 0.5f * imgf;
            const float gh_iou = (gb - gt) * 0.5f * imgf;
            g0 = (r0 + gx_iou) * (nx * (1.0f - nx));
            g1 = (r1 + gy_iou) * (ny * (1.0f - ny));
            g2 = r2 + gw_iou * (nw * (1.0f - nw));
            g3 = r3 + gh_iou * (nh * (1.0f - nh));
            gc = 2.0f * dc * (conf * (1.0f - conf));
        } else {
            // the other predictors of the cell carry the no-object term (model.py:270-274)
            acc_conf += ln * (conf * conf);
            gc = 2.0f * ln * conf * (conf * (1.0f - conf));
        }
        float* q = g + (size_t)(5 * lane) * S2;
        q[0] = g0 * inv_gb;
        q[S2] = g1 * inv_gb;
        q[2 * (size_t)S2] = g2 * inv_gb;
        q[3 * (size_t)S2] = g3 * inv_gb;
        q[4 * (size_t)S2] = gc * inv_gb;
    }
    // ---- classes (model.py:278-294): squared error on the softmax, class j0 + GW u by
    // lane j0.  The first 4 GW classes sit in registers (t0, q0: fetched on entry);
    // a cell with more re-reads the rest in every pass.
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + GW * u < nc) m = fmaxf(m, t0[u]);
    for (int j0 = lane + 4 * GW; j0 < nc; j0 += 4 * GW) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = j0 + GW * u < nc ? logit(c0 + j0 + GW * u) : m;
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(m, t[u]);
    }
    m = group_maxf<GW>(m);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        t0[u] = exp_fast(t0[u] - m);              // from here on: the exponentials
        if (lane + GW * u < nc) s += t0[u];
    }
    for (int j0 = lane + 4 * GW; j0 < nc; j0 += 4 * GW) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = j0 + GW * u < nc ? logit(c0 + j0 + GW * u) : m;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + GW * u < nc) s += exp_fast(t[u] - m);
    }
    s = group_sum<GW>(s);
    const float rs = rcp_fast(s);
    float dot = 0.f, l2 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        t0[u] *= rs;                              // the probabilities
        if (lane + GW * u < nc) {
            const float d = t0[u] - q0[u];
            l2 += d * d;
            dot += 2.0f * d * t0[u];
        }
    }
    for (int j0 = lane + 4 * GW; j0 < nc; j0 += 4 * GW) {
        float t[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t[u] = j0 + GW * u < nc ? logit(c0 + j0 + GW * u) : m;
            q[u] = j0 + GW * u < nc ? gcls[j0 + GW * u] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + GW * u < nc) {
                const float pj = exp_fast(t[u] - m) * rs;
                const float d = pj - q[u];
                l2 += d * d;
                dot += 2.0f * d * pj;
            }
        }
    }
    dot = group_sum<GW>(dot);
    acc_cls += l2;     // per lane; the workgroup reduction adds the lanes
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + GW * u < nc)
            g[(size_t)(c0 + lane + GW * u) * S2] = t0[u] * (2.0f * (t0[u] - q0[u]) - dot) * inv_gb;
    for (int j0 = lane + 4 * GW; j0 < nc; j0 += 4 * GW) {
        float t[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t[u] = j0 + GW * u < nc ? logit(c0 + j0 + GW * u) : m;
            q[u] = j0 + GW * u < nc ? gcls[j0 + GW * u] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + GW * u < nc) {
                const float pj = exp_fast(t[u] - m) * rs;
                g[(size_t)(c0 + j0 + GW * u) * S2] = pj * (2.0f * (pj - q[u]) - dot) * inv_gb;
            }
        }
    }
}

// reads a cell's channels from the head tensor on use
template <bool BIAS>
struct GlobalLogits {
    const float* __restrict__ p;
    const float* __restrict__ b;   // the cell's bias values, channels S2 apart (BIAS only)
    int S2;
    __device__ __forceinline__ float operator()(int ch) const {
        if constexpr (BIAS) return p[ch * S2] + b[ch * S2];
        else return p[ch * S2];
    }
};

// wave -> workgroup reduction of the three partial sums in a fixed order; one
// row of `partials` per workgroup
__device__ __forceinline__ void loss_block_reduce(float acc_coord, float acc_conf, float acc_cls,
                                                  float* __restrict__ partials, int row) {
    __shared__ float red[LOSS_THREADS / 64][3];
    acc_coord = wave_sum(acc_coord);
    acc_conf = wave_sum(acc_conf);
    acc_cls = wave_sum(acc_cls);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave][0] = acc_coord;
        red[wave][1] = acc_conf;
        red[wave][2] = acc_cls;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < LOSS_THREADS / 64; ++w) s += red[w][threadIdx.x];
        partials[(size_t)row * 4 + threadIdx.x] = s;
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Called by all LOSS_THREADS threads of one workgroup; see loss_finalize_kernel.
// Four rows per thread are in flight per round trip.  loss_terms may be null.
__device__ __forceinline__ void loss_sum_rows(const float* __restrict__ partials, int nrows,
                                              float inv_gb, float* __restrict__ loss_terms,
                                              float* __restrict__ log_ring, int log_capacity,
                                              int* __restrict__ log_cursor) {
    __shared__ double red64[LOSS_THREADS / 64][3];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const float4* rows = reinterpret_cast<const float4*>(partials);
    for (int r0 = threadIdx.x; r0 < nrows; r0 += 4 * LOSS_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * LOSS_THREADS;
            v[u] = r < nrows ? rows[r] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 += (double)v[u].x;
            a1 += (double)v[u].y;
            a2 += (double)v[u].z;
        }
    }
    a0 = wave_sum_f64(a0);
    a1 = wave_sum_f64(a1);
    a2 = wave_sum_f64(a2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red64[wave][0] = a0;
        red64[wave][1] = a1;
        red64[wave][2] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0, f = 0.0, k = 0.0;
#pragma unroll
        for (int w = 0; w < LOSS_THREADS / 64; ++w) {
            c += red64[w][0];
            f += red64[w][1];
            k += red64[w][2];
        }
        const double tot = (c + f) + k;
        if (loss_terms) {
            loss_terms[0] = (float)c;
            loss_terms[1] = (float)f;
            loss_terms[2] = (float)k;
            loss_terms[3] = (float)tot;
            loss_terms[4] = (float)(tot * (double)inv_gb);
        }
        if (log_ring) {
            const int cur = *log_cursor;
            float* row = log_ring + (size_t)(cur % log_capacity) * 4;
            row[0] = (float)c;
            row[1] = (float)f;
            row[2] = (float)k;
            row[3] = (float)tot;
            *log_cursor = cur + 1;
        }
    }
}

// Pipelined loss step: a call leaves its rows in one half of the workspace and
// its first workgroup adds the rows the PREVIOUS call left in the other half,
// so no second kernel sits between two steps.  A half starts with a header:
// word 0 = number of rows + 1 (0: nothing pending), word 1 = inv_gb of that call.
struct LossPipe {
    int* own_hdr;            // header of the half this call fills (null: flush only)
    int* prev_hdr;           // header of the half the previous call filled
    const float* prev_rows;
    float* terms_prev;       // receives the previous call's five loss terms; may be null
    float* log_ring;
    int* log_cursor;
    int log_capacity;
};

// Run by all threads of the logging workgroup.  Kernel boundaries order it after
// the call that wrote the rows; nothing else in this launch touches either header.
__device__ __forceinline__ void loss_pipe_log(const LossPipe& p, int own_rows, float own_inv_gb) {
    if (p.own_hdr && threadIdx.x == 0) {
        p.own_hdr[0] = own_rows + 1;
        p.own_hdr[1] = __float_as_int(own_inv_gb);
    }
    const int pending = p.prev_hdr[0];
    if (pending <= 0) return;
    const float inv = __int_as_float(p.prev_hdr[1]);
    loss_sum_rows(p.prev_rows, pending - 1, inv, p.terms_prev, p.log_ring, p.log_capacity,
                  p.log_cursor);
    if (threadIdx.x == 0) p.prev_hdr[0] = 0;   // consumed (after the barrier inside the sum)
}

// Run-time shape (any S, Bx, C within the limits): channels are read on use.
// BIAS: `pred` is the last layer's output before its bias (model.py:133-135) and
// `bias` (one image's worth of values, same layout) is added as the logits are read.
// NHWC: the head tensor and its gradient are cell-major, (batch, S, S, D) -- a cell's D
// values are consecutive and cells are D floats apart (the memory of a channels-last
// (batch, D, S, S) tensor); otherwise channel-major as everywhere else.  An image is
// the same contiguous block of D * S * S floats either way, so the tile serves both.
// TILE: the workgroup's gradients -- one contiguous block of the tensor -- are
// collected in LDS (dynamic, loss_generic_tile_floats() floats) and leave in
// 16-byte stores; `grad` must be 16-byte aligned and the workgroup must own at
// most LOSS_TILE_MAX_ITEMS cells.  Otherwise every lane stores its cell's
// gradients directly, channel by channel.
constexpr int LOSS_TILE_MAX_ITEMS = 1024;
constexpr int LOSS_TILE_MAX_BYTES = 38 * 1024;    // four workgroups per CU with the lists
inline long long loss_generic_tile_floats(int imgs_per_block, int S, int nb, int nc) {
    return (long long)imgs_per_block * (5 * nb + nc) * S * S + 4;
}

template <bool PIPE, bool BIAS, bool TILE, bool NHWC = false>
__global__ __launch_bounds__(LOSS_THREADS) void loss_fwd_bwd_generic_kernel(
    const float* __restrict__ pred, const float* __restrict__ bias,
    const float* __restrict__ gt_xywh,
    const float* __restrict__ gt_cls, const uint8_t* __restrict__ obj_mask,
    float* __restrict__ grad, float* __restrict__ partials, long long batch, int S, int nb,
    int nc, int imgs_per_block, float cellf, float imgf, float lc, float ln, float inv_gb,
    LossPipe pipe) {
    int bid = blockIdx.x;
    if constexpr (PIPE) {
        // workgroup 0 logs the previous call, the others are this call's row 0, 1, ...
        if (bid == 0) {
            loss_pipe_log(pipe, (int)gridDim.x - 1, inv_gb);
            return;
        }
        --bid;
    }
    const int S2 = S * S;
    const long long D = 5 * nb + nc;
    // floats between two channels of a cell, and between two cells of a channel
    const int chs = NHWC ? 1 : S2;
    const int cls_ = NHWC ? (int)D : 1;
    const long long img0 = (long long)bid * imgs_per_block;
    const int nitems = imgs_per_block * S2;       // <= max(LOSS_THREADS, YH_MAX_S^2) = 4096
    // Pass 1, one lane per cell: a cell without an object is finished (its confidence
    // logits in, D gradients out); a cell with one is listed, per wavefront and in
    // cell order, so that the order of every sum is the same from run to run.
    // Pass 2: the listed cells, a group of lanes per cell (loss_obj_cell_wave): 16 or
    // 32 lanes while they cover the classes in one batch of loads, else a wavefront.
    constexpr int NW = LOSS_THREADS / 64;
    // items a wavefront can see: 64 per round
    constexpr int LIST_CAP = (TILE ? LOSS_TILE_MAX_ITEMS : 4096) / NW;
    __shared__ unsigned short olist[NW][LIST_CAP];
    extern __shared__ __attribute__((aligned(16))) float gtile[];   // TILE only
    // `pad` lines the LDS copy up with the 16-byte grid of the global addresses
    const long long gbase = img0 * D * S2;
    const int pad = TILE ? (int)(gbase & 3) : 0;
    const int DS2 = (int)D * S2;
    __shared__ int wcount[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc_coord = 0.f, acc_conf = 0.f, acc_cls = 0.f;
    int listed = 0;                               // wave-uniform
    for (int item0 = 0; item0 < nitems; item0 += LOSS_THREADS) {
        const int item = item0 + threadIdx.x;
        bool has_obj = false;
        if (item < nitems) {
            const int gi = item / S2;
            const int cell = item - gi * S2;
            const long long img = img0 + gi;
            if (img < batch) {
                has_obj = obj_mask[img * S2 + cell] != 0;
                if (!has_obj) {
                    const long long off = img * D * S2 + cell * cls_;
                    const GlobalLogits<BIAS> logit{pred + off, BIAS ? bias + cell : nullptr, chs};
                    float* g = grad + off;
                    if constexpr (TILE) g = gtile + pad + gi * DS2 + cell * cls_;
                    loss_noobj_cell_rt(logit, g, chs, nb, nc, ln, inv_gb, acc_conf);
                }
            }
        }
        const unsigned long long om = __ballot(has_obj);
        if (has_obj)
            olist[wave][listed + __popcll(om & ((1ull << lane) - 1ull))] = (unsigned short)item;
        listed += __popcll(om);
    }
    if (lane == 0) wcount[wave] = listed;
    __syncthreads();
    {
        int cnt[NW];
        int total_obj = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) { cnt[q] = wcount[q]; total_obj += cnt[q]; }
        auto listed_cell = [&](int j0, auto gw) {
            constexpr int GW = decltype(gw)::value;
            int j = j0, src = 0;
#pragma unroll
            for (int q = 0; q < NW - 1; ++q)
                if (src == q && j >= cnt[q]) { j -= cnt[q]; src = q + 1; }
            const int item = olist[src][j];
            const int gi = item / S2;
            const int cell = item - gi * S2;
            const long long img = img0 + gi;
            const long long tcell = img * S2 + cell;
            const long long off = img * D * S2 + cell * cls_;
            const GlobalLogits<BIAS> logit{pred + off, BIAS ? bias + cell : nullptr, chs};
            float* g = grad + off;
            if constexpr (TILE) g = gtile + pad + gi * DS2 + cell * cls_;
            loss_obj_cell_wave<GW>(logit, g, gt_xywh + tcell * 4, gt_cls + tcell * nc, cell, S, chs,
                                   nb, nc, cellf, imgf, lc, ln, inv_gb, acc_coord, acc_conf,
                                   acc_cls);
        };
        if (nc <= 64) {
            for (int j0 = (int)threadIdx.x / 16; j0 < total_obj; j0 += LOSS_THREADS / 16)
                listed_cell(j0, std::integral_constant<int, 16>{});
        } else if (nc <= 128) {
            for (int j0 = (int)threadIdx.x / 32; j0 < total_obj; j0 += LOSS_THREADS / 32)
                listed_cell(j0, std::integral_constant<int, 32>{});
        } else {
            for (int j0 = wave; j0 < total_obj; j0 += NW)
                listed_cell(j0, std::integral_constant<int, 64>{});
        }
    }
    loss_block_reduce(acc_coord, acc_conf, acc_cls, partials, bid);   // has a barrier after the LDS writes
    if constexpr (TILE) {
        const long long left = batch - img0;
        const int n = (int)(left < imgs_per_block ? left : imgs_per_block) * DS2;   // floats of this block
        float* gout = grad + gbase;
        const int head = (4 - pad) & 3;                             // floats up to the 16-byte grid
        if ((int)threadIdx.x < head && (int)threadIdx.x < n)
            gout[threadIdx.x] = gtile[pad + threadIdx.x];
        const int n4 = n > head ? (n - head) >> 2 : 0;
        const float4* s4 = reinterpret_cast<const float4*>(gtile + pad + head);
        float4* g4 = reinterpret_cast<float4*>(gout + head);
        for (int i = threadIdx.x; i < n4; i += LOSS_THREADS) store_stream(g4 + i, s4[i]);
        const int done = head + 4 * n4;
        if ((int)threadIdx.x < n - done) gout[done + threadIdx.x] = gtile[pad + done + threadIdx.x];
    }
}

// Run-time shape, channel-major, a workgroup's block too large for the LDS tile (S=14,
// Bx=5, C=80: 82 KB per image).  Storing the gradients lane by lane leaves them as
// 4-byte pieces in rows of S*S floats that end anywhere in a cache line; here the block
// still leaves in 16-byte non-temporal stores, LOSS_ROLL_FLOATS of it at a time.  What a
// piece of the block holds is known without a tile of the whole block: zeros, except the
// confidence gradients of the cells without an object (kept per cell and predictor in
// LDS, `cg`) and the D gradients of each cell with one (`og`, one row per listed cell,
// written by pass 2 with a channel stride of 1).  Per piece: zero the window, drop the
// confidence values and the object cells' values that fall into it (every thread
// holds the block offsets of its few entries in registers), store.  A workgroup with
// more object cells than `og` has rows for stores everything directly instead.
constexpr int LOSS_ROLL_FLOATS = 4096;          // the window: 16 KB
constexpr int LOSS_ROLL_ENTRIES = 8;            // conf / object entries a thread holds
constexpr int LOSS_ROLL_CAP_FLOATS = LOSS_ROLL_ENTRIES * LOSS_THREADS;   // of cg, and of og
constexpr unsigned int LOSS_ROLL_SKIP = 0x7FC0DEADu;   // cg entry of a cell that holds an object

template <bool BIAS, bool NHWC = false, bool PIPE = false>
__global__ __launch_bounds__(LOSS_THREADS) void loss_fwd_bwd_roll_kernel(
    const float* __restrict__ pred, const float* __restrict__ bias,
    const float* __restrict__ gt_xywh, const float* __restrict__ gt_cls,
    const uint8_t* __restrict__ obj_mask, float* __restrict__ grad, float* __restrict__ partials,
    long long batch, int S, int nb, int nc, int imgs_per_block, float cellf, float imgf, float lc,
    float ln, float inv_gb, int og_cap, LossPipe pipe) {
    int bid = blockIdx.x;
    if constexpr (PIPE) {
        // workgroup 0 logs the previous call, the others are this call's row 0, 1, ...
        if (bid == 0) {
            loss_pipe_log(pipe, (int)gridDim.x - 1, inv_gb);
            return;
        }
        --bid;
    }
    const int S2 = S * S;
    const int D = 5 * nb + nc;
    const int DS2 = D * S2;
    // floats between two channels of a cell and between two cells of a channel (NHWC: the
    // block is cell-major, see loss_fwd_bwd_generic_kernel)
    const int chs = NHWC ? 1 : S2;
    const int cls_ = NHWC ? D : 1;
    const long long img0 = (long long)bid * imgs_per_block;
    const int nitems = imgs_per_block * S2;      // * nb <= LOSS_ROLL_CAP_FLOATS (the host checks)
    constexpr int NW = LOSS_THREADS / 64;
    constexpr int LIST_CAP = 4096 / NW;
    __shared__ unsigned short olist[NW][LIST_CAP];
    __shared__ int wcount[NW];
    extern __shared__ __attribute__((aligned(16))) float rollmem[];
    float* win = rollmem;                                   // LOSS_ROLL_FLOATS
    float* og = win + LOSS_ROLL_FLOATS;                     // og_cap * D
    float* cg = og + (size_t)og_cap * D;                    // nitems * nb
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc_coord = 0.f, acc_conf = 0.f, acc_cls = 0.f;
    int listed = 0;
    // pass 1: the confidence gradients of the cells without an object; the others are listed
    for (int item0 = 0; item0 < nitems; item0 += LOSS_THREADS) {
        const int item = item0 + threadIdx.x;
        bool has_obj = false;
        if (item < nitems) {
            const int gi = item / S2;
            const int cell = item - gi * S2;
            const long long img = img0 + gi;
            if (img < batch) {
                has_obj = obj_mask[img * S2 + cell] != 0;
                float* c = cg + (size_t)item * nb;
                if (!has_obj) {
                    const GlobalLogits<BIAS> logit{pred + img * (long long)DS2 + cell * cls_,
                                                   BIAS ? bias + cell : nullptr, chs};
                    for (int k0 = 0; k0 < nb; k0 += 4) {
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = k0 + u < nb ? logit(5 * (k0 + u) + 4) : 0.0f;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (k0 + u < nb) {
                                const float conf = sigmoidf_(v[u]);
                                acc_conf += ln * (conf * conf);
                                c[k0 + u] = 2.0f * ln * conf * (conf * (1.0f - conf)) * inv_gb;
                            }
                        }
                    }
                } else {
                    for (int k = 0; k < nb; ++k) c[k] = __uint_as_float(LOSS_ROLL_SKIP);
                }
            }
        }
        const unsigned long long om = __ballot(has_obj);
        if (has_obj)
            olist[wave][listed + __popcll(om & ((1ull << lane) - 1ull))] = (unsigned short)item;
        listed += __popcll(om);
    }
    if (lane == 0) wcount[wave] = listed;
    __syncthreads();
    int cnt[NW];
    int total_obj = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) { cnt[q] = wcount[q]; total_obj += cnt[q]; }
    const bool direct = total_obj > og_cap;      // the same for the whole workgroup
    auto listed_item = [&](int j0) -> int {
        int j = j0, src = 0;
#pragma unroll
        for (int q = 0; q < NW - 1; ++q)
            if (src == q && j >= cnt[q]) { j -= cnt[q]; src = q + 1; }
        return olist[src][j];
    };
    if (direct) {
        // every cell's D values straight to memory first (zeros and the confidence
        // gradients; an object cell gets zeros here and its values from pass 2 below)
        for (int item = threadIdx.x; item < nitems; item += LOSS_THREADS) {
            const int gi = item / S2;
            const int cell = item - gi * S2;
            if (img0 + gi < batch) {
                float* g = grad + (img0 + gi) * (long long)DS2 + cell * cls_;
                const float* c = cg + (size_t)item * nb;
                for (int ch = 0; ch < D; ++ch) {
                    float v = 0.0f;
                    if (ch < 5 * nb && ch % 5 == 4) {
                        v = c[ch / 5];
                        if (__float_as_uint(v) == LOSS_ROLL_SKIP) v = 0.0f;
                    }
                    g[(size_t)ch * chs] = v;
                }
            }
        }
        __syncthreads();     // those stores are performed before pass 2 stores to the same words
    }
    // pass 2: the listed cells, a group of lanes per cell
    auto listed_cell = [&](int j0, auto gw) {
        constexpr int GW = decltype(gw)::value;
        const int item = listed_item(j0);
        const int gi = item / S2;
        const int cell = item - gi * S2;
        const long long img = img0 + gi;
        const long long tcell = img * S2 + cell;
        const long long off = img * (long long)DS2 + cell * cls_;
        const GlobalLogits<BIAS> logit{pred + off, BIAS ? bias + cell : nullptr, chs};
        float* g = direct ? grad + off : og + (size_t)j0 * D;
        loss_obj_cell_wave<GW>(logit, g, gt_xywh + tcell * 4, gt_cls + tcell * nc, cell, S,
                               direct ? chs : 1, nb, nc, cellf, imgf, lc, ln, inv_gb, acc_coord,
                               acc_conf, acc_cls);
    };
    if (nc <= 64) {
        for (int j0 = (int)threadIdx.x / 16; j0 < total_obj; j0 += LOSS_THREADS / 16)
            listed_cell(j0, std::integral_constant<int, 16>{});
    } else if (nc <= 128) {
        for (int j0 = (int)threadIdx.x / 32; j0 < total_obj; j0 += LOSS_THREADS / 32)
            listed_cell(j0, std::integral_constant<int, 32>{});
    } else {
        for (int j0 = wave; j0 < total_obj; j0 += NW)
            listed_cell(j0, std::integral_constant<int, 64>{});
    }
    loss_block_reduce(acc_coord, acc_conf, acc_cls, partials, bid);   // has a barrier after the LDS writes
    if (direct) return;
    // ---- the block leaves, a window at a time
    const long long gbase = img0 * (long long)DS2;
    const long long left = batch - img0;
    const int n = (int)(left < imgs_per_block ? left : imgs_per_block) * DS2;   // floats of this block
    float* gout = grad + gbase;
    // block offsets and values of this thread's entries: confidence gradients ...
    int ce[LOSS_ROLL_ENTRIES], oe[LOSS_ROLL_ENTRIES];
    float cv[LOSS_ROLL_ENTRIES], ov[LOSS_ROLL_ENTRIES];
#pragma unroll
    for (int u = 0; u < LOSS_ROLL_ENTRIES; ++u) {
        const int q = u * LOSS_THREADS + (int)threadIdx.x;
        ce[u] = -1; cv[u] = 0.f; oe[u] = -1; ov[u] = 0.f;
        if (q < nitems * nb) {
            const int item = q / nb, k = q - item * nb;
            const int gi = item / S2, cell = item - gi * S2;
            const float v = cg[q];
            if (img0 + gi < batch && __float_as_uint(v) != LOSS_ROLL_SKIP) {
                ce[u] = gi * DS2 + (5 * k + 4) * chs + cell * cls_;
                cv[u] = v;
            }
        }
        // ... and the values of the object cells
        if (q < total_obj * D) {
            const int j0 = q / D, ch = q - j0 * D;
            const int item = listed_item(j0);
            const int gi = item / S2, cell = item - gi * S2;
            oe[u] = gi * DS2 + ch * chs + cell * cls_;
            ov[u] = og[q];
        }
    }
    // position of the block's first float on the 16-byte grid of the ADDRESSES: the buffer
    // itself may be only 4-byte aligned (this kernel also serves such buffers), and every
    // 16-byte store below must land on that grid
    const int pad = (int)((reinterpret_cast<uintptr_t>(gout) >> 2) & 3);
    const int head = n < ((4 - pad) & 3) ? n : ((4 - pad) & 3);   // floats up to the 16-byte grid
    const int body = ((n - head) >> 2) << 2;                      // floats that leave as float4
    // one window [c0, c0 + len) of the block: zero, drop the entries in, store
    // The barriers between the steps order LDS accesses only.  __syncthreads() would also
    // wait for the window's global stores to be acknowledged -- a trip to HBM per window.
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto window = [&](int c0, int len, bool vec) {
        lds_barrier();                                            // the last window has been read
        for (int i = threadIdx.x; i < (len + 3) / 4; i += LOSS_THREADS)
            reinterpret_cast<float4*>(win)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        lds_barrier();
#pragma unroll
        for (int u = 0; u < LOSS_ROLL_ENTRIES; ++u) {
            if (ce[u] >= c0 && ce[u] < c0 + len) win[ce[u] - c0] = cv[u];
            if (oe[u] >= c0 && oe[u] < c0 + len) win[oe[u] - c0] = ov[u];
        }
        lds_barrier();
        if (vec) {
            float4* g4 = reinterpret_cast<float4*>(gout + c0);
            for (int i = threadIdx.x; i < len / 4; i += LOSS_THREADS)
                store_stream(g4 + i, reinterpret_cast<const float4*>(win)[i]);
        } else {
            if ((int)threadIdx.x < len) gout[c0 + threadIdx.x] = win[threadIdx.x];
        }
    };
    if (head > 0) window(0, head, false);
    for (int c0 = head; c0 < head + body; c0 += LOSS_ROLL_FLOATS)
        window(c0, head + body - c0 < LOSS_ROLL_FLOATS ? head + body - c0 : LOSS_ROLL_FLOATS, true);
    if (n - head - body > 0) window(head + body, n - head - body, false);
}

// Compile-time shape.  A lane keeps the logits it needs in registers; with S*S
// known, every channel of a cell is a constant byte offset from the cell's
// address (no per-channel address arithmetic).

// LDS floats of the gradient tile of one workgroup
template <int S, int BX, int C>
constexpr int loss_tile_floats() {
    return loss_ipb_for(S * S) * (5 * BX + C) * S * S + 4;
}

// Pass 2 of the compile-time-shape loss, by all threads of the workgroup after the barrier
// behind pass 1.  `olist` holds the listed cells (item = image-in-workgroup * S*S + cell)
// per wavefront, 64 entries each and in ascending item order, `wcount` their number per
// wavefront; the j0-th listed cell of the workgroup is worked on by thread j0 (box and
// confidence work; with two predictors by threads 2 j0 and 2 j0 + 1, a predictor each) and by
// thread LOSS_THREADS/2 + j0 (class work): several lanes share what would be one long path.
// The cell's logits wait in the tile slots its gradients then take
// (`tile` = the tile's first float of the workgroup's block).  `gt(w, j, tcell, gx, gc)` sets
// the addresses of the cell's ground-truth box and class row, given the wavefront whose list
// holds it, its position there and its cell index in the batch.
template <int S, int BX, int C, class GT>
__device__ __forceinline__ void loss_pass2(const int* olist, const int* wcount, float* tile,
                                           long long img0, const GT& gt, float cellf, float imgf,
                                           float lc, float ln, float inv_gb, float& acc_coord,
                                           float& acc_conf, float& acc_cls) {
    constexpr int D = 5 * BX + C;
    constexpr int S2 = S * S;
    int cnt[LOSS_THREADS / 64];
    int total_obj = 0;
#pragma unroll
    for (int q = 0; q < LOSS_THREADS / 64; ++q) { cnt[q] = wcount[q]; total_obj += cnt[q]; }
    constexpr int HALF = LOSS_THREADS / 2;
    const bool cls_role = (int)threadIdx.x >= HALF;
    if constexpr (BX == 2) {
        // Two predictors: the box work of a cell goes to a PAIR of lanes, a predictor each
        // (loss_obj_cell_box_pair) -- thread 2 j0 + k for predictor k of the j0-th listed cell.
        if (!cls_role) {
            const int k = (int)threadIdx.x & 1;
            for (int j0 = (int)threadIdx.x >> 1; j0 < total_obj; j0 += HALF / 2) {
                int j = j0;
                int src = 0;
#pragma unroll
                for (int q = 0; q < LOSS_THREADS / 64 - 1; ++q)
                    if (src == q && j >= cnt[q]) { j -= cnt[q]; src = q + 1; }
                const int item = olist[src * 64 + j];
                const int gi = item / S2;
                const int cell = item - gi * S2;
                const long long tcell = (img0 + gi) * S2 + cell;
                float* gl = tile + gi * (D * S2) + cell + (5 * k) * S2;
                float v5[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) v5[u] = gl[u * S2];
                auto logit5 = [&](int u) -> float { return v5[u]; };
                const float *gx, *gc;
                gt(src, j, tcell, gx, gc);
                loss_obj_cell_box_pair(logit5, gl, gx, cell, S, S2, k, cellf, imgf, lc, ln, inv_gb, acc_coord,
                                       acc_conf);
            }
            return;
        }
    }
    for (int j0 = (int)threadIdx.x - (cls_role ? HALF : 0); j0 < total_obj; j0 += HALF) {
        int j = j0;
        int src = 0;
#pragma unroll
        for (int q = 0; q < LOSS_THREADS / 64 - 1; ++q)
            if (src == q && j >= cnt[q]) { j -= cnt[q]; src = q + 1; }
        const int item = olist[src * 64 + j];
        const int gi = item / S2;
        const int cell = item - gi * S2;
        const long long img = img0 + gi;
        const long long tcell = img * S2 + cell;
        float* gl = tile + gi * (D * S2) + cell;
        float v[D];
#pragma unroll
        for (int ch = 0; ch < D; ++ch) v[ch] = gl[ch * S2];
        auto logit = [&](int ch) -> float { return v[ch]; };
        const float *gx, *gc;
        gt(src, j, tcell, gx, gc);
        if (cls_role)
            loss_obj_cell<C, 2>(logit, gl, gx, gc, cell, S, S2, BX, C, cellf, imgf, lc, ln, inv_gb,
                                acc_coord, acc_conf, acc_cls);
        else
            loss_obj_cell<C, 1>(logit, gl, gx, gc, cell, S, S2, BX, C, cellf, imgf, lc, ln, inv_gb,
                                acc_coord, acc_conf, acc_cls);
    }
}

// The workgroup's gradient tile (`sgrad`, its block starting `pad` floats in) leaves for
// grad[gbase ...] in 16-byte non-temporal stores, whole cache lines per wavefront
// instruction; the few floats before and after the 16-byte grid go one by one.
template <int S, int BX, int C>
__device__ __forceinline__ void loss_tile_store(const float* sgrad, int pad, float* __restrict__ grad,
                                                long long gbase, long long img0, long long batch) {
    constexpr int D = 5 * BX + C;
    constexpr int S2 = S * S;
    constexpr int IPB = loss_ipb_for(S2);
    const long long left = batch - img0;
    const int n = (int)(left < IPB ? left : IPB) * (D * S2);   // floats of this block
    float* gout = grad + gbase;
    const int head = (4 - pad) & 3;                             // floats up to the 16-byte grid
    if ((int)threadIdx.x < head && (int)threadIdx.x < n)
        gout[threadIdx.x] = sgrad[pad + threadIdx.x];
    const int n4 = n > head ? (n - head) >> 2 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(sgrad + pad + head);
    float4* g4 = reinterpret_cast<float4*>(gout + head);
    for (int i = threadIdx.x; i < n4; i += LOSS_THREADS) store_stream(g4 + i, s4[i]);
    const int done = head + 4 * n4;
    if ((int)threadIdx.x < n - done) gout[done + threadIdx.x] = sgrad[pad + done + threadIdx.x];
}

// The work of workgroup `bid` of a grid of `nblocks` (with PIPE, workgroup 0 logs
// the previous call and the others are this call's rows 0, 1, ...); `sgrad` is an
// LDS tile of loss_tile_floats() floats, 16-byte aligned.
template <int S, int BX, int C, bool PIPE, bool BIAS>
__device__ __forceinline__ void loss_block(
    const float* __restrict__ pred, const float* __restrict__ bias,
    const float* __restrict__ gt_xywh,
    const float* __restrict__ gt_cls, const uint8_t* __restrict__ obj_mask,
    float* __restrict__ grad, float* __restrict__ partials, long long batch,
    float cellf, float imgf, float lc, float ln, float inv_gb,
    const LossPipe& pipe, int bid, int nblocks, float* sgrad) {
    if constexpr (PIPE) {
        if (bid == 0) {
            loss_pipe_log(pipe, nblocks - 1, inv_gb);
            return;
        }
        --bid;
    }
    constexpr int D = 5 * BX + C;
    constexpr int S2 = S * S;
    constexpr int IPB = loss_ipb_for(S2);      // images per workgroup (the host's choice too)
    const long long img0 = (long long)bid * IPB;
    const int nitems = IPB * S2;
    // The workgroup's gradients are one contiguous block of the tensor.  The lanes
    // leave theirs in LDS and the block goes out in 16-byte stores, whole cache
    // lines per wavefront instruction, instead of 30 four-byte stores per lane in
    // runs of S*S floats.  `pad` lines the LDS copy up with the 16-byte grid of
    // the global addresses.
    const long long gbase = img0 * (long long)D * S2;
    const int pad = (int)(gbase & 3);
    YH_LSTAMP(0);

    // Pass 1, one lane per cell: cells without an object are finished here (two
    // confidence logits in, their gradients into the tile); cells with one are only
    // listed, per wavefront, in cell order.  Pass 2: the listed cells -- about 6 %
    // of all -- are worked on by the first lanes of the workgroup, so the long
    // object path runs in one wavefront per workgroup instead of in every one.
    __shared__ int olist[LOSS_THREADS];
    __shared__ int wcount[LOSS_THREADS / 64];
    static_assert(IPB * S2 <= LOSS_THREADS, "one cell per lane");
    // The ground truth of a listed cell is requested by the lane that owns the cell, together
    // with the cell's logits (one trip to memory), and waits in LDS for pass 2: per wavefront
    // LOSS_GT_SLOTS slots of 4 + C floats, in list order; a cell beyond them is fetched by its
    // pass-2 lanes (a second, dependent trip).
    constexpr bool kPark = LOSS_GT_SLOTS > 0 && C % 4 == 0;
    constexpr int GTF = 4 + C;
    __shared__ __attribute__((aligned(16))) float gtpark[kPark ? (LOSS_THREADS / 64) * LOSS_GT_SLOTS * GTF : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc_coord = 0.f, acc_conf = 0.f, acc_cls = 0.f;
    {
        const int item = threadIdx.x;
        const int gi = item / S2;
        const int cell = item - gi * S2;
        const long long img = img0 + gi;
        const bool mine = item < nitems && img < batch;
        const long long tcell = mine ? img * S2 + cell : 0;
        const float* p = pred + (mine ? img * (long long)D * S2 + cell : 0);
        // every cell needs its confidence logits: requested together with the mask
        float vc[BX];
#pragma unroll
        for (int k = 0; k < BX; ++k) vc[k] = mine ? p[(5 * k + 4) * S2] : 0.0f;
        const bool has_obj = mine && obj_mask[tcell] != 0;
        // the cells that hold an object are listed per wavefront, in cell order
        const unsigned long long omask = __ballot(has_obj);
        const int opos = __popcll(omask & ((1ull << lane) - 1ull));
        if (has_obj) olist[wave * 64 + opos] = threadIdx.x;
        if (lane == 0) wcount[wave] = __popcll(omask);
        if (has_obj) {
            // the cell's logits are requested now and wait for pass 2 in the very tile
            // slots their gradients will take, its ground truth in its slot of the wavefront's
            // list (one memory round trip less on the path of the two wavefronts that do pass 2)
            float v[D];
#pragma unroll
            for (int ch = 0; ch < D; ++ch) {
                if (ch < 5 * BX && ch % 5 == 4) v[ch] = vc[ch / 5];
                else v[ch] = p[ch * S2];
            }
            if constexpr (kPark) {
                if (opos < LOSS_GT_SLOTS) {
                    float4 gq[GTF / 4];
                    gq[0] = *reinterpret_cast<const float4*>(gt_xywh + tcell * 4);
                    const float4* gc4 = reinterpret_cast<const float4*>(gt_cls + tcell * C);
#pragma unroll
                    for (int q = 0; q < C / 4; ++q) gq[1 + q] = gc4[q];
                    float4* slot = reinterpret_cast<float4*>(gtpark) + (wave * LOSS_GT_SLOTS + opos) * (GTF / 4);
#pragma unroll
                    for (int q = 0; q < GTF / 4; ++q) slot[q] = gq[q];
                }
            }
            if constexpr (BIAS) {
                const float* bq = bias + cell;
#pragma unroll
                for (int ch = 0; ch < D; ++ch) v[ch] += bq[ch * S2];
            }
            float* gl = sgrad + pad + gi * (D * S2) + cell;
#pragma unroll
            for (int ch = 0; ch < D; ++ch) gl[ch * S2] = v[ch];
        } else if (mine) {
            if constexpr (BIAS) {
#pragma unroll
                for (int k = 0; k < BX; ++k) vc[k] += bias[(5 * k + 4) * S2 + cell];
            }
            auto logit = [&](int ch) -> float { return vc[ch / 5]; };   // asked for 5k+4 only
            loss_noobj_cell(logit, sgrad + pad + gi * (D * S2) + cell, S2, BX, C, ln, inv_gb,
                            acc_conf);
        }
    }
    YH_LSTAMP(1);
    __syncthreads();
    YH_LSTAMP(2);
    loss_pass2<S, BX, C>(olist, wcount, sgrad + pad, img0,
                         [&](int w, int j, long long tcell, const float*& gx, const float*& gc) {
                             if (kPark && j < LOSS_GT_SLOTS) {
                                 gx = gtpark + (w * LOSS_GT_SLOTS + j) * GTF;
                                 gc = gx + 4;
                             } else {
                                 gx = gt_xywh + tcell * 4;
                                 gc = gt_cls + tcell * C;
                             }
                         },
                         cellf, imgf, lc, ln, inv_gb, acc_coord, acc_conf, acc_cls);
    YH_LSTAMP(3);
    loss_block_reduce(acc_coord, acc_conf, acc_cls, partials, bid);   // has a barrier after the LDS writes
    YH_LSTAMP(4);
    loss_tile_store<S, BX, C>(sgrad, pad, grad, gbase, img0, batch);
    YH_LSTAMP(5);
}

template <int S, int BX, int C, bool PIPE, bool BIAS>
__global__ __launch_bounds__(LOSS_THREADS, 4) void loss_fwd_bwd_kernel(
    const float* __restrict__ pred, const float* __restrict__ bias,
    const float* __restrict__ gt_xywh,
    const float* __restrict__ gt_cls, const uint8_t* __restrict__ obj_mask,
    float* __restrict__ grad, float* __restrict__ partials, long long batch,
    int imgs_per_block, float cellf, float imgf, float lc, float ln, float inv_gb,
    LossPipe pipe) {
    (void)imgs_per_block;
    __shared__ __attribute__((aligned(16))) float sgrad[loss_tile_floats<S, BX, C>()];
    loss_block<S, BX, C, PIPE, BIAS>(pred, bias, gt_xywh, gt_cls, obj_mask, grad, partials, batch,
                                     cellf, imgf, lc, ln, inv_gb, pipe, (int)blockIdx.x,
                                     (int)gridDim.x, sgrad);
}

// One workgroup of LOSS_THREADS threads: adds the rows in a fixed order (fp64
// accumulation) and writes coord, conf, class, total and total * inv_gb.  If
// log_ring is given the four sums also go to row (*log_cursor % log_capacity) of
// the device-side loss log and the cursor advances (it lives on the device, so a
// replayed graph keeps logging).
__global__ __launch_bounds__(LOSS_THREADS) void loss_finalize_kernel(
    const float* __restrict__ partials, int nrows, float inv_gb, float* __restrict__ loss_terms,
    float* __restrict__ log_ring, int log_capacity, int* __restrict__ log_cursor) {
    loss_sum_rows(partials, nrows, inv_gb, loss_terms, log_ring, log_capacity, log_cursor);
}

// The pipelined loss step on its own: logs what the last call left behind in its
// half, then looks at the other half as well (at most one holds a pending step
// when the phases alternated as documented).
__global__ __launch_bounds__(LOSS_THREADS) void loss_pipe_flush_kernel(LossPipe last,
                                                                       LossPipe other) {
    loss_pipe_log(last, 0, 0.f);
    __syncthreads();
    loss_pipe_log(other, 0, 0.f);
}

// Column sums of a (nrows, ncols) matrix in a fixed order.  Block (x, y) covers
// columns [64 x, 64 x + 64) of rows [y * rows_per_slice, (y + 1) * rows_per_slice):
// wavefront w adds rows w, w + 4, ... of the slice (eight loads in flight), the
// four sums are added in wavefront order, the result goes to row y of `out`.  A
// second launch over those rows finishes.  Used for the gradient of the last
// layer's bias: the sum of grad_pred over the batch.
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out, long long nrows,
                                                         int ncols, int rows_per_slice) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const long long r0 = (long long)blockIdx.y * rows_per_slice;
    const long long r1 = r0 + rows_per_slice < nrows ? r0 + rows_per_slice : nrows;
    float acc = 0.f;
    if (col < ncols) {
        long long r = r0 + w;
        const float* p = in + r * ncols + col;
        for (; r + 28 < r1; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long long)(4 * u) * ncols];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
            p += 32LL * ncols;
        }
        for (; r < r1; r += 4) {
            acc += *p;
            p += 4LL * ncols;
        }
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < ncols)
        out[(long long)blockIdx.y * ncols + col] =
            ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// grad[i] *= *scale, skipped entirely when *scale == 1 (the loss.backward() case).
__global__ __launch_bounds__(256) void scale_grad_kernel(float* __restrict__ grad,
                                                         const float* __restrict__ scale,
                                                         long long n) {
    const float s = *scale;
    if (s == 1.0f) return;
    const long long n4 = n / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float4* g4 = reinterpret_cast<float4*>(grad);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = g4[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        g4[i] = v;
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        grad[i] *= s;
}

}  // namespace yh
