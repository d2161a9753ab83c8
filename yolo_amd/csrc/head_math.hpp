// Scalar device math of the YOLOv1 head: sigmoid, grid decode, IoU forward and
// backward.  Every function restates one reference primitive in fp32 with the
// reference's operation order; the translation unit is compiled with
// -ffp-contract=off so no multiply-add is fused and the IoU bits match a CPU
// evaluation of the same expression.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace yh {

struct Box {
    float l, t, r, b;
};

// v_max_f32 / v_min_f32 on values as they are: fmaxf/fminf would first quiet a
// signalling NaN in each operand (one more instruction per operand)
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Streaming (non-temporal) stores for outputs that no workgroup reads again: the
// lines leave the L2 as they fill instead of waiting there for the write-back at
// the end of the kernel, which otherwise sits between two kernels of a stream
// (24 MB of gradients per launch at batch 4096).  Worth it only where a wavefront
// instruction covers whole 128-byte lines; partial lines cost a read-modify-write.
typedef float yh_v4f __attribute__((ext_vector_type(4)));
typedef float yh_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream(float4* dst, const float4& q) {
    const yh_v4f v = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(v, reinterpret_cast<yh_v4f*>(dst));
}
__device__ __forceinline__ void store_stream(float2* dst, const float2& q) {
    const yh_v2f v = {q.x, q.y};
    __builtin_nontemporal_store(v, reinterpret_cast<yh_v2f*>(dst));
}

// exp(x) on the hardware base-2 exponential (v_exp_f32, 1 ulp).  The product
// x*log2(e) is split with a fused multiply-add so that neither its rounding nor
// the rounding of the constant reaches the result: about 2 ulp on [-87, 88.72],
// at 7 instructions instead of the ~15 of a correctly rounded expf.  The
// argument is held inside [-104, 88.72] (one v_med3_f32): above, v_exp_f32
// returns +inf and inf * r + inf is NaN whenever the residual r is negative;
// far below, x*log2(e) itself overflows and r becomes inf - inf.  exp(-104)
// is below the smallest denormal and exp(88.72) = 3.39e38 is finite; the
// callers form 1 / (1 + exp) or exp of a value <= 0, so the plateaus cost less
// than 3e-39 absolute.
__device__ __forceinline__ float exp_fast(float x_in) {
    constexpr float kLog2e = 1.44269502162933349609375f;   // fp32(log2 e)
    constexpr float kLog2eLo = 1.925963033500011e-08f;      // log2 e - kLog2e
    constexpr float kLn2 = 0.693147182464599609375f;
    const float x = __builtin_amdgcn_fmed3f(x_in, -104.0f, 88.72f);
    const float t = x * kLog2e;
    const float r = __builtin_fmaf(x, kLog2e, -t) + x * kLog2eLo;   // what t lost
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, r * kLn2, e);
}

// 1/x on v_rcp_f32 (1 ulp)
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

// torch.sigmoid, ~3 ulp
__device__ __forceinline__ float sigmoidf_(float x) { return rcp_fast(1.0f + exp_fast(-x)); }

// denormalize_xywh (model.py:167-183) then xywh_to_ltrb (model.py:150-155) for one
// box of the cell whose grid offsets are (xoff, yoff) = cell_size * (i, j).
// ul/ut/ur/ub receive the edges before the one-sided clips (the backward needs
// to know whether a clip was active).
__device__ __forceinline__ Box decode_box(float nx, float ny, float nw, float nh, float xoff,
                                          float yoff, float cellf, float imgf, float& ul,
                                          float& ut, float& ur, float& ub) {
    const float x = nx * cellf + xoff;
    const float y = ny * cellf + yoff;
    const float hw = (nw * imgf) / 2.0f;
    const float hh = (nh * imgf) / 2.0f;
    ul = x - hw;
    ut = y - hh;
    ur = x + hw;
    ub = y + hh;
    Box o;
    o.l = fmaxf(ul, 0.0f);
    o.t = fmaxf(ut, 0.0f);
    o.r = fminf(ur, imgf);
    o.b = fminf(ub, imgf);
    return o;
}

__device__ __forceinline__ Box decode_box(float nx, float ny, float nw, float nh, float xoff,
                                          float yoff, float cellf, float imgf) {
    float a, b, c, d;
    return decode_box(nx, ny, nw, nh, xoff, yoff, cellf, imgf, a, b, c, d);
}

// get_area (eval.py:22-32)
__device__ __forceinline__ float box_area(const Box& p) {
    return fmaxf(p.r - p.l, 0.0f) * fmaxf(p.b - p.t, 0.0f);
}

// get_iou (eval.py:53-65) for one pair: inter / ((a1 + a2) - inter), 0/0 -> 0.
__device__ __forceinline__ float iou_pair(const Box& p, const Box& g) {
    const float a1 = box_area(p);
    const float a2 = box_area(g);
    const float iw = fmaxf(fminf(p.r, g.r) - fmaxf(p.l, g.l), 0.0f);
    const float ih = fmaxf(fminf(p.b, g.b) - fmaxf(p.t, g.t), 0.0f);
    const float inter = iw * ih;
    const float uni = (a1 + a2) - inter;
    return uni == 0.0f ? 0.0f : inter / uni;
}

// "iou_pair(p, g) > thr" without the division where that is safe: outside the band
// [lo, hi] = thr * (1 -+ 2e-6) the products lo * union and hi * union decide (each is
// rounded once, 6e-8 relative, far inside the band); a lane inside the band, or with a
// union of zero or near the denormals, takes the IEEE division.  thr <= 0 or not
// finite: lo = -inf, hi = +inf, every lane divides.  Same answer as iou_pair(...) > thr,
// always.
struct IouBand {
    float thr, lo, hi;
};
__device__ __forceinline__ IouBand make_iou_band(float thr) {
    IouBand b = {thr, -INFINITY, INFINITY};
    if (thr > 0.0f && thr < 1e30f) {
        b.lo = thr * (1.0f - 2e-6f);
        b.hi = thr * (1.0f + 2e-6f);
    }
    return b;
}
// As iou_above_band below with the two areas (box_area of each box) handed in, for a loop that
// keeps one box for many tests; a lane that is unsure divides, the others skip the division.
__device__ __forceinline__ bool iou_above_band_areas(const Box& p, float a1, const Box& g, float a2,
                                                     const IouBand& band) {
    const float iw = fmaxf(vmin_raw(p.r, g.r) - vmax_raw(p.l, g.l), 0.0f);
    const float ih = fmaxf(vmin_raw(p.b, g.b) - vmax_raw(p.t, g.t), 0.0f);
    const float inter = iw * ih;
    const float uni = (a1 + a2) - inter;
    // inter > hi * uni (rounded once, 6e-8 relative) puts the quotient above thr with room to
    // spare, inter < lo * uni below it.  The 1e-30 added on either side widens the band by
    // nothing for boxes of ordinary size and keeps a union of zero or near the denormals
    // (boxes of 1e-15 px, where a product no longer carries 24 bits) inside it; so does a
    // band of +-inf.  A lane inside the band divides.
    const float t_hi = __builtin_fmaf(band.hi, uni, 1e-30f);
    const float t_lo = __builtin_fmaf(band.lo, uni, -1e-30f);
    bool above = inter > t_hi;
    if (!(inter > t_hi) && !(inter < t_lo)) above = (uni == 0.0f ? 0.0f : inter / uni) > band.thr;
    return above;
}
// The same answer with one branch region instead of two nested ones (both compares first, then
// the lanes that are unsure divide): for loops where the test is most of the body.
__device__ __forceinline__ bool iou_above_band_flat(const Box& p, float a1, const Box& g, float a2,
                                                    const IouBand& band) {
    const float iw = fmaxf(vmin_raw(p.r, g.r) - vmax_raw(p.l, g.l), 0.0f);
    const float ih = fmaxf(vmin_raw(p.b, g.b) - vmax_raw(p.t, g.t), 0.0f);
    const float inter = iw * ih;
    const float uni = (a1 + a2) - inter;
    bool above = inter > __builtin_fmaf(band.hi, uni, 1e-30f);
    const bool unsure = !above & !(inter < __builtin_fmaf(band.lo, uni, -1e-30f));
    if (unsure) above = (uni == 0.0f ? 0.0f : inter / uni) > band.thr;
    return above;
}
__device__ __forceinline__ bool iou_above_band(const Box& p, const Box& g, const IouBand& band) {
    return iou_above_band_areas(p, box_area(p), g, box_area(g), band);
}

// share of the gradient that max(a, b) (or min) sends to a: all of it when a
// wins, half on an exact tie (torch.maximum / torch.minimum backward).
__device__ __forceinline__ float pick_max(float a, float b) {
    return a > b ? 1.0f : (a == b ? 0.5f : 0.0f);
}
__device__ __forceinline__ float pick_min(float a, float b) {
    return a < b ? 1.0f : (a == b ? 0.5f : 0.0f);
}

// d iou_pair(p, g) / d p, scaled by the upstream gradient `giou` (g is constant).
// torch.clip(x, min=0) passes the gradient where x >= 0.
__device__ __forceinline__ void iou_pair_bwd(const Box& p, const Box& g, float giou, float& gl,
                                             float& gt, float& gr, float& gb) {
    const float w1 = p.r - p.l, h1 = p.b - p.t;
    const float cw1 = fmaxf(w1, 0.0f), ch1 = fmaxf(h1, 0.0f);
    const float a1 = cw1 * ch1;
    const float a2 = box_area(g);
    const float il = fmaxf(p.l, g.l), it = fmaxf(p.t, g.t);
    const float ir = fminf(p.r, g.r), ib = fminf(p.b, g.b);
    const float iw = ir - il, ih = ib - it;
    const float ciw = fmaxf(iw, 0.0f), cih = fmaxf(ih, 0.0f);
    const float inter = ciw * cih;
    const float uni = (a1 + a2) - inter;
    gl = gt = gr = gb = 0.0f;
    if (uni == 0.0f) return;  // torch.where routes no gradient into the quotient
    // iou = inter / uni, uni = a1 + a2 - inter
    const float ru = rcp_fast(uni);
    const float g_uni = -giou * inter * (ru * ru);
    const float g_inter = giou * ru - g_uni;
    const float g_a1 = g_uni;
    // intersection extent
    const float g_iw = iw >= 0.0f ? g_inter * cih : 0.0f;
    const float g_ih = ih >= 0.0f ? g_inter * ciw : 0.0f;
    gr += g_iw * pick_min(p.r, g.r);
    gl -= g_iw * pick_max(p.l, g.l);
    gb += g_ih * pick_min(p.b, g.b);
    gt -= g_ih * pick_max(p.t, g.t);
    // own area
    const float g_w1 = w1 >= 0.0f ? g_a1 * ch1 : 0.0f;
    const float g_h1 = h1 >= 0.0f ? g_a1 * cw1 : 0.0f;
    gr += g_w1;
    gl -= g_w1;
    gb += g_h1;
    gt -= g_h1;
}

// monotone map of a float to an unsigned key (larger float -> larger key)
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- wave64 helpers -----------------------------------------------------------
// Lane i of a row of 16 receives the value of lane perm(i) of the same row through
// the data-parallel-primitive path of the VALU (no LDS round trip): quad_perm
// [1,0,3,2] and [2,3,0,1] are lane ^ 1 and lane ^ 2, row_half_mirror is 7 - i within
// each half row, row_mirror is 15 - i.  Every lane of the row must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum / maximum over each row of 16 lanes, delivered to all 16
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_row<0xB1>(v);
    v += dpp_row<0x4E>(v);
    v += dpp_row<0x141>(v);
    v += dpp_row<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_maxf(float v) {
    v = fmaxf(v, dpp_row<0xB1>(v));
    v = fmaxf(v, dpp_row<0x4E>(v));
    v = fmaxf(v, dpp_row<0x141>(v));
    v = fmaxf(v, dpp_row<0x140>(v));
    return v;
}
// Exclusive prefix sum over the 64 lanes on the DPP path: a shifted-add scan
// inside each row of 16, then lane 15 of rows 0 and 2 into rows 1 and 3
// (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).  All lanes active.
__device__ __forceinline__ int wave_excl_scan(int v) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return x - v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace yh
