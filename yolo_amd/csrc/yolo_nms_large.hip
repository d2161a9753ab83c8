// yh_decode_nms_ws / yh_nms_ws: the NMS entry points with a workspace, which take
// more candidates per image than the LDS-resident kernels (global-memory path with a
// segmented radix sort, one segment per image).  Up to YH_MAX_CANDIDATES they forward to
// yh_decode_nms / yh_decode_nms_linear / yh_nms and need no workspace.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstring>

#include "../../include/yolo_head.h"
#include "nms_large_kernels.hpp"
#include "radix_sort.hpp"

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

struct LargeLayout {
    size_t keys_a, keys_b, boxes, hist, total;
};

LargeLayout large_layout(int64_t batch, int N, bool own_boxes) {
    LargeLayout l = {};
    size_t off = 0;
    const size_t k8 = up256((size_t)batch * N * 8);
    l.keys_a = off; off += k8;
    l.keys_b = off; off += k8;
    if (own_boxes) { l.boxes = off; off += up256((size_t)batch * N * 16); }
    l.hist = off; off += up256(yh::radix_hist_words(batch, N) * 4);
    l.total = off;
    return l;
}

// The keys are class | inverted score | flat index and lie in flat-index order, so a stable sort
// on the class and score bits alone orders them: four passes for the score and one for every
// eight bits the classes take (class_bits: an upper bound on the bits of a valid class).  Keys
// that are not valid are all ones: the largest digit in every pass, so they come out last.
int sort_and_suppress(const LargeLayout& l, char* ws, const float4* boxes, const float* scores,
                      float* dets, int32_t* keep_count, int32_t* keep_idx, int64_t batch, int N,
                      int class_bits, float iou_thresh, hipStream_t st) {
    const int bit_hi = 48 + 8 * ((class_bits + 7) / 8);
    int err = 0;
    const unsigned long long* sorted =
        yh::radix_sort_u64((unsigned long long*)(ws + l.keys_a), (unsigned long long*)(ws + l.keys_b),
                           (unsigned int*)(ws + l.hist), batch, N, 16, bit_hi, st, &err);
    if (!sorted) return err;
    hipLaunchKernelGGL(yh::nms_large_kernel, dim3((unsigned)batch), dim3(256), 0, st, sorted, boxes, scores,
                       dets, keep_count, keep_idx, N, iou_thresh);
    return launch_status();
}

inline int bits_for(int n_values) {     // bits that hold 0 .. n_values - 1
    int b = 0;
    while ((1 << b) < n_values) ++b;
    return b;
}

}  // namespace

extern "C" {

size_t yh_nms_workspace_bytes(int64_t batch, int N, int own_boxes) {
    if (batch < 1 || N <= YH_MAX_CANDIDATES || N > YH_MAX_CANDIDATES_WS) return 0;
    return large_layout(batch, N, own_boxes != 0).total;
}

int yh_decode_nms_ws(const float* pred, const float* bias, float* dets, int32_t* keep_count,
                     int32_t* keep_idx, int mode, float conf_thresh, float iou_thresh,
                     int64_t batch, int S, int Bx, int C, int img_size, void* workspace,
                     size_t workspace_bytes, void* stream) {
    const int64_t N = (int64_t)S * S * Bx;
    if (S < 1 || S > YH_MAX_S || Bx < 1 || Bx > YH_MAX_BX || N <= YH_MAX_CANDIDATES)
        return bias ? yh_decode_nms_linear(pred, bias, dets, keep_count, keep_idx, mode, conf_thresh,
                                           iou_thresh, batch, S, Bx, C, img_size, stream)
                    : yh_decode_nms(pred, dets, keep_count, keep_idx, mode, conf_thresh, iou_thresh,
                                    batch, S, Bx, C, img_size, stream);
    if (batch < 0 || C < 1 || C > YH_MAX_C || img_size < 1 || img_size % S != 0) return YH_ERR_SHAPE;
    if (batch > ((int64_t)1 << 40) / N) return YH_ERR_SHAPE;
    if (mode != YH_MODE_REFERENCE && mode != YH_MODE_DENSE) return YH_ERR_ARG;
    if (isnan(conf_thresh) || isnan(iou_thresh)) return YH_ERR_ARG;
    if (batch == 0) return YH_OK;
    if (!pred || !dets || !keep_count) return YH_ERR_NULL;
    if (!aligned(pred, 4) || !aligned(bias, 4) || !aligned(dets, 8) || !aligned(keep_count, 4) ||
        !aligned(keep_idx, 4))
        return YH_ERR_ALIGN;
    const LargeLayout l = large_layout(batch, (int)N, true);
    if (!workspace || workspace_bytes < l.total) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 256)) return YH_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float4* boxes = (float4*)(ws + l.boxes);
    unsigned long long* keys = (unsigned long long*)(ws + l.keys_a);
    int64_t blocks = (batch * S * S + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const float cellf = (float)(img_size / S), imgf = (float)img_size;
    if (bias)
        hipLaunchKernelGGL(yh::nms_large_decode_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st,
                           pred, bias, boxes, keys, (long long)batch, S, Bx, C, mode, cellf, imgf,
                           conf_thresh);
    else
        hipLaunchKernelGGL(yh::nms_large_decode_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st,
                           pred, bias, boxes, keys, (long long)batch, S, Bx, C, mode, cellf, imgf,
                           conf_thresh);
    if (const int rc = launch_status()) return rc;
    return sort_and_suppress(l, ws, boxes, nullptr, dets, keep_count, keep_idx, batch, (int)N,
                             bits_for(C), iou_thresh, st);
}

int yh_nms_ws(const float* cand_ltrb, const float* cand_score, const int32_t* cand_cls,
              const uint8_t* cand_present, float* dets, int32_t* keep_count, int32_t* keep_idx,
              float conf_thresh, float iou_thresh, int64_t batch, int N, void* workspace,
              size_t workspace_bytes, void* stream) {
    if (N <= YH_MAX_CANDIDATES)
        return yh_nms(cand_ltrb, cand_score, cand_cls, cand_present, dets, keep_count, keep_idx,
                      conf_thresh, iou_thresh, batch, N, stream);
    if (batch < 0) return YH_ERR_SHAPE;
    if (N > YH_MAX_CANDIDATES_WS) return YH_ERR_UNSUPPORTED;
    if (batch > ((int64_t)1 << 40) / N) return YH_ERR_SHAPE;
    if (isnan(conf_thresh) || isnan(iou_thresh)) return YH_ERR_ARG;
    if (batch == 0) return YH_OK;
    if (!cand_ltrb || !cand_score || !cand_cls || !dets || !keep_count) return YH_ERR_NULL;
    if (!aligned(cand_ltrb, 16) || !aligned(cand_score, 4) || !aligned(cand_cls, 4) ||
        !aligned(dets, 8) || !aligned(keep_count, 4) || !aligned(keep_idx, 4))
        return YH_ERR_ALIGN;
    const LargeLayout l = large_layout(batch, N, false);
    if (!workspace || workspace_bytes < l.total) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 256)) return YH_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int64_t blocks = (batch * N + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(yh::nms_large_cand_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       cand_score, cand_cls, cand_present, (unsigned long long*)(ws + l.keys_a),
                       (long long)batch, N, conf_thresh);
    if (const int rc = launch_status()) return rc;
    return sort_and_suppress(l, ws, (const float4*)cand_ltrb, cand_score, dets, keep_count, keep_idx,
                             batch, N, 15, iou_thresh, st);      // classes up to 32,767
}

}  // extern "C"
