// yh_decode_nms_ws / yh_nms_ws: the NMS entry points with a workspace, which take
// more candidates per image than the LDS-resident kernels (global-memory path with
// the rocPRIM segmented radix sort).  Up to YH_MAX_CANDIDATES they forward to
// yh_decode_nms / yh_decode_nms_linear / yh_nms and need no workspace.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "../../include/yolo_head.h"
#include "nms_large_kernels.hpp"

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

using OffsetIt = rocprim::transform_iterator<rocprim::counting_iterator<unsigned int>,
                                             yh::NmsSegmentOffset, unsigned int>;

// images sorted per rocPRIM call: its sizes are 32-bit
inline int64_t images_per_sort(int64_t batch, int N) {
    int64_t g = ((int64_t)1 << 30) / N;
    if (g < 1) g = 1;
    return g < batch ? g : batch;
}

hipError_t sort_images(void* temp, size_t& temp_bytes, const unsigned long long* in,
                       unsigned long long* out, int64_t images, int N, hipStream_t st) {
    const OffsetIt begin(rocprim::counting_iterator<unsigned int>(0), yh::NmsSegmentOffset{(unsigned)N, 0});
    const OffsetIt end(rocprim::counting_iterator<unsigned int>(0), yh::NmsSegmentOffset{(unsigned)N, 1});
    return rocprim::segmented_radix_sort_keys(temp, temp_bytes, in, out, (unsigned int)(images * N),
                                              (unsigned int)images, begin, end, 0, 64, st);
}

size_t sort_temp_bytes(int64_t batch, int N) {
    size_t bytes = 0;
    const hipError_t e = sort_images(nullptr, bytes, nullptr, nullptr, images_per_sort(batch, N), N, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return up256((size_t)images_per_sort(batch, N) * N * 16 + (1u << 20));   // no device to ask
    }
    return up256(bytes);
}

struct LargeLayout {
    size_t keys_in, keys_out, boxes, temp, temp_bytes, total;
};

LargeLayout large_layout(int64_t batch, int N, bool own_boxes) {
    LargeLayout l = {};
    size_t off = 0;
    const size_t k8 = up256((size_t)batch * N * 8);
    l.keys_in = off; off += k8;
    l.keys_out = off; off += k8;
    if (own_boxes) { l.boxes = off; off += up256((size_t)batch * N * 16); }
    l.temp_bytes = sort_temp_bytes(batch, N);
    l.temp = off; off += l.temp_bytes;
    l.total = off;
    return l;
}

int sort_and_suppress(const LargeLayout& l, char* ws, const float4* boxes, const float* scores,
                      float* dets, int32_t* keep_count, int32_t* keep_idx, int64_t batch, int N,
                      float iou_thresh, hipStream_t st) {
    const unsigned long long* kin = (const unsigned long long*)(ws + l.keys_in);
    unsigned long long* kout = (unsigned long long*)(ws + l.keys_out);
    const int64_t per = images_per_sort(batch, N);
    for (int64_t i0 = 0; i0 < batch; i0 += per) {
        const int64_t n = batch - i0 < per ? batch - i0 : per;
        size_t tb = l.temp_bytes;
        const hipError_t e = sort_images(ws + l.temp, tb, kin + i0 * N, kout + i0 * N, n, N, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(yh::nms_large_kernel, dim3((unsigned)batch), dim3(256), 0, st, kout, boxes, scores,
                       dets, keep_count, keep_idx, N, iou_thresh);
    return launch_status();
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
    unsigned long long* keys = (unsigned long long*)(ws + l.keys_in);
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
                             iou_thresh, st);
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
                       cand_score, cand_cls, cand_present, (unsigned long long*)(ws + l.keys_in),
                       (long long)batch, N, conf_thresh);
    if (const int rc = launch_status()) return rc;
    return sort_and_suppress(l, ws, (const float4*)cand_ltrb, cand_score, dets, keep_count, keep_idx,
                             batch, N, iou_thresh, st);
}

}  // extern "C"
