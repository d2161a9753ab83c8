// yh_map_11point and its workspace size: the mAP entry point of include/yolo_head.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstring>

#include "../../include/yolo_head.h"
#include "map_kernels.hpp"
#include "radix_sort.hpp"

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

// two buffers of sort words for the predictions (and for the ground truths when they are
// sorted by class), the true-positive classes, and the histogram of the radix sort
struct MapLayout {
    size_t words_a, words_b, tpcls, gwords_a, gwords_b, hist, total;
};

MapLayout map_layout(int64_t P, int64_t G) {
    MapLayout l = {};
    size_t off = 0;
    const size_t p8 = up256((size_t)(P < 1 ? 1 : P) * 8);
    l.words_a = off; off += p8;
    l.words_b = off; off += p8;
    l.tpcls = off; off += up256((size_t)(P < 1 ? 1 : P) * 4);
    const bool sorted_gt = G > yh::MAP_GT_DIRECT;
    if (sorted_gt) {
        const size_t g8 = up256((size_t)G * 8);
        l.gwords_a = off; off += g8;
        l.gwords_b = off; off += g8;
    }
    const size_t hp = yh::radix_hist_words(1, P < 1 ? 1 : P), hg = sorted_gt ? yh::radix_hist_words(1, G) : 0;
    l.hist = off; off += up256((hp > hg ? hp : hg) * 4);
    l.total = off;
    return l;
}

}  // namespace

extern "C" {

size_t yh_map_workspace_bytes(int64_t P, int64_t G) {
    if (P < 0 || G < 0 || P > YH_MAP_MAX_ROWS || G > YH_MAP_MAX_ROWS) return 0;
    return map_layout(P, G).total;
}

int yh_map_11point(const float* pred, const float* gt, int64_t P, int64_t G, int n_classes,
                   float iou_thresh, double* ap_out, double* map_out, void* workspace,
                   size_t workspace_bytes, void* stream) {
    if (P < 0 || G < 0 || n_classes < 1 || n_classes > 65535) return YH_ERR_SHAPE;
    if (P > YH_MAP_MAX_ROWS || G > YH_MAP_MAX_ROWS) return YH_ERR_UNSUPPORTED;
    if (isnan(iou_thresh)) return YH_ERR_ARG;
    if (!ap_out || !map_out || (P > 0 && !pred) || (G > 0 && !gt)) return YH_ERR_NULL;
    const MapLayout l = map_layout(P, G);
    if (!workspace || workspace_bytes < l.total) return YH_ERR_WORKSPACE;
    if (!aligned(pred, 4) || !aligned(gt, 4) || !aligned(ap_out, 8) || !aligned(map_out, 8) ||
        !aligned(workspace, 256))
        return YH_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned long long* words_a = (unsigned long long*)(ws + l.words_a);
    unsigned long long* words_b = (unsigned long long*)(ws + l.words_b);
    unsigned int* hist = (unsigned int*)(ws + l.hist);
    int* tpcls = (int*)(ws + l.tpcls);
    if (P <= yh::MAP_SMALL_P && G <= yh::MAP_GT_DIRECT && n_classes <= yh::MAP_SMALL_CLASSES) {
        // a small call: everything in one workgroup, one launch (the workspace is not used)
        hipLaunchKernelGGL(yh::map_small_kernel, dim3(1), dim3(yh::MAP_SMALL_P), 0, st, pred, gt, (int)P,
                           (int)G, n_classes, iou_thresh, ap_out, map_out);
        return launch_status();
    }
    const unsigned long long* order = words_a;   // the ranked sort words (low half: the row)
    unsigned long long* spare = words_b;         // the buffer the sort leaves free
    if (P > 0) {
        const unsigned grid = (unsigned)((P + 255) / 256);
        if (G > yh::MAP_GT_DIRECT) {
            unsigned long long* ga = (unsigned long long*)(ws + l.gwords_a);
            unsigned long long* gb = (unsigned long long*)(ws + l.gwords_b);
            hipLaunchKernelGGL(yh::map_gt_keys_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st,
                               gt, (long long)G, n_classes, ga);
            if (const int rc = launch_status()) return rc;
            int err = 0;
            // classes are below 65,536: two passes over bits 32..47
            const unsigned long long* gsorted = yh::radix_sort_u64(ga, gb, hist, 1, G, 32, 48, st, &err);
            if (!gsorted) return err;
            hipLaunchKernelGGL((yh::map_tp_key_kernel<true, 1>), dim3(grid), dim3(256), 0, st, pred, gt,
                               (long long)P, (long long)G, n_classes, iou_thresh, gsorted, tpcls, words_a);
        } else if (P <= 16384) {
            // few predictions: four slices of a workgroup share the scan of the ground truths
            hipLaunchKernelGGL((yh::map_tp_key_kernel<false, 4>), dim3((unsigned)((P + 63) / 64)), dim3(256),
                               0, st, pred, gt, (long long)P, (long long)G, n_classes, iou_thresh, nullptr,
                               tpcls, words_a);
        } else {
            hipLaunchKernelGGL((yh::map_tp_key_kernel<false, 1>), dim3(grid), dim3(256), 0, st, pred, gt,
                               (long long)P, (long long)G, n_classes, iou_thresh, nullptr, tpcls, words_a);
        }
        if (const int rc = launch_status()) return rc;
        int err = 0;
        unsigned long long* sorted = yh::radix_sort_u64(words_a, words_b, hist, 1, P, 32, 64, st, &err);
        if (!sorted) return err;
        order = sorted;
        spare = sorted == words_a ? words_b : words_a;
    }
    // a long ranked list is gathered once (into the buffer the sort leaves free)
    int* ranked = nullptr;
    if (P > 65536) {
        ranked = (int*)spare;
        hipLaunchKernelGGL(yh::map_gather_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                           tpcls, order, (long long)P, ranked);
        if (const int rc = launch_status()) return rc;
    }
    hipLaunchKernelGGL(yh::map_ap_kernel, dim3((unsigned)n_classes), dim3(yh::MAP_AP_THREADS), 0, st,
                       tpcls, order, ranked, (long long)P, (long long)G, ap_out);
    if (const int rc = launch_status()) return rc;
    hipLaunchKernelGGL(yh::map_mean_kernel, dim3(1), dim3(64), 0, st, ap_out, n_classes, map_out);
    return launch_status();
}

}  // extern "C"
