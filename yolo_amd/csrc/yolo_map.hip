// yh_map_11point and its workspace size: the mAP entry point of
// include/yolo_head.h, in a translation unit of its own because it pulls in the
// rocPRIM device radix sort.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/yolo_head.h"
#include "map_kernels.hpp"

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

// temporary storage rocPRIM wants for a pair sort of n 32-bit keys (a host-side query)
size_t sort_temp_bytes(size_t n) {
    if (n == 0) return 0;
    size_t bytes = 0;
    unsigned int* k = nullptr;
    const hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, bytes, k, k, k, k, n, 0, 32, (hipStream_t)0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return up256(16 * n + (1u << 20));   // no device to ask: a generous bound
    }
    return bytes;
}

struct MapLayout {
    size_t keys_in, keys_out, idx_in, idx_out, tpcls, gkeys_in, gkeys_out, gidx_in, gidx_out, temp,
        temp_bytes, total;
};

MapLayout map_layout(int64_t P, int64_t G) {
    MapLayout l = {};
    size_t off = 0;
    const size_t p4 = up256((size_t)(P < 1 ? 1 : P) * 4);
    l.keys_in = off; off += p4;
    l.keys_out = off; off += p4;
    l.idx_in = off; off += p4;
    l.idx_out = off; off += p4;
    l.tpcls = off; off += p4;
    const bool sorted_gt = G > yh::MAP_GT_DIRECT;
    if (sorted_gt) {
        const size_t g4 = up256((size_t)G * 4);
        l.gkeys_in = off; off += g4;
        l.gkeys_out = off; off += g4;
        l.gidx_in = off; off += g4;
        l.gidx_out = off; off += g4;
    }
    const size_t tp = sort_temp_bytes((size_t)P), tg = sorted_gt ? sort_temp_bytes((size_t)G) : 0;
    l.temp_bytes = up256(tp > tg ? tp : tg);
    l.temp = off; off += l.temp_bytes;
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
    unsigned int* keys_in = (unsigned int*)(ws + l.keys_in);
    unsigned int* keys_out = (unsigned int*)(ws + l.keys_out);
    unsigned int* idx_in = (unsigned int*)(ws + l.idx_in);
    unsigned int* idx_out = (unsigned int*)(ws + l.idx_out);
    int* tpcls = (int*)(ws + l.tpcls);
    void* temp = ws + l.temp;
    size_t temp_bytes = l.temp_bytes;
    if (P <= yh::MAP_SMALL_P && G <= yh::MAP_GT_DIRECT && n_classes <= yh::MAP_SMALL_CLASSES) {
        // a small call: everything in one workgroup, one launch (the workspace is not used)
        hipLaunchKernelGGL(yh::map_small_kernel, dim3(1), dim3(yh::MAP_SMALL_P), 0, st, pred, gt, (int)P,
                           (int)G, n_classes, iou_thresh, ap_out, map_out);
        return launch_status();
    }
    if (P > 0) {
        const unsigned grid = (unsigned)((P + 255) / 256);
        if (G > yh::MAP_GT_DIRECT) {
            unsigned int* gk_in = (unsigned int*)(ws + l.gkeys_in);
            unsigned int* gk_out = (unsigned int*)(ws + l.gkeys_out);
            unsigned int* gi_in = (unsigned int*)(ws + l.gidx_in);
            unsigned int* gi_out = (unsigned int*)(ws + l.gidx_out);
            hipLaunchKernelGGL(yh::map_gt_keys_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st,
                               gt, (long long)G, n_classes, gk_in, gi_in);
            if (const int rc = launch_status()) return rc;
            const hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, gk_in, gk_out, gi_in,
                                                           gi_out, (size_t)G, 0, 32, st);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((yh::map_tp_key_kernel<true, 1>), dim3(grid), dim3(256), 0, st, pred, gt,
                               (long long)P, (long long)G, n_classes, iou_thresh, gk_out, gi_out, tpcls,
                               keys_in, idx_in);
        } else if (P <= 16384) {
            // few predictions: four slices of a workgroup share the scan of the ground truths
            hipLaunchKernelGGL((yh::map_tp_key_kernel<false, 4>), dim3((unsigned)((P + 63) / 64)), dim3(256),
                               0, st, pred, gt, (long long)P, (long long)G, n_classes, iou_thresh, nullptr,
                               nullptr, tpcls, keys_in, idx_in);
        } else {
            hipLaunchKernelGGL((yh::map_tp_key_kernel<false, 1>), dim3(grid), dim3(256), 0, st, pred, gt,
                               (long long)P, (long long)G, n_classes, iou_thresh, nullptr, nullptr, tpcls,
                               keys_in, idx_in);
        }
        if (const int rc = launch_status()) return rc;
        const hipError_t e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, idx_in,
                                                            idx_out, (size_t)P, 0, 32, st);
        if (e != hipSuccess) return (int)e;
    }
    // a long ranked list is gathered once (into the unsorted keys' place, free after the sort)
    int* ranked = nullptr;
    if (P > 65536) {
        ranked = (int*)keys_in;
        hipLaunchKernelGGL(yh::map_gather_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                           tpcls, idx_out, (long long)P, ranked);
        if (const int rc = launch_status()) return rc;
    }
    hipLaunchKernelGGL(yh::map_ap_kernel, dim3((unsigned)n_classes), dim3(yh::MAP_AP_THREADS), 0, st,
                       tpcls, idx_out, ranked, (long long)P, (long long)G, ap_out);
    if (const int rc = launch_status()) return rc;
    hipLaunchKernelGGL(yh::map_mean_kernel, dim3(1), dim3(64), 0, st, ap_out, n_classes, map_out);
    return launch_status();
}

}  // extern "C"
