// libyolohead: the C ABI declared in include/yolo_head.h.  Host side only checks
// arguments and enqueues kernels on the caller's stream; it never allocates,
// synchronises or keeps state, so every entry point can be captured into a
// HIP graph.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/yolo_head.h"
#include "decode_kernels.hpp"
#include "loss_kernels.hpp"
#include "nms_kernels.hpp"
#include "step_kernels.hpp"
#include "per_image_kernels.hpp"
#include "target_kernels.hpp"
#include "box_kernels.hpp"

// Loss rows of yh_loss_fwd_bwd_static: device memory that belongs to the library
// image itself, one copy per GPU at that GPU's own address.  In host code the
// name is only a handle for hipGetSymbolAddress, never a device address.
static __device__ __align__(256) float scratch[YH_STATIC_SCRATCH_BYTES / sizeof(float)];

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int check_shape(int64_t batch, int S, int Bx, int C, int img_size) {
    if (batch < 0 || S < 1 || S > YH_MAX_S || Bx < 1 || Bx > YH_MAX_BX || C < 1 ||
        C > YH_MAX_C || img_size < 1)
        return YH_ERR_SHAPE;
    // cell_size = img_size // S (model.py:114) must tile the image exactly, as the
    // reference's arange(0, img_size, cell_size) grid requires (model.py:172-174)
    if (img_size % S != 0) return YH_ERR_SHAPE;
    if (batch * (int64_t)S * S > (int64_t)1 << 40) return YH_ERR_SHAPE;
    return YH_OK;
}

inline int loss_imgs_per_block(int S) {
    const int S2 = S * S;
    return S2 <= yh::LOSS_THREADS ? yh::loss_ipb_for(S2) : 1;
}

inline int64_t loss_blocks(int64_t batch, int S) {
    const int g = loss_imgs_per_block(S);
    return (batch + g - 1) / g;
}

inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

inline int round16(int x) { return (x + 15) & ~15; }

// LDS of one gfx950 CU; a launch that asks for more is refused on the host
constexpr size_t kLdsBudget = 160 * 1024;

// workgroups of 256 threads for an elementwise kernel with a grid-stride loop
inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > (1 << 20)) b = 1 << 20;
    return (unsigned)b;
}

// LDS layout of one image's region in nms_kernel
void nms_layout(yh::NmsArgs& a, int slot_pad, int imgs_per_block = 1) {
    a.off_keys = 0;
    a.off_box = slot_pad * 8;
    a.off_meta = slot_pad * 24;
    a.off_seg = slot_pad * 32;
    int off = slot_pad * (8 + 16 + 8 + 4);
    a.off_alive = off;
    off += round16((slot_pad / 64) * 8);
    a.off_misc = off;
    off += 128;
    a.off_ccls = off;
    off += round16(a.S * a.S * 4);
    a.off_hist = off;     // class histogram / bucket cursors and bucket starts (fused path)
    off += round16(a.nc * 4);
    a.off_segst = off;
    off += round16(a.nc * 4);
    // kill and precedence tables of the one-wavefront path: 2 x 64 words of 64 bits
    // (2 x 128 slots x 2 words with two candidates per lane)
    a.off_kill = off;
    off += slot_pad == 128 ? 4096 : 1024;
    // a cell-major image is staged through LDS when the blocks of a workgroup's images are
    // small enough to leave room for a second workgroup on the CU
    a.off_stage = -1;
    const int block_bytes = (5 * a.nb + a.nc) * a.S * a.S * 4;
    // (with a candidate per predictor the tables are large already and a block above 8 KB costs
    // a workgroup per CU: 16 lanes per cell read it from memory instead, see nms_block)
    const int stage_budget = a.mode == YH_MODE_DENSE ? 8 * 1024 * imgs_per_block : 48 * 1024;
    if (a.cell_major && (long long)block_bytes * imgs_per_block <= stage_budget) {
        a.off_stage = off;
        off += round16(block_bytes);
    }
    a.region_bytes = off;
}

// row slices of the first column-sum pass: about 32 rows each, at most 1024 slices
int colsum_slices(int64_t nrows) {
    int64_t s = (nrows + 31) / 32;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

template <int SRC, int S_CT, int BX_CT, int C_CT, bool CM = false>
int launch_nms_shape(yh::NmsArgs a, hipStream_t st) {
#define YH_NMS_CASE(NT, ITEMS)                                                              \
    {                                                                                       \
        constexpr int threads = yh::nms_block_threads(NT);                                  \
        nms_layout(a, NT * ITEMS, threads / NT);                                            \
        const int ipb = threads / NT;                                                       \
        const unsigned grid = (unsigned)((a.batch + ipb - 1) / ipb);                        \
        const size_t lds = (size_t)a.region_bytes * ipb;                                    \
        if (lds > kLdsBudget) return YH_ERR_UNSUPPORTED;                                    \
        hipLaunchKernelGGL((yh::nms_kernel<NT, ITEMS, SRC, S_CT, BX_CT, C_CT, CM>), dim3(grid), \
                           dim3(threads), lds, st, a);                                      \
        return launch_status();                                                             \
    }
    // a compile-time shape has two possible slot counts (S*S and S*S*Bx): only the
    // cases those need are instantiated
    constexpr int kCells = S_CT * S_CT, kBoxes = S_CT * S_CT * BX_CT;
#define YH_NMS_FITS(LO, HI) \
    (S_CT == 0 || (kCells > LO && kCells <= HI) || (kBoxes > LO && kBoxes <= HI))
    if constexpr (YH_NMS_FITS(0, 64)) if (a.nslot <= 64) YH_NMS_CASE(64, 1)
    if constexpr (YH_NMS_FITS(64, 128)) if (a.nslot <= 128) YH_NMS_CASE(64, 2)
    if constexpr (YH_NMS_FITS(128, 256)) if (a.nslot <= 256) YH_NMS_CASE(256, 1)
    if constexpr (YH_NMS_FITS(256, 512)) if (a.nslot <= 512) YH_NMS_CASE(256, 2)
    if constexpr (YH_NMS_FITS(512, 1024)) if (a.nslot <= 1024) {
        // A compile-time shape whose class logits can be shared out over more threads than
        // cells (the stress shape S=14, Bx=5, C=80): 512 threads per image at 64 registers,
        // so that four images still share a CU, each with twice the wavefronts (43.9 against
        // 46.8 us at batch 1024, 21.9 against 29.9 at 256, 162.7 against 182.6 at 4096)
        if constexpr (S_CT > 0 && 2 * S_CT * S_CT <= 512 && C_CT >= 32 && C_CT % 16 == 0) {
            if (a.mode == YH_MODE_DENSE) {
                nms_layout(a, 1024, 1);
                if ((size_t)a.region_bytes > kLdsBudget) return YH_ERR_UNSUPPORTED;
                hipLaunchKernelGGL((yh::nms_kernel_wide<2, SRC, S_CT, BX_CT, C_CT>),
                                   dim3((unsigned)a.batch), dim3(512), (size_t)a.region_bytes, st, a);
                return launch_status();
            }
        }
        YH_NMS_CASE(256, 4)
    }
    if constexpr (YH_NMS_FITS(1024, 2048)) if (a.nslot <= 2048) YH_NMS_CASE(256, 8)
    return YH_ERR_UNSUPPORTED;
#undef YH_NMS_FITS
#undef YH_NMS_CASE
}

template <int SRC>
int launch_nms(const yh::NmsArgs& a, hipStream_t st) {
    if (a.batch == 0) return YH_OK;
    if (a.nslot > YH_MAX_CANDIDATES || a.nout > YH_MAX_CANDIDATES) return YH_ERR_UNSUPPORTED;
    if constexpr (SRC == 0) {
        // the two shapes the head is benchmarked on get compile-time channel counts
        // (channel-major tensors only: their channel offsets are compile-time constants)
        if (a.cell_major) return launch_nms_shape<0, 0, 0, 0, true>(a, st);
        if (a.S == 7 && a.nb == 2 && a.nc == 20) return launch_nms_shape<0, 7, 2, 20>(a, st);
        if (a.S == 14 && a.nb == 5 && a.nc == 80) return launch_nms_shape<0, 14, 5, 80>(a, st);
        if (a.S == 7 && a.nb == 2 && a.nc == 80) return launch_nms_shape<0, 7, 2, 80>(a, st);
    }
    if constexpr (SRC == 2) {
        if (a.S == 7 && a.nb == 2 && a.nc == 20) return launch_nms_shape<2, 7, 2, 20>(a, st);
    }
    return launch_nms_shape<SRC, 0, 0, 0>(a, st);
}

}  // namespace

extern "C" {

int yh_version(void) { return YH_VERSION; }

#ifdef YH_PHASE_PROBE
// probe builds only: where loss_block leaves its time stamps (nullptr: nowhere)
int yh_probe_set_loss_stamps(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(yh::yh_loss_stamps), &p, sizeof(p));
}
#endif

const char* yh_strerror(int rc) {
    switch (rc) {
        case YH_OK: return "ok";
        case YH_ERR_NULL: return "yolo_head: a required pointer is NULL";
        case YH_ERR_SHAPE:
            return "yolo_head: shape out of range (need batch >= 0, 1 <= S <= 64, 1 <= Bx <= 16, "
                   "1 <= C <= 4096, img_size a positive multiple of S)";
        case YH_ERR_WORKSPACE: return "yolo_head: workspace missing or too small";
        case YH_ERR_ALIGN: return "yolo_head: pointer not aligned as documented";
        case YH_ERR_ARG: return "yolo_head: invalid scalar argument";
        case YH_ERR_UNSUPPORTED:
            return "yolo_head: more candidates per image (or rows) than this entry point takes, or a shape "
                   "whose per-image tables do not fit the 160 KB of LDS; see the limits in yolo_head.h "
                   "(yh_decode_nms_ws / yh_nms_ws take up to 65,536 candidates per image)";
        case YH_ERR_SCRATCH:
            return "yolo_head: the batch needs more loss rows than the static scratch array holds "
                   "(or its device address could not be resolved)";
        default: break;
    }
    if (rc > 0) return hipGetErrorString((hipError_t)rc);
    return "yolo_head: unknown error code";
}

size_t yh_workspace_bytes(int op, int64_t batch, int S, int Bx, int C) {
    (void)Bx; (void)C;
    if (op == YH_OP_MAP) return yh_map_workspace_bytes(batch, 0);   // few ground truths; else ask that function
    if ((op != YH_OP_LOSS && op != YH_OP_LOSS_PIPELINED) || batch < 0 || S < 1 || S > YH_MAX_S)
        return 0;
    const int64_t rows = loss_blocks(batch, S);
    const size_t bytes = (size_t)(rows < 1 ? 1 : rows) * 4 * sizeof(float);
    // pipelined: two halves, each a 16-byte header and the rows
    return op == YH_OP_LOSS_PIPELINED ? 2 * (16 + bytes) : bytes;
}

}  // extern "C"

namespace {

// Argument checks and launch shared by yh_loss_partials (pipe == nullptr, rows at
// `rows_base`) and the pipelined step (workgroup 0 logs the previous call).
int loss_launch(const float* pred, const float* bias, const float* gt_xywh, const float* gt_cls,
                const uint8_t* obj_mask, float* grad_pred, float* rows_base, int64_t batch,
                int S, int Bx, int C, int img_size, float lambda_coord, float lambda_noobj,
                float inv_global_batch, const yh::LossPipe* pipe, hipStream_t st,
                bool nhwc = false) {
    if (batch > 0 && (!pred || !gt_xywh || !gt_cls || !obj_mask || !grad_pred)) return YH_ERR_NULL;
    if (nhwc && (bias || pipe)) return YH_ERR_ARG;
    if (!aligned(gt_xywh, 16) || !aligned(pred, 4) || !aligned(grad_pred, 4) ||
        !aligned(gt_cls, 4))
        return YH_ERR_ALIGN;
    if (isnan(lambda_coord) || isnan(lambda_noobj) || isnan(inv_global_batch)) return YH_ERR_ARG;

    const int64_t rows = loss_blocks(batch, S);
    if (rows == 0 && !pipe) return YH_OK;
    const int g = loss_imgs_per_block(S);
    const float cellf = (float)(img_size / S);
    const float imgf = (float)img_size;
    if (!aligned(bias, 4)) return YH_ERR_ALIGN;
    const bool fast = !nhwc && S == 7 && Bx == 2 && C == 20 && aligned(gt_cls, 16) && aligned(grad_pred, 16);
    const yh::LossPipe pp = pipe ? *pipe : yh::LossPipe{};
    const dim3 grid((unsigned)(rows + (pipe ? 1 : 0)));
    // run-time shapes: the gradients of a workgroup go through an LDS tile when they fit
    const long long tile_bytes = yh::loss_generic_tile_floats(g, S, Bx, C) * 4;
    const bool tile = !fast && aligned(grad_pred, 16) && tile_bytes <= yh::LOSS_TILE_MAX_BYTES &&
                      g * S * S <= yh::LOSS_TILE_MAX_ITEMS;
    const size_t lds = tile ? (size_t)tile_bytes : 0;
#define YH_LOSS_GENERIC(PIPE, BIAS, TILE)                                                        \
    hipLaunchKernelGGL((yh::loss_fwd_bwd_generic_kernel<PIPE, BIAS, TILE>), grid,                \
                       dim3(yh::LOSS_THREADS), lds, st, pred, bias, gt_xywh, gt_cls, obj_mask,   \
                       grad_pred, rows_base, (long long)batch, S, Bx, C, g, cellf, imgf,         \
                       lambda_coord, lambda_noobj, inv_global_batch, pp);
#define YH_LOSS_LAUNCH(PIPE)                                                                     \
    if (fast && bias)                                                                            \
        hipLaunchKernelGGL((yh::loss_fwd_bwd_kernel<7, 2, 20, PIPE, true>), grid,                \
                           dim3(yh::LOSS_THREADS), 0, st, pred, bias, gt_xywh, gt_cls, obj_mask, \
                           grad_pred, rows_base, (long long)batch, g, cellf, imgf, lambda_coord, \
                           lambda_noobj, inv_global_batch, pp);                                  \
    else if (fast)                                                                               \
        hipLaunchKernelGGL((yh::loss_fwd_bwd_kernel<7, 2, 20, PIPE, false>), grid,               \
                           dim3(yh::LOSS_THREADS), 0, st, pred, bias, gt_xywh, gt_cls, obj_mask, \
                           grad_pred, rows_base, (long long)batch, g, cellf, imgf, lambda_coord, \
                           lambda_noobj, inv_global_batch, pp);                                  \
    else if (bias && tile)                                                                       \
        YH_LOSS_GENERIC(PIPE, true, true)                                                        \
    else if (bias)                                                                               \
        YH_LOSS_GENERIC(PIPE, true, false)                                                       \
    else if (tile)                                                                               \
        YH_LOSS_GENERIC(PIPE, false, true)                                                       \
    else                                                                                         \
        YH_LOSS_GENERIC(PIPE, false, false)
    // channel-major blocks too large for the tile: the rolling-window kernel, when a thread's
    // share of the per-cell confidence gradients and of the object cells' rows fits its registers
    const int D = 5 * Bx + C;
    const int og_cap = yh::LOSS_ROLL_CAP_FLOATS / D;
    const bool roll = !fast && !tile && og_cap >= 4 &&
                      (long long)g * S * S * Bx <= yh::LOSS_ROLL_CAP_FLOATS;
    if (roll) {
        const size_t roll_lds = ((size_t)yh::LOSS_ROLL_FLOATS + (size_t)og_cap * D + (size_t)g * S * S * Bx) * 4;
#define YH_LOSS_ROLL(BIAS, NHWC, PIPE)                                                              \
        hipLaunchKernelGGL((yh::loss_fwd_bwd_roll_kernel<BIAS, NHWC, PIPE>), grid, dim3(yh::LOSS_THREADS), \
                           roll_lds, st, pred, bias, gt_xywh, gt_cls, obj_mask, grad_pred, rows_base,  \
                           (long long)batch, S, Bx, C, g, cellf, imgf, lambda_coord, lambda_noobj,     \
                           inv_global_batch, og_cap, pp);
        if (nhwc) YH_LOSS_ROLL(false, true, false)
        else if (bias && pipe) YH_LOSS_ROLL(true, false, true)
        else if (bias) YH_LOSS_ROLL(true, false, false)
        else if (pipe) YH_LOSS_ROLL(false, false, true)
        else YH_LOSS_ROLL(false, false, false)
#undef YH_LOSS_ROLL
    } else if (nhwc) {
        if (tile)
            hipLaunchKernelGGL((yh::loss_fwd_bwd_generic_kernel<false, false, true, true>), grid,
                               dim3(yh::LOSS_THREADS), lds, st, pred, bias, gt_xywh, gt_cls, obj_mask,
                               grad_pred, rows_base, (long long)batch, S, Bx, C, g, cellf, imgf,
                               lambda_coord, lambda_noobj, inv_global_batch, pp);
        else
            hipLaunchKernelGGL((yh::loss_fwd_bwd_generic_kernel<false, false, false, true>), grid,
                               dim3(yh::LOSS_THREADS), lds, st, pred, bias, gt_xywh, gt_cls, obj_mask,
                               grad_pred, rows_base, (long long)batch, S, Bx, C, g, cellf, imgf,
                               lambda_coord, lambda_noobj, inv_global_batch, pp);
    } else if (pipe) { YH_LOSS_LAUNCH(true) } else { YH_LOSS_LAUNCH(false) }
#undef YH_LOSS_LAUNCH
#undef YH_LOSS_GENERIC
    return launch_status();
}

// the two halves of a pipelined workspace
struct PipeHalves {
    char* half[2];
};

int pipe_halves(void* workspace, size_t workspace_bytes, size_t need_per_half, PipeHalves& h) {
    if (!workspace) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return YH_ERR_ALIGN;
    const size_t stride = (workspace_bytes / 2) & ~(size_t)15;
    if (stride < need_per_half) return YH_ERR_WORKSPACE;
    h.half[0] = (char*)workspace;
    h.half[1] = (char*)workspace + stride;
    return YH_OK;
}

int check_log(const float* loss_terms, const float* log_ring, int log_capacity,
              const int32_t* log_cursor) {
    if (log_ring && (!log_cursor || log_capacity < 1)) return YH_ERR_ARG;
    if (!aligned(loss_terms, 4) || !aligned(log_ring, 4) || !aligned(log_cursor, 4))
        return YH_ERR_ALIGN;
    return YH_OK;
}

}  // namespace

extern "C" {

int yh_loss_partials(const float* pred, const float* gt_xywh, const float* gt_cls,
                     const uint8_t* obj_mask, float* grad_pred, void* workspace,
                     size_t workspace_bytes, int64_t batch, int S, int Bx, int C, int img_size,
                     float lambda_coord, float lambda_noobj, float inv_global_batch,
                     void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (!workspace) return YH_ERR_WORKSPACE;
    if (workspace_bytes < yh_workspace_bytes(YH_OP_LOSS, batch, S, Bx, C)) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return YH_ERR_ALIGN;
    return loss_launch(pred, nullptr, gt_xywh, gt_cls, obj_mask, grad_pred, (float*)workspace,
                       batch, S, Bx, C, img_size, lambda_coord, lambda_noobj, inv_global_batch,
                       nullptr, (hipStream_t)stream);
}

int yh_loss_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < 64) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return YH_ERR_ALIGN;
    return (int)hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream);
}

int yh_loss_fwd_bwd_pipelined(const float* pred, const float* gt_xywh, const float* gt_cls,
                              const uint8_t* obj_mask, float* grad_pred, float* loss_terms_prev,
                              float* log_ring, int log_capacity, int32_t* log_cursor,
                              void* workspace, size_t workspace_bytes, int phase, int64_t batch,
                              int S, int Bx, int C, int img_size, float lambda_coord,
                              float lambda_noobj, float inv_global_batch, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (phase != 0 && phase != 1) return YH_ERR_ARG;
    if (const int rc = check_log(loss_terms_prev, log_ring, log_capacity, log_cursor)) return rc;
    PipeHalves h;
    const size_t need = yh_workspace_bytes(YH_OP_LOSS_PIPELINED, batch, S, Bx, C) / 2;
    if (const int rc = pipe_halves(workspace, workspace_bytes, need, h)) return rc;
    char* own = h.half[phase];
    char* prev = h.half[1 - phase];
    const yh::LossPipe pipe{(int*)own, (int*)prev, (const float*)(prev + 16), loss_terms_prev,
                            log_ring, (int*)log_cursor, log_capacity};
    return loss_launch(pred, nullptr, gt_xywh, gt_cls, obj_mask, grad_pred, (float*)(own + 16),
                       batch, S, Bx, C, img_size, lambda_coord, lambda_noobj, inv_global_batch,
                       &pipe, (hipStream_t)stream);
}

int yh_loss_pipelined_flush(void* workspace, size_t workspace_bytes, int phase, float* loss_terms,
                            float* log_ring, int log_capacity, int32_t* log_cursor,
                            void* stream) {
    if (phase != 0 && phase != 1) return YH_ERR_ARG;
    if (const int rc = check_log(loss_terms, log_ring, log_capacity, log_cursor)) return rc;
    PipeHalves h;
    if (const int rc = pipe_halves(workspace, workspace_bytes, 32, h)) return rc;
    yh::LossPipe pipe[2];
    for (int i = 0; i < 2; ++i) {
        char* half = h.half[i == 0 ? phase : 1 - phase];
        pipe[i] = yh::LossPipe{nullptr, (int*)half, (const float*)(half + 16), loss_terms,
                               log_ring, (int*)log_cursor, log_capacity};
    }
    hipLaunchKernelGGL(yh::loss_pipe_flush_kernel, dim3(1), dim3(yh::LOSS_THREADS), 0,
                       (hipStream_t)stream, pipe[0], pipe[1]);
    return launch_status();
}

int yh_loss_finalize(const void* workspace, size_t workspace_bytes, int64_t batch, int S,
                     float inv_global_batch, float* loss_terms, float* log_ring,
                     int log_capacity, int32_t* log_cursor, void* stream) {
    if (batch < 0 || S < 1 || S > YH_MAX_S) return YH_ERR_SHAPE;
    if (!loss_terms) return YH_ERR_NULL;
    if (!workspace || workspace_bytes < yh_workspace_bytes(YH_OP_LOSS, batch, S, 1, 1))
        return YH_ERR_WORKSPACE;
    if (log_ring && (!log_cursor || log_capacity < 1)) return YH_ERR_ARG;
    if (!aligned(workspace, 16) || !aligned(loss_terms, 4) || !aligned(log_ring, 4) ||
        !aligned(log_cursor, 4))
        return YH_ERR_ALIGN;
    if (isnan(inv_global_batch)) return YH_ERR_ARG;
    hipLaunchKernelGGL(yh::loss_finalize_kernel, dim3(1), dim3(yh::LOSS_THREADS), 0,
                       (hipStream_t)stream,
                       (const float*)workspace, (int)loss_blocks(batch, S), inv_global_batch,
                       loss_terms, log_ring, log_capacity, (int*)log_cursor);
    return launch_status();
}

// Device address of `scratch` on the current device, resolved once per device
// (with that device current) and kept; nullptr with *err set on failure.
static float* static_scratch_ptr(int* err) {
    constexpr int kMaxDevices = 64;
    static std::atomic<float*> ptrs[kMaxDevices];
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { *err = (int)e; return nullptr; }
    if (dev < 0 || dev >= kMaxDevices) { *err = YH_ERR_SCRATCH; return nullptr; }
    float* p = ptrs[dev].load(std::memory_order_acquire);
    if (!p) {
        // two threads may both get here: they resolve the same address
        void* sym = nullptr;
        e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sym) { *err = e != hipSuccess ? (int)e : YH_ERR_SCRATCH; return nullptr; }
        p = (float*)sym;
        ptrs[dev].store(p, std::memory_order_release);
    }
    return p;
}

int yh_loss_fwd_bwd_static(const float* pred, const float* gt_xywh, const float* gt_cls,
                           const uint8_t* obj_mask, float* grad_pred, float* loss_terms,
                           int64_t batch, int S, int Bx, int C, int img_size, float lambda_coord,
                           float lambda_noobj, float inv_global_batch, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (!loss_terms) return YH_ERR_NULL;
    // one row of 16 bytes per workgroup must fit the static array
    if (yh_workspace_bytes(YH_OP_LOSS, batch, S, Bx, C) > (size_t)YH_STATIC_SCRATCH_BYTES)
        return YH_ERR_SCRATCH;
    int err = YH_OK;
    float* rows = static_scratch_ptr(&err);
    if (!rows) return err;
    // every row the finalize kernel reads is written by this call's first kernel:
    // nothing in the array has to start at zero
    if (const int rc = loss_launch(pred, nullptr, gt_xywh, gt_cls, obj_mask, grad_pred, rows, batch,
                                   S, Bx, C, img_size, lambda_coord, lambda_noobj,
                                   inv_global_batch, nullptr, (hipStream_t)stream))
        return rc;
    return yh_loss_finalize(rows, YH_STATIC_SCRATCH_BYTES, batch, S, inv_global_batch, loss_terms,
                            nullptr, 0, nullptr, stream);
}

int yh_loss_fwd_bwd(const float* pred, const float* gt_xywh, const float* gt_cls,
                    const uint8_t* obj_mask, float* grad_pred, float* loss_terms,
                    void* workspace, size_t workspace_bytes, int64_t batch, int S, int Bx,
                    int C, int img_size, float lambda_coord, float lambda_noobj,
                    float inv_global_batch, void* stream) {
    if (!loss_terms) {
        const int rc = check_shape(batch, S, Bx, C, img_size);
        return rc ? rc : YH_ERR_NULL;
    }
    if (const int rc = yh_loss_partials(pred, gt_xywh, gt_cls, obj_mask, grad_pred, workspace,
                                        workspace_bytes, batch, S, Bx, C, img_size, lambda_coord,
                                        lambda_noobj, inv_global_batch, stream))
        return rc;
    return yh_loss_finalize(workspace, workspace_bytes, batch, S, inv_global_batch, loss_terms,
                            nullptr, 0, nullptr, stream);
}

int yh_loss_fwd_bwd_nhwc(const float* pred, const float* gt_xywh, const float* gt_cls,
                         const uint8_t* obj_mask, float* grad_pred, float* loss_terms,
                         void* workspace, size_t workspace_bytes, int64_t batch, int S, int Bx,
                         int C, int img_size, float lambda_coord, float lambda_noobj,
                         float inv_global_batch, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (!loss_terms) return YH_ERR_NULL;
    if (!workspace) return YH_ERR_WORKSPACE;
    if (workspace_bytes < yh_workspace_bytes(YH_OP_LOSS, batch, S, Bx, C)) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return YH_ERR_ALIGN;
    if (const int rc = loss_launch(pred, nullptr, gt_xywh, gt_cls, obj_mask, grad_pred,
                                   (float*)workspace, batch, S, Bx, C, img_size, lambda_coord,
                                   lambda_noobj, inv_global_batch, nullptr, (hipStream_t)stream, true))
        return rc;
    return yh_loss_finalize(workspace, workspace_bytes, batch, S, inv_global_batch, loss_terms,
                            nullptr, 0, nullptr, stream);
}

size_t yh_bias_grad_workspace_bytes(int64_t nrows, int ncols) {
    if (nrows < 0 || ncols < 1) return 0;
    return (size_t)colsum_slices(nrows) * (size_t)ncols * sizeof(float);
}

int yh_bias_grad(const float* grad, float* grad_bias, void* workspace, size_t workspace_bytes,
                 int64_t nrows, int ncols, void* stream) {
    if (nrows < 0 || ncols < 1) return YH_ERR_SHAPE;
    if (!grad_bias || (nrows > 0 && !grad)) return YH_ERR_NULL;
    if (!workspace || workspace_bytes < yh_bias_grad_workspace_bytes(nrows, ncols))
        return YH_ERR_WORKSPACE;
    if (!aligned(grad, 4) || !aligned(grad_bias, 4) || !aligned(workspace, 4)) return YH_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int slices = colsum_slices(nrows);
    const int per = nrows == 0 ? 1 : (int)((nrows + slices - 1) / slices);
    const unsigned gx = (unsigned)((ncols + 63) / 64);
    // pass 1: `slices` partial rows; pass 2: their sum (also writes zeros when nrows == 0)
    if (nrows > 0) {
        hipLaunchKernelGGL(yh::column_sum_kernel, dim3(gx, (unsigned)slices), dim3(256), 0, st, grad,
                           (float*)workspace, (long long)nrows, ncols, per);
        if (const int rc = launch_status()) return rc;
    }
    hipLaunchKernelGGL(yh::column_sum_kernel, dim3(gx, 1), dim3(256), 0, st,
                       (const float*)workspace, grad_bias, (long long)(nrows > 0 ? slices : 0),
                       ncols, slices);
    return launch_status();
}

int yh_loss_fwd_bwd_linear(const float* z, const float* bias, const float* gt_xywh,
                           const float* gt_cls, const uint8_t* obj_mask, float* grad_z,
                           float* loss_terms, void* workspace, size_t workspace_bytes,
                           int64_t batch, int S, int Bx, int C, int img_size, float lambda_coord,
                           float lambda_noobj, float inv_global_batch, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (!loss_terms || !bias) return YH_ERR_NULL;
    if (!workspace) return YH_ERR_WORKSPACE;
    if (workspace_bytes < yh_workspace_bytes(YH_OP_LOSS, batch, S, Bx, C)) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return YH_ERR_ALIGN;
    if (const int rc = loss_launch(z, bias, gt_xywh, gt_cls, obj_mask, grad_z, (float*)workspace,
                                   batch, S, Bx, C, img_size, lambda_coord, lambda_noobj,
                                   inv_global_batch, nullptr, (hipStream_t)stream))
        return rc;
    return yh_loss_finalize(workspace, workspace_bytes, batch, S, inv_global_batch, loss_terms,
                            nullptr, 0, nullptr, stream);
}

int yh_scale_grad(float* grad, const float* scale, int64_t n, void* stream) {
    if (n < 0) return YH_ERR_SHAPE;
    if (n == 0) return YH_OK;
    if (!grad || !scale) return YH_ERR_NULL;
    if (!aligned(grad, 16) || !aligned(scale, 4)) return YH_ERR_ALIGN;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(yh::scale_grad_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, grad, scale, (long long)n);
    return launch_status();
}

int yh_pairwise_boxes(const float* ltrb1, const float* ltrb2, float* out, int64_t N, int64_t M,
                      int kind, void* stream) {
    if (N < 0 || M < 0 || N > ((int64_t)1 << 31) || M > ((int64_t)1 << 31)) return YH_ERR_SHAPE;
    if (kind != YH_BOX_IOU && kind != YH_BOX_GIOU && kind != YH_BOX_INTERSECTION &&
        kind != YH_BOX_ENCLOSING)
        return YH_ERR_ARG;
    if (N == 0 || M == 0) return YH_OK;
    if (!ltrb1 || !ltrb2 || !out) return YH_ERR_NULL;
    if (!aligned(ltrb1, 16) || !aligned(ltrb2, 16) || !aligned(out, 4)) return YH_ERR_ALIGN;
    int64_t bx = (M + 255) / 256;
    if (bx > 64) bx = 64;
    int64_t by = 4096 / bx;   // about 4096 workgroups in all; each loops over its share of the rows
    if (by > N) by = N;
    const dim3 grid((unsigned)bx, (unsigned)by);
    hipStream_t st = (hipStream_t)stream;
    if (kind == YH_BOX_IOU)
        hipLaunchKernelGGL(yh::pairwise_boxes_kernel<yh::BOX_IOU>, grid, dim3(256), 0, st, ltrb1,
                           ltrb2, out, (long long)N, (long long)M);
    else if (kind == YH_BOX_GIOU)
        hipLaunchKernelGGL(yh::pairwise_boxes_kernel<yh::BOX_GIOU>, grid, dim3(256), 0, st, ltrb1,
                           ltrb2, out, (long long)N, (long long)M);
    else if (kind == YH_BOX_ENCLOSING)
        hipLaunchKernelGGL(yh::pairwise_boxes_kernel<yh::BOX_ENCLOSING>, grid, dim3(256), 0, st,
                           ltrb1, ltrb2, out, (long long)N, (long long)M);
    else
        hipLaunchKernelGGL(yh::pairwise_boxes_kernel<yh::BOX_INTERSECTION>, grid, dim3(256), 0, st,
                           ltrb1, ltrb2, out, (long long)N, (long long)M);
    return launch_status();
}

int yh_box_area(const float* ltrb, float* out, int64_t n, void* stream) {
    if (n < 0) return YH_ERR_SHAPE;
    if (n == 0) return YH_OK;
    if (!ltrb || !out) return YH_ERR_NULL;
    if (!aligned(ltrb, 16) || !aligned(out, 4)) return YH_ERR_ALIGN;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(yh::box_area_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, ltrb, out, (long long)n);
    return launch_status();
}

int yh_encode_targets(const float* gt_ltrb, const int32_t* gt_label, const int32_t* gt_count,
                      float* gt_xywh, float* gt_cls, uint8_t* obj_mask, int64_t batch, int M,
                      int S, int C, int img_size, void* stream) {
    if (const int rc = check_shape(batch, S, 1, C, img_size)) return rc;
    if (M < 0 || M > 65536) return YH_ERR_SHAPE;
    if (batch == 0) return YH_OK;
    if (!gt_count || !gt_xywh || !gt_cls || !obj_mask || (M > 0 && (!gt_ltrb || !gt_label)))
        return YH_ERR_NULL;
    if (!aligned(gt_ltrb, 16) || !aligned(gt_xywh, 16) || !aligned(gt_label, 4) ||
        !aligned(gt_count, 4) || !aligned(gt_cls, 4))
        return YH_ERR_ALIGN;
    const int vec4 = (C % 4 == 0 && aligned(gt_cls, 16)) ? 1 : 0;
    const int g = loss_imgs_per_block(S);
    const size_t lds = (size_t)g * M * 24 + (size_t)g * S * S * 4;
    // the tiled kernel keeps the boxes of its g images in LDS: small grids with many
    // boxes per image (S = 1: g = 256) take the general kernel instead
    if (M >= 1 && M <= 64 && lds <= 48 * 1024) {
        hipLaunchKernelGGL(yh::encode_targets_tiled_kernel, dim3((unsigned)((batch + g - 1) / g)),
                           dim3(256), lds, (hipStream_t)stream, gt_ltrb, gt_label, gt_count,
                           gt_xywh, gt_cls, obj_mask, (long long)batch, M, S, C,
                           (float)(img_size / S), (float)img_size, g, vec4);
        return launch_status();
    }
    const int64_t total = batch * S * S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(yh::encode_targets_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, gt_ltrb, gt_label, gt_count, gt_xywh, gt_cls, obj_mask,
                       (long long)batch, M, S, C, (float)(img_size / S), (float)img_size, vec4);
    return launch_status();
}

int yh_decode_grid(const float* pred, float* ltrb, float* conf, float* cls_prob,
                   int64_t* cls_idx, int64_t* best_k, float* best_ltrb, uint8_t* conf_mask,
                   float conf_thresh, int64_t batch, int S, int Bx, int C, int img_size,
                   void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (batch == 0) return YH_OK;
    if (!pred) return YH_ERR_NULL;
    if (!aligned(pred, 4) || !aligned(ltrb, 16) || !aligned(best_ltrb, 16) ||
        !aligned(cls_idx, 8) || !aligned(best_k, 8) || !aligned(conf, 4) || !aligned(cls_prob, 4))
        return YH_ERR_ALIGN;
    if (conf_mask && isnan(conf_thresh)) return YH_ERR_ARG;
    const int64_t total = batch * S * S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(yh::decode_grid_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, pred, ltrb, conf, cls_prob, (long long*)cls_idx,
                       (long long*)best_k, best_ltrb, conf_mask, conf_thresh, (long long)batch,
                       S, Bx, C, (float)(img_size / S), (float)img_size);
    return launch_status();
}

int yh_norm_xywh(const float* pred, float* norm_xywh, int64_t batch, int S, int Bx, int C,
                 void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, S)) return rc;
    if (batch == 0) return YH_OK;
    if (!pred || !norm_xywh) return YH_ERR_NULL;
    if (!aligned(pred, 4) || !aligned(norm_xywh, 16)) return YH_ERR_ALIGN;
    hipLaunchKernelGGL(yh::norm_xywh_kernel, dim3(grid_for(batch * S * S)), dim3(256), 0,
                       (hipStream_t)stream, pred, norm_xywh, (long long)batch, S, Bx, C);
    return launch_status();
}

int yh_denormalize_xywh(const float* norm_xywh, float* xywh, int64_t batch, int S, int Bx,
                        int img_size, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, 1, img_size)) return rc;
    if (batch == 0) return YH_OK;
    if (!norm_xywh || !xywh) return YH_ERR_NULL;
    if (!aligned(norm_xywh, 16) || !aligned(xywh, 16)) return YH_ERR_ALIGN;
    const int64_t n = batch * S * S * Bx;
    hipLaunchKernelGGL(yh::denormalize_xywh_kernel, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, norm_xywh, xywh, (long long)n, S, Bx,
                       (float)(img_size / S), (float)img_size);
    return launch_status();
}

int yh_xywh_to_ltrb(const float* xywh, float* ltrb, int64_t n, int img_size, void* stream) {
    if (n < 0 || img_size < 1) return YH_ERR_SHAPE;
    if (n == 0) return YH_OK;
    if (!xywh || !ltrb) return YH_ERR_NULL;
    if (!aligned(xywh, 16) || !aligned(ltrb, 16)) return YH_ERR_ALIGN;
    hipLaunchKernelGGL(yh::xywh_to_ltrb_kernel, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, xywh, ltrb, (long long)n, (float)img_size);
    return launch_status();
}

int yh_responsibility_mask(const float* pred, const float* gt_xywh, const uint8_t* obj_mask,
                           uint8_t* resp_mask, int64_t batch, int S, int Bx, int C, int img_size,
                           void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (batch == 0) return YH_OK;
    if (!pred || !gt_xywh || !obj_mask || !resp_mask) return YH_ERR_NULL;
    if (!aligned(pred, 4) || !aligned(gt_xywh, 16)) return YH_ERR_ALIGN;
    hipLaunchKernelGGL(yh::responsibility_mask_kernel, dim3(grid_for(batch * S * S)), dim3(256), 0,
                       (hipStream_t)stream, pred, gt_xywh, obj_mask, resp_mask, (long long)batch,
                       S, Bx, C, (float)(img_size / S), (float)img_size);
    return launch_status();
}

int yh_decode_candidates(const float* pred, float* cand_ltrb, float* cand_score,
                         int32_t* cand_cls, uint8_t* cand_present, int mode, int64_t batch,
                         int S, int Bx, int C, int img_size, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (mode != YH_MODE_REFERENCE && mode != YH_MODE_DENSE) return YH_ERR_ARG;
    if (batch == 0) return YH_OK;
    if (!pred || !cand_ltrb || !cand_score || !cand_cls || !cand_present) return YH_ERR_NULL;
    if (!aligned(pred, 4) || !aligned(cand_ltrb, 16) || !aligned(cand_score, 4) ||
        !aligned(cand_cls, 4))
        return YH_ERR_ALIGN;
    const int64_t total = batch * S * S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(yh::decode_candidates_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, pred, cand_ltrb, cand_score, cand_cls, cand_present,
                       (long long)batch, S, Bx, C, mode, (float)(img_size / S), (float)img_size);
    return launch_status();
}

}  // extern "C"

namespace {

int decode_nms_launch(const float* pred, const float* bias, float* dets, int32_t* keep_count,
                      int32_t* keep_idx, int mode, float conf_thresh, float iou_thresh,
                      int64_t batch, int S, int Bx, int C, int img_size, bool with_bias,
                      void* stream, bool nhwc = false, int nout = 0) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (mode != YH_MODE_REFERENCE && mode != YH_MODE_DENSE) return YH_ERR_ARG;
    if (isnan(conf_thresh) || isnan(iou_thresh)) return YH_ERR_ARG;
    if ((int64_t)S * S * Bx > YH_MAX_CANDIDATES) return YH_ERR_UNSUPPORTED;
    if (with_bias && !bias) return YH_ERR_NULL;
    if (batch == 0) return YH_OK;
    if (!pred || !dets || !keep_count) return YH_ERR_NULL;
    if (!aligned(pred, 4) || !aligned(bias, 4) || !aligned(dets, 8) || !aligned(keep_count, 4) ||
        !aligned(keep_idx, 4))
        return YH_ERR_ALIGN;
    yh::NmsArgs a = {};
    a.pred = pred;
    a.bias = bias;
    a.dets = dets;
    a.keep_count = keep_count;
    a.keep_idx = keep_idx;
    a.batch = batch;
    a.S = S; a.nb = Bx; a.nc = C; a.mode = mode;
    a.cell_major = nhwc ? 1 : 0;
    a.nslot = mode == YH_MODE_REFERENCE ? S * S : S * S * Bx;
    // rows of dets per image: S*S*Bx unless the caller sized the tensor itself (at least one
    // row per candidate of the mode)
    if (nout != 0 && (nout < a.nslot || nout > YH_MAX_CANDIDATES)) return YH_ERR_ARG;
    a.nout = nout != 0 ? nout : S * S * Bx;
    a.cellf = (float)(img_size / S);
    a.imgf = (float)img_size;
    a.conf_thr = conf_thresh;
    a.iou_thr = iou_thresh;
#ifdef YH_PHASE_PROBE
    { const char* e = getenv("YH_NMS_STOP"); a.dbg_stop = e ? atoi(e) : 0; }
    { const char* e = getenv("YH_NMS_STAMPS"); a.stamps = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    return with_bias ? launch_nms<2>(a, (hipStream_t)stream) : launch_nms<0>(a, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int yh_decode_nms(const float* pred, float* dets, int32_t* keep_count, int32_t* keep_idx,
                  int mode, float conf_thresh, float iou_thresh, int64_t batch, int S, int Bx,
                  int C, int img_size, void* stream) {
    return decode_nms_launch(pred, nullptr, dets, keep_count, keep_idx, mode, conf_thresh,
                             iou_thresh, batch, S, Bx, C, img_size, false, stream);
}

int yh_decode_nms_rows(const float* pred, float* dets, int32_t* keep_count, int32_t* keep_idx,
                       int mode, float conf_thresh, float iou_thresh, int rows, int64_t batch, int S,
                       int Bx, int C, int img_size, void* stream) {
    if (rows < 1) return YH_ERR_ARG;
    return decode_nms_launch(pred, nullptr, dets, keep_count, keep_idx, mode, conf_thresh,
                             iou_thresh, batch, S, Bx, C, img_size, false, stream, false, rows);
}

int yh_decode_nms_nhwc(const float* pred, float* dets, int32_t* keep_count, int32_t* keep_idx,
                       int mode, float conf_thresh, float iou_thresh, int64_t batch, int S, int Bx,
                       int C, int img_size, void* stream) {
    return decode_nms_launch(pred, nullptr, dets, keep_count, keep_idx, mode, conf_thresh,
                             iou_thresh, batch, S, Bx, C, img_size, false, stream, true);
}

int yh_decode_nms_linear(const float* z, const float* bias, float* dets, int32_t* keep_count,
                         int32_t* keep_idx, int mode, float conf_thresh, float iou_thresh,
                         int64_t batch, int S, int Bx, int C, int img_size, void* stream) {
    return decode_nms_launch(z, bias, dets, keep_count, keep_idx, mode, conf_thresh, iou_thresh,
                             batch, S, Bx, C, img_size, true, stream);
}

int yh_nms(const float* cand_ltrb, const float* cand_score, const int32_t* cand_cls,
           const uint8_t* cand_present, float* dets, int32_t* keep_count, int32_t* keep_idx,
           float conf_thresh, float iou_thresh, int64_t batch, int N, void* stream) {
    if (batch < 0 || N < 1) return YH_ERR_SHAPE;
    if (N > YH_MAX_CANDIDATES) return YH_ERR_UNSUPPORTED;
    if (isnan(conf_thresh) || isnan(iou_thresh)) return YH_ERR_ARG;
    if (batch == 0) return YH_OK;
    if (!cand_ltrb || !cand_score || !cand_cls || !dets || !keep_count) return YH_ERR_NULL;
    if (!aligned(cand_ltrb, 16) || !aligned(cand_score, 4) || !aligned(cand_cls, 4) ||
        !aligned(dets, 8) || !aligned(keep_count, 4) || !aligned(keep_idx, 4))
        return YH_ERR_ALIGN;
    yh::NmsArgs a = {};
    a.c_ltrb = cand_ltrb;
    a.c_score = cand_score;
    a.c_cls = cand_cls;
    a.c_present = cand_present;
    a.dets = dets;
    a.keep_count = keep_count;
    a.keep_idx = keep_idx;
    a.batch = batch;
    a.S = 1; a.nb = 1; a.nc = 1; a.mode = YH_MODE_DENSE;
    a.nslot = N;
    a.nout = N;
    a.conf_thr = conf_thresh;
    a.iou_thr = iou_thresh;
    return launch_nms<1>(a, (hipStream_t)stream);
}

}  // extern "C"

namespace {

// Both halves of a step on one stream.  The tuned shape in reference mode takes ONE kernel
// that reads the head tensor once (head_step_kernel); every other case is the loss launch
// followed by the decode + NMS launch -- the same bits either way.  `pipe` as in loss_launch.
int head_step_launch(const float* pred, const float* gt_xywh, const float* gt_cls,
                     const uint8_t* obj_mask, float* grad_pred, float* dets, int32_t* keep_count,
                     int32_t* keep_idx, float* rows_base, const yh::LossPipe* pipe, int mode,
                     float conf_thresh, float iou_thresh, int rows, int64_t batch, int S, int Bx, int C,
                     int img_size, float lambda_coord, float lambda_noobj, float inv_global_batch,
                     hipStream_t st) {
    if (mode != YH_MODE_REFERENCE && mode != YH_MODE_DENSE) return YH_ERR_ARG;
    if (isnan(conf_thresh) || isnan(iou_thresh) || rows < 0) return YH_ERR_ARG;
    if ((int64_t)S * S * Bx > YH_MAX_CANDIDATES) return YH_ERR_UNSUPPORTED;
    const int nslot = mode == YH_MODE_REFERENCE ? S * S : S * S * Bx;
    if (rows != 0 && (rows < nslot || rows > YH_MAX_CANDIDATES)) return YH_ERR_ARG;
    if (batch > 0 && (!pred || !gt_xywh || !gt_cls || !obj_mask || !grad_pred || !dets || !keep_count))
        return YH_ERR_NULL;
    if (!aligned(gt_xywh, 16) || !aligned(pred, 4) || !aligned(grad_pred, 4) || !aligned(gt_cls, 4) ||
        !aligned(dets, 8) || !aligned(keep_count, 4) || !aligned(keep_idx, 4))
        return YH_ERR_ALIGN;
    if (isnan(lambda_coord) || isnan(lambda_noobj) || isnan(inv_global_batch)) return YH_ERR_ARG;
    const bool one_kernel = S == 7 && Bx == 2 && C == 20 && mode == YH_MODE_REFERENCE &&
                            aligned(gt_cls, 16) && aligned(grad_pred, 16);
    if (!one_kernel) {
        if (const int rc = loss_launch(pred, nullptr, gt_xywh, gt_cls, obj_mask, grad_pred, rows_base,
                                       batch, S, Bx, C, img_size, lambda_coord, lambda_noobj,
                                       inv_global_batch, pipe, st))
            return rc;
        return decode_nms_launch(pred, nullptr, dets, keep_count, keep_idx, mode, conf_thresh,
                                 iou_thresh, batch, S, Bx, C, img_size, false, st, false, rows);
    }
    const int64_t nrows = loss_blocks(batch, S);
    if (nrows == 0 && !pipe) return YH_OK;
    yh::StepArgs a = {};
    a.pred = pred; a.gt_xywh = gt_xywh; a.gt_cls = gt_cls; a.obj_mask = obj_mask;
    a.grad = grad_pred; a.partials = rows_base;
    a.dets = dets; a.keep_count = keep_count; a.keep_idx = keep_idx;
    a.batch = batch;
    a.nout = rows != 0 ? rows : S * S * Bx;
    a.cellf = (float)(img_size / S); a.imgf = (float)img_size;
    a.lc = lambda_coord; a.ln = lambda_noobj; a.inv_gb = inv_global_batch;
    a.conf_thr = conf_thresh; a.iou_thr = iou_thresh;
    if (pipe) a.pipe = *pipe;
    const dim3 grid((unsigned)(nrows + (pipe ? 1 : 0)));
    if (pipe)
        hipLaunchKernelGGL((yh::head_step_kernel<7, 2, 20, true>), grid, dim3(yh::LOSS_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((yh::head_step_kernel<7, 2, 20, false>), grid, dim3(yh::LOSS_THREADS), 0, st, a);
    return launch_status();
}

}  // namespace

extern "C" {

int yh_head_step(const float* pred, const float* gt_xywh, const float* gt_cls,
                 const uint8_t* obj_mask, float* grad_pred, float* loss_terms, float* dets,
                 int32_t* keep_count, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                 int mode, float conf_thresh, float iou_thresh, int rows, int64_t batch, int S, int Bx,
                 int C, int img_size, float lambda_coord, float lambda_noobj, float inv_global_batch,
                 void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (!loss_terms) return YH_ERR_NULL;
    if (!workspace) return YH_ERR_WORKSPACE;
    if (workspace_bytes < yh_workspace_bytes(YH_OP_LOSS, batch, S, Bx, C)) return YH_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return YH_ERR_ALIGN;
    if (const int rc = head_step_launch(pred, gt_xywh, gt_cls, obj_mask, grad_pred, dets, keep_count,
                                        keep_idx, (float*)workspace, nullptr, mode, conf_thresh,
                                        iou_thresh, rows, batch, S, Bx, C, img_size, lambda_coord,
                                        lambda_noobj, inv_global_batch, (hipStream_t)stream))
        return rc;
    return yh_loss_finalize(workspace, workspace_bytes, batch, S, inv_global_batch, loss_terms,
                            nullptr, 0, nullptr, stream);
}

int yh_head_step_pipelined(const float* pred, const float* gt_xywh, const float* gt_cls,
                           const uint8_t* obj_mask, float* grad_pred, float* dets,
                           int32_t* keep_count, int32_t* keep_idx, float* loss_terms_prev,
                           float* log_ring, int log_capacity, int32_t* log_cursor, void* workspace,
                           size_t workspace_bytes, int phase, int mode, float conf_thresh,
                           float iou_thresh, int rows, int64_t batch, int S, int Bx, int C,
                           int img_size, float lambda_coord, float lambda_noobj,
                           float inv_global_batch, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (phase != 0 && phase != 1) return YH_ERR_ARG;
    if (const int rc = check_log(loss_terms_prev, log_ring, log_capacity, log_cursor)) return rc;
    PipeHalves h;
    const size_t need = yh_workspace_bytes(YH_OP_LOSS_PIPELINED, batch, S, Bx, C) / 2;
    if (const int rc = pipe_halves(workspace, workspace_bytes, need, h)) return rc;
    char* own = h.half[phase];
    char* prev = h.half[1 - phase];
    const yh::LossPipe pipe{(int*)own, (int*)prev, (const float*)(prev + 16), loss_terms_prev,
                            log_ring, (int*)log_cursor, log_capacity};
    return head_step_launch(pred, gt_xywh, gt_cls, obj_mask, grad_pred, dets, keep_count, keep_idx,
                            (float*)(own + 16), &pipe, mode, conf_thresh, iou_thresh, rows, batch, S, Bx,
                            C, img_size, lambda_coord, lambda_noobj, inv_global_batch,
                            (hipStream_t)stream);
}

int yh_loss_per_image(const float* pred, const float* gt_xywh, const float* gt_cls,
                      const uint8_t* obj_mask, float* per_image, int64_t batch, int S, int Bx, int C,
                      int img_size, float lambda_coord, float lambda_noobj, void* stream) {
    if (const int rc = check_shape(batch, S, Bx, C, img_size)) return rc;
    if (batch == 0) return YH_OK;
    if (!pred || !gt_xywh || !gt_cls || !obj_mask || !per_image) return YH_ERR_NULL;
    if (!aligned(gt_xywh, 16) || !aligned(pred, 4) || !aligned(gt_cls, 4) || !aligned(per_image, 4))
        return YH_ERR_ALIGN;
    if (isnan(lambda_coord) || isnan(lambda_noobj)) return YH_ERR_ARG;
    const int ipb = S * S <= 64 ? 4 : 1;
    hipLaunchKernelGGL(yh::loss_per_image_kernel, dim3((unsigned)((batch + ipb - 1) / ipb)), dim3(256), 0,
                       (hipStream_t)stream, pred, gt_xywh, gt_cls, obj_mask, per_image, (long long)batch, S,
                       Bx, C, ipb, (float)(img_size / S), (float)img_size, lambda_coord, lambda_noobj);
    return launch_status();
}

}  // extern "C"
