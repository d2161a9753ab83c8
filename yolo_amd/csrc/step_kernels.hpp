// One launch for both halves of a step on the head tensor: the loss forward + backward
// (model.py:296-317, the gradient of loss.backward(), train.py:89-93) and decode +
// confidence threshold + per-class NMS in reference mode (model.py:379-394): what an
// evaluation pass wants from one forward output -- the validation loss of train.py:100-110
// and the detections that go to eval.get_map (eval.py:145).  The two
// kernels it replaces (loss_fwd_bwd_kernel, nms_kernel<64, 1, ...>) each read the head
// tensor -- the loss kernel in effect all of it: the confidence channels of every cell and
// every channel of the cells with an object touch nearly every 128-byte line -- so one
// kernel that reads it ONCE moves a quarter less.
//
// Geometry: both kernels already work on four images per 256-thread workgroup.  Here
// wavefront w of a workgroup owns image w of its block and lane c < S*S owns cell c:
//   1. the lane requests the cell's 5 Bx + C logits and its object-mask byte (one round trip);
//      a lane whose cell holds an object then requests the cell's ground truth;
//   2. loss pass 1 as in loss_block: a cell without an object is finished (confidence
//      gradients and zeros into the LDS gradient tile), a cell with one parks its logits
//      in the tile slots its gradients will take and is listed;
//   3. the wavefront runs the NMS of its image from the logits in registers
//      (cell_candidate_from_registers + nms_wave_one_per_lane) and stores the detection
//      rows -- this is what the ground truth's trip to memory hides behind;
//   4. the ground truth, arrived by now, is parked in LDS (in the wavefront's NMS tables,
//      which are dead), one barrier, and loss pass 2 / the row sums / the tile store run as
//      in loss_block (loss_pass2, loss_block_reduce, loss_tile_store).
// Every value is computed by the code the two kernels use, per cell in the same order, and
// the partial sums meet in the same lanes in the same order: gradients, loss rows,
// detection rows, counts and indices are the bits of the two separate launches.
#pragma once
#include "loss_kernels.hpp"
#include "nms_kernels.hpp"

namespace yh {

struct StepArgs {
    const float* pred;
    const float* gt_xywh;
    const float* gt_cls;
    const uint8_t* obj_mask;
    float* grad;
    float* partials;
    float* dets;
    int32_t* keep_count;
    int32_t* keep_idx;      // may be null
    long long batch;
    int nout;               // rows of dets per image
    float cellf, imgf, lc, ln, inv_gb, conf_thr, iou_thr;
    LossPipe pipe;
};

// per wavefront: the NMS tables of nms_wave_one_per_lane for up to 64 classes --
// class masks 512 B, kill and precedence words 1024 B, boxes 1024 B, scores 256 B, class
// row offsets 256 B -- and, once they are dead, the parked ground truth of the
// wavefront's listed cells (STEP_GT_SLOTS slots of 4 + C floats; a cell beyond them is
// fetched from memory in pass 2)
constexpr int STEP_WAVE_LDS_BYTES = 3072;

template <int C>
constexpr int step_gt_slots() {
    return STEP_WAVE_LDS_BYTES / ((4 + C) * 4);
}

template <int S, int BX, int C, bool PIPE>
__global__ __launch_bounds__(LOSS_THREADS, 4) void head_step_kernel(const StepArgs a) {
    constexpr int D = 5 * BX + C;
    constexpr int S2 = S * S;
    constexpr int IPB = loss_ipb_for(S2);
    constexpr int NW = LOSS_THREADS / 64;
    static_assert(IPB == NW && S2 <= 64, "a wavefront per image, a lane per cell");
    static_assert(C <= 64 && C % 4 == 0, "class table of 64 words; class rows read as float4");
    constexpr int GT_FLOATS = 4 + C;                 // a parked ground truth: box, class row
    constexpr int GT_SLOTS = step_gt_slots<C>();
    __shared__ __attribute__((aligned(16))) float sgrad[loss_tile_floats<S, BX, C>()];
    __shared__ __attribute__((aligned(16))) unsigned char wlds[NW][STEP_WAVE_LDS_BYTES];
    __shared__ int olist[LOSS_THREADS];
    __shared__ int wcount[NW];
    __shared__ float sconf[LOSS_THREADS];            // pass 1's confidence term per item

    int bid = blockIdx.x;
    if constexpr (PIPE) {
        // workgroup 0 logs the previous call, the others are this call's rows 0, 1, ...
        if (bid == 0) {
            loss_pipe_log(a.pipe, (int)gridDim.x - 1, a.inv_gb);
            return;
        }
        --bid;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long img0 = (long long)bid * IPB;
    const long long img = img0 + wave;
    const long long gbase = img0 * (long long)D * S2;
    const int pad = (int)(gbase & 3);
    const bool live = lane < S2 && img < a.batch;    // this lane owns a cell
    const int cell = lane < S2 ? lane : 0;
    const int item = wave * S2 + cell;               // the cell's index in the workgroup's block
    YH_LSTAMP(0);

    // ---- 1. the cell's logits, its mask byte, then its ground truth --------------------
    float v[D];
#pragma unroll
    for (int ch = 0; ch < D; ++ch) v[ch] = 0.0f;
    bool has_obj = false;
    float4 gq[GT_FLOATS / 4];
#pragma unroll
    for (int q = 0; q < GT_FLOATS / 4; ++q) gq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const long long tcell = img * S2 + cell;
        const uint8_t m = a.obj_mask[tcell];
        const float* p = a.pred + img * (long long)D * S2 + cell;
#pragma unroll
        for (int ch = 0; ch < D; ++ch) v[ch] = p[ch * S2];
        has_obj = m != 0;
        if (has_obj) {
            gq[0] = *reinterpret_cast<const float4*>(a.gt_xywh + tcell * 4);
            const float4* gc = reinterpret_cast<const float4*>(a.gt_cls + tcell * C);
#pragma unroll
            for (int q = 0; q < C / 4; ++q) gq[1 + q] = gc[q];
        }
    }

    // ---- 2. loss pass 1 ---------------------------------------------------------------
    float conf_term = 0.f;
    if (live) {
        float* gl = sgrad + pad + wave * (D * S2) + cell;
        if (has_obj) {
#pragma unroll
            for (int ch = 0; ch < D; ++ch) gl[ch * S2] = v[ch];
        } else {
            auto logit = [&](int ch) -> float { return v[ch]; };
            loss_noobj_cell(logit, gl, S2, BX, C, a.ln, a.inv_gb, conf_term);
        }
    }
    const unsigned long long omask = __ballot(has_obj);
    const int opos = __popcll(omask & ((1ull << lane) - 1ull));
    if (has_obj) olist[wave * 64 + opos] = item;
    if (lane == 0) wcount[wave] = __popcll(omask);
    if (lane < S2) sconf[item] = conf_term;
    if (threadIdx.x >= IPB * S2) sconf[threadIdx.x] = 0.f;   // the lanes beyond the block's items
    YH_LSTAMP(1);

    // ---- 3. NMS of this wavefront's image ------------------------------------------------
    unsigned char* region = wlds[wave];
    if (img < a.batch) {                                   // wave-uniform
        Box cbox = {0.f, 0.f, 0.f, 0.f};
        float cscore = 0.f;
        int ccl = 0, cidx = 0;
        bool cvalid = false;
        if (lane < S2) {
            int k;
            cell_candidate_from_registers<BX, C>(v, cell, S, a.cellf, a.imgf, cbox, cscore, ccl, k);
            cidx = cell * BX + k;
            cvalid = cscore >= a.conf_thr;
        }
        WaveNmsTables tb;
        tb.cmask = reinterpret_cast<unsigned long long*>(region);
        tb.killt = reinterpret_cast<unsigned long long*>(region + 512);
        tb.hight = tb.killt + 64;
        tb.sbox = reinterpret_cast<float4*>(region + 1536);
        tb.sscore = reinterpret_cast<float*>(region + 2560);
        tb.cbase = reinterpret_cast<int*>(region + 2816);
        nms_wave_one_per_lane(tb, C, cbox, cscore, ccl, cidx, cvalid, a.iou_thr,
                              a.dets + img * (long long)a.nout * 6,
                              a.keep_idx ? a.keep_idx + img * (long long)a.nout : nullptr,
                              a.keep_count + img, a.nout, [](int) {});
    }

    YH_LSTAMP(2);
    // ---- 4. the ground truth is parked where the NMS tables were; pass 2 ---------------------
    group_sync<64>();                                      // this wavefront's table reads are done
    if (has_obj && opos < GT_SLOTS) {
        float4* slot = reinterpret_cast<float4*>(region) + opos * (GT_FLOATS / 4);
#pragma unroll
        for (int q = 0; q < GT_FLOATS / 4; ++q) slot[q] = gq[q];
    }
    __syncthreads();
    YH_LSTAMP(3);
    float acc_coord = 0.f, acc_cls = 0.f;
    float acc_conf = sconf[threadIdx.x];                   // as if thread i had done item i in pass 1
    loss_pass2<S, BX, C>(olist, wcount, sgrad + pad, img0,
                         [&](int w, int j, long long tcell, const float*& gx, const float*& gc) {
                             if (j < GT_SLOTS) {
                                 gx = reinterpret_cast<const float*>(wlds[w]) + j * GT_FLOATS;
                                 gc = gx + 4;
                             } else {
                                 gx = a.gt_xywh + tcell * 4;
                                 gc = a.gt_cls + tcell * C;
                             }
                         },
                         a.cellf, a.imgf, a.lc, a.ln, a.inv_gb, acc_coord, acc_conf, acc_cls);
    YH_LSTAMP(4);
    loss_block_reduce(acc_coord, acc_conf, acc_cls, a.partials, bid);   // has a barrier after the LDS writes
    YH_LSTAMP(5);
    loss_tile_store<S, BX, C>(sgrad, pad, a.grad, gbase, img0, a.batch);
    YH_LSTAMP(6);
}

}  // namespace yh
