// Decode + confidence threshold + per-class greedy NMS, one image per group of
// NT threads (NT = 64: one wavefront per image, four images per workgroup, no
// workgroup barriers; NT = 256: one workgroup per image).
//
// Phases per image, all on chip:
//   A  candidates into registers (decoded from the head tensor, or loaded)
//   B  64-bit sort key (class | inverted score | slot) per candidate; every
//      candidate counts the keys below its own (its rank) and the keys of a
//      lower class (the start of its class segment) from broadcast LDS reads
//   C  scatter to rank order in LDS: classes become contiguous segments, each
//      in descending score order
//   D  greedy suppression, all classes at once.  One wavefront per image: in step
//      s every candidate looks at the s-th member of its own segment (the pivot);
//      if the pivot is still alive and IoU > threshold the candidate dies.
//      Liveness is one 64-bit ballot per 64 positions, held in scalar registers
//      and rebuilt after every step, so a pivot's bit is final before it is
//      used; the loop ends when no live candidate has a pivot left.  One
//      workgroup per image: every (earlier, later) pair inside a block of 64
//      members of a segment is tested once, the pairs spread evenly over the
//      threads, and the blocks are then settled by a fixed point that the lanes
//      of a wavefront reach with ballots (see the code).
//   E  kept rows get their output slot from a prefix popcount over the
//      liveness words and are stored straight from registers (consecutive
//      survivors write consecutive 24-byte rows); the tail is zero-filled.
#pragma once
#include "head_math.hpp"

namespace yh {

constexpr int NMS_BLOCK = 256;
// bucketed ranking: a class bucket up to this size is ranked by counting (n * n comparisons), a
// larger one by chunks of 64 ranked that way plus binary searches between the chunks, whose
// fixed cost (two barriers, the searches) pays from about 200 keys on
#ifndef YH_RANK_COUNT_MAX
#define YH_RANK_COUNT_MAX 192
#endif
constexpr int RANK_COUNT_MAX = YH_RANK_COUNT_MAX;
// threads of a workgroup: 256, or the image's whole group when that is larger
constexpr int nms_block_threads(int nt) { return nt > NMS_BLOCK ? nt : NMS_BLOCK; }

struct NmsArgs {
    // source 0: the head tensor; source 2: the head tensor before the last layer's
    // bias was added, plus that bias (one image's worth of values, same layout)
    const float* pred;
    const float* bias;
    // source 1: caller-supplied candidates
    const float* c_ltrb;
    const float* c_score;
    const int32_t* c_cls;
    const uint8_t* c_present;
    // outputs
    float* dets;
    int32_t* keep_count;
    int32_t* keep_idx;
    long long batch;
    int S, nb, nc, mode;
    // 1: the head tensor is cell-major, (batch, S, S, D) -- a cell's D values consecutive
    // (run-time-shape kernels only); 0: channel-major
    int cell_major;
    int nslot;  // candidate slots per image (S*S in reference mode, S*S*nb dense, N for source 1)
    int nout;   // rows of dets per image
    float cellf, imgf, conf_thr, iou_thr;
    // per-image LDS region: byte offsets of the arrays inside it, and its size
    int off_stage;   // cell-major only: LDS copy of the image's block (-1: read from memory)
    int off_keys, off_box, off_meta, off_seg, off_alive, off_misc, off_ccls, off_hist, off_segst, off_kill;
    int region_bytes;
    int dbg_stop;  // builds with -DYH_PHASE_PROBE only: leave after this phase (0 = run everything)
    // builds with -DYH_PHASE_PROBE only: 16 time stamps per workgroup (100 MHz counter)
    unsigned long long* stamps;
};

#ifdef YH_PHASE_PROBE
#define YH_STAMP(i)                                                                        \
    do {                                                                                   \
        if (a.stamps && threadIdx.x == 0)                                                  \
            a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define YH_STAMP(i) do {} while (0)
#endif

// ---- decode primitives shared by every kernel that forms candidates --------------
// `logit(ch)` returns channel ch of the cell: from registers when the shape is a
// compile-time one (all channels fetched up front, one memory round trip), from
// memory otherwise.  The arithmetic is the same either way, so every kernel
// that decodes a cell produces the same bits.

// reads the cell's channels from the head tensor on use
template <bool BIAS>
struct MemLogitsT {
    const float* __restrict__ p;
    const float* __restrict__ b;   // the cell's bias values, channels S2 apart (BIAS only)
    int S2;
    __device__ __forceinline__ float operator()(int ch) const {
        if constexpr (BIAS) return p[ch * S2] + b[ch * S2];
        else return p[ch * S2];
    }
};
using MemLogits = MemLogitsT<false>;

// class of a cell: argmax of the class logits, first index on ties
template <class L>
__device__ __forceinline__ int cell_class(const L& logit, int c0, int nc) {
    int best = 0;
    float m = logit(c0);
#pragma unroll
    for (int j = 1; j < nc; ++j) {
        const float v = logit(c0 + j);
        if (v > m) { m = v; best = j; }
    }
    return best;
}

// run-time class count: the logits are fetched eight at a time so that one cell
// costs nc/8 memory round trips instead of nc
template <class L>
__device__ __forceinline__ int cell_class_mem(const L& logit, int c0, int nc) {
    int best = 0;
    float m = logit(c0);
    for (int j0 = 1; j0 < nc; j0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (j0 + u < nc) ? logit(c0 + j0 + u) : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + u < nc && t[u] > m) { m = t[u]; best = j0 + u; }
    }
    return best;
}

// predictor with the higher confidence, first index on ties (torch.max, model.py:387)
template <class L>
__device__ __forceinline__ int cell_best_conf(const L& logit, int nb, float& conf_best) {
    int best = 0;
    conf_best = sigmoidf_(logit(4));
#pragma unroll
    for (int k = 1; k < nb; ++k) {
        const float c = sigmoidf_(logit(5 * k + 4));
        if (c > conf_best) { conf_best = c; best = k; }
    }
    return best;
}

// box of the predictor whose x,y,w,h logits are given, in pixels (model.py:185-188)
__device__ __forceinline__ Box box_from_logits(float lx, float ly, float lw, float lh, int cell,
                                               int S, float cellf, float imgf) {
    return decode_box(sigmoidf_(lx), sigmoidf_(ly), sigmoidf_(lw), sigmoidf_(lh),
                      cellf * (float)(cell / S), cellf * (float)(cell % S), cellf, imgf);
}

template <class L>
__device__ __forceinline__ Box cell_box(const L& logit, int k, int cell, int S, float cellf,
                                        float imgf) {
    return box_from_logits(logit(5 * k + 0), logit(5 * k + 1), logit(5 * k + 2), logit(5 * k + 3),
                           cell, S, cellf, imgf);
}

// The candidate of a cell in reference mode (model.py:384-394) from the cell's 5 BX + C
// logits held in registers: the predictor with the higher confidence, its box, its
// confidence and the cell's class.
template <int BX, int C>
__device__ __forceinline__ void cell_candidate_from_registers(float (&v)[5 * BX + C], int cell, int S,
                                                              float cellf, float imgf, Box& box,
                                                              float& conf, int& cls, int& k) {
    auto logit = [&](int ch) -> float { return v[ch]; };
    k = cell_best_conf(logit, BX, conf);
    // The box logits of EVERY predictor are wanted in registers here.  Without this the
    // compiler turns "select among loaded values" into "load from a selected address", and
    // the four loads of the chosen box become a second, dependent trip to memory behind the
    // confidence compare.
#pragma unroll
    for (int q = 0; q < BX; ++q)
        asm volatile("" : "+v"(v[5 * q]), "+v"(v[5 * q + 1]), "+v"(v[5 * q + 2]), "+v"(v[5 * q + 3]));
    float lx = v[0], ly = v[1], lw = v[2], lh = v[3];
#pragma unroll
    for (int q = 1; q < BX; ++q)
        if (k == q) { lx = v[5 * q]; ly = v[5 * q + 1]; lw = v[5 * q + 2]; lh = v[5 * q + 3]; }
    box = box_from_logits(lx, ly, lw, lh, cell, S, cellf, imgf);
    cls = cell_class(logit, 5 * BX, C);
}

template <int NT>
__device__ __forceinline__ void group_sync() {
    if constexpr (NT == 64) {
        // one wavefront: LDS operations of a wave complete in order, only the
        // compiler must not reorder across this point
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// LDS tables of the sort-free NMS of one wavefront (one candidate per lane)
struct WaveNmsTables {
    unsigned long long* cmask;   // class -> mask of the lanes that hold a valid candidate of it (nc_tab words)
    unsigned long long* killt;   // per lane: lanes ahead of it in output order whose IoU with it exceeds the threshold
    unsigned long long* hight;   // per lane: lanes of its class ahead of it in output order
    float4* sbox;                // per lane: its box
    float* sscore;               // per lane: its score
    int* cbase;                  // per class: first output row (nc_tab ints, at least 64)
};

// Per-class greedy NMS of one image by ONE wavefront, one candidate per lane, at most 128
// classes (nc_tab), no sort.  The lanes of a class find each other through the class ->
// lane-mask table (one 64-bit OR per lane).  Every pair of a class is then looked at once,
// the pairs of a class of n spread evenly over its members: a member takes the (n-1)/2
// members that follow it cyclically in lane order (one more for the first half of an even
// class), so the loop runs about half as long as the largest class.  For a pair, the member
// that comes later in the output order (lower score, higher index on equal scores) gets the
// other one's bit in its `higher` word and, when their IoU exceeds the threshold, in its
// `killers` word (LDS ORs).  Greedy NMS is the fixed point "kept iff no killer is kept",
// reached in as many ballot rounds as the longest chain.  The output row is the number of
// kept lanes of lower classes (per-class counts from the mask table, prefix-summed across
// lanes) plus the kept lanes in `higher`.  Rows go to `dets` (nout rows of 6 floats, kept
// rows first, zero rows after them), flat indices to `kidx` (may be null), the count to
// *count.  All 64 lanes of the wavefront must call; `stamp(i)` marks phase boundaries in
// probe builds.
template <class STAMP>
__device__ __forceinline__ void nms_wave_one_per_lane(const WaveNmsTables& tb, int nc_tab, const Box& cbox,
                                                      float myscore, int ccl, int cidx, bool valid,
                                                      float iou_thr, float* dets, int32_t* kd,
                                                      int32_t* count, int nout, const STAMP& stamp) {
    constexpr int NT = 64;
    const int lane = threadIdx.x & 63;
    const bool small = nc_tab <= 64;
    unsigned long long* cmask = tb.cmask;
    unsigned long long* killt = tb.killt;
    unsigned long long* hight = tb.hight;
    float4* sbox = tb.sbox;
    float* sscore = tb.sscore;
    const unsigned long long me = 1ull << lane;
    cmask[lane] = 0ull;
    if (!small && lane + 64 < nc_tab) cmask[lane + 64] = 0ull;   // the table has nc_tab words
    killt[lane] = 0ull;
    hight[lane] = 0ull;
    sbox[lane] = make_float4(cbox.l, cbox.t, cbox.r, cbox.b);
    sscore[lane] = myscore;
    group_sync<NT>();
    if (valid) atomicOr(&cmask[ccl], me);
    group_sync<NT>();
    const unsigned long long mine = valid ? cmask[ccl] : 0ull;
    const int n = __popcll(mine);
    const int p = __popcll(mine & (me - 1ull));
    int h = (n - 1) >> 1;                       // -1 for a lane without a candidate
    if (n > 0 && (n & 1) == 0 && p < (n >> 1)) ++h;
    // bit k of `rot` is lane (lane + 1 + k) & 63: the members after this one, cyclically
    const int sh = (lane + 1) & 63;
    unsigned long long rot = sh ? (mine >> sh) | (mine << (64 - sh)) : mine;
    const IouBand band = make_iou_band(iou_thr);
    while (__ballot(h > 0)) {
        if (h > 0) {
            --h;
            const int j = (lane + __ffsll((long long)rot)) & 63;    // ffs = k + 1
            rot &= rot - 1ull;
            const float sj = sscore[j];
            const float4 b = sbox[j];
            const bool j_first = sj > myscore || (sj == myscore && j < lane);
            const int later = j_first ? lane : j;
            const unsigned long long first_bit = 1ull << (j_first ? j : lane);
            atomicOr(&hight[later], first_bit);
            if (iou_above_band({b.x, b.y, b.z, b.w}, cbox, band)) atomicOr(&killt[later], first_bit);
        }
    }
    group_sync<NT>();
    stamp(3);
    const unsigned long long killers = killt[lane], higher = hight[lane];
    unsigned long long undec = __ballot(valid), kept = 0ull;
    while (undec) {
        const bool und = (undec >> lane) & 1ull;
        const bool die = und && (killers & kept) != 0ull;
        const bool keep = und && !die && (killers & undec) == 0ull;
        const unsigned long long km = __ballot(keep);
        kept |= km;
        undec &= ~(km | __ballot(die));
    }
    stamp(4);
    // kept lanes per class -> first output row of each class
    int* cbase = tb.cbase;
    if (small) {
        cbase[lane] = wave_excl_scan(lane < nc_tab ? __popcll(cmask[lane] & kept) : 0);
    } else {
        // two classes per lane
        const int c0 = 2 * lane;
        const int n0 = c0 < nc_tab ? __popcll(cmask[c0] & kept) : 0;
        const int n1 = c0 + 1 < nc_tab ? __popcll(cmask[c0 + 1] & kept) : 0;
        const int base = wave_excl_scan(n0 + n1);
        cbase[c0] = base;
        cbase[c0 + 1] = base + n0;
    }
    group_sync<NT>();
    const bool iskept = (kept >> lane) & 1ull;
    const int pos = iskept ? cbase[ccl] + __popcll(higher & kept) : 0;
    const int total = __popcll(kept);
    float2* dst2 = reinterpret_cast<float2*>(dets);
    if (iskept) {
        float2* row = dst2 + pos * 3;
        store_stream(row + 0, make_float2(cbox.l, cbox.t));
        store_stream(row + 1, make_float2(cbox.r, cbox.b));
        store_stream(row + 2, make_float2(myscore, (float)ccl));
        if (kd) kd[pos] = cidx;
    }
    for (int i = total * 3 + lane; i < nout * 3; i += NT) store_stream(dst2 + i, make_float2(0.0f, 0.0f));
    if (kd)
        for (int i = total + lane; i < nout; i += NT) kd[i] = -1;
    if (lane == 0) *count = total;
}

// Bit ii of the result: the box at sbox[base + ii] overlaps `later` (area `alater`) by more than
// the threshold, for ii < len <= 16.  Four boxes a step; a box is requested again as soon as it has
// been used, two tests before it is needed.  The reads run up to four positions past base + len:
// the caller's array must be followed by readable LDS (the image's region is).
__device__ __forceinline__ unsigned int iou_hits_run(const float4* sbox, int base, int len, const Box& later,
                                                     float alater, const IouBand& band) {
    unsigned int hits = 0u;
    auto test = [&](const float4& b, int ii) {
        const Box eb{b.x, b.y, b.z, b.w};
        hits |= (iou_above_band_flat(eb, box_area(eb), later, alater, band) ? 1u : 0u) << ii;
    };
    float4 b0 = sbox[base], b1 = sbox[base + 1], b2 = sbox[base + 2], b3 = sbox[base + 3];
    int ii = 0;
    for (; ii + 3 < len; ii += 4) {
        test(b0, ii);
        test(b1, ii + 1);
        b0 = sbox[base + ii + 4];
        b1 = sbox[base + ii + 5];
        test(b2, ii + 2);
        test(b3, ii + 3);
        b2 = sbox[base + ii + 6];
        b3 = sbox[base + ii + 7];
    }
    if (ii < len) test(b0, ii);
    if (ii + 1 < len) test(b1, ii + 1);
    if (ii + 2 < len) test(b2, ii + 2);
    return hits;
}

// The same with two boxes in flight instead of four (eight registers less; reads run up to two
// positions past base + len)
__device__ __forceinline__ unsigned int iou_hits_run2(const float4* sbox, int base, int len, const Box& later,
                                                      float alater, const IouBand& band) {
    unsigned int hits = 0u;
    auto test = [&](const float4& b, int ii) {
        const Box eb{b.x, b.y, b.z, b.w};
        hits |= (iou_above_band_flat(eb, box_area(eb), later, alater, band) ? 1u : 0u) << ii;
    };
    float4 b0 = sbox[base], b1 = sbox[base + 1];
    int ii = 0;
    for (; ii + 1 < len; ii += 2) {
        test(b0, ii);
        b0 = sbox[base + ii + 2];
        test(b1, ii + 1);
        b1 = sbox[base + ii + 3];
    }
    if (ii < len) test(b0, ii);
    return hits;
}

// SRC 0: candidates decoded from the head tensor (S_CT, BX_CT, C_CT > 0: grid
// size and predictor/class counts fixed at compile time, so every channel of a
// cell is a constant offset; 0: run-time); SRC 1: candidates loaded from memory.
// The work of workgroup `block_id` of the NMS grid; `smem` is its LDS
// (region_bytes per image it holds).
template <int NT, int ITEMS, int SRC, int S_CT, int BX_CT, int C_CT, bool CM = false>
__device__ __forceinline__ void nms_block(const NmsArgs& a, int block_id, unsigned char* smem) {
    static_assert(!CM || (S_CT == 0 && SRC == 0), "cell-major tensors take the run-time-shape path");
    constexpr int IPB = nms_block_threads(NT) / NT;  // images per workgroup
    constexpr bool kHead = SRC == 0 || SRC == 2;   // candidates decoded from the head tensor
    constexpr bool kBias = SRC == 2;               // ... to which a bias is added on the way in
    // class-bucketed ranking: a workgroup per image, class count known and bounded
    constexpr bool kBucket = NT != 64 && kHead;
    constexpr bool kWave = NT == 64;               // one wavefront per image
    constexpr int NW = NT / 64;          // wavefronts per image
    const int sub = threadIdx.x / NT;
    const int t = threadIdx.x % NT;
    const int lane = threadIdx.x & 63;
    const int wv = t >> 6;  // wavefront within the image group
    // (one image per workgroup: the image index is uniform, a scalar register pair)
    const long long img = IPB == 1 ? (long long)block_id : (long long)block_id * IPB + sub;
    // NT == 64: a whole wavefront leaves and no workgroup barrier is used below;
    // NT == 256: the grid has exactly one workgroup per image, so this never fires.
    if (img >= a.batch) return;

    unsigned char* region = smem + (size_t)sub * a.region_bytes;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(region + a.off_keys);
    float4* sbox = reinterpret_cast<float4*>(region + a.off_box);
    // per sorted position: {score bits, class << 16 | flat index}, and the first
    // position of its class segment
    int2* smeta = reinterpret_cast<int2*>(region + a.off_meta);
    int* sseg = reinterpret_cast<int*>(region + a.off_seg);
    unsigned long long* alive = reinterpret_cast<unsigned long long*>(region + a.off_alive);
    int* misc = reinterpret_cast<int*>(region + a.off_misc);  // [0] valid count, [1] largest position in a segment, [2] block heads queued, [3] members queued in a round; from [8]: scan carries
    int* ccls = reinterpret_cast<int*>(region + a.off_ccls);

    const int nslot = a.nslot;
    const int S = S_CT > 0 ? S_CT : a.S;
    const int S2 = S * S;

    YH_STAMP(0);
    // ---- phase A: candidates ---------------------------------------------------
    Box cbox[ITEMS];
    float cscore[ITEMS];
    int ccl[ITEMS], cidx[ITEMS];
    bool cvalid[ITEMS];
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        cbox[e] = {0.f, 0.f, 0.f, 0.f};
        cscore[e] = 0.f; ccl[e] = 0; cidx[e] = 0; cvalid[e] = false;
    }
    if (t == 0) { misc[0] = 0; misc[1] = 0; misc[4] = 0; }   // [4]: a bucket above RANK_COUNT_MAX is there (bucketed ranking)
    if constexpr (!kWave) {
        // bins and cursors of the pair phase's row sort (the tables of the one-wavefront path lie here)
        if (t < 160) reinterpret_cast<int*>(region + a.off_kill)[t] = 0;
    }
    if constexpr (kBucket) {
        // the class histogram starts at zero; with a candidate per predictor the barrier
        // behind the class argmax orders this before the first count
        int* hist0 = reinterpret_cast<int*>(region + a.off_hist);
        for (int c = t; c < a.nc; c += NT) hist0[c] = 0;
    }

    if constexpr (kHead) {
        constexpr bool kStatic = BX_CT > 0 && C_CT > 0;
        // all channels of a cell in registers only while that stays cheap
        constexpr bool kStaticCell = kStatic && (5 * BX_CT + C_CT <= 96);
        const int nb = kStatic ? BX_CT : a.nb;
        const int nc = kStatic ? C_CT : a.nc;
        const long long D = 5 * nb + nc;
        const float* pimg = a.pred + img * D * S2;
        // floats between two channels of a cell and between two cells of a channel
        const int chs = CM ? 1 : S2;
        const int cells = CM ? (int)D : 1;
        bool cls_ready = false;    // the cells' classes are in ccls already
        if constexpr (CM) {
            // A lane's cell is D floats away from its neighbour's: read from memory cell by
            // cell, a wavefront load would touch a line per lane.  The image's block comes in
            // through LDS in consecutive 4-byte loads instead (when the host found room for
            // it) and the lanes pick their cells up from there.
            if (a.off_stage >= 0) {
                float* stage = reinterpret_cast<float*>(region + a.off_stage);
                const int nfl = (int)D * S2;
                for (int i = t; i < nfl; i += NT) stage[i] = pimg[i];
                group_sync<NT>();
                pimg = stage;
            } else {
                // No room: 16 lanes share a cell's class logits, each wavefront load covering
                // 64 consecutive bytes of four cells, and settle the argmax among themselves
                // (first index on ties; a NaN counts only at index 0, as in cell_class_mem).
                const int l16 = t & 15;
                const int rounds = (S2 + (NT >> 4) - 1) / (NT >> 4);
                for (int r = 0; r < rounds; ++r) {
                    const int cell_raw = r * (NT >> 4) + (t >> 4);
                    const int cell = min(cell_raw, S2 - 1);
                    const float* p = pimg + (size_t)cell * cells + 5 * nb;
                    float m = l16 == 0 ? p[0] : -INFINITY;
                    int best = l16 == 0 ? 0 : 0x7fffffff;
                    for (int j0 = l16 == 0 ? 16 : l16; j0 < nc; j0 += 64) {
                        float v4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v4[u] = j0 + 16 * u < nc ? p[j0 + 16 * u] : 0.0f;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (j0 + 16 * u < nc && v4[u] > m) { m = v4[u]; best = j0 + 16 * u; }
                    }
#pragma unroll
                    for (int o = 8; o >= 1; o >>= 1) {
                        const float om = __shfl_xor(m, o, 16);
                        const int ob = __shfl_xor(best, o, 16);
                        if (om > m || (om == m && ob < best)) { m = om; best = ob; }
                    }
                    if (l16 == 0 && cell_raw < S2) ccls[cell] = best;
                }
                group_sync<NT>();
                cls_ready = true;
            }
        }
        if (a.mode == 0) {
            // slot = cell; the higher-confidence predictor is the candidate
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                const int cell = e * NT + t;
                if (cell < nslot) {
                    const float* p = pimg + cell * cells;
                    float cb;
                    int k;
                    if constexpr (kStaticCell) {
                        // every channel of the cell in flight at once
                        float v[5 * BX_CT + C_CT];
#pragma unroll
                        for (int ch = 0; ch < 5 * BX_CT + C_CT; ++ch) v[ch] = p[ch * S2];
                        if constexpr (kBias) {
                            const float* bq = a.bias + cell;
#pragma unroll
                            for (int ch = 0; ch < 5 * BX_CT + C_CT; ++ch) v[ch] += bq[ch * S2];
                        }
                        cell_candidate_from_registers<BX_CT, C_CT>(v, cell, S, a.cellf, a.imgf, cbox[e], cb,
                                                                   ccl[e], k);
                    } else {
                        const MemLogitsT<kBias> logit{p, kBias ? a.bias + cell : nullptr, chs};
                        k = cell_best_conf(logit, nb, cb);
                        cbox[e] = cell_box(logit, k, cell, S, a.cellf, a.imgf);
                        ccl[e] = cls_ready ? ccls[cell] : cell_class_mem(logit, 5 * nb, nc);
                    }
                    cscore[e] = cb;
                    cidx[e] = cell * nb + k;
                    cvalid[e] = cb >= a.conf_thr;
                }
            }
        } else {
            // slot = cell*nb + k; the class is computed once per cell
            constexpr bool kSplit = kStatic && NT >= 2 * (S_CT * S_CT) && C_CT >= 32 && C_CT % 16 == 0;
            float blog[ITEMS][5];     // kStatic: the candidates' own logits, requested early
            if constexpr (kSplit) {
                // More threads than cells: a cell's classes are shared out in slices of 16, a
                // (slice, cell) pair per work item, every item's 16 logits requested before
                // the first is looked at -- one round trip, as when a lane takes a whole
                // cell, at a fraction of the registers.  The slices meet in LDS; the lowest
                // slice wins a tie (first index).
                constexpr int NSL = C_CT / 16;                       // slices per cell
                constexpr int NITEM = NSL * S_CT * S_CT;             // (slice, cell) pairs
                constexpr int PER = (NITEM + NT - 1) / NT;           // per thread
                float* smax = reinterpret_cast<float*>(keys);        // the keys are formed later
                int* sarg = reinterpret_cast<int*>(smax + NITEM);
                static_assert(NITEM * 8 <= NT * ITEMS * 8, "the partial results fit the key array");
                // the thread's own candidates' box and confidence logits travel with the class logits
#pragma unroll
                for (int e = 0; e < ITEMS; ++e) {
                    const int n = e * NT + t;
                    const int cell = n < nslot ? n / BX_CT : 0;
                    const int k = n < nslot ? n - cell * BX_CT : 0;
                    const float* q = pimg + cell + (size_t)(5 * k) * S2;
#pragma unroll
                    for (int u = 0; u < 5; ++u) blog[e][u] = q[u * S2];
                    if constexpr (kBias) {
                        const float* bq = a.bias + cell + (size_t)(5 * k) * S2;
#pragma unroll
                        for (int u = 0; u < 5; ++u) blog[e][u] += bq[u * S2];
                    }
                }
                float v[PER][16];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int w = t + NT * i;
                    const int sl = w < NITEM ? w / (S_CT * S_CT) : 0;
                    const int cell = w < NITEM ? w - sl * (S_CT * S_CT) : 0;
                    const float* p = pimg + cell + (size_t)(5 * BX_CT + 16 * sl) * S2;
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[i][u] = p[u * S2];
                    if constexpr (kBias) {
                        const float* bq = a.bias + cell + (size_t)(5 * BX_CT + 16 * sl) * S2;
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[i][u] += bq[u * S2];
                    }
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int w = t + NT * i;
                    if (w < NITEM) {
                        const int sl = w / (S_CT * S_CT);
                        float m = v[i][0];
                        int best = 0;
#pragma unroll
                        for (int u = 1; u < 16; ++u)
                            if (v[i][u] > m) { m = v[i][u]; best = u; }
                        smax[w] = m;
                        sarg[w] = 16 * sl + best;
                    }
                }
                group_sync<NT>();
                if (t < S2) {
                    float m = smax[t];
                    int best = sarg[t];
#pragma unroll
                    for (int sl = 1; sl < NSL; ++sl) {
                        const float x = smax[sl * S2 + t];
                        if (x > m) { m = x; best = sarg[sl * S2 + t]; }
                    }
                    ccls[t] = best;
                }
            } else {
            if constexpr (kStatic) {
                // the candidates' own logits are requested together with the class logits:
                // one round trip to memory instead of two
#pragma unroll
                for (int e = 0; e < ITEMS; ++e) {
                    const int n = e * NT + t;
                    const int cell = n < nslot ? n / BX_CT : 0;
                    const int k = n < nslot ? n - cell * BX_CT : 0;
                    const float* q = pimg + cell + (size_t)(5 * k) * S2;
#pragma unroll
                    for (int u = 0; u < 5; ++u) blog[e][u] = q[u * S2];
                    if constexpr (kBias) {
                        const float* bq = a.bias + cell + (size_t)(5 * k) * S2;
#pragma unroll
                        for (int u = 0; u < 5; ++u) blog[e][u] += bq[u * S2];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {   // S*S <= nslot <= NT*ITEMS
                const int cell = e * NT + t;
                if (cell >= S2 || cls_ready) continue;
                const float* p = pimg + cell * cells;
                if constexpr (kStatic) {
                    float v[C_CT];
#pragma unroll
                    for (int j = 0; j < C_CT; ++j) v[j] = p[(5 * BX_CT + j) * S2];
                    if constexpr (kBias) {
                        const float* bq = a.bias + cell;
#pragma unroll
                        for (int j = 0; j < C_CT; ++j) v[j] += bq[(5 * BX_CT + j) * S2];
                    }
                    auto logit = [&](int ch) -> float { return v[ch]; };
                    ccls[cell] = cell_class(logit, 0, C_CT);
                } else {
                    ccls[cell] = cell_class_mem(
                        MemLogitsT<kBias>{p, kBias ? a.bias + cell : nullptr, chs}, 5 * nb, nc);
                }
            }
            }
            group_sync<NT>();
            YH_STAMP(1);
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                const int n = e * NT + t;
                if (n < nslot) {
                    const int cell = n / nb;
                    const int k = n - cell * nb;
                    const MemLogitsT<kBias> q{pimg + cell * cells + (5 * k) * chs,
                                              kBias ? a.bias + cell + (5 * k) * S2 : nullptr, chs};
                    float lx, ly, lw, lh, lc;
                    if constexpr (kStatic) {
                        lx = blog[e][0]; ly = blog[e][1]; lw = blog[e][2]; lh = blog[e][3]; lc = blog[e][4];
                    } else {
                        lx = q(0); ly = q(1); lw = q(2); lh = q(3); lc = q(4);
                    }
                    cbox[e] = box_from_logits(lx, ly, lw, lh, cell, S, a.cellf, a.imgf);
                    cscore[e] = sigmoidf_(lc);
                    ccl[e] = ccls[cell];
                    cidx[e] = n;
                    cvalid[e] = cscore[e] >= a.conf_thr;
                }
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int n = e * NT + t;
            if (n < nslot) {
                const long long gi = img * nslot + n;
                const float4 bx = reinterpret_cast<const float4*>(a.c_ltrb)[gi];
                cbox[e] = {bx.x, bx.y, bx.z, bx.w};
                cscore[e] = a.c_score[gi];
                ccl[e] = a.c_cls[gi];
                cidx[e] = n;
                const bool present = a.c_present ? a.c_present[gi] != 0 : true;
                cvalid[e] = present && cscore[e] >= a.conf_thr && ccl[e] >= 0 && ccl[e] <= 32767;
            }
        }
    }

#ifdef YH_PHASE_PROBE
    if (a.dbg_stop == 1) { if (cvalid[0] && cscore[0] == 12345.f) a.keep_count[img] = ccl[0] + (int)cbox[0].l; return; }
#endif
    // One wavefront per image, one candidate per lane, at most 128 classes: no sort
    // (nms_wave_one_per_lane).
    if constexpr (kWave && ITEMS == 1) {
        // candidates handed over in memory carry any class up to 32,767: the table in the
        // key array serves them when every valid one is below 64
        int nc_tab;
        if constexpr (kHead) nc_tab = C_CT > 0 ? C_CT : a.nc;
        else nc_tab = __ballot(cvalid[0] && ccl[0] >= 64) ? 129 : 64;
        if (nc_tab <= 128) {
            // the class -> lane-mask table: the key array for up to 64 classes, else the
            // (adjacent) histogram and bucket-start arrays, 2 x 4 bytes per class
            const bool small = nc_tab <= 64;
            WaveNmsTables tb;
            tb.cmask = small ? keys : reinterpret_cast<unsigned long long*>(region + a.off_hist);
            tb.killt = reinterpret_cast<unsigned long long*>(region + a.off_kill);
            tb.hight = tb.killt + 64;
            tb.sbox = sbox;
            tb.sscore = reinterpret_cast<float*>(smeta);
            tb.cbase = small ? sseg : reinterpret_cast<int*>(keys);
            YH_STAMP(2);
            nms_wave_one_per_lane(tb, nc_tab, cbox[0], cscore[0], ccl[0], cidx[0], cvalid[0], a.iou_thr,
                                  a.dets + img * (long long)a.nout * 6,
                                  a.keep_idx ? a.keep_idx + img * (long long)a.nout : nullptr,
                                  a.keep_count + img, a.nout, [&](int i) { YH_STAMP(i); (void)i; });
            YH_STAMP(15);
            return;
        }
    }

    // The same without a sort for two candidates per lane (65 to 128 slots, e.g. a
    // candidate per predictor at S=7, Bx=2): slot s = e*64 + lane, every lane mask is
    // 128 bits in two words (word s >> 6, bit s & 63), the class table holds two words
    // per class (up to 128 classes) and the kill / precedence tables two words per slot.
    if constexpr (kWave && ITEMS == 2) {
        int nc_tab;
        if constexpr (kHead) nc_tab = C_CT > 0 ? C_CT : a.nc;
        else nc_tab = __ballot((cvalid[0] && ccl[0] >= 64) || (cvalid[1] && ccl[1] >= 64)) ? 129 : 64;
        if (nc_tab <= 128) {
            // class -> slot-mask table, a word per class for each half of the slots: both in
            // the key array for up to 64 classes, else the second half's words in the
            // (adjacent) histogram and bucket-start arrays, 2 x 4 bytes per class
            const bool small = nc_tab <= 64;
            unsigned long long* cm0 = keys;
            unsigned long long* cm1 =
                small ? keys + 64 : reinterpret_cast<unsigned long long*>(region + a.off_hist);
            unsigned long long* killt = reinterpret_cast<unsigned long long*>(region + a.off_kill);
            unsigned long long* hight = killt + 256;
            float* sscore = reinterpret_cast<float*>(smeta);
            YH_STAMP(2);
            const unsigned long long me = 1ull << lane;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = e * 64 + lane;
                keys[s] = 0ull;
                killt[2 * s] = 0ull; killt[2 * s + 1] = 0ull;
                hight[2 * s] = 0ull; hight[2 * s + 1] = 0ull;
                sbox[s] = make_float4(cbox[e].l, cbox[e].t, cbox[e].r, cbox[e].b);
                sscore[s] = cscore[e];
            }
            if (!small) {
                if (lane < nc_tab) cm1[lane] = 0ull;
                if (lane + 64 < nc_tab) cm1[lane + 64] = 0ull;
            }
            group_sync<NT>();
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (cvalid[e]) atomicOr(e == 0 ? &cm0[ccl[e]] : &cm1[ccl[e]], me);
            group_sync<NT>();
            int h[2];
            unsigned long long rlo[2], rhi[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = e * 64 + lane;
                const unsigned long long m0 = cvalid[e] ? cm0[ccl[e]] : 0ull;
                const unsigned long long m1 = cvalid[e] ? cm1[ccl[e]] : 0ull;
                const int n = __popcll(m0) + __popcll(m1);
                const int p = e == 0 ? __popcll(m0 & (me - 1ull)) : __popcll(m0) + __popcll(m1 & (me - 1ull));
                h[e] = (n - 1) >> 1;                    // -1 for a slot without a candidate
                if (n > 0 && (n & 1) == 0 && p < (n >> 1)) ++h[e];
                // bit k of the rotated mask is slot (s + 1 + k) & 127: the members after this one, cyclically
                const int sh = (s + 1) & 127;
                const unsigned long long lo = sh >= 64 ? m1 : m0, hi = sh >= 64 ? m0 : m1;
                const int r = sh & 63;
                rlo[e] = r ? (lo >> r) | (hi << (64 - r)) : lo;
                rhi[e] = r ? (hi >> r) | (lo << (64 - r)) : hi;
            }
            const IouBand band = make_iou_band(a.iou_thr);
            while (__ballot(h[0] > 0 || h[1] > 0)) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (h[e] > 0) {
                        --h[e];
                        int k1;                          // distance to the next member, 1-based
                        if (rlo[e]) { k1 = __ffsll((long long)rlo[e]); rlo[e] &= rlo[e] - 1ull; }
                        else { k1 = 64 + __ffsll((long long)rhi[e]); rhi[e] &= rhi[e] - 1ull; }
                        const int s = e * 64 + lane;
                        const int j = (s + k1) & 127;
                        const float sj = sscore[j];
                        const float4 b = sbox[j];
                        const bool j_first = sj > cscore[e] || (sj == cscore[e] && j < s);
                        const int later = j_first ? s : j;
                        const int first = j_first ? j : s;
                        const unsigned long long first_bit = 1ull << (first & 63);
                        atomicOr(&hight[2 * later + (first >> 6)], first_bit);
                        if (iou_above_band({b.x, b.y, b.z, b.w}, cbox[e], band))
                            atomicOr(&killt[2 * later + (first >> 6)], first_bit);
                    }
                }
            }
            group_sync<NT>();
            YH_STAMP(3);
            unsigned long long kl[2][2], hg[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = e * 64 + lane;
                kl[e][0] = killt[2 * s]; kl[e][1] = killt[2 * s + 1];
                hg[e][0] = hight[2 * s]; hg[e][1] = hight[2 * s + 1];
            }
            unsigned long long und[2] = {__ballot(cvalid[0]), __ballot(cvalid[1])};
            unsigned long long kept[2] = {0ull, 0ull};
            while (und[0] | und[1]) {
                bool keep[2], die[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bool u = (und[e] >> lane) & 1ull;
                    die[e] = u && ((kl[e][0] & kept[0]) | (kl[e][1] & kept[1])) != 0ull;
                    keep[e] = u && !die[e] && ((kl[e][0] & und[0]) | (kl[e][1] & und[1])) == 0ull;
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const unsigned long long km = __ballot(keep[e]);
                    kept[e] |= km;
                    und[e] &= ~(km | __ballot(die[e]));
                }
            }
            YH_STAMP(4);
            // kept slots per class -> first output row of each class
            int* cbase = sseg;
            if (small) {
                cbase[lane] = wave_excl_scan(
                    lane < nc_tab ? __popcll(cm0[lane] & kept[0]) + __popcll(cm1[lane] & kept[1]) : 0);
            } else {
                // two classes per lane
                const int c0 = 2 * lane;
                const int n0 = c0 < nc_tab ? __popcll(cm0[c0] & kept[0]) + __popcll(cm1[c0] & kept[1]) : 0;
                const int n1 = c0 + 1 < nc_tab ? __popcll(cm0[c0 + 1] & kept[0]) + __popcll(cm1[c0 + 1] & kept[1]) : 0;
                const int base = wave_excl_scan(n0 + n1);
                cbase[c0] = base;
                cbase[c0 + 1] = base + n0;
            }
            group_sync<NT>();
            const int total = __popcll(kept[0]) + __popcll(kept[1]);
            const int nout = a.nout;
            float2* dst2 = reinterpret_cast<float2*>(a.dets + img * (long long)nout * 6);
            int32_t* kd = a.keep_idx ? a.keep_idx + img * (long long)nout : nullptr;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if ((kept[e] >> lane) & 1ull) {
                    const int pos = cbase[ccl[e]] + __popcll(hg[e][0] & kept[0]) + __popcll(hg[e][1] & kept[1]);
                    float2* row = dst2 + pos * 3;
                    store_stream(row + 0, make_float2(cbox[e].l, cbox[e].t));
                    store_stream(row + 1, make_float2(cbox[e].r, cbox[e].b));
                    store_stream(row + 2, make_float2(cscore[e], (float)ccl[e]));
                    if (kd) kd[pos] = cidx[e];
                }
            }
            for (int i = total * 3 + t; i < nout * 3; i += NT) store_stream(dst2 + i, make_float2(0.0f, 0.0f));
            if (kd)
                for (int i = total + t; i < nout; i += NT) kd[i] = -1;
            if (t == 0) a.keep_count[img] = total;
            YH_STAMP(15);
            return;
        }
    }

    // ---- phase B: keys, rank, segment start ---------------------------------------
    unsigned long long key[ITEMS], clsbase[ITEMS];
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        const int n = e * NT + t;
        const unsigned int sk = ~float_to_ordered(cscore[e] + 0.0f);  // descending score
        clsbase[e] = (unsigned long long)(unsigned int)ccl[e] << 48;
        key[e] = cvalid[e] ? (clsbase[e] | ((unsigned long long)sk << 16) | (unsigned int)n)
                           : ~0ull;
        if constexpr (!kBucket && !kWave) keys[n] = key[e];
    }
    int nvalid_w = 0;   // kWave: valid candidates of the image (wave-uniform)
    if constexpr (kWave) {
        // only valid candidates are ranked: their keys go to the front of the array
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const unsigned long long vm = __ballot(cvalid[e]);
            if (cvalid[e]) keys[nvalid_w + __popcll(vm & ((1ull << lane) - 1ull))] = key[e];
            nvalid_w += __popcll(vm);
        }
    }
    group_sync<NT>();
    YH_STAMP(2);
    int rank[ITEMS], seg[ITEMS];
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) { rank[e] = 0; seg[e] = 0; }
    if constexpr (kWave) {
        // every lane reads the same LDS address per iteration (a broadcast); unrolled
        // so that several reads are in flight.  The class segment starts are found
        // after the scatter, from the sorted order itself.
#pragma unroll 8
        for (int j = 0; j < nvalid_w; ++j) {
            const unsigned long long kj = keys[j];
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) rank[e] += kj < key[e] ? 1 : 0;
        }
    } else if constexpr (kBucket) {
        // Many candidates per image: bucket them by class first, so that a candidate
        // is ranked against the members of its own class only (N * segment length
        // comparisons instead of N * N).
        int* hist = reinterpret_cast<int*>(region + a.off_hist);    // per class: count, then fill cursor
        int* segst = reinterpret_cast<int*>(region + a.off_segst);  // per class: first sorted position
        int* wtot = misc + 8;                                       // per wavefront: scan carry
        const int nc = a.nc;
        if (a.mode == 0) group_sync<NT>();   // no barrier since the histogram was cleared
#pragma unroll
        for (int e = 0; e < ITEMS; ++e)
            if (cvalid[e]) atomicAdd(&hist[ccl[e]], 1);
        group_sync<NT>();
        YH_STAMP(3);
        // exclusive prefix sum over the classes
        if (nc <= 128) {
            // two classes per lane: the first wavefront does it alone, no carries, one barrier less
            if (wv == 0) {
                const int h0 = lane < nc ? hist[lane] : 0;
                const int h1 = 64 + lane < nc ? hist[64 + lane] : 0;
                const int e0 = wave_excl_scan(h0);
                const int e1 = wave_excl_scan(h1) + __builtin_amdgcn_readlane(e0 + h0, 63);
                if (lane < nc) { segst[lane] = e0; hist[lane] = 0; }              // the count becomes
                if (64 + lane < nc) { segst[64 + lane] = e1; hist[64 + lane] = 0; }   // the bucket's fill cursor
            }
            YH_STAMP(4);
        } else {
            // a contiguous chunk of classes per thread, scan per wavefront (DPP), carries through LDS
            const int chunk = (nc + NT - 1) / NT;
            const int cbeg = t * chunk;
            int local = 0;
            for (int q = 0; q < chunk; ++q)
                if (cbeg + q < nc) local += hist[cbeg + q];
            const int incl = wave_excl_scan(local) + local;
            if (lane == 63) wtot[wv] = incl;
            group_sync<NT>();
            YH_STAMP(4);
            int run = incl - local;
            {
                // carries of the wavefronts before mine: a lane per wavefront, one scan
                static_assert(NW <= 64, "a lane per wavefront");
                const int wc = lane < NW ? wtot[lane] : 0;
                run += __builtin_amdgcn_readlane(wave_excl_scan(wc), __builtin_amdgcn_readfirstlane(wv));
            }
            for (int q = 0; q < chunk; ++q) {
                const int c = cbeg + q;
                if (c < nc) {
                    const int h = hist[c];
                    segst[c] = run;
                    run += h;
                    hist[c] = 0;   // becomes the fill cursor of the bucket
                }
            }
        }
        group_sync<NT>();
        YH_STAMP(5);
#pragma unroll
        for (int e = 0; e < ITEMS; ++e)
            if (cvalid[e]) keys[segst[ccl[e]] + atomicAdd(&hist[ccl[e]], 1)] = key[e];
        group_sync<NT>();
        YH_STAMP(6);
        // Rank inside the bucket; the bucket order itself does not matter.  A thread ranks
        // the keys at its bucket POSITIONS, not its own candidates: neighbouring lanes are
        // then in the same bucket, read the same key in every step (one LDS access for all
        // of them) and run loops of the same length -- a crowded class keeps its own
        // wavefronts busy, not every wavefront of the image.  The rank goes back to the
        // candidate's owner through LDS (the slot index is in the key).
        int* rankof = sseg;                       // not written before phase C
        const int nbucketed = segst[nc - 1] + hist[nc - 1];
        // A bucket of n > RANK_COUNT_MAX keys would cost n * n comparisons that way.  Its keys are
        // ranked inside their chunk of 64 here (the same counting loop), written back in that
        // order, and a key's rank is then its place in its own chunk plus, for every other chunk,
        // the number of smaller keys there -- a binary search of seven steps, the chunks'
        // searches in step.
        unsigned long long bigkey[ITEMS];     // a key of such a bucket ...
        int bigpos[ITEMS];                    // ... and its position in chunk order (-1: none)
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int p = e * NT + t;
            bigpos[e] = -1;
            bigkey[e] = 0ull;
            if (p < nbucketed) {
                const unsigned long long kp = keys[p];
                const int c = (int)(kp >> 48);
                const int s0 = segst[c];
                const int n = hist[c];
                const bool big = n > RANK_COUNT_MAX;
                const int first = big ? s0 + ((p - s0) & ~63) : s0;           // of the keys counted here
                const int ncount = big ? min(64, s0 + n - first) : n;
                int cnt = 0;
                for (int j = 0; j < ncount; ++j) cnt += keys[first + j] < kp ? 1 : 0;
                if (!big) rankof[(int)(kp & 0xffffull)] = s0 + cnt;
                else {
                    bigkey[e] = kp;
                    bigpos[e] = first + cnt;
                    if (p == s0) misc[4] = 1;
                }
            }
        }
        group_sync<NT>();
        if (misc[4] != 0) {    // the same for every thread of the image
#pragma unroll
            for (int e = 0; e < ITEMS; ++e)
                if (bigpos[e] >= 0) keys[bigpos[e]] = bigkey[e];
            group_sync<NT>();
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                if (bigpos[e] < 0) continue;
                const unsigned long long kp = bigkey[e];
                const int c = (int)(kp >> 48);
                const int s0 = segst[c], n = hist[c];
                const int nch = (n + 63) >> 6;
                const int own = (bigpos[e] - s0) >> 6;
                int rk = s0 + ((bigpos[e] - s0) & 63);
                for (int q0 = 0; q0 < nch; q0 += 4) {
                    int lo[4], m[4], base[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = q0 + u;
                        lo[u] = 0;
                        base[u] = s0 + 64 * q - 1;
                        m[u] = (q < nch && q != own) ? min(64, n - 64 * q) : 0;
                    }
#pragma unroll
                    for (int step = 64; step >= 1; step >>= 1) {
                        unsigned long long kv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            kv[u] = lo[u] + step <= m[u] ? keys[base[u] + lo[u] + step] : ~0ull;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (kv[u] < kp) lo[u] += step;
                    }
                    rk += (lo[0] + lo[1]) + (lo[2] + lo[3]);
                }
                rankof[(int)(kp & 0xffffull)] = rk;
            }
            group_sync<NT>();
        }
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            if (cvalid[e]) {
                rank[e] = rankof[e * NT + t];
                seg[e] = segst[ccl[e]];
            }
        }
        group_sync<NT>();     // phase C writes the segment starts over the ranks
    } else {
#pragma unroll 8
        for (int j = 0; j < nslot; ++j) {
            const unsigned long long kj = keys[j];
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                rank[e] += kj < key[e] ? 1 : 0;
                seg[e] += kj < clsbase[e] ? 1 : 0;
            }
        }
    }

#ifdef YH_PHASE_PROBE
    if (a.dbg_stop == 2) { if (rank[0] == 12345) a.keep_count[img] = seg[0]; return; }
#endif
    YH_STAMP(7);
    // ---- phase C: scatter to rank order ---------------------------------------------
    int nv_local = 0, mp_local = 0;
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        if (cvalid[e]) {
            sbox[rank[e]] = make_float4(cbox[e].l, cbox[e].t, cbox[e].r, cbox[e].b);
            smeta[rank[e]] = make_int2(__float_as_int(cscore[e]), (ccl[e] << 16) | cidx[e]);
            if constexpr (!kWave) sseg[rank[e]] = seg[e];   // kWave: found from the sorted order below
            mp_local = max(mp_local, rank[e] - seg[e]);
        }
        if constexpr (!kWave) nv_local += __popcll(__ballot(cvalid[e]));
    }
    if constexpr (!kWave) {
        mp_local = wave_max(mp_local);
        if (lane == 0) {
            atomicAdd(&misc[0], nv_local);
            atomicMax(&misc[1], mp_local);
        }
    }
    group_sync<NT>();
    YH_STAMP(8);
    const int nvalid = kWave ? nvalid_w : misc[0];
    int maxpos = kWave ? 0 : misc[1];

#ifdef YH_PHASE_PROBE
    if (a.dbg_stop == 3) return;
#endif
    // ---- phase D: greedy suppression over sorted positions ----------------------------
    Box mybox[ITEMS];
    int4 mymeta[ITEMS];
    bool al[ITEMS];
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        const int i = e * NT + t;
        al[e] = i < nvalid;
        mybox[e] = {0.f, 0.f, 0.f, 0.f};
        mymeta[e] = make_int4(0, 0, 0, 0);
        if (al[e]) {
            // a workgroup per image: the sorted boxes stay in LDS to the end and are read
            // again where they are needed, instead of living in registers through phase D
            if constexpr (NT == 64) {
                const float4 b = sbox[i];
                mybox[e] = {b.x, b.y, b.z, b.w};
            }
            if constexpr (kWave) {
                const int2 m = smeta[i];
                mymeta[e] = make_int4(m.x, m.y >> 16, m.y & 0xffff, 0);
            } else {
                // score, class and flat index stay in LDS until the survivors are written out
                mymeta[e] = make_int4(0, 0, 0, sseg[i]);
            }
        }
    }
    if constexpr (kWave) {
        // Lane = sorted position: a class segment starts where the class changes, so
        // the start of my segment is the highest such position at or below mine.
        unsigned long long heads[ITEMS];
        int prev_last = -1;   // class of the last position of the previous item
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int c = mymeta[e].y;
            int pc = __shfl_up(c, 1, 64);
            if (lane == 0) pc = prev_last;
            heads[e] = __ballot(al[e] && (e * NT + lane == 0 || c != pc));
            prev_last = __shfl(c, 63, 64);
        }
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const unsigned long long mine = heads[e] & ((2ull << lane) - 1ull);
            int sg = mine ? e * NT + 63 - __clzll(mine) : -1;
#pragma unroll
            for (int q = ITEMS - 1; q >= 0; --q)
                if (q < e && sg < 0 && heads[q]) sg = q * NT + 63 - __clzll(heads[q]);
            mymeta[e].w = al[e] ? sg : 0;
        }
    }
    int pos_out[ITEMS];   // output row of a survivor
    int total = 0;        // survivors of the image
    if constexpr (NT == 64) {
        // the liveness words of the image are wave-uniform: keep them in scalar registers
        unsigned long long aw[ITEMS];
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) aw[e] = __ballot(al[e]);
        // a step is needed while some candidate that is still alive has a member of its
        // segment left to be tested against (one ballot; no longest-segment reduction)
        for (int step = 0;; ++step) {
            bool more = false;
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) more |= al[e] && e * NT + t - mymeta[e].w > step;
            if (!__ballot(more)) break;
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                const int i = e * NT + t;
                if (al[e] && i - mymeta[e].w > step) {
                    const int pv = mymeta[e].w + step;  // pivot: step-th member of my class segment
                    unsigned long long w = aw[0];
#pragma unroll
                    for (int q = 1; q < ITEMS; ++q) w = (pv >> 6) == q ? aw[q] : w;
                    if ((w >> (pv & 63)) & 1ull) {
                        const float4 b = sbox[pv];
                        const Box pb = {b.x, b.y, b.z, b.w};
                        if (iou_pair(pb, mybox[e]) > a.iou_thr) al[e] = false;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) aw[e] = __ballot(al[e]);
        }
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            pos_out[e] = total + __popcll(aw[e] & ((1ull << lane) - 1ull));
            total += __popcll(aw[e]);
        }
    } else {
        // A workgroup per image.  A class segment is cut into blocks of 64 members.  The
        // pairs (earlier member i, later member p) inside a block, of all blocks, are
        // spread evenly over the threads, whatever the block lengths; a pair whose IoU
        // exceeds the threshold sets bit (index of i in the block) in the word of p -- the
        // word belongs to the lane that tests p's pairs, so it is filled with plain stores.
        // The blocks are then settled: a member stays when no member that stays is in its
        // word (a fixed point, reached with ballots -- see below).  No barrier per member, and
        // every lane of the pair loop has work.  A segment of more than 64 members takes a
        // round per block: after its k-th block is settled the members of the later blocks
        // test themselves against that block's survivors (a member suppressed this way gets
        // its OWN bit set in its word, which reads as "gone" when its block's turn comes).
        // Most images need one round.
        const int nrounds = (maxpos >> 6) + 1;
        unsigned long long* kills = keys;     // the keys are not needed any more
        // The pair phase works on ROWS: a later member p of a block against its r earlier
        // members, up to 16 of them in a pass (a member with more has its row in several
        // passes, each filling its own 16-bit slice of p's word).  The rows are sorted by falling r (a counting sort over the 63 lengths) and
        // the wavefronts draw them 64 at a time from a queue, the first pass first and the long
        // rows first: the lanes of a wavefront hold rows of nearly the same length, walk them
        // in step (next earlier member: an address increment) and leave the loop together,
        // and no wavefront is left with the long rows.  The row list takes the place of
        // the segment starts, which every thread has read into a register before the barrier
        // below.
        int* rowlist = sseg;
        // [0..63] rows of length 63 - bin, [64..127] fill cursors, [128] queue head; cleared at entry
        int* rowhist = reinterpret_cast<int*>(region + a.off_kill);
        int r[ITEMS];
        // my block of the segment, and its first position (i - r[e])
        auto blk = [&](int e) { return (e * NT + t - r[e] - mymeta[e].w) >> 6; };
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int i = e * NT + t;
            const int pos = al[e] ? i - mymeta[e].w : 0;
            r[e] = pos & 63;                      // members of my block ahead of me
            if (al[e]) kills[i] = 0ull;
            if (r[e] > 0) atomicAdd(&rowhist[63 - r[e]], 1);
        }
        group_sync<NT>();
        YH_STAMP(9);
        const int wv_s = __builtin_amdgcn_readfirstlane(wv);
        // every wavefront scans the 64 bins for itself: no barrier between scan and scatter
        const int hv = rowhist[lane];
        const int hex = wave_excl_scan(hv);
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int i = e * NT + t;
            const int first = __shfl(hex, 63 - r[e], 64);     // all lanes take part
            if (r[e] > 0) rowlist[first + atomicAdd(&rowhist[64 + 63 - r[e]], 1)] = i | (r[e] << 16);
        }
        group_sync<NT>();
        YH_STAMP(10);
#ifdef YH_PHASE_PROBE
        if (a.dbg_stop == 5) return;
#endif
        {
            const IouBand band = make_iou_band(a.iou_thr);
            // rows in pass 0 (all), 1 (longer than 16), 2 (longer than 32), 3 (longer than 48):
            // each a prefix of the sorted list; and the groups of 64 each pass has
            const int n0 = __builtin_amdgcn_readlane(hex + hv, 63), n1 = __builtin_amdgcn_readlane(hex, 47);
            const int n2 = __builtin_amdgcn_readlane(hex, 31), n3 = __builtin_amdgcn_readlane(hex, 15);
            const int g0 = (n0 + 63) >> 6, g1 = (n1 + 63) >> 6, g2 = (n2 + 63) >> 6, g3 = (n3 + 63) >> 6;
            for (;;) {
                int q = 0;
                if (lane == 0) q = atomicAdd(&rowhist[128], 1);
                q = __builtin_amdgcn_readfirstlane(q);
                // Ticket -> (group, pass), group by group: group g of pass k holds rows of at most
                // min(r(g) - 16 k, 16) members, falling in g and in k, so the long units go out first
                // (g3 <= g2 <= g1 <= g0: the first g3 groups are in four passes, the next in three, ...).
                int pass, g;
                if (q < 4 * g3) { g = q >> 2; pass = q & 3; }
                else {
                    q -= 4 * g3;
                    if (q < 3 * (g2 - g3)) { const int d = q / 3; g = g3 + d; pass = q - 3 * d; }
                    else {
                        q -= 3 * (g2 - g3);
                        if (q < 2 * (g1 - g2)) { g = g2 + (q >> 1); pass = q & 1; }
                        else { g = g1 + q - 2 * (g1 - g2); pass = 0; }
                    }
                }
                if (g >= g0) break;
                const int n = pass == 0 ? n0 : pass == 1 ? n1 : pass == 2 ? n2 : n3;
                const int w = (g << 6) + lane;
                if (w < n) {
                    const int ent = rowlist[w];
                    const int p = ent & 0xffff, rp = ent >> 16;
                    const int len = min(rp - 16 * pass, 16);
                    const float4 bp = sbox[p];
                    const Box later{bp.x, bp.y, bp.z, bp.w};
                    const float abp = box_area(later);
                    const int base = p - rp + 16 * pass;
                    const unsigned int hits = iou_hits_run(sbox, base, len, later, abp, band);
                    // p's word, this pass's slice: the earlier members that overlap p
                    reinterpret_cast<unsigned short*>(kills + p)[pass] = (unsigned short)hits;
                }
            }
        }
        group_sync<NT>();
        YH_STAMP(11);
#ifdef YH_PHASE_PROBE
        if (a.dbg_stop == 6) return;
#endif
        YH_STAMP(12);
        const IouBand xband = make_iou_band(a.iou_thr);
        for (int round = 0; round < nrounds; ++round) {
            // The greedy pass over the blocks of this round, all of them at once: "a member stays
            // iff no member that stays is in its word" is a fixed point that the 64 lanes of a
            // wavefront reach with ballots, in as many steps as the longest chain of the image
            // -- nothing walks a block member by member.  A wavefront holds 64 consecutive
            // positions; a block (64 members at most) lies in one such chunk or in two.  Pass 1
            // settles the members whose block starts in their own chunk (all their earlier
            // members are lanes of the same wavefront), pass 2, behind a barrier, the members
            // whose block started in the chunk before: those earlier members' fate is in that
            // chunk's liveness word by then.  A member's own bit in its word says that an earlier
            // block's survivor has suppressed it already.
            bool late[ITEMS];
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                const bool part = al[e] && blk(e) == round;
                const unsigned long long w = part ? kills[e * NT + t] : 0ull;
                const bool gone = (w >> r[e]) & 1ull;
                const int d = r[e] - lane;          // > 0: the block started d lanes before this chunk
                late[e] = part && !gone && d > 0;
                if (part && gone) al[e] = false;
                bool mine = part && !gone && d <= 0;
                // the earlier members that overlap this one, as lanes of this chunk
                const unsigned long long in_chunk = mine ? (w & ((1ull << r[e]) - 1ull)) << -d : 0ull;
                unsigned long long open = __ballot(mine), kept = 0ull;
                while (open) {
                    const bool die = mine && (in_chunk & kept) != 0ull;
                    const bool keep = mine && !die && (in_chunk & open) == 0ull;
                    const unsigned long long km = __ballot(keep);
                    kept |= km;
                    open &= ~(km | __ballot(die));
                    if (die) al[e] = false;
                    if (die || keep) mine = false;
                }
                const unsigned long long lw = __ballot(al[e]);
                if (lane == 0) alive[e * NW + wv] = lw;
            }
            group_sync<NT>();
            YH_STAMP(13);
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                if (__ballot(late[e]) == 0ull) continue;       // the same for the whole wavefront
                bool mine = late[e];
                const int d = mine ? r[e] - lane : 1;
                // (the word again: nothing 64 bits wide is carried over the barrier)
                const unsigned long long earlier = mine ? kills[e * NT + t] & ((1ull << r[e]) - 1ull) : 0ull;
                const unsigned long long in_chunk = earlier >> d;
                const unsigned long long before_chunk = (earlier & ((1ull << d) - 1ull)) << (64 - d);
                const unsigned long long prev = alive[e * NW + wv - 1];   // a late member is not in chunk 0
                if (mine && (before_chunk & prev) != 0ull) { al[e] = false; mine = false; }
                unsigned long long open = __ballot(mine), kept = 0ull;
                while (open) {
                    const bool die = mine && (in_chunk & kept) != 0ull;
                    const bool keep = mine && !die && (in_chunk & open) == 0ull;
                    const unsigned long long km = __ballot(keep);
                    kept |= km;
                    open &= ~(km | __ballot(die));
                    if (die) al[e] = false;
                    if (die || keep) mine = false;
                }
                const unsigned long long lw = __ballot(al[e]);
                if (lane == 0) alive[e * NW + wv] = lw;     // this wavefront's own word
            }
            if (t == 0) misc[3] = 0;   // members queued for the test against this block's survivors
            group_sync<NT>();
            if (round + 1 < nrounds) {
                // The survivors of block `round` against the members of the later blocks of
                // their segment (that block is full: 64 positions from its start).  The later
                // members queue up.
                int* xlist = sseg;           // the segment starts are dead since the pair loop
#pragma unroll
                for (int e = 0; e < ITEMS; ++e) {
                    if (al[e] && blk(e) > round)
                        xlist[atomicAdd(&misc[3], 1)] = (e * NT + t) | (mymeta[e].w << 16);
                }
                group_sync<NT>();
                // A work item is a queued member against 16 of the block's 64 pivots: the run of
                // tests of the pair phase, masked with the pivots that stayed.
                const int nx = misc[3];
                for (int it = t; it < 4 * nx; it += NT) {
                    const int ent = xlist[it >> 2];
                    const int i = ent & 0xffff, seg0 = ent >> 16;
                    const int pb = seg0 + (round << 6) + ((it & 3) << 4);   // the chunk's first pivot
                    const int wd = pb >> 6, sh = pb & 63;
                    unsigned long long stay = alive[wd] >> sh;
                    if (sh > 48) stay |= alive[wd + 1] << (64 - sh);
                    const unsigned int stay16 = (unsigned int)stay & 0xffffu;
                    if (stay16 == 0u) continue;
                    const float4 mb = sbox[i];
                    const Box later{mb.x, mb.y, mb.z, mb.w};
                    if (iou_hits_run2(sbox, pb, 16, later, box_area(later), xband) & stay16) {
                        // the same values whichever chunk's lane gets here
                        kills[i] = 1ull << ((i - seg0) & 63);     // its own bit: gone when its block's turn comes
                        atomicAnd(&alive[i >> 6], ~(1ull << (i & 63)));
                    }
                }
                group_sync<NT>();
#pragma unroll
                for (int e = 0; e < ITEMS; ++e)
                    if (al[e]) al[e] = (alive[e * NW + wv] >> lane) & 1ull;
            }
        }
        YH_STAMP(14);
        // the survivors' boxes, scores, classes and flat indices come back from LDS along with
        // the liveness words
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            if (al[e]) {
                const float4 b = sbox[e * NT + t];
                const int2 m = smeta[e * NT + t];
                mybox[e] = {b.x, b.y, b.z, b.w};
                mymeta[e] = make_int4(m.x, m.y >> 16, m.y & 0xffff, mymeta[e].w);
            }
        }
        // survivors before each chunk of 64 positions: a lane per chunk, one scan per wavefront
        const int ac = lane < ITEMS * NW ? __popcll(alive[lane]) : 0;
        const int aex = wave_excl_scan(ac);
        total = __builtin_amdgcn_readlane(aex + ac, ITEMS * NW - 1);
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int myw = e * NW + wv_s;
            pos_out[e] = __popcll(alive[myw] & ((1ull << lane) - 1ull)) + __builtin_amdgcn_readlane(aex, myw);
        }
    }
#ifdef YH_PHASE_PROBE
    if (a.dbg_stop == 4) { if (al[0] && mybox[0].l == 12345.f) a.keep_count[img] = 1; return; }
#endif

    // ---- phase E: survivors go straight to their output rows; the rest is zero-filled ----
    // A row is 24 bytes, so 8-byte stores are always aligned; consecutive survivors
    // write consecutive rows.
    const int nout = a.nout;
    float2* dst2 = reinterpret_cast<float2*>(a.dets + img * (long long)nout * 6);
    int32_t* kd = a.keep_idx ? a.keep_idx + img * (long long)nout : nullptr;
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        if (al[e]) {
            float2* row = dst2 + pos_out[e] * 3;
            store_stream(row + 0, make_float2(mybox[e].l, mybox[e].t));
            store_stream(row + 1, make_float2(mybox[e].r, mybox[e].b));
            store_stream(row + 2, make_float2(__int_as_float(mymeta[e].x), (float)mymeta[e].y));
            if (kd) kd[pos_out[e]] = mymeta[e].z;
        }
    }
    for (int i = total * 3 + t; i < nout * 3; i += NT) store_stream(dst2 + i, make_float2(0.0f, 0.0f));
    if (kd)
        for (int i = total + t; i < nout; i += NT) kd[i] = -1;
    if (t == 0) a.keep_count[img] = total;
    YH_STAMP(15);
}

template <int NT, int ITEMS, int SRC, int S_CT, int BX_CT, int C_CT, bool CM = false>
__global__ __launch_bounds__(nms_block_threads(NT)) void nms_kernel(const NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    nms_block<NT, ITEMS, SRC, S_CT, BX_CT, C_CT, CM>(a, (int)blockIdx.x, smem);
}

// 512 threads per image at eight wavefronts per SIMD (at most 64 registers a lane):
// four images share a CU as with nms_kernel<256, 4>, each with twice the wavefronts
template <int ITEMS, int SRC, int S_CT, int BX_CT, int C_CT>
__global__ __launch_bounds__(512, 8) void nms_kernel_wide(const NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    nms_block<512, ITEMS, SRC, S_CT, BX_CT, C_CT>(a, (int)blockIdx.x, smem);
}

// ---- candidates to global memory (the values nms_kernel<.,.,0> forms in registers) ---
static __global__ __launch_bounds__(256) void decode_candidates_kernel(
    const float* __restrict__ pred, float* __restrict__ c_ltrb, float* __restrict__ c_score,
    int32_t* __restrict__ c_cls, uint8_t* __restrict__ c_present, long long batch, int S,
    int nb, int nc, int mode, float cellf, float imgf) {
    const int S2 = S * S;
    const long long total = batch * S2;
    const long long D = 5 * nb + nc;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long img = g / S2;
        const int cell = (int)(g - img * S2);
        const float* p = pred + img * D * S2 + cell;
        const MemLogits logit{p, nullptr, S2};
        const int cls = cell_class_mem(logit, 5 * nb, nc);
        float cb;
        const int best = cell_best_conf(logit, nb, cb);
        for (int k = 0; k < nb; ++k) {
            const long long o = g * nb + k;
            const Box b = cell_box(logit, k, cell, S, cellf, imgf);
            reinterpret_cast<float4*>(c_ltrb)[o] = make_float4(b.l, b.t, b.r, b.b);
            c_score[o] = sigmoidf_(logit(5 * k + 4));
            c_cls[o] = cls;
            c_present[o] = (mode == 1 || k == best) ? 1 : 0;
        }
    }
}

}  // namespace yh
