// 11-point mAP of eval.py:86-149 at dataset size:
//   1  (many ground truths only) ground truths sorted by class, so that a
//      prediction meets the ones of its own class only
//   2  one lane per prediction: true-positive flag (IoU against the ground truths
//      of its class, scaled by 8 as eval.py:99 does) -> the class index of a true
//      positive, else -1; and its sort key, the confidence as an ordered integer
//   3  device radix sort (radix_sort.hpp) of 64-bit words -- the inverted key in the high
//      half, the row number in the low half -- on the high half, ascending and stable: the
//      confidence-descending order of eval.py:121 with equal confidences in row order
//   4  one workgroup per class: prefix count of that class's true positives over
//      the ranked list, precision/recall in fp64 and the 11 recall-threshold maxima
//   5  one lane: the mean over classes in class order (the reference's own
//      summation order, so that the float64 result is the same bits)
#pragma once
#include "head_math.hpp"

namespace yh {

constexpr int MAP_GT_DIRECT = 2048;   // up to here every prediction scans all ground truths

// class of a row as an index, or -1 when it is not an integer in [0, n_classes)
__device__ __forceinline__ int map_class_index(float clsf, int n_classes) {
    const int ci = (int)clsf;
    return ((float)ci == clsf && ci >= 0 && ci < n_classes) ? ci : -1;
}

// sort word of a ground truth: its class (0xFFFF when it has none: n_classes <= 65,535) above
// its row number
__global__ __launch_bounds__(256) void map_gt_keys_kernel(const float* __restrict__ gt, long long G,
                                                          int n_classes,
                                                          unsigned long long* __restrict__ words) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int ci = map_class_index(gt[g * 5 + 4], n_classes);
    words[g] = ((unsigned long long)(ci < 0 ? 0xFFFFu : (unsigned int)ci) << 32) | (unsigned int)g;
}

// eval.py:96-100: a prediction is a true positive when ANY ground truth of its class
// has 8 * IoU(gt, pred) >= threshold.  A prediction whose class is not an integer
// in [0, n_classes) belongs to no class's list (eval.py:132 compares with the loop
// index), so its flag is never looked at.
// SORTED: gsorted holds the ground truths' sort words in ascending order (class in the high
// half, row number in the low half); else the ground truths pass through LDS, 256 at a time, and the
// SL wavefront-aligned slices of a workgroup (256 / SL predictions each) share the
// scan: slice s takes every SL-th ground truth, the flags are OR-ed in LDS.  Few
// predictions then still fill the machine.
template <bool SORTED, int SL>
__global__ __launch_bounds__(256) void map_tp_key_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, long long P, long long G,
    int n_classes, float iou_thresh, const unsigned long long* __restrict__ gsorted,
    int* __restrict__ tpcls, unsigned long long* __restrict__ words) {
    constexpr int PB = 256 / SL;                       // predictions per workgroup
    const int pl = threadIdx.x % PB, sl = threadIdx.x / PB;
    const long long p = (long long)blockIdx.x * PB + pl;
    const bool live = p < P;
    Box pb = {0.f, 0.f, 0.f, 0.f};
    float conf = 0.f, clsf = -1.f;
    if (live) {
        const float* row = pred + p * 6;
        pb = {row[0], row[1], row[2], row[3]};
        conf = row[4];
        clsf = row[5];
    }
    const int ci = live ? map_class_index(clsf, n_classes) : -1;
    bool tp = false;
    if constexpr (SORTED) {
        if (ci >= 0) {
            // first ground truth of class ci
            long long lo = 0, hi = G;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if ((unsigned int)(gsorted[mid] >> 32) < (unsigned int)ci) lo = mid + 1; else hi = mid;
            }
            for (long long g = lo; g < G && (unsigned int)(gsorted[g] >> 32) == (unsigned int)ci; ++g) {
                const float* gr = gt + (long long)(unsigned int)gsorted[g] * 5;
                const Box gb = {gr[0], gr[1], gr[2], gr[3]};
                tp = tp || (iou_pair(gb, pb) * 8.0f >= iou_thresh);
            }
        }
    } else {
        __shared__ float sgt[256 * 5];
        __shared__ int anytp[PB];
        if (SL > 1 && sl == 0) anytp[pl] = 0;
        for (long long g0 = 0; g0 < G; g0 += 256) {
            const int n = (int)(G - g0 < 256 ? G - g0 : 256);
            __syncthreads();
            for (int i = threadIdx.x; i < n * 5; i += 256) sgt[i] = gt[g0 * 5 + i];
            __syncthreads();
            if (ci >= 0) {
                for (int g = sl; g < n; g += SL) {
                    if (sgt[g * 5 + 4] == clsf) {
                        const Box gb = {sgt[g * 5], sgt[g * 5 + 1], sgt[g * 5 + 2], sgt[g * 5 + 3]};
                        tp = tp || (iou_pair(gb, pb) * 8.0f >= iou_thresh);
                    }
                }
            }
        }
        if constexpr (SL > 1) {
            if (tp) atomicOr(&anytp[pl], 1);
            __syncthreads();
            tp = anytp[pl] != 0;
            if (sl != 0) return;
        }
    }
    if (live) {
        tpcls[p] = tp ? ci : -1;
        // descending sort key; -0 counts as +0 and a NaN confidence ranks first, above +inf,
        // as torch.argsort(descending=True) ranks it (eval.py:121).  The word is sorted
        // ascending on its high half: the inverted key
        const unsigned int ckey = conf != conf ? 0xFFFFFFFFu : float_to_ordered(conf + 0.0f);
        words[p] = ((unsigned long long)(~ckey) << 32) | (unsigned int)p;
    }
}

// ranked[k] = class of the k-th ranked prediction if it is a true positive, else -1
__global__ __launch_bounds__(256) void map_gather_kernel(const int* __restrict__ tpcls,
                                                         const unsigned long long* __restrict__ order,
                                                         long long P, int* __restrict__ ranked) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < P) ranked[k] = tpcls[(unsigned int)order[k]];
}

// mean of the 11 interpolated precisions as np.sum forms it (eight pairwise-combined
// accumulators, then the tail), divided by 11 (eval.py:142)
__device__ __forceinline__ double map_mean_of_11(const double* p) {
    double r = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    r += p[8];
    r += p[9];
    r += p[10];
    return r / 11.0;
}

// Precision and recall of one true positive (fp64, eval.py:124-125) folded into the 11
// recall-threshold maxima (np.linspace(0, 1, 11)).  tp = true positives of the class up
// to and including this one, k = its 0-based rank among all predictions.
__device__ __forceinline__ void map_fold_maxima(long long tp, long long k, long long G, double* b) {
    const double prec = (double)tp / (double)(k + 1);
    const double rec = (double)tp / (double)G;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const double thr = i == 10 ? 1.0 : (double)i * 0.1;
        if (rec >= thr && prec > b[i]) b[i] = prec;
    }
}

// One workgroup of 16 wavefronts per class.  A wavefront owns a contiguous span of
// the ranked list and walks it 64 positions at a time (lane = position, coalesced):
// the running count of the class's true positives is a ballot + popcount, carried
// as a scalar; pass 1 gives every wavefront its starting count, pass 2 forms
// precision and recall (fp64) at the true positives -- between two of them the
// count stays and the precision only falls, so nothing else can raise a maximum
// -- and keeps the 11 recall-threshold maxima per lane; positive doubles order as
// integers, so the lanes merge theirs with an LDS integer max.
// The low half of order[k] is the row of the prediction ranked k-th; `ranked` (may be null)
// is the gathered list tpcls[row], used when given.
constexpr int MAP_AP_THREADS = 1024;
__global__ __launch_bounds__(MAP_AP_THREADS) void map_ap_kernel(
    const int* __restrict__ tpcls, const unsigned long long* __restrict__ order,
    const int* __restrict__ ranked, long long P, long long G, double* __restrict__ ap_out) {
    constexpr int NW = MAP_AP_THREADS / 64;
    __shared__ long long wcnt[NW];
    __shared__ unsigned long long lbest[11];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 11) lbest[threadIdx.x] = 0ull;
    const long long span = ((P + NW - 1) / NW + 63) / 64 * 64;
    const long long k0 = w * span;
    const long long k1 = k0 + span < P ? k0 + span : P;
    auto is_tp = [&](long long k) -> bool {
        if (k >= k1) return false;
        return (ranked ? ranked[k] : tpcls[(unsigned int)order[k]]) == c;
    };
    const bool short_span = span <= 256;     // the flags of pass 1 stay in registers for pass 2
    bool f0 = false, f1 = false, f2 = false, f3 = false;
    long long count = 0;
    for (long long base = k0; base < k1; base += 256) {
        // four rounds in flight
        f0 = is_tp(base + lane); f1 = is_tp(base + 64 + lane); f2 = is_tp(base + 128 + lane);
        f3 = is_tp(base + 192 + lane);
        count += __popcll(__ballot(f0)) + __popcll(__ballot(f1)) + __popcll(__ballot(f2)) +
                 __popcll(__ballot(f3));
    }
    if (lane == 0) wcnt[w] = count;
    __syncthreads();
    long long carry = 0;
    for (int q = 0; q < w; ++q) carry += wcnt[q];
    double b[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) b[i] = 0.0;
    if (count > 0) {
        int round = 0;
        for (long long base = k0; base < k1; base += 64, ++round) {
            const long long k = base + lane;
            const bool v = short_span ? (round == 0 ? f0 : round == 1 ? f1 : round == 2 ? f2 : f3)
                                      : is_tp(k);
            const unsigned long long m = __ballot(v);
            if (m == 0ull) continue;
            if (v) {
                const long long tp = carry + __popcll(m & ((2ull << lane) - 1ull));
                map_fold_maxima(tp, k, G, b);
            }
            carry += __popcll(m);
        }
#pragma unroll
        for (int i = 0; i < 11; ++i)
            if (b[i] > 0.0) atomicMax(&lbest[i], (unsigned long long)__double_as_longlong(b[i]));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double p[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) p[i] = __longlong_as_double((long long)lbest[i]);
        ap_out[c] = map_mean_of_11(p);
    }
}

// The whole metric in ONE workgroup for a small call (P <= MAP_SMALL_P predictions,
// G <= MAP_GT_DIRECT ground truths, n_classes <= MAP_SMALL_CLASSES): one launch in
// place of four.  Thread = prediction: true-positive flag against the ground truths
// (through LDS, 256 at a time); a 64-bit key -- confidence as an ordered integer,
// then the inverted row number (equal confidences in row order), then the payload
// (class of a true positive + 1, else 0) in the low bits, which the unique bits
// above keep out of the comparison; bitonic sort of the 1,024 keys, descending, a key
// per thread in a register (partners 64 or more apart meet through LDS, closer ones
// through wavefront shuffles); then a wavefront per class walks the ranked payloads
// as map_ap_kernel does, and one lane adds the class values in class order.  Same
// arithmetic, same bits as the four-kernel path.
constexpr int MAP_SMALL_P = 1024;
constexpr int MAP_SMALL_CLASSES = 1024;
__global__ __launch_bounds__(MAP_SMALL_P) void map_small_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, int P, int G, int n_classes,
    float iou_thresh, double* __restrict__ ap_out, double* __restrict__ map_out) {
    constexpr int NT = MAP_SMALL_P, NW = NT / 64;
    __shared__ unsigned long long skey[NT];
    __shared__ float sgt[256 * 5];
    __shared__ int srk[NT];
    __shared__ unsigned long long lbest[NW][11];
    __shared__ double sap[MAP_SMALL_CLASSES];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const bool live = t < P;
    Box pb = {0.f, 0.f, 0.f, 0.f};
    float conf = 0.f, clsf = -1.f;
    if (live) {
        const float2* row = reinterpret_cast<const float2*>(pred + (long long)t * 6);   // 24-byte rows
        const float2 r0 = row[0], r1 = row[1], r2 = row[2];
        pb = {r0.x, r0.y, r1.x, r1.y};
        conf = r2.x;
        clsf = r2.y;
    }
    const int ci = live ? map_class_index(clsf, n_classes) : -1;
    bool tp = false;
    for (int g0 = 0; g0 < G; g0 += 256) {
        const int n = G - g0 < 256 ? G - g0 : 256;
        __syncthreads();
        for (int i = t; i < n * 5; i += NT) sgt[i] = gt[(long long)g0 * 5 + i];
        __syncthreads();
        // The ground truths of the prediction's class first, as bit masks (a compare
        // each), then IoU only for those: a wavefront runs as many IoU rounds as its
        // busiest lane has matches, not as there are ground truths.
        for (int q0 = 0; q0 < n; q0 += 64) {
            const int nq = n - q0 < 64 ? n - q0 : 64;
            unsigned long long match = 0ull;
            if (ci >= 0)
                for (int g = 0; g < nq; ++g)
                    match |= (unsigned long long)(sgt[(q0 + g) * 5 + 4] == clsf) << g;
            while (__ballot(match != 0ull)) {
                if (match) {
                    const int g = q0 + __ffsll((long long)match) - 1;
                    match &= match - 1ull;
                    const Box gb = {sgt[g * 5], sgt[g * 5 + 1], sgt[g * 5 + 2], sgt[g * 5 + 3]};
                    tp = tp || (iou_pair(gb, pb) * 8.0f >= iou_thresh);
                }
            }
        }
    }
    // [63:32] confidence (-0 counts as +0, a NaN ranks first as in torch.argsort), [31:21]
    // inverted row, [20:0] payload; rows beyond P carry zeros above an inverted row that is
    // lower than any real one
    const unsigned int ckey = !live ? 0u : (conf != conf ? 0xFFFFFFFFu : float_to_ordered(conf + 0.0f));
    const unsigned int payload = tp ? (unsigned int)(ci + 1) : 0u;
    unsigned long long v = ((unsigned long long)ckey << 32) |
                           ((unsigned long long)((unsigned int)(NT - 1 - t) & 0x7FFu) << 21) | payload;
    for (int k = 2; k <= NT; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long o;
            if (j >= 64) {
                __syncthreads();
                skey[t] = v;
                __syncthreads();
                o = skey[t ^ j];
            } else {
                o = __shfl_xor(v, j, 64);
            }
            const bool keep_max = ((t & j) == 0) == ((t & k) == 0);
            v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    srk[t] = (int)(v & 0x1FFFFFu);
    __syncthreads();
    // A wavefront per class.  First the ranks of the class's true positives are listed
    // (ballot + popcount, in rank order); then lane j takes the j-th of them -- j + 1 true
    // positives at that rank -- so the fp64 quotients are formed for true positives only.
    int* hits = reinterpret_cast<int*>(skey) + w * 64;    // the sort is over: 64 ranks per wavefront
    for (int c = w; c < n_classes; c += NW) {
        if (lane < 11) lbest[w][lane] = 0ull;
        double b[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) b[i] = 0.0;
        int nh = 0;           // ranks listed and not yet folded (at most 64)
        long long done = 0;   // true positives already folded
        auto fold_listed = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < nh) map_fold_maxima(done + lane + 1, hits[lane], G, b);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            done += nh;
            nh = 0;
        };
        for (int base = 0; base < P; base += 64) {
            const int k = base + lane;
            const bool hit = k < P && srk[k] == c + 1;
            const unsigned long long m = __ballot(hit);
            const int cnt = __popcll(m);
            if (nh + cnt > 64) fold_listed();
            if (hit) hits[nh + __popcll(m & ((1ull << lane) - 1ull))] = k;
            nh += cnt;
        }
        if (nh > 0) fold_listed();
        if (done > 0) {
#pragma unroll
            for (int i = 0; i < 11; ++i)
                if (b[i] > 0.0) atomicMax(&lbest[w][i], (unsigned long long)__double_as_longlong(b[i]));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double p[11];
#pragma unroll
            for (int i = 0; i < 11; ++i) p[i] = __longlong_as_double((long long)lbest[w][i]);
            const double ap = map_mean_of_11(p);
            ap_out[c] = ap;
            sap[c] = ap;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (t == 0) {
        double s = 0.0;
        for (int c = 0; c < n_classes; ++c) s += sap[c];          // eval.py:146-148
        *map_out = s / (double)n_classes;                         // eval.py:149
    }
}

__global__ void map_mean_kernel(const double* __restrict__ ap, int n_classes,
                                double* __restrict__ map_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int c = 0; c < n_classes; ++c) s += ap[c];         // eval.py:146-148
        *map_out = s / (double)n_classes;                       // eval.py:149
    }
}

}  // namespace yh
