// 11-point mAP of eval.py:86-149 in three small launches:
//   1  one lane per prediction: true-positive flag (pairwise IoU against the
//      ground truths of its class, scaled by 8 as eval.py:99 does) and its rank
//      by confidence; the class index of a true positive (else -1) is scattered
//      to the rank position
//   2  one workgroup per class: prefix count of that class's true positives over
//      the ranked list, precision/recall in fp64 and the 11 recall-threshold maxima
//   3  one lane: the mean over classes in class order
#pragma once
#include "head_math.hpp"

namespace yh {

__global__ __launch_bounds__(256) void map_rank_tp_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ gt,
                                                          long long P, long long G,
                                                          float iou_thresh,
                                                          int* __restrict__ ranked) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* row = pred + p * 6;
    const Box pb = {row[0], row[1], row[2], row[3]};
    const float conf = row[4], clsf = row[5];
    // eval.py:96-100: same class and 8 * IoU(gt, pred) >= threshold, for any ground truth
    bool tp = false;
    for (long long g = 0; g < G; ++g) {
        const float* gr = gt + g * 5;
        if (gr[4] == clsf) {
            const Box gb = {gr[0], gr[1], gr[2], gr[3]};
            tp = tp || (iou_pair(gb, pb) * 8.0f >= iou_thresh);
        }
    }
    // eval.py:121: position in the confidence-descending order (ties: lower index first)
    long long rank = 0;
    for (long long q = 0; q < P; ++q) {
        const float cq = pred[q * 6 + 4];
        rank += (cq > conf || (cq == conf && q < p)) ? 1 : 0;
    }
    const int ci = (int)clsf;
    ranked[rank] = (tp && (float)ci == clsf) ? ci : -1;
}

__global__ __launch_bounds__(256) void map_ap_kernel(const int* __restrict__ ranked, long long P,
                                                     long long G, double* __restrict__ ap_out) {
    __shared__ int cnt[256];
    __shared__ double best[256][11];
    const int c = blockIdx.x;
    const int t = threadIdx.x;
    const long long chunk = (P + 255) / 256;
    const long long k0 = t * chunk;
    const long long k1 = k0 + chunk < P ? k0 + chunk : P;
    int local = 0;
    for (long long k = k0; k < k1; ++k) local += ranked[k] == c ? 1 : 0;
    cnt[t] = local;
    __syncthreads();
    long long tp = 0;                       // true positives ranked before this chunk
    for (int w = 0; w < t; ++w) tp += cnt[w];
    double b[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) b[i] = 0.0;
    for (long long k = k0; k < k1; ++k) {
        tp += ranked[k] == c ? 1 : 0;
        const double prec = (double)tp / (double)(k + 1);      // eval.py:124
        const double rec = (double)tp / (double)G;             // eval.py:125 (0/0 -> NaN: never >=)
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const double thr = i == 10 ? 1.0 : (double)i * 0.1;   // np.linspace(0, 1, 11)
            if (rec >= thr && prec > b[i]) b[i] = prec;
        }
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) best[t][i] = b[i];
    __syncthreads();
    if (t < 11) {
        double m = 0.0;
        for (int w = 0; w < 256; ++w) m = best[w][t] > m ? best[w][t] : m;
        best[0][t] = m;
    }
    __syncthreads();
    if (t == 0) {
        const double* p = best[0];
        // np.sum over 11 values: eight pairwise-combined accumulators, then the tail
        double r = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        r += p[8];
        r += p[9];
        r += p[10];
        ap_out[c] = r / 11.0;                                   // eval.py:142
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
