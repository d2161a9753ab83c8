/*
 * The C ABI used from plain C: no PyTorch, no C++.  Device buffers come from
 * hipMalloc, inputs from a small linear congruential generator (the GPU test
 * tests/test_c_abi_demo_gpu.py feeds the same numbers through the Python binding
 * and compares).  Prints one line:
 *   loss <total/B> coord <sum> conf <sum> cls <sum> kept <detections> gradsum <sum |g|>
 *
 * Build (done by `make -C yolo_amd/csrc demo`):
 *   hipcc -x c examples/c_abi_demo.c -Iinclude -Lyolo_amd/lib -lyolohead -o build/c_abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "yolo_head.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_YH(x) do { int rc_ = (x); if (rc_ != YH_OK) { \
    fprintf(stderr, "%s: %s\n", #x, yh_strerror(rc_)); return 3; } } while (0)

static uint32_t lcg_state = 12345u;
static float lcg_uniform(void) {           /* in [0, 1) with 24 random bits */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
}

int main(int argc, char** argv) {
    const int S = 7, Bx = 2, C = 20, img = 448, D = 5 * Bx + C, S2 = S * S;
    const int64_t B = argc > 1 ? atoll(argv[1]) : 64;
    const size_t n_pred = (size_t)B * D * S2, n_cell = (size_t)B * S2;

    float* pred = malloc(n_pred * 4);
    float* gxywh = calloc(n_cell * 4, 4);
    float* gcls = calloc(n_cell * C, 4);
    uint8_t* mask = calloc(n_cell, 1);
    for (size_t i = 0; i < n_pred; ++i) pred[i] = 6.0f * lcg_uniform() - 3.0f;
    for (size_t c = 0; c < n_cell; ++c) {
        if (lcg_uniform() < 0.0625f) {     /* an object in about one cell of sixteen */
            mask[c] = 1;
            for (int q = 0; q < 4; ++q) gxywh[c * 4 + q] = 0.05f + 0.9f * lcg_uniform();
            gcls[c * C + (int)(lcg_uniform() * C)] = 1.0f;
        }
    }

    float *d_pred, *d_gxywh, *d_gcls, *d_grad, *d_terms, *d_dets;
    uint8_t* d_mask;
    int32_t* d_count;
    void* d_ws;
    const size_t ws_bytes = yh_workspace_bytes(YH_OP_LOSS, B, S, Bx, C);
    CHECK_HIP(hipMalloc((void**)&d_pred, n_pred * 4));
    CHECK_HIP(hipMalloc((void**)&d_gxywh, n_cell * 16));
    CHECK_HIP(hipMalloc((void**)&d_gcls, n_cell * C * 4));
    CHECK_HIP(hipMalloc((void**)&d_mask, n_cell));
    CHECK_HIP(hipMalloc((void**)&d_grad, n_pred * 4));
    CHECK_HIP(hipMalloc((void**)&d_terms, 5 * 4));
    CHECK_HIP(hipMalloc((void**)&d_dets, (size_t)B * S2 * Bx * 6 * 4));
    CHECK_HIP(hipMalloc((void**)&d_count, (size_t)B * 4));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_pred, pred, n_pred * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_gxywh, gxywh, n_cell * 16, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_gcls, gcls, n_cell * C * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_mask, mask, n_cell, hipMemcpyHostToDevice));

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_YH(yh_loss_fwd_bwd(d_pred, d_gxywh, d_gcls, d_mask, d_grad, d_terms, d_ws, ws_bytes, B, S,
                             Bx, C, img, 5.0f, 0.5f, 1.0f / (float)B, stream));
    CHECK_YH(yh_decode_nms(d_pred, d_dets, d_count, NULL, YH_MODE_REFERENCE, 0.5f, 0.5f, B, S, Bx, C,
                           img, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    float terms[5];
    float* grad = malloc(n_pred * 4);
    int32_t* count = malloc((size_t)B * 4);
    CHECK_HIP(hipMemcpy(terms, d_terms, sizeof terms, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(grad, d_grad, n_pred * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(count, d_count, (size_t)B * 4, hipMemcpyDeviceToHost));
    double gsum = 0.0;
    long kept = 0;
    for (size_t i = 0; i < n_pred; ++i) gsum += fabs((double)grad[i]);
    for (int64_t b = 0; b < B; ++b) kept += count[b];
    printf("loss %.9g coord %.9g conf %.9g cls %.9g kept %ld gradsum %.9g\n", terms[4], terms[0],
           terms[1], terms[2], kept, gsum);
    /* an argument error comes back as a code, not as a crash */
    if (yh_loss_fwd_bwd(NULL, d_gxywh, d_gcls, d_mask, d_grad, d_terms, d_ws, ws_bytes, B, S, Bx, C,
                        img, 5.0f, 0.5f, 1.0f, stream) != YH_ERR_NULL)
        return 4;
    return 0;
}
