// Shared between decoder.hip (inference) and train.hip (training iteration).
#pragma once
#include <string.h>
#include <vector>
#include "common.h"
#include "excel_internal.h"
#include "../../include/excel_hip.h"

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct excel_decoder {
    excel_decoder_config cfg;
    std::vector<excel_fuse_layer_weights> fuse;
    std::vector<excel_decoder_block_weights> blocks;
    excel_decoder_weights w;
};

static inline GemmArgs ga0(const float* A, const float* B, float* C, const float* bias, const float* res, int M, int N, int K, int lda, int ldb,
                           int ldc, int ldr, int act) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.res = res;
    g.M = M; g.N = N; g.K = K; g.Kld = (K + 3) / 4 * 4;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.act = act; g.out_mode = GEMM_OUT_PLAIN; g.alpha = 1.f; g.zdiv = 1;
    return g;
}

#define TRYD(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// decoder.hip
int excel_launch_dec_softmax(float* s, long long rows, int P, int Pp, int causal, hipStream_t st);
// [B, R, Cc] (row pitch ld) -> [B, Cc, Rp]: out[b][c][r] = in[b][r][c]; columns r in [R, Rp) are zero-filled
int excel_launch_dec_transpose(const float* in, float* out, int B, int R, int Cc, int ld, int Rp, hipStream_t st);
