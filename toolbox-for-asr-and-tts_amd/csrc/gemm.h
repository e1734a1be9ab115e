// Parameters of the 16-bit (fp16 / bf16 operands, fp32 accumulate) MFMA GEMM (gemm.hip).
#pragma once

#include "common.h"

namespace asrfe {

enum GemmEpilogue {
    EPI_BF16 = 0,       // out 16-bit = acc + bias
    EPI_BF16_RELU = 1,  // out 16-bit = relu(acc + bias)
    EPI_F32_ADD = 2,    // out f32  = acc + bias (+ add16) (+ add32)
    EPI_RELU_STAT = 4,  // EPI_BF16_RELU that also leaves stat_out (below)
    EPI_F32_NORM = 5,   // EPI_F32_ADD that applies the LayerNorm of its A operand's rows after the product (stat_in, below)
    EPI_ARGMAX = 3,     // no product is stored: out = (value, column) pairs, [M][4 N / 256] x 8 bytes, the largest
                        // acc + bias among each 64-column strip of a row's columns below n_valid (lowest column on a tie;
                        // a strip with nothing above -inf, or only NaN, reports column 0x7fffffff)
};

struct GemmParams {
    const uint16_t* A;  // 16-bit [M][lda]
    int lda;
    const uint16_t* W;  // 16-bit [N][ldw]
    int ldw;
    const float* bias;  // [N]
    int M, N, K;
    void* out;          // [M][ldo]
    int ldo;
    const uint16_t* add16;  // optional 16-bit [M][ld16] (same element type as A)
    int ld16;
    const float* add32;     // optional f32 [M][ld32]
    int ld32;
    // A LayerNorm between two products, applied after the fact.  The first product (EPI_RELU_STAT) leaves, per
    // 64-column strip and row, the sum and the sum of squares of what it stores: stat_out [N / 64][M][2] (strip-major: a
    // wave stores 32 rows of one strip as one 256-byte run).  launch_row_stats_finish turns them into (mean,
    // 1 / sqrt(var + eps)) per row, and the second product (EPI_F32_NORM) takes the unnormalised rows as its A operand
    // and finishes with
    //     out = (acc - mean colsum[n]) rstd + post_bias[n]        (colsum[n] = sum_k W[n][k])
    // with (mean, rstd) of row m from stat_in [M][2]; bias must then be zeros.  (n(a) W^T = (a W^T - mean 1 W^T) / sigma:
    // the normalised operand never exists in memory.)
    float* stat_out;
    const float* stat_in;
    const float* colsum;
    const float* post_bias;
    int n_valid;               // EPI_ARGMAX: columns at or past it do not take part (0: all N do)
    int n_tiles_m, n_tiles_n;  // filled by launch_gemm
    int flags;                 // bit 0: skip the epilogue stores (timing experiments); bits 4-5: store cache policy;
                               // bits 8-9: force the tile height (1: 256 rows, 2: 128 rows; 0: launch_gemm chooses)
};

int32_t launch_gemm(asrfe_ctx* ctx, const GemmParams& p, int epi, hipStream_t s);
// partial [n_strips][M][2] (sum, sum of squares over 64 columns each) -> row_norm [M][2] = (mean, 1 / sqrt(var + eps))
int32_t launch_row_stats_finish(asrfe_ctx* ctx, const float* partial, int n_strips, int M, float eps, float* row_norm, hipStream_t s);
int32_t prepare_gemm(asrfe_ctx* ctx);  // per-device kernel attributes; called by asrfe_create
// the same product for a few hundred rows at most (skinny.hip): N a multiple of 32, K a multiple of 512 up to 3072
int32_t launch_skinny_gemm(asrfe_ctx* ctx, const GemmParams& p, int epi, hipStream_t s);

}  // namespace asrfe
