// Parameters of the 16-bit (fp16 / bf16 operands, fp32 accumulate) MFMA GEMM (gemm.hip).
#pragma once

#include "common.h"

namespace asrfe {

enum GemmEpilogue {
    EPI_BF16 = 0,       // out 16-bit = acc + bias
    EPI_BF16_RELU = 1,  // out 16-bit = relu(acc + bias)
    EPI_F32_ADD = 2,    // out f32  = acc + bias (+ add16) (+ add32)
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
    int n_valid;               // EPI_ARGMAX: columns at or past it do not take part (0: all N do)
    int n_tiles_m, n_tiles_n;  // filled by launch_gemm
    int flags;                 // bit 0: skip the epilogue stores (timing experiments); bits 4-5: store cache policy;
                               // bits 8-9: force the tile height (1: 256 rows, 2: 128 rows; 0: launch_gemm chooses)
};

int32_t launch_gemm(asrfe_ctx* ctx, const GemmParams& p, int epi, hipStream_t s);
int32_t prepare_gemm(asrfe_ctx* ctx);  // per-device kernel attributes; called by asrfe_create
// the same product for a few hundred rows at most (skinny.hip): N a multiple of 32, K a multiple of 512 up to 3072
int32_t launch_skinny_gemm(asrfe_ctx* ctx, const GemmParams& p, int epi, hipStream_t s);

}  // namespace asrfe
