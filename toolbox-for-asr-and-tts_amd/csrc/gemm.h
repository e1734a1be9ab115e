// Parameters of the bf16 MFMA GEMM (gemm.hip).
#pragma once

#include "common.h"

namespace asrfe {

enum GemmEpilogue {
    EPI_BF16 = 0,       // out bf16 = acc + bias
    EPI_BF16_RELU = 1,  // out bf16 = relu(acc + bias)
    EPI_F32_ADD = 2,    // out f32  = acc + bias (+ add16) (+ add32)
};

struct GemmParams {
    const uint16_t* A;  // bf16 [M][lda]
    int lda;
    const uint16_t* W;  // bf16 [N][ldw]
    int ldw;
    const float* bias;  // [N]
    int M, N, K;
    void* out;          // [M][ldo]
    int ldo;
    const uint16_t* add16;  // optional 16-bit [M][ld16] (same element type as A)
    int ld16;
    const float* add32;     // optional f32 [M][ld32]
    int ld32;
    int n_tiles_m, n_tiles_n;  // filled by launch_gemm
    int flags;                 // bit 0: skip the epilogue stores (timing experiments)
};

int32_t launch_gemm(asrfe_ctx* ctx, const GemmParams& p, int epi, hipStream_t s);

}  // namespace asrfe
