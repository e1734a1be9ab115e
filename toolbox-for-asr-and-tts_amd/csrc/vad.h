// FSMN voice-activity network: weights held by the context, kernel parameter blocks and launchers.
#pragma once

#include <vector>

#include "common.h"

namespace asrfe {

struct VadWeights {
    asrfe_vad_config cfg;
    int in_pad;    // input_dim rounded up to a multiple of 128 (the GEMM's K step)
    int w_pad;     // width every hidden activation is padded to: a multiple of 256 (the GEMM's N tile); padding is zero
    int proj_pad;  // proj_dim rounded up to a multiple of 128
    uint16_t* w_in1; float* b_in1;  // [w_pad][in_pad]
    uint16_t* w_in2; float* b_in2;  // [w_pad][w_pad]
    struct Block {
        uint16_t* w_lin;            // [w_pad][w_pad], rows >= proj_dim zero (no bias)
        float* taps;                // [lorder][proj_dim], tap-major
        uint16_t* w_aff; float* b_aff;  // [w_pad][proj_pad]
    };
    std::vector<Block> blocks;
    uint16_t* w_out1; float* b_out1;  // [w_pad][w_pad]
    uint16_t* w_out2; float* b_out2;  // [w_pad][w_pad]
    float* zero_bias;                 // [w_pad]
};

struct VadFsmnParams {
    const uint16_t* p;     // [B * T][ldp] projected rows
    int ldp;
    const float* cache;    // [B][cache_stride] floats, this layer's [hist][C] block at the front; or null (zeros)
    int64_t cache_stride;
    const float* taps;     // [lorder][C]
    int T, C, lorder;
    uint16_t* q;           // [B * T][ldq], columns >= C zero
    int ldq;
};

int32_t launch_vad_cast(asrfe_ctx* ctx, const float* in, int64_t M, int K, uint16_t* out, int ld, hipStream_t s);
int32_t launch_vad_fsmn(asrfe_ctx* ctx, const VadFsmnParams& p, int B, hipStream_t s);
// cache[b] <- the last hist rows of (cache[b] ++ p[b][0 .. len_b))
int32_t launch_vad_cache_update(asrfe_ctx* ctx, const uint16_t* p, int ldp, const int32_t* lens, int B, int T, int C, int hist,
                                float* cache, int64_t cache_stride, hipStream_t s);
int32_t launch_vad_softmax(asrfe_ctx* ctx, const float* logits, int ld, int64_t M, int n, float* scores, hipStream_t s);
int32_t launch_frame_decibel(asrfe_ctx* ctx, const void* pcm, int pcm_dtype, int64_t stride, const int32_t* lens, int B, int t_max,
                             float* out, hipStream_t s);
void free_vad(asrfe_ctx* ctx);

}  // namespace asrfe
