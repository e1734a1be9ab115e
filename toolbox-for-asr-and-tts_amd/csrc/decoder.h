// CIF predictor + Paraformer SAN-M decoder: weights held by the context, kernel parameter blocks and launchers.
#pragma once

#include <vector>

#include "common.h"

namespace asrfe {

struct DecLayerWeights {
    // feed-forward: w_2(LayerNorm(relu(w_1 norm1(x)))); norm1 is folded into w_1 / b_1 and the hidden LayerNorm into
    // w_2, which thereby gets a bias
    uint16_t* w1; float* b1;    // [ff][512], [ff]
    uint16_t* w2; float* b2;    // [512][ff], [512]
    float* c2;                  // [512]: row sums of w2 as stored (the GEMM applies the hidden LayerNorm after the product)
    float* n2_g; float* n2_b;   // norm2 scale / shift (applied in the FSMN kernel)
    float* taps;                // FSMN [11][512], tap-major
    uint16_t* wq; float* bq;    // [512][512] with norm3 folded in, [512]
    uint16_t* wkv; float* bkv;  // [1024][512], [1024]
    uint16_t* wo; float* bo;    // [512][512], [512]
};

struct DecoderWeights {
    asrfe_decoder_config cfg;
    uint16_t* wc_hi; uint16_t* wc_lo; float* bc;  // predictor convolution as [512][3 * 512] (tap-major columns), 16-bit head + remainder
    float* w_alpha; float b_alpha;                // predictor output layer [512], bias
    float* zero_bias;                             // zeros, max(2048, vocab_pad) long
    std::vector<DecLayerWeights> layers;
    uint16_t* w1_last; float* b1_last; uint16_t* w2_last; float* b2_last;  // the feed-forward-only closing block
    float* c2_last;
    uint16_t* w_out; float* b_out;                // [vocab_pad][512] with after_norm folded in, [vocab_pad]
    int vocab_pad;                                // vocab rounded up to a multiple of 256 (GEMM tile width)
};

struct CifParams {
    const float* enc;       // [B][Tp][512]
    const float* alphas;    // [B][Tp] (zero at and past lens[b])
    const int32_t* lens;    // [B]
    int Tp, n_max;
    float threshold, tail_threshold;
    float* embeds;          // [B][n_max][512]
    int32_t* token_num;     // [B]
    float* alphas_out;      // optional [B][Tp + 1]: alphas with the tail weight at position lens[b]
    int32_t* n_fires;       // optional [B]
};

struct DecFsmnParams {
    const float* t;         // [B][N][512] feed-forward output
    float* x;               // [B][N][512] residual stream, updated in place
    const float* gamma; const float* beta;  // norm2
    const float* taps;      // [11][512]
    const int32_t* token_num;
    int N;
    float eps;
    uint16_t* h;            // optional [B][N][512], 16-bit: the plain LayerNorm (no scale / shift) of the updated rows
};

int32_t launch_pred_split(asrfe_ctx* ctx, const float* enc, const int32_t* lens, int B, int Tp, uint16_t* hi, uint16_t* lo, hipStream_t s);
int32_t launch_pred_alpha(asrfe_ctx* ctx, const float* conv, const float* w, float bias, float smooth, float noise,
                          const int32_t* lens, int B, int Tp, float* alphas, hipStream_t s);
int32_t launch_cif(asrfe_ctx* ctx, const CifParams& p, int B, hipStream_t s);
// fires [B][n_max]: the frame of each token's fire (-1 behind the fires), from alphas [B][Tp + 1] with the tail in place
int32_t launch_cif_fire_frames(asrfe_ctx* ctx, const float* alphas, int B, int Tp, int n_max, float thr, int32_t* fires,
                               hipStream_t s);
int32_t launch_ln_hidden(asrfe_ctx* ctx, const uint16_t* in, uint16_t* out, int M, int width, float eps, hipStream_t s);
int32_t launch_dec_fsmn(asrfe_ctx* ctx, const DecFsmnParams& p, int B, hipStream_t s);
// the same from (value, column) pairs, ld per row, that the vocabulary projection's arg-max epilogue leaves
int32_t launch_argmax_pairs(asrfe_ctx* ctx, const void* pairs, int ld, int B, int N, const int32_t* token_num, int32_t* ids,
                            hipStream_t s);
int32_t launch_argmax(asrfe_ctx* ctx, const float* logits, int ld, int vocab, int B, int N, const int32_t* token_num,
                      int32_t* ids, hipStream_t s);

}  // namespace asrfe
