// HiFi-GAN generator (mel -> waveform): weights held by the context, kernel parameter blocks and launchers.
#pragma once

#include <vector>

#include "common.h"

namespace asrfe {

// One convolution as an implicit GEMM on the matrix cores: out[co][q] = sum over (tap, ci) of W[co][tap * Ci + ci] *
// x[q + off0 + tap * dtap][ci], activations time-major with the channels contiguous ([B][T][C], 16-bit, already passed
// through the leaky ReLU that precedes every convolution of the generator).  A transposed convolution of stride u is u
// such products ("phases"): output sample u q + r - pad takes taps r, r + u, ... from inputs q, q - 1, ...
struct ConvParams {
    const uint16_t* x;     // [B][Tin][Ci]
    const uint16_t* w;     // [phases][Co][taps * Ci]
    const float* bias;     // [Co]
    int B, Tin, Ci, Co, taps;
    int off0, dtap;        // input time of tap j at position q: q + off0 + j * dtap (outside [0, Tin): zero)
    int phases, q_count;   // positions per phase
    int out_mul, out_add0, out_add_step;  // output time = q * out_mul + out_add0 + phase * out_add_step (outside [0, Tout): dropped)
    int Tout;
    float slope;           // leaky ReLU applied to what y16 receives (and, with act_in, to x while it is staged)
    uint16_t* y16;         // optional [B][Tout][Co]: lrelu(result)
    const void* res_in;    // optional [B][Tout][Co] added to the result: f32, or 16-bit with res16
    void* res_out;         // optional [B][Tout][Co]: the result, f32, or rounded to 16-bit with res16
    int res16;             // the residual stream is kept in the 16-bit operand type (fp16 contexts): res_in / res_out are
                           // 16-bit arrays of the raw (not activated) values
    int act_in;            // x holds raw values: the leaky ReLU in front of the convolution is applied while x is staged
                           // (fp16 operands only)
    int ci_tile;           // input channels staged per pass (filled by launch_conv_mfma)
    int flags;             // timing experiments (ASRFE_VOC_ABLATE; read by a -DASRFE_VOC_TIMING build only, bit 16 by the launcher)
};

struct VocConv {
    uint16_t* w; float* bias;
    int Ci, Co, k, dil;
};
struct VocoderWeights {
    asrfe_vocoder_config cfg;
    VocConv pre;
    struct Stage {
        uint16_t* up_w; float* up_b;  // [u][Co][(k / u) * Ci]
        int Ci, Co, k, u;
        std::vector<VocConv> c1, c2;  // per residual block j and pair m: index j * 3 + m
    };
    std::vector<Stage> stages;
    float* post_w; float* post_b;     // conv_post as f32 [7][C_last], [1]
    int c_last, hop;
};

int32_t launch_conv_mfma(asrfe_ctx* ctx, const ConvParams& p, hipStream_t s);
// a residual pair (dilated convolution, plain convolution, both C -> C over k taps) as one launch on the f32 stream:
// out = x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2; built for 32 and 64 channels (conv_pair_supported)
bool conv_pair_supported(int C, int k, int dil);
// (res16: x and out are 16-bit arrays, see ConvParams)
int32_t launch_conv_pair(asrfe_ctx* ctx, const void* x, void* out, int res16, const VocConv& c1, const VocConv& c2, int B, int T,
                         float slope, hipStream_t s);
int32_t prepare_conv(asrfe_ctx* ctx);  // per-device kernel attributes; called by asrfe_load_vocoder_weights
// mel f32 [B][n_mels][T] -> 16-bit [B][T][n_mels]
int32_t launch_mel_to_ntc(asrfe_ctx* ctx, const float* mel, int B, int n_mels, int T, uint16_t* out, hipStream_t s);
// y16 = lrelu((a + b + ...) * scale, slope) over n_in arrays (at most 4) of n elements
// (res16: the inputs are 16-bit arrays)
int32_t launch_scale_act(asrfe_ctx* ctx, const void* const* in, int res16, int n_in, float scale, float slope, int64_t n, uint16_t* y16,
                         hipStream_t s);
// wav[b][t] = tanh(bias + sum over (tap < 7, ci) w[tap][ci] x[b][t + tap - 3][ci])
int32_t launch_conv_post(asrfe_ctx* ctx, const uint16_t* x, const float* w, const float* bias, int B, int T, int C,
                         float* wav, hipStream_t s);
void free_vocoder(asrfe_ctx* ctx);

}  // namespace asrfe
