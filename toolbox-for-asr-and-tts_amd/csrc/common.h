// Shared declarations for libasrfe (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "asrfe.h"

namespace asrfe {
struct DecoderWeights;  // decoder.h
struct VocoderWeights;  // vocoder.h
struct VadWeights;      // vad.h

// ---- frontend geometry the kernels are built for -------------------------------------------
constexpr int kFrameLen = 400;
constexpr int kFrameShift = 160;
constexpr int kNfft = 512;
constexpr int kNc = 256;          // complex points of the packed real FFT
constexpr int kMaxMels = 80;
constexpr int kMaxFeatDim = 640;  // lfr_m * n_mels the fused frontend kernel keeps CMVN constants for
constexpr int kMelGroups = 5;     // filters handled per lane: m = lane + 16 * g
constexpr int kMelMaxTerms = 64;  // sum over groups of the widest filter in each group
constexpr int kWinPad = 416;      // 13 * 32 window slots, zero past kFrameLen

// Device tables of the frontend, built on the host in asrfe_create.
struct FrontendTables {
    float* window;    // [kWinPad]
    float2* tw256;    // [16][16]  W256^(n2*k1), row n2
    float2* tw512;    // [129]     W512^k
    float* mel_w;     // [kMelMaxTerms / 4][16][4] weights: 4-term chunk, lane, term (includes the 1/4 power scale)
    int32_t* mel_sb;  // [kMelGroups][16] first bin of filter lane + 16 g
    int32_t mel_gw[kMelGroups];  // widest filter of each group, rounded up to a multiple of 4 terms
    int32_t mel_go[kMelGroups];  // first term of each group in mel_w (a multiple of 4)
    float* cmvn_shift;  // [lfr_m * n_mels]
    float* cmvn_scale;
};

struct FbankParams {
    const void* pcm;
    int64_t pcm_stride;
    const int32_t* pcm_lens;
    int32_t batch;
    int32_t t_max;       // rows per clip in logmel (fbank) / frames covered (fused: 6 * tp_max)
    float* logmel;       // [B, t_max, n_mels]     (fbank mode)
    int32_t* n_frames;   // [B]
    void* feats;         // [B, tp_max, lfr_m*n_mels] (fused mode)
    int32_t tp_max;
    int32_t* lfr_lens;   // [B]
    int32_t n_mels;
    int32_t lfr_m, lfr_n;
    int32_t aligned16;   // pcm base and row stride allow 16-byte vector loads
    FrontendTables tab;
};

// ---- encoder ----------------------------------------------------------------------------------
struct LayerWeights {
    // device pointers (library-owned)
    // norm1 and norm2 have no parameters of their own here: their scale and shift are folded into the linear
    // layer behind them (w_qkv / b_qkv and w_ffn1 / b_ffn1) when the weights are loaded
    uint16_t* w_qkv; float* b_qkv;     // 16-bit [1536][kin_pad], f32 [1536]
    float* w_fsmn;                     // f32 [11][512]  (tap-major)
    uint16_t* w_out; float* b_out;     // bf16 [512][512]
    uint16_t* w_ffn1; float* b_ffn1;   // bf16 [2048][512]
    uint16_t* w_ffn2; float* b_ffn2;   // bf16 [512][2048]
    // w_out, w_ffn1 and w_ffn2 again as the stream of 16 KiB LDS images the fused post-attention kernel
    // consumes, in the order it consumes them (layout: post_attn.hip); with stream_has_qkv the NEXT layer's w_qkv
    // follows, so that the kernel can project that layer's q, k, v as well
    uint16_t* w_stream;
    bool stream_has_qkv;
};

}  // namespace asrfe

struct asrfe_ctx {
    asrfe_config cfg;
    int device;
    asrfe::FrontendTables tab;
    bool cmvn_loaded;
    bool enc_loaded;
    int kin_pad;  // layer-0 K padded to a multiple of 128
    int num_cus;  // compute units of the device (persistent kernels launch one workgroup per CU)
    bool unfused_ffn;  // ASRFE_UNFUSED_FFN=1 in the environment: LayerNorm + two GEMM launches instead of the fused kernel
    bool separate_qkv;  // ASRFE_SEPARATE_QKV=1: every layer launches its own QKV projection
    bool row_major_x;   // ASRFE_ROW_MAJOR_X=1: the residual stream stays row-major between post-attention launches
    bool mfma16;        // the post-attention kernel on v_mfma_f32_16x16x32 (post_attn16.hip): the weight stream is packed
                        // for it when the weights are loaded.  ASRFE_MFMA16=0 keeps the 32x32x16 kernel; so does any plan
                        // that needs the row-major residual stream or a layer without the fused QKV projection
    int small_rows;     // encoder calls of at most this many rows take the latency plan (0: never)
    std::vector<asrfe::LayerWeights> layers;
    float* after_g;
    float* after_b;
    float* pe_inv;  // [input_size/2] inverse timescales of the sinusoidal position code
    asrfe::VocoderWeights* voc = nullptr;  // HiFi-GAN generator weights (asrfe_load_vocoder_weights), else null
    asrfe::DecoderWeights* dec = nullptr;  // CIF predictor + decoder weights (asrfe_load_decoder_weights), else null
    asrfe::VadWeights* vad = nullptr;      // FSMN voice-activity network (asrfe_load_vad_weights), else null
    std::vector<void*> owned;  // every device allocation, freed in asrfe_destroy
    // optional per-kernel event timing (asrfe_profile_begin / _end)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_events;  // start/stop pairs
    std::vector<int> prof_kind;           // kind of each recorded pair
    size_t prof_used = 0;                 // pairs recorded so far
};

namespace asrfe {
enum ProfKind { PK_FBANK = 0, PK_FRONTEND, PK_LFR_CMVN, PK_PE, PK_LN, PK_GEMM_QKV, PK_FSMN, PK_ATTN, PK_GEMM_OUT,
                PK_GEMM_FFN1, PK_GEMM_FFN2, PK_FFN_FUSED };
// Brackets one launch with events when profiling is on; a no-op otherwise.
struct ProfScope {
    asrfe_ctx* ctx;
    hipStream_t s;
    size_t slot;
    bool on;
    ProfScope(asrfe_ctx* c, int kind, hipStream_t st) : ctx(c), s(st), slot(0), on(false) {
        if (c && c->prof_on && (c->prof_used + 1) * 2 <= c->prof_events.size()) {
            slot = c->prof_used++;
            c->prof_kind[slot] = kind;
            on = hipEventRecord(c->prof_events[2 * slot], st) == hipSuccess;
        }
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(ctx->prof_events[2 * slot + 1], s);
    }
};
int32_t set_err(asrfe_ctx* ctx, int32_t code, const std::string& msg);
int32_t launch_fbank(asrfe_ctx* ctx, const FbankParams& p, int pcm_dtype, bool fused, int out_dtype, hipStream_t s);
int32_t launch_lfr_cmvn(asrfe_ctx* ctx, const float* logmel, const int32_t* n_frames, int batch, int t_max,
                        int tp_max, void* feats, int out_dtype, int32_t* lens_out, hipStream_t s);
void free_vocoder(asrfe_ctx* ctx);  // vocoder.cpp
void free_decoder(asrfe_ctx* ctx);  // decoder.cpp: drops the predictor / decoder weight table (device memory is in `owned`)
void free_vad(asrfe_ctx* ctx);      // vad.cpp
int32_t prepare_static_scratch(asrfe_ctx* ctx);  // static_scratch.hip: device address of the library's scratch array
int64_t encoder_workspace_bytes(const asrfe_ctx* ctx, int batch, int tp);
int32_t encoder_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int batch, int tp, void* out,
                        int out_dtype, void* ws, int64_t ws_bytes, hipStream_t s);
}  // namespace asrfe
