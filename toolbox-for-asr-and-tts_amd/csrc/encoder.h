// Parameter blocks and launchers of the encoder's non-GEMM kernels.
#pragma once

#include "common.h"

namespace asrfe {

// 4-head attention (d_k = 128) of Tq query rows per utterance onto Tk key / value rows per utterance, of which the
// first lens[b] are valid.  Encoder self-attention: q, k, v are column blocks of one [B*Tp][1536] buffer (attn_self);
// decoder cross-attention: q comes from the token rows, k and v from the projected encoder frames.
struct AttnParams {
    const uint16_t* q;    // 16-bit [B*Tq][ld_q], head h at columns h*128 .. +127
    int ld_q, Tq;
    const uint16_t* k;    // 16-bit [B*Tk][ld_kv], head h at columns h*128 .. +127 from this pointer
    const uint16_t* v;    // same row stride as k
    int ld_kv, Tk;
    const int32_t* lens;  // [B] valid keys per utterance
    uint16_t* out;        // 16-bit [B*Tq][ld_out] attention output
    int ld_out;
    uint16_t* fsmn;       // self-attention only (Tq == Tk), else null: 16-bit [B*Tk][512] FSMN memory block of v
    const float* w_fsmn;  // f32 [11][512] taps, tap-major (with fsmn)
    int B;
    int n_qtiles;         // filled by launch_attention
    float scale_log2;     // log2(e) / sqrt(d_k)
    int flags;            // timing experiments only (results are wrong when set)
};
// the encoder's layout: qkv [B*Tp][ld] = q | k | v
inline AttnParams attn_self(const uint16_t* qkv, int ld, int B, int Tp, const int32_t* lens, uint16_t* out, int ld_out) {
    AttnParams a{};
    a.q = qkv; a.ld_q = ld; a.Tq = Tp;
    a.k = qkv + 512; a.v = qkv + 1024; a.ld_kv = ld; a.Tk = Tp;
    a.lens = lens; a.out = out; a.ld_out = ld_out; a.B = B;
    return a;
}
int32_t launch_attention(asrfe_ctx* ctx, const AttnParams& p, hipStream_t s);

// pe[t][i] = sin((t+1) inv[i]) for i < D/2, cos((t+1) inv[i - D/2]) otherwise
int32_t launch_pe_table(asrfe_ctx* ctx, const float* inv, int Tp, int D, float* pe, hipStream_t s);

// h = LayerNorm(feats * xscale + pe[t]) over D columns -> 16-bit [M][ldh], columns D..ldh-1 zero; g / bta may be
// null (scale 1, shift 0)
int32_t launch_ln_input(asrfe_ctx* ctx, const float* feats, const float* pe, float xscale, const float* g,
                        const float* bta, float eps, int M, int Tp, int D, uint16_t* hout, int ldh, hipStream_t s);

// h = LayerNorm(x) over 512 columns -> bf16 [M][512]
int32_t launch_ln512(asrfe_ctx* ctx, const float* x, const float* g, const float* bta, float eps, int M, uint16_t* hout,
                     hipStream_t s);

// out = LayerNorm(x) over 512 columns, rows t >= lens[b] zero; f32 or bf16
int32_t launch_ln_final(asrfe_ctx* ctx, const float* x, const float* g, const float* bta, float eps, int B, int Tp,
                        const int32_t* lens, void* out, int out_dtype, hipStream_t s);

// Everything after the attention kernel in one launch (post_attn.hip):
//     x_mid = x + fsmn + att Wo^T + b_o,   x_out = x_mid + W2 relu(W1 n(x_mid) + b1) + b2,   y = n(x_out) as 16-bit
// where n(.) normalises a row to zero mean and unit variance (the LayerNorm scale and shift live in the weights).
// x may be null (layer 0: no residual into the attention block); x_out may alias x.  With qkv set (and y null) the
// kernel goes on to the next layer's projection, qkv = n(x_out) Wqkv^T + b_qkv, whose weights close the stream.
constexpr int kPostAttnQkvMaxFf = 5632;  // linear_units up to which the QKV output fits the LDS budget
struct FfnParams {
    const uint16_t* att;   // [M][512] attention output, 16-bit
    const uint16_t* fsmn;  // [M][512] FSMN memory, 16-bit
    const float* x;        // [M][512] or null
    float* x_out;          // [M][512]
    int tiled;             // x (and, with qkv, x_out) in the kernel's register order instead of row-major: per 32-row
                           // group 16 tiles x 4 quads of 64 lanes x 16 bytes, lane l + 32 h of (tile nt, quad g) holding
                           // columns 32 nt + 8 g + 4 h .. + 3 of row l (post_attn.hip), or lane 16 q + c of run 2 a + mh of
                           // block nt holding columns 32 nt + 16 a + 4 q .. + 3 of row 16 mh + c (post_attn16.hip); the
                           // buffers then span whole 128-row tiles
    const uint16_t* wstream;  // Wo, W1 and W2 as the kernel's piece stream (LayerWeights::w_stream)
    const float* bo;       // [512]
    const float* b1;       // [ff]
    const float* b2;       // [512]
    void* y;               // optional 16-bit [M][ldy] in the compute dtype
    int ldy;
    const float* bqkv;     // [1536] folded bias of the next layer's QKV projection (with qkv)
    void* qkv;             // optional 16-bit [M][ldq]: q | k | v of the next layer
    int ldq;
    float eps;
    int M, ff;
    int flags;             // timing experiments only (results are wrong): bit 0 skips the stores
    long long* trace;      // timing build only: cycle-counter stamps of one wave (flags bit 8)
};
int32_t launch_post_attn(asrfe_ctx* ctx, const FfnParams& p, hipStream_t s);
// the same kernel on v_mfma_f32_16x16x32 (post_attn16.hip): tiled residual stream only (qkv output or x_out alone), and a
// weight stream whose accumulator-fed operands are packed in ITS k order (asrfe_ctx::mfma16)
int32_t launch_post_attn16(asrfe_ctx* ctx, const FfnParams& p, hipStream_t s);
int32_t prepare_post_attn16(asrfe_ctx* ctx);
int32_t prepare_post_attn(asrfe_ctx* ctx);  // per-device kernel attributes; called by asrfe_create

}  // namespace asrfe
