// CIF predictor + Paraformer SAN-M decoder: weight loading, workspace carving and the launch sequences.
//
// Predictor (per batch):   A = [x(t-1) | x(t) | x(t+1)] as 16-bit head + remainder      pred_split kernel
//                          conv = A_hi Wc_hi^T + A_lo Wc_hi^T + A_hi Wc_lo^T + b          three MFMA GEMM launches, f32
//                          alphas = relu(sigmoid(relu(conv) . w + b) smooth - noise)      pred_alpha kernel
//                          embeds, token_num = integrate-and-fire(x, alphas + tail)       cif kernel
// Decoder block:           h = n(x) (norm1 folded into w_1)                               ln kernel, 16-bit
//                          f = relu(h W1^T + b1)                                          GEMM, 16-bit [M][ff]
//                          g = n(f) (hidden LayerNorm folded into w_2)                    ln_hidden kernel
//                          t = g W2^T + b2                                                GEMM, f32
//                          x += mask (conv11(mask norm2(t)) + mask norm2(t))              dec_fsmn kernel
//                          q = n(x) Wq^T + bq (norm3 folded);  kv = mem Wkv^T + bkv       two GEMMs, 16-bit
//                          att = softmax(q k^T / sqrt(128), keys < lens[b]) v             attention kernel (cross mode)
//                          x += att Wo^T + bo                                             GEMM, f32 in place
// Closing block:           t = g W2^T + b2 of the same feed-forward (no residual), n(t) with after_norm folded into
//                          the output layer, logits = n(t) Wout^T + bout (f32), arg-max per token row.
// M = batch * n_max token rows; the residual stream x stays fp32, MFMA operands are the context's 16-bit type.
#include "decoder.h"

#include <cmath>
#include <cstdio>

#include "encoder.h"
#include "gemm.h"
#include "weights.h"

using namespace asrfe;

namespace asrfe {
void free_decoder(asrfe_ctx* ctx) {
    delete ctx->dec;
    ctx->dec = nullptr;
}
}  // namespace asrfe

extern "C" int32_t asrfe_default_decoder_config(asrfe_decoder_config* c) {
    if (!c) return ASRFE_ERR_BAD_ARG;
    c->num_blocks = 16;
    c->linear_units = 2048;
    c->kernel_size = 11;
    c->vocab_size = 8404;
    c->ln_eps = 1e-12f;
    c->threshold = 1.0f;
    c->tail_threshold = 0.45f;
    c->smooth_factor = 1.0f;
    c->noise_threshold = 0.0f;
    return ASRFE_OK;
}

static const char* kWho = "asrfe_load_decoder_weights";

static int32_t dec_range_refusal(asrfe_ctx* ctx, const std::string& what, double bound) {
    char buf[288];
    std::snprintf(buf, sizeof(buf),
                  "asrfe_load_decoder_weights: %s can reach %.4g, beyond the fp16 range (65504): load this model with "
                  "enc_compute_dtype = ASRFE_DT_BF16",
                  what.c_str(), bound);
    return set_err(ctx, ASRFE_ERR_RANGE, buf);
}

// a linear layer behind a LayerNorm: fold, prove the fp16 range of its output when asked, upload as 16-bit + f32 bias
static int32_t up_folded(asrfe_ctx* ctx, const TensorMap& m, const std::string& wname, const std::string& bname,
                         const std::string& gname, const std::string& betaname, int64_t n, int64_t k, int64_t n_pad,
                         bool prove, const std::string& what, uint16_t** w_dev, float** b_dev, float** rowsum_dev = nullptr) {
    const HostTensor *tw, *tc = nullptr, *tg, *tb;
    int32_t rc;
    if ((rc = need(ctx, m, wname, {n, k}, &tw, kWho)) || (rc = need(ctx, m, gname, {k}, &tg, kWho)) ||
        (rc = need(ctx, m, betaname, {k}, &tb, kWho)))
        return rc;
    if (!bname.empty() && (rc = need(ctx, m, bname, {n}, &tc, kWho))) return rc;
    std::vector<float> wf, cf;
    fold_layernorm(tw, tc, tg, tb, wf, cf);
    if (prove && ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) {
        const double bound = linear_bound(wf, cf, n, k, 0, n);
        if (bound > kF16Max) return dec_range_refusal(ctx, what, bound);
    }
    cf.resize(static_cast<size_t>(n_pad), 0.0f);
    if ((rc = up_16(ctx, wf.data(), n, k, n_pad, k, w_dev)) || (rc = upload(ctx, cf, b_dev))) return rc;
    if (rowsum_dev) {
        // sums of the weights as the device holds them (rounded to the 16-bit operand type)
        const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
        std::vector<float> rs(static_cast<size_t>(n_pad), 0.0f);
        for (int64_t r = 0; r < n; ++r) {
            double acc = 0.0;
            for (int64_t i = 0; i < k; ++i) {
                const float w = wf[r * k + i];
                acc += f16 ? f16_to_f32(f16_rne(w)) : bf16_to_f32(bf16_rne(w));
            }
            rs[r] = static_cast<float>(acc);
        }
        if ((rc = upload(ctx, rs, rowsum_dev))) return rc;
    }
    return ASRFE_OK;
}

extern "C" int32_t asrfe_load_decoder_weights(asrfe_ctx* ctx, const asrfe_decoder_config* cfg, const asrfe_tensor_desc* descs,
                                              int32_t n) {
    if (!ctx || !cfg || !descs || n <= 0) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_decoder_weights: null argument");
    if (ctx->dec) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_decoder_weights: weights already loaded");
    if (ctx->cfg.enc_output_size != 512) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_decoder_weights: encoder width must be 512");
    if (cfg->num_blocks < 1 || cfg->kernel_size != 11 || cfg->vocab_size < 1 || cfg->linear_units % 512 != 0 ||
        cfg->linear_units < 512 || cfg->linear_units > 4096 || cfg->threshold != 1.0f)
        return set_err(ctx, ASRFE_ERR_BAD_ARG,
                       "asrfe_load_decoder_weights: num_blocks >= 1, kernel_size 11, threshold 1.0, linear_units a multiple "
                       "of 512 up to 4096");
    TensorMap m;
    int32_t rc;
    if ((rc = tensor_map(ctx, descs, n, "", kWho, m))) return rc;
    const int64_t d = 512, ff = cfg->linear_units, ks = cfg->kernel_size, V = cfg->vocab_size;
    DecoderWeights* W = new DecoderWeights();
    W->cfg = *cfg;
    W->vocab_pad = static_cast<int>((V + 255) / 256 * 256);
    ctx->dec = W;  // (freed with the context; on failure the partial state is dropped below)
    auto fail = [&](int32_t code) {
        free_decoder(ctx);
        return code;
    };
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    const HostTensor* t;

    // ---- predictor: Conv1d(512, 512, 3) as a [512][3 * 512] matrix whose column block j is tap j, split into a
    //      16-bit head and a 16-bit remainder
    if ((rc = need(ctx, m, "predictor.cif_conv1d.weight", {d, d, 3}, &t, kWho))) return fail(rc);
    {
        std::vector<uint16_t> hi(static_cast<size_t>(d * 3 * d)), lo(hi.size());
        for (int64_t o = 0; o < d; ++o)
            for (int64_t c = 0; c < d; ++c)
                for (int64_t j = 0; j < 3; ++j) {
                    const float w = t->data[(o * d + c) * 3 + j];
                    const uint16_t h = f16 ? f16_rne(w) : bf16_rne(w);
                    const float hf = f16 ? f16_to_f32(h) : bf16_to_f32(h);
                    const size_t idx = static_cast<size_t>(o * 3 * d + j * d + c);
                    hi[idx] = h;
                    lo[idx] = f16 ? f16_rne(w - hf) : bf16_rne(w - hf);
                }
        if ((rc = upload(ctx, hi, &W->wc_hi)) || (rc = upload(ctx, lo, &W->wc_lo))) return fail(rc);
    }
    if ((rc = need(ctx, m, "predictor.cif_conv1d.bias", {d}, &t, kWho)) || (rc = up_f32(ctx, t, &W->bc))) return fail(rc);
    if ((rc = need(ctx, m, "predictor.cif_output.weight", {1, d}, &t, kWho)) || (rc = up_f32(ctx, t, &W->w_alpha))) return fail(rc);
    if ((rc = need(ctx, m, "predictor.cif_output.bias", {1}, &t, kWho))) return fail(rc);
    W->b_alpha = t->data[0];
    {
        std::vector<float> z(static_cast<size_t>(std::max<int64_t>(std::max<int64_t>(ff, 2 * d), W->vocab_pad)), 0.0f);
        if ((rc = upload(ctx, z, &W->zero_bias))) return fail(rc);
    }

    // ---- decoder blocks
    auto ffn = [&](const std::string& p, uint16_t** w1, float** b1, uint16_t** w2, float** b2, float** c2) -> int32_t {
        int32_t r;
        if ((r = up_folded(ctx, m, p + ".feed_forward.w_1.weight", p + ".feed_forward.w_1.bias", p + ".norm1.weight",
                           p + ".norm1.bias", ff, d, ff, true, p + " feed-forward hidden layer", w1, b1)))
            return r;
        return up_folded(ctx, m, p + ".feed_forward.w_2.weight", "", p + ".feed_forward.norm.weight",
                         p + ".feed_forward.norm.bias", d, ff, d, false, "", w2, b2, c2);
    };
    W->layers.assign(cfg->num_blocks, DecLayerWeights{});
    for (int i = 0; i < cfg->num_blocks; ++i) {
        const std::string p = "decoder.decoders." + std::to_string(i);
        DecLayerWeights& L = W->layers[i];
        if ((rc = ffn(p, &L.w1, &L.b1, &L.w2, &L.b2, &L.c2))) return fail(rc);
        if ((rc = need(ctx, m, p + ".norm2.weight", {d}, &t, kWho)) || (rc = up_f32(ctx, t, &L.n2_g))) return fail(rc);
        if ((rc = need(ctx, m, p + ".norm2.bias", {d}, &t, kWho)) || (rc = up_f32(ctx, t, &L.n2_b))) return fail(rc);
        if ((rc = need(ctx, m, p + ".self_attn.fsmn_block.weight", {d, 1, ks}, &t, kWho))) return fail(rc);
        {
            std::vector<float> w(static_cast<size_t>(ks * d));
            for (int64_t ch = 0; ch < d; ++ch)
                for (int64_t j = 0; j < ks; ++j) w[j * d + ch] = t->data[ch * ks + j];
            if ((rc = upload(ctx, w, &L.taps))) return fail(rc);
        }
        if ((rc = up_folded(ctx, m, p + ".src_attn.linear_q.weight", p + ".src_attn.linear_q.bias", p + ".norm3.weight",
                            p + ".norm3.bias", d, d, d, true, p + " cross-attention q", &L.wq, &L.bq)))
            return fail(rc);
        // k / v are linear in the encoder output, which is itself a LayerNorm output (after_norm): bounded likewise,
        // but its scale and shift live in the encoder weights, so no proof is attempted here (the kernels saturate)
        if ((rc = need(ctx, m, p + ".src_attn.linear_k_v.weight", {2 * d, d}, &t, kWho)) ||
            (rc = up_16(ctx, t->data, 2 * d, d, 2 * d, d, &L.wkv)))
            return fail(rc);
        if ((rc = need(ctx, m, p + ".src_attn.linear_k_v.bias", {2 * d}, &t, kWho)) || (rc = up_f32(ctx, t, &L.bkv))) return fail(rc);
        if ((rc = need(ctx, m, p + ".src_attn.linear_out.weight", {d, d}, &t, kWho)) || (rc = up_16(ctx, t->data, d, d, d, d, &L.wo)))
            return fail(rc);
        if ((rc = need(ctx, m, p + ".src_attn.linear_out.bias", {d}, &t, kWho)) || (rc = up_f32(ctx, t, &L.bo))) return fail(rc);
    }
    if ((rc = ffn("decoder.decoders3.0", &W->w1_last, &W->b1_last, &W->w2_last, &W->b2_last, &W->c2_last))) return fail(rc);
    if ((rc = up_folded(ctx, m, "decoder.output_layer.weight", "decoder.output_layer.bias", "decoder.after_norm.weight",
                        "decoder.after_norm.bias", V, d, W->vocab_pad, false, "", &W->w_out, &W->b_out)))
        return fail(rc);
    ASRFE_HIP_TRY(ctx, hipDeviceSynchronize());
    return ASRFE_OK;
}

// ---- workspace ------------------------------------------------------------------------------------------------------
namespace {
struct DecBuffers {
    uint16_t* a_hi;   // [M][1536]: the middle column block is the 16-bit copy of the encoder output (the decoder's memory)
    uint16_t* a_lo;   // [M][1536]
    float* conv;      // [M][512]
    float* alphas;    // [M]
    uint16_t* kv;     // [M][1024]
    float* x;         // [Md][512] residual stream
    float* t;         // [Md][512] feed-forward output
    uint16_t* h;      // [Md][512]
    uint16_t* f;      // [Md][ff]
    uint16_t* g;      // [Md][ff]
    float* stats;     // [ff / 64][Md][2]: per 64-column strip of f, the sum and the sum of squares
    float* row_norm;  // [Md][2]: mean and 1 / sqrt(var + eps) of f's rows
    uint16_t* q;      // [Md][512]
    uint16_t* att;    // [Md][512]
    float* logits;    // [Md][vocab_pad / 64] (value, column) pairs; [Md][vocab_pad] f32 when Md takes the latency plan
    int64_t bytes;
};

DecBuffers dec_carve(const asrfe_ctx* ctx, void* ws, int B, int Tp, int n_max) {
    const int64_t M = static_cast<int64_t>(B) * Tp, Md = static_cast<int64_t>(B) * n_max;
    const int64_t ff = ctx->dec->cfg.linear_units;
    char* base = static_cast<char*>(ws);
    int64_t off = 0;
    auto take = [&](int64_t nbytes) {
        char* p = base + off;
        off += (nbytes + 255) / 256 * 256;
        return p;
    };
    DecBuffers b;
    b.a_hi = reinterpret_cast<uint16_t*>(take(M * 1536 * 2));
    b.a_lo = reinterpret_cast<uint16_t*>(take(M * 1536 * 2));
    b.conv = reinterpret_cast<float*>(take(M * 512 * 4));
    b.alphas = reinterpret_cast<float*>(take(M * 4));
    b.kv = reinterpret_cast<uint16_t*>(take(M * 1024 * 2));
    b.x = reinterpret_cast<float*>(take(Md * 512 * 4));
    b.t = reinterpret_cast<float*>(take(Md * 512 * 4));
    b.h = reinterpret_cast<uint16_t*>(take(Md * 512 * 2));
    b.f = reinterpret_cast<uint16_t*>(take(Md * ff * 2));
    b.g = reinterpret_cast<uint16_t*>(take(Md * ff * 2));
    b.stats = reinterpret_cast<float*>(take(Md * (ff / 64) * 8));
    b.row_norm = reinterpret_cast<float*>(take(Md * 8));
    b.q = reinterpret_cast<uint16_t*>(take(Md * 512 * 2));
    b.att = reinterpret_cast<uint16_t*>(take(Md * 512 * 2));
    // without a logits buffer of the caller's the vocabulary projection keeps one (value, column) pair per 64-column
    // strip for the arg-max; only calls small enough for the latency plan store whole rows here
    const int64_t small = ctx->cfg.enc_small_batch_rows;
    b.logits = reinterpret_cast<float*>(take(Md > small ? Md * (ctx->dec->vocab_pad / 64) * 8 : Md * ctx->dec->vocab_pad * 4));
    b.bytes = off;
    return b;
}

int32_t check_common(asrfe_ctx* ctx, const char* who, int32_t batch, int32_t tp, int32_t n_max, const void* ws, int64_t ws_bytes) {
    if (!ctx->dec) return set_err(ctx, ASRFE_ERR_NOT_LOADED, std::string(who) + ": decoder weights not loaded");
    if (batch < 0 || batch > 16384 || tp < 0 || tp > 8192 || n_max < 1 || n_max > 8193)
        return set_err(ctx, ASRFE_ERR_SHAPE, std::string(who) + ": batch in 0..16384, tp in 0..8192, n_max in 1..8193");
    if (static_cast<int64_t>(batch) * tp > (1 << 24) || static_cast<int64_t>(batch) * n_max > (1 << 22))
        return set_err(ctx, ASRFE_ERR_SHAPE, std::string(who) + ": batch * tp or batch * n_max too large");
    if (batch == 0 || tp == 0) return ASRFE_OK;
    if (!ws || reinterpret_cast<uintptr_t>(ws) % 256 != 0) return set_err(ctx, ASRFE_ERR_WORKSPACE, std::string(who) + ": workspace must be 256-byte aligned");
    if (ws_bytes < dec_carve(ctx, nullptr, batch, tp, n_max).bytes) return set_err(ctx, ASRFE_ERR_WORKSPACE, std::string(who) + ": workspace too small");
    return ASRFE_OK;
}

int32_t gemm(asrfe_ctx* ctx, const uint16_t* A, int lda, const uint16_t* Wt, int ldw, const float* bias, int M, int N, int K,
             void* out, int ldo, const float* add32, int epi, hipStream_t s) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.bias = bias; g.M = M; g.N = N; g.K = K; g.out = out; g.ldo = ldo;
    g.add32 = add32; g.ld32 = ldo;
    // a handful of rows (one short utterance): the latency plan's product, which spreads the weights over the chip
    // instead of giving one workgroup 256 rows
    const int small = ctx->cfg.enc_small_batch_rows;
    if (M <= small && K % 512 == 0 && K <= 3072 && N % 32 == 0) return launch_skinny_gemm(ctx, g, epi, s);
    return launch_gemm(ctx, g, epi, s);
}
}  // namespace

extern "C" int64_t asrfe_decoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t tp, int32_t n_max) {
    if (!ctx || !ctx->dec || batch < 0 || tp < 0 || n_max < 0) return -1;
    return dec_carve(ctx, nullptr, batch, tp, n_max).bytes;
}

extern "C" int32_t asrfe_decoder_logits_stride(const asrfe_ctx* ctx) { return ctx && ctx->dec ? ctx->dec->vocab_pad : -1; }

extern "C" int32_t asrfe_cif_predictor(asrfe_ctx* ctx, const float* enc_out, const int32_t* lens, int32_t batch, int32_t tp,
                                       int32_t n_max, float* embeds, int32_t* token_num, float* alphas_out, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    int32_t rc = check_common(ctx, "asrfe_cif_predictor", batch, tp, n_max, workspace, workspace_bytes);
    if (rc || batch == 0 || tp == 0) return rc;
    if (!enc_out || !lens || !embeds || !token_num) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_cif_predictor: null argument");
    if (reinterpret_cast<uintptr_t>(enc_out) % 16 != 0 || reinterpret_cast<uintptr_t>(embeds) % 16 != 0)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_cif_predictor: enc_out / embeds must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DecoderWeights& W = *ctx->dec;
    const DecBuffers b = dec_carve(ctx, workspace, batch, tp, n_max);
    const int M = batch * tp;
    if ((rc = launch_pred_split(ctx, enc_out, lens, batch, tp, b.a_hi, b.a_lo, s))) return rc;
    // head x head carries the bias; the two cross terms are added on top (the remainder x remainder term is below 2^-22)
    if ((rc = gemm(ctx, b.a_hi, 1536, W.wc_hi, 1536, W.bc, M, 512, 1536, b.conv, 512, nullptr, EPI_F32_ADD, s))) return rc;
    if ((rc = gemm(ctx, b.a_lo, 1536, W.wc_hi, 1536, W.zero_bias, M, 512, 1536, b.conv, 512, b.conv, EPI_F32_ADD, s))) return rc;
    if ((rc = gemm(ctx, b.a_hi, 1536, W.wc_lo, 1536, W.zero_bias, M, 512, 1536, b.conv, 512, b.conv, EPI_F32_ADD, s))) return rc;
    if ((rc = launch_pred_alpha(ctx, b.conv, W.w_alpha, W.b_alpha, W.cfg.smooth_factor, W.cfg.noise_threshold, lens, batch, tp,
                                b.alphas, s)))
        return rc;
    CifParams c{};
    c.enc = enc_out; c.alphas = b.alphas; c.lens = lens; c.Tp = tp; c.n_max = n_max;
    c.threshold = W.cfg.threshold; c.tail_threshold = W.cfg.tail_threshold;
    c.embeds = embeds; c.token_num = token_num; c.alphas_out = alphas_out; c.n_fires = nullptr;
    return launch_cif(ctx, c, batch, s);
}

extern "C" int32_t asrfe_cif_fire_frames(asrfe_ctx* ctx, const float* alphas, int32_t batch, int32_t tp, int32_t n_max,
                                         int32_t* fire_frames, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!ctx->dec) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_cif_fire_frames: decoder weights not loaded");
    if (batch < 0 || batch > 16384 || tp < 0 || tp > 8192 || n_max < 1 || n_max > 8193)
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_cif_fire_frames: batch in 0..16384, tp in 0..8192, n_max in 1..8193");
    if (batch == 0) return ASRFE_OK;
    if (!alphas || !fire_frames) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_cif_fire_frames: null argument");
    return launch_cif_fire_frames(ctx, alphas, batch, tp, n_max, ctx->dec->cfg.threshold, fire_frames,
                                  static_cast<hipStream_t>(stream));
}

extern "C" int32_t asrfe_decoder_forward(asrfe_ctx* ctx, const float* enc_out, const int32_t* lens, int32_t batch, int32_t tp,
                                         const float* embeds, const int32_t* token_num, int32_t n_max, int32_t* token_ids,
                                         float* logits, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    int32_t rc = check_common(ctx, "asrfe_decoder_forward", batch, tp, n_max, workspace, workspace_bytes);
    if (rc || batch == 0 || tp == 0) return rc;
    if (!enc_out || !lens || !embeds || !token_num || !token_ids) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_decoder_forward: null argument");
    if (reinterpret_cast<uintptr_t>(embeds) % 16 != 0 || reinterpret_cast<uintptr_t>(enc_out) % 16 != 0 ||
        (logits && reinterpret_cast<uintptr_t>(logits) % 16 != 0))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_decoder_forward: enc_out / embeds / logits must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DecoderWeights& W = *ctx->dec;
    const DecBuffers b = dec_carve(ctx, workspace, batch, tp, n_max);
    const int M = batch * tp, Md = batch * n_max;
    const int ff = W.cfg.linear_units;
    const float eps = W.cfg.ln_eps;
    // the cross-attention memory: the 16-bit head of the encoder frames, zero past lens[b], [M][512] with row stride 1536
    // (the middle column block of the predictor's operand).  Rebuilt from enc_out here: the workspace may be another one
    // than the predictor's, or have been used in between.
    if ((rc = launch_pred_split(ctx, enc_out, lens, batch, tp, b.a_hi, b.a_lo, s))) return rc;
    const uint16_t* mem16 = b.a_hi + 512;
    if (hipMemcpyAsync(b.x, embeds, static_cast<size_t>(Md) * 512 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return set_err(ctx, ASRFE_ERR_HIP, "asrfe_decoder_forward: copy of the embeddings failed");

    // Rows for the general GEMM: the hidden LayerNorm is applied behind the second product from row statistics that
    // the first product's epilogue leaves (no pass over the [Md][ff] hidden layer in between).
    const bool norm_after = Md > ctx->cfg.enc_small_batch_rows;
    auto feed_forward = [&](const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* c2) -> int32_t {
        int32_t r;
        if ((r = launch_ln512(ctx, b.x, nullptr, nullptr, eps, Md, b.h, s))) return r;
        if (norm_after) {
            GemmParams g1{};
            g1.A = b.h; g1.lda = 512; g1.W = w1; g1.ldw = 512; g1.bias = b1; g1.M = Md; g1.N = ff; g1.K = 512;
            g1.out = b.f; g1.ldo = ff; g1.stat_out = b.stats;
            if ((r = launch_gemm(ctx, g1, EPI_RELU_STAT, s))) return r;
            if ((r = launch_row_stats_finish(ctx, b.stats, ff / 64, Md, eps, b.row_norm, s))) return r;
            GemmParams g2{};
            g2.A = b.f; g2.lda = ff; g2.W = w2; g2.ldw = ff; g2.bias = W.zero_bias; g2.M = Md; g2.N = 512; g2.K = ff;
            g2.out = b.t; g2.ldo = 512; g2.ld32 = 512;
            g2.stat_in = b.row_norm; g2.colsum = c2; g2.post_bias = b2;
            return launch_gemm(ctx, g2, EPI_F32_NORM, s);
        }
        if ((r = gemm(ctx, b.h, 512, w1, 512, b1, Md, ff, 512, b.f, ff, nullptr, EPI_BF16_RELU, s))) return r;
        if ((r = launch_ln_hidden(ctx, b.f, b.g, Md, ff, eps, s))) return r;
        return gemm(ctx, b.g, ff, w2, ff, b2, Md, 512, ff, b.t, 512, nullptr, EPI_F32_ADD, s);
    };
    for (const DecLayerWeights& L : W.layers) {
        if ((rc = feed_forward(L.w1, L.b1, L.w2, L.b2, L.c2))) return rc;
        DecFsmnParams fp{};
        fp.t = b.t; fp.x = b.x; fp.gamma = L.n2_g; fp.beta = L.n2_b; fp.taps = L.taps; fp.token_num = token_num;
        fp.N = n_max; fp.eps = eps;
        fp.h = b.h;  // the LayerNorm in front of the query projection leaves the same kernel
        if ((rc = launch_dec_fsmn(ctx, fp, batch, s))) return rc;
        if ((rc = gemm(ctx, b.h, 512, L.wq, 512, L.bq, Md, 512, 512, b.q, 512, nullptr, EPI_BF16, s))) return rc;
        if ((rc = gemm(ctx, mem16, 1536, L.wkv, 512, L.bkv, M, 1024, 512, b.kv, 1024, nullptr, EPI_BF16, s))) return rc;
        AttnParams a{};
        a.q = b.q; a.ld_q = 512; a.Tq = n_max;
        a.k = b.kv; a.v = b.kv + 512; a.ld_kv = 1024; a.Tk = tp;
        a.lens = lens; a.out = b.att; a.ld_out = 512; a.B = batch;
        a.scale_log2 = 1.4426950408889634f / std::sqrt(128.0f);
        if ((rc = launch_attention(ctx, a, s))) return rc;
        if ((rc = gemm(ctx, b.att, 512, L.wo, 512, L.bo, Md, 512, 512, b.x, 512, b.x, EPI_F32_ADD, s))) return rc;
    }
    // closing block: feed-forward without residual, after_norm (folded into the output layer), vocabulary projection
    if ((rc = feed_forward(W.w1_last, W.b1_last, W.w2_last, W.b2_last, W.c2_last))) return rc;
    if ((rc = launch_ln512(ctx, b.t, nullptr, nullptr, eps, Md, b.h, s))) return rc;
    // The caller does not ask for the logits and the rows take the general GEMM: its epilogue keeps the largest value
    // of every 64-column strip with its column instead of storing 554 MB (at 64 x 256 rows) for the arg-max to read back.
    // The values compared are the ones the stored logits would hold, so the ids are the same.
    if (!logits && Md > ctx->cfg.enc_small_batch_rows) {
        GemmParams g{};
        g.A = b.h; g.lda = 512; g.W = W.w_out; g.ldw = 512; g.bias = W.b_out; g.M = Md; g.N = W.vocab_pad; g.K = 512;
        g.out = b.logits; g.ldo = W.vocab_pad / 64; g.n_valid = W.cfg.vocab_size;
        if ((rc = launch_gemm(ctx, g, EPI_ARGMAX, s))) return rc;
        return launch_argmax_pairs(ctx, b.logits, W.vocab_pad / 64, batch, n_max, token_num, token_ids, s);
    }
    float* lg = logits ? logits : b.logits;
    if ((rc = gemm(ctx, b.h, 512, W.w_out, 512, W.b_out, Md, W.vocab_pad, 512, lg, W.vocab_pad, nullptr, EPI_F32_ADD, s))) return rc;
    return launch_argmax(ctx, lg, W.vocab_pad, W.cfg.vocab_size, batch, n_max, token_num, token_ids, s);
}
