// HiFi-GAN generator: weight loading (convolution kernels re-laid out as implicit-GEMM matrices), workspace and the
// launch sequence.
//
//   x = conv_pre(mel);  per stage i: a = lrelu(x);  x = ConvTranspose(a) [u phases];  for each residual block j:
//   three times { t = lrelu(conv_dilated(lrelu(x_j))); x_j = conv(t) + x_j };  x = mean_j x_j;  wav = tanh(conv_post(lrelu(x))).
// Activations are time-major with contiguous channels: the residual stream in f32, what a convolution reads in the
// context's 16-bit type with the leaky ReLU that precedes every convolution already applied by the producer.
#include "vocoder.h"

#include <cstdio>
#include <cstdlib>

#include "weights.h"

using namespace asrfe;

namespace asrfe {
void free_vocoder(asrfe_ctx* ctx) {
    delete ctx->voc;
    ctx->voc = nullptr;
}
}  // namespace asrfe

extern "C" int32_t asrfe_default_vocoder_config(asrfe_vocoder_config* c) {
    if (!c) return ASRFE_ERR_BAD_ARG;
    std::memset(c, 0, sizeof(*c));
    c->n_mels = 80;
    c->initial_channel = 512;
    c->n_upsamples = 4;
    const int rates[4] = {8, 8, 2, 2}, kernels[4] = {16, 16, 4, 4};
    for (int i = 0; i < 4; ++i) {
        c->upsample_rates[i] = rates[i];
        c->upsample_kernels[i] = kernels[i];
    }
    c->n_resblocks = 3;
    const int rk[3] = {3, 7, 11};
    for (int j = 0; j < 3; ++j) {
        c->resblock_kernels[j] = rk[j];
        c->resblock_dilations[j][0] = 1;
        c->resblock_dilations[j][1] = 3;
        c->resblock_dilations[j][2] = 5;
    }
    c->lrelu_slope = 0.1f;
    c->final_slope = 0.01f;
    return ASRFE_OK;
}

static const char* kVocWho = "asrfe_load_vocoder_weights";

// Conv1d weight [Co][Ci][k] -> [Co][k * Ci] (k index = tap * Ci + ci), 16-bit
static int32_t up_conv(asrfe_ctx* ctx, const TensorMap& m, const std::string& name, int64_t co, int64_t ci, int64_t k, int dil,
                       VocConv* out) {
    const HostTensor *tw, *tb;
    int32_t rc;
    if ((rc = need(ctx, m, name + ".weight", {co, ci, k}, &tw, kVocWho)) || (rc = need(ctx, m, name + ".bias", {co}, &tb, kVocWho)))
        return rc;
    std::vector<float> w(static_cast<size_t>(co * k * ci));
    for (int64_t o = 0; o < co; ++o)
        for (int64_t c = 0; c < ci; ++c)
            for (int64_t j = 0; j < k; ++j) w[(o * k + j) * ci + c] = tw->data[(o * ci + c) * k + j];
    if ((rc = up_16(ctx, w.data(), co, k * ci, co, k * ci, &out->w)) || (rc = up_f32(ctx, tb, &out->bias))) return rc;
    out->Ci = static_cast<int>(ci); out->Co = static_cast<int>(co); out->k = static_cast<int>(k); out->dil = dil;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_load_vocoder_weights(asrfe_ctx* ctx, const asrfe_vocoder_config* cfg, const asrfe_tensor_desc* descs,
                                              int32_t n) {
    if (!ctx || !cfg || !descs || n <= 0) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: null argument");
    if (ctx->voc) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: weights already loaded");
    const asrfe_vocoder_config& c = *cfg;
    if (c.n_mels < 16 || c.n_mels % 16 != 0 || c.n_upsamples < 1 || c.n_upsamples > 8 || c.n_resblocks < 1 || c.n_resblocks > 4)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: n_mels a multiple of 16, 1..8 upsampling stages, 1..4 residual blocks per stage");
    int64_t ch = c.initial_channel;
    int hop = 1;
    for (int i = 0; i < c.n_upsamples; ++i) {
        const int u = c.upsample_rates[i], k = c.upsample_kernels[i];
        if (u < 1 || k < u || k % u != 0 || (k - u) % 2 != 0)
            return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: an upsampling kernel must be a multiple of its stride (k - stride even)");
        if (ch % 64 != 0) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: every stage's channel count must be a multiple of 32");
        ch /= 2;
        hop *= u;
    }
    for (int j = 0; j < c.n_resblocks; ++j)
        if (c.resblock_kernels[j] < 1 || c.resblock_kernels[j] % 2 == 0)
            return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: residual-block kernels must be odd");
    TensorMap m;
    int32_t rc;
    if ((rc = tensor_map(ctx, descs, n, "", kVocWho, m))) return rc;
    if ((rc = prepare_conv(ctx))) return rc;
    VocoderWeights* W = new VocoderWeights();
    W->cfg = c;
    W->hop = hop;
    ctx->voc = W;
    auto fail = [&](int32_t code) {
        free_vocoder(ctx);
        return code;
    };
    if ((rc = up_conv(ctx, m, "conv_pre", c.initial_channel, c.n_mels, 7, 1, &W->pre))) return fail(rc);
    ch = c.initial_channel;
    W->stages.resize(c.n_upsamples);
    for (int i = 0; i < c.n_upsamples; ++i) {
        VocoderWeights::Stage& S = W->stages[i];
        const int64_t ci = ch, co = ch / 2, k = c.upsample_kernels[i], u = c.upsample_rates[i], tp = k / u;
        const HostTensor *tw, *tb;
        const std::string un = "ups." + std::to_string(i);
        if ((rc = need(ctx, m, un + ".weight", {ci, co, k}, &tw, kVocWho)) || (rc = need(ctx, m, un + ".bias", {co}, &tb, kVocWho)))
            return fail(rc);
        // ConvTranspose1d weight [Ci][Co][k] -> u phase matrices [Co][(k / u) * Ci]: phase r, tap j is kernel index r + u j
        std::vector<float> w(static_cast<size_t>(u * co * tp * ci));
        for (int64_t r = 0; r < u; ++r)
            for (int64_t o = 0; o < co; ++o)
                for (int64_t j = 0; j < tp; ++j)
                    for (int64_t cc = 0; cc < ci; ++cc)
                        w[((r * co + o) * tp + j) * ci + cc] = tw->data[(cc * co + o) * k + r + u * j];
        if ((rc = up_16(ctx, w.data(), u * co, tp * ci, u * co, tp * ci, &S.up_w)) || (rc = up_f32(ctx, tb, &S.up_b))) return fail(rc);
        S.Ci = static_cast<int>(ci); S.Co = static_cast<int>(co); S.k = static_cast<int>(k); S.u = static_cast<int>(u);
        S.c1.resize(c.n_resblocks * 3);
        S.c2.resize(c.n_resblocks * 3);
        for (int j = 0; j < c.n_resblocks; ++j)
            for (int q = 0; q < 3; ++q) {
                const std::string p = "resblocks." + std::to_string(i * c.n_resblocks + j);
                const int d = c.resblock_dilations[j][q];
                if (d < 1) return fail(set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_vocoder_weights: dilations must be >= 1"));
                if ((rc = up_conv(ctx, m, p + ".convs1." + std::to_string(q), co, co, c.resblock_kernels[j], d, &S.c1[j * 3 + q])) ||
                    (rc = up_conv(ctx, m, p + ".convs2." + std::to_string(q), co, co, c.resblock_kernels[j], 1, &S.c2[j * 3 + q])))
                    return fail(rc);
            }
        ch = co;
    }
    {
        const HostTensor *tw, *tb;
        if ((rc = need(ctx, m, "conv_post.weight", {1, ch, 7}, &tw, kVocWho)) || (rc = need(ctx, m, "conv_post.bias", {1}, &tb, kVocWho)))
            return fail(rc);
        std::vector<float> w(static_cast<size_t>(7 * ch));
        for (int64_t cc = 0; cc < ch; ++cc)
            for (int64_t j = 0; j < 7; ++j) w[j * ch + cc] = tw->data[cc * 7 + j];
        if ((rc = upload(ctx, w, &W->post_w)) || (rc = up_f32(ctx, tb, &W->post_b))) return fail(rc);
        W->c_last = static_cast<int>(ch);
    }
    ASRFE_HIP_TRY(ctx, hipDeviceSynchronize());
    return ASRFE_OK;
}

namespace {
struct VocBuffers {
    uint16_t* mel16;  // [B][T][n_mels]
    uint16_t* a[4];   // 16-bit activations: stage input, convolution scratch, ping, pong
    void* r[3];       // residual stream (f32, or the 16-bit operand type: voc_res16): stage's, ping, pong
    void* out[4];     // the same type: the output of each of a stage's residual blocks
    int64_t bytes;
};
// fp16 contexts keep the generator's residual stream in fp16: every consumer of it rounds to fp16 anyway (a convolution's
// operand) or adds at most three convolution outputs to it before the next rounding; ASRFE_VOC_RES32=1 keeps f32
bool voc_res16(const asrfe_ctx* ctx) {
    static const bool off = [] { const char* e = getenv("ASRFE_VOC_RES32"); return e && atoi(e) != 0; }();
    return ctx->cfg.enc_compute_dtype == ASRFE_DT_F16 && !off;
}
VocBuffers voc_carve(const asrfe_ctx* ctx, void* ws, int B, int T) {
    const VocoderWeights& W = *ctx->voc;
    int64_t worst = static_cast<int64_t>(B) * T * W.cfg.initial_channel, len = T, ch = W.cfg.initial_channel;
    for (const auto& S : W.stages) {
        len *= S.u;
        ch = S.Co;
        worst = std::max<int64_t>(worst, static_cast<int64_t>(B) * len * ch);
    }
    char* base = static_cast<char*>(ws);
    int64_t off = 0;
    auto take = [&](int64_t nbytes) {
        char* p = base + off;
        off += (nbytes + 255) / 256 * 256;
        return p;
    };
    VocBuffers b;
    b.mel16 = reinterpret_cast<uint16_t*>(take(static_cast<int64_t>(B) * T * W.cfg.n_mels * 2));
    for (int i = 0; i < 4; ++i) b.a[i] = reinterpret_cast<uint16_t*>(take(worst * 2));
    const int64_t rs = voc_res16(ctx) ? 2 : 4;
    for (int i = 0; i < 3; ++i) b.r[i] = take(worst * rs);
    for (int i = 0; i < 4; ++i) b.out[i] = i < W.cfg.n_resblocks ? take(worst * rs) : nullptr;
    b.bytes = off;
    return b;
}
}  // namespace

extern "C" int64_t asrfe_vocoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t t_mel) {
    if (!ctx || !ctx->voc || batch < 0 || t_mel < 0) return -1;
    return voc_carve(ctx, nullptr, batch, t_mel).bytes;
}

extern "C" int32_t asrfe_vocoder_hop(const asrfe_ctx* ctx) { return ctx && ctx->voc ? ctx->voc->hop : -1; }

extern "C" int32_t asrfe_vocoder_forward(asrfe_ctx* ctx, const float* mel, int32_t batch, int32_t t_mel, float* wav, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!ctx->voc) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_vocoder_forward: vocoder weights not loaded");
    if (batch < 0 || batch > 4096 || t_mel < 0 || t_mel > (1 << 20)) return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_vocoder_forward: batch in 0..4096, t_mel in 0..2^20");
    if (batch == 0 || t_mel == 0) return ASRFE_OK;
    const VocoderWeights& W = *ctx->voc;
    if (static_cast<int64_t>(batch) * t_mel * W.hop * 32 >= (int64_t{1} << 40)) return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_vocoder_forward: too many samples");
    if (!mel || !wav || !workspace) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_vocoder_forward: null argument");
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return set_err(ctx, ASRFE_ERR_WORKSPACE, "asrfe_vocoder_forward: workspace must be 256-byte aligned");
    if (workspace_bytes < voc_carve(ctx, nullptr, batch, t_mel).bytes) return set_err(ctx, ASRFE_ERR_WORKSPACE, "asrfe_vocoder_forward: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const VocBuffers b = voc_carve(ctx, workspace, batch, t_mel);
    const float slope = W.cfg.lrelu_slope;
    int32_t rc;
    if ((rc = launch_mel_to_ntc(ctx, mel, batch, W.cfg.n_mels, t_mel, b.mel16, s))) return rc;

    const int res16 = voc_res16(ctx) ? 1 : 0;
    auto conv = [&](const VocConv& cv, const uint16_t* x, int T, float act_slope, uint16_t* y16, const void* res_in,
                    void* res_out, int act_in = 0) -> int32_t {
        ConvParams p{};
        p.res16 = res16; p.act_in = act_in;
        p.x = x; p.w = cv.w; p.bias = cv.bias; p.B = batch; p.Tin = T; p.Ci = cv.Ci; p.Co = cv.Co; p.taps = cv.k;
        p.off0 = -((cv.k - 1) / 2) * cv.dil; p.dtap = cv.dil; p.phases = 1; p.q_count = T;
        p.out_mul = 1; p.out_add0 = 0; p.out_add_step = 0; p.Tout = T;
        p.slope = act_slope; p.y16 = y16; p.res_in = res_in; p.res_out = res_out;
        return launch_conv_mfma(ctx, p, s);
    };
    // conv_pre: only its activated output is needed (the first thing a stage does is lrelu)
    if ((rc = conv(W.pre, b.mel16, t_mel, slope, b.a[0], nullptr, nullptr))) return rc;
    int T = t_mel;
    const int nrb = W.cfg.n_resblocks;
    for (size_t i = 0; i < W.stages.size(); ++i) {
        const VocoderWeights::Stage& S = W.stages[i];
        const int Tout = T * S.u;
        // Stages with few channels run every residual pair as one launch on the f32 stream (no 16-bit copies between
        // the convolutions); ASRFE_VOC_NO_FUSE=1 keeps the two-launch form everywhere (the tests compare the two)
        const char* nf = getenv("ASRFE_VOC_NO_FUSE");  // (read per call: a test switches it inside one process)
        bool fused = !(nf && atoi(nf) != 0);
        for (int j = 0; j < nrb * 3 && fused; ++j) {
            const VocConv &a = S.c1[j], &c = S.c2[j];
            fused = a.Ci == S.Co && a.Co == S.Co && c.Ci == S.Co && c.Co == S.Co && a.k == c.k && c.dil == 1 &&
                    conv_pair_supported(S.Co, a.k, a.dil);
        }
        {   // transposed convolution: u phases of k / u taps; raw result to the residual stream, activated copy beside it
            // (nobody reads the copy in a fused stage)
            ConvParams p{};
            p.x = b.a[0]; p.w = S.up_w; p.bias = S.up_b; p.B = batch; p.Tin = T; p.Ci = S.Ci; p.Co = S.Co; p.taps = S.k / S.u;
            p.off0 = 0; p.dtap = -1; p.phases = S.u; p.q_count = T + 1;
            p.out_mul = S.u; p.out_add0 = -((S.k - S.u) / 2); p.out_add_step = 1; p.Tout = Tout;
            p.slope = slope; p.y16 = (fused || res16) ? nullptr : b.a[1]; p.res_out = b.r[0]; p.res16 = res16;
            if ((rc = launch_conv_mfma(ctx, p, s))) return rc;
        }
        T = Tout;
        for (int j = 0; j < nrb; ++j) {
            const void* xr = b.r[0];
            const uint16_t* xa = b.a[1];
            for (int q = 0; q < 3; ++q) {
                const bool last = q == 2;  // the block's output goes to its own buffer; nothing reads its activated copy
                void* ro = last ? b.out[j] : b.r[1 + (q & 1)];
                if (fused) {
                    if ((rc = launch_conv_pair(ctx, xr, ro, res16, S.c1[j * 3 + q], S.c2[j * 3 + q], batch, T, slope, s))) return rc;
                    xr = ro;
                    continue;
                }
                if (res16) {
                    // the 16-bit stream is its own operand: the first convolution activates it while staging, the second
                    // adds it and stores the raw sum in the same type (10 bytes per element and pair instead of 16)
                    if ((rc = conv(S.c1[j * 3 + q], static_cast<const uint16_t*>(xr), T, slope, b.a[0], nullptr, nullptr, 1))) return rc;
                    if ((rc = conv(S.c2[j * 3 + q], b.a[0], T, slope, nullptr, xr, ro))) return rc;
                    xr = ro;
                    continue;
                }
                if ((rc = conv(S.c1[j * 3 + q], xa, T, slope, b.a[0], nullptr, nullptr))) return rc;
                uint16_t* ao = last ? nullptr : b.a[2 + (q & 1)];
                if ((rc = conv(S.c2[j * 3 + q], b.a[0], T, slope, ao, xr, ro))) return rc;
                xr = ro;
                xa = ao;
            }
        }
        // mean over the residual blocks, then the leaky ReLU in front of the next transposed convolution (or, with its
        // own slope, in front of the output convolution)
        const bool final_stage = i + 1 == W.stages.size();
        if ((rc = launch_scale_act(ctx, b.out, res16, nrb, 1.0f / static_cast<float>(nrb), final_stage ? W.cfg.final_slope : slope,
                                   static_cast<int64_t>(batch) * T * S.Co, b.a[0], s)))
            return rc;
    }
    return launch_conv_post(ctx, b.a[0], W.post_w, W.post_b, batch, T, W.c_last, wav, s);
}
