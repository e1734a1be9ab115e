// FSMN voice-activity network behind the C ABI: weight loading, workspace layout and the forward pass.
// Stands in for the network inside vad_model.generate(input=audio_np, cache=..., is_final=False, chunk_size=...)
// (/root/reference/voice-service/app/services/voice_interface.py:1585-1590): FunASR's FSMN encoder of the model
// speech_fsmn_vad_zh-cn-16k-common-pytorch (in_linear1, in_linear2, ReLU, fsmn_layers x [linear, fsmn_block, affine,
// ReLU], out_linear1, out_linear2, softmax).  Every product runs on the encoder's MFMA GEMM with all widths padded to its
// tile sizes (zeros in the padding), the memory block and the softmax in kernels of their own (vad_kernels.hip).
#include "vad.h"

#include "gemm.h"
#include "weights.h"

using namespace asrfe;

namespace {
constexpr const char* kWho = "asrfe_load_vad_weights";
int round_up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace

void asrfe::free_vad(asrfe_ctx* ctx) {
    delete ctx->vad;
    ctx->vad = nullptr;
}

extern "C" int32_t asrfe_default_vad_config(asrfe_vad_config* cfg) {
    if (!cfg) return ASRFE_ERR_BAD_ARG;
    cfg->input_dim = 400;
    cfg->input_affine_dim = 140;
    cfg->fsmn_layers = 4;
    cfg->linear_dim = 250;
    cfg->proj_dim = 128;
    cfg->lorder = 20;
    cfg->output_affine_dim = 140;
    cfg->output_dim = 248;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_load_vad_weights(asrfe_ctx* ctx, const asrfe_vad_config* cfg, const asrfe_tensor_desc* descs, int32_t n) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!cfg || !descs || n <= 0) return set_err(ctx, ASRFE_ERR_BAD_ARG, std::string(kWho) + ": null argument");
    if (ctx->vad) return set_err(ctx, ASRFE_ERR_BAD_ARG, std::string(kWho) + ": weights are already loaded in this context");
    const asrfe_vad_config& c = *cfg;
    const int widest = std::max(std::max(c.input_affine_dim, c.linear_dim), std::max(c.output_affine_dim, c.output_dim));
    if (c.input_dim < 8 || c.input_dim > 4096 || c.input_dim % 8 != 0 || c.input_affine_dim < 1 || c.linear_dim < 1 ||
        c.output_affine_dim < 1 || c.output_dim < 1 || widest > 1024 || c.proj_dim < 1 || c.proj_dim > 256 ||
        c.fsmn_layers < 1 || c.fsmn_layers > 64 || c.lorder < 1 || c.lorder > 33 || (c.lorder - 1) * c.proj_dim > 8192)
        return set_err(ctx, ASRFE_ERR_SHAPE, std::string(kWho) + ": input_dim a multiple of 8 up to 4096, widths up to 1024, "
                                             "proj_dim up to 256, lorder up to 33, 1..64 layers");
    TensorMap m;
    int32_t rc = tensor_map(ctx, descs, n, "", kWho, m);
    if (rc) return rc;
    VadWeights* W = new VadWeights();
    W->cfg = c;
    W->in_pad = round_up(c.input_dim, 128);
    W->w_pad = round_up(widest, 256);
    W->proj_pad = round_up(c.proj_dim, 128);
    auto fail = [&](int32_t r) {
        delete W;
        return r;
    };
    const HostTensor* t = nullptr;
    // a linear layer [rows][k] -> 16-bit [w_pad][k_pad] with its bias [w_pad] (zeros where the layer has none)
    auto linear = [&](const std::string& name, int rows, int k, int k_pad, bool has_bias, uint16_t** w, float** b) -> int32_t {
        int32_t r;
        if ((r = need(ctx, m, name + ".weight", {rows, k}, &t, kWho)) || (r = up_16(ctx, t->data, rows, k, W->w_pad, k_pad, w))) return r;
        if (!has_bias) return ASRFE_OK;
        if ((r = need(ctx, m, name + ".bias", {rows}, &t, kWho))) return r;
        std::vector<float> v(static_cast<size_t>(W->w_pad), 0.0f);
        std::copy(t->data, t->data + rows, v.begin());
        return upload(ctx, v, b);
    };
    if ((rc = linear("encoder.in_linear1.linear", c.input_affine_dim, c.input_dim, W->in_pad, true, &W->w_in1, &W->b_in1))) return fail(rc);
    if ((rc = linear("encoder.in_linear2.linear", c.linear_dim, c.input_affine_dim, W->w_pad, true, &W->w_in2, &W->b_in2))) return fail(rc);
    W->blocks.resize(static_cast<size_t>(c.fsmn_layers));
    for (int i = 0; i < c.fsmn_layers; ++i) {
        VadWeights::Block& B = W->blocks[static_cast<size_t>(i)];
        const std::string p = "encoder.fsmn." + std::to_string(i);
        if ((rc = linear(p + ".linear.linear", c.proj_dim, c.linear_dim, W->w_pad, false, &B.w_lin, nullptr))) return fail(rc);
        if ((rc = need(ctx, m, p + ".fsmn_block.conv_left.weight", {c.proj_dim, 1, c.lorder, 1}, &t, kWho))) return fail(rc);
        std::vector<float> taps(static_cast<size_t>(c.lorder) * c.proj_dim);
        for (int ch = 0; ch < c.proj_dim; ++ch)
            for (int j = 0; j < c.lorder; ++j) taps[static_cast<size_t>(j) * c.proj_dim + ch] = t->data[ch * c.lorder + j];
        if ((rc = upload(ctx, taps, &B.taps))) return fail(rc);
        if ((rc = linear(p + ".affine.linear", c.linear_dim, c.proj_dim, W->proj_pad, true, &B.w_aff, &B.b_aff))) return fail(rc);
    }
    if ((rc = linear("encoder.out_linear1.linear", c.output_affine_dim, c.linear_dim, W->w_pad, true, &W->w_out1, &W->b_out1))) return fail(rc);
    if ((rc = linear("encoder.out_linear2.linear", c.output_dim, c.output_affine_dim, W->w_pad, true, &W->w_out2, &W->b_out2))) return fail(rc);
    {
        std::vector<float> z(static_cast<size_t>(W->w_pad), 0.0f);
        if ((rc = upload(ctx, z, &W->zero_bias))) return fail(rc);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(set_err(ctx, ASRFE_ERR_HIP, std::string(kWho) + ": " + hipGetErrorString(e)));
    ctx->vad = W;
    return ASRFE_OK;
}

// ---- workspace ------------------------------------------------------------------------------------------------------
namespace {
struct VadBuffers {
    uint16_t* a0;    // [M][in_pad]
    uint16_t* h0;    // [M][w_pad]
    uint16_t* h1;    // [M][w_pad]
    uint16_t* p;     // [M][w_pad] projection of a block (proj_dim valid columns)
    uint16_t* q;     // [M][proj_pad] memory-block output
    float* logits;   // [M][w_pad]
    int64_t bytes;
};

VadBuffers vad_carve(const asrfe_ctx* ctx, void* ws, int64_t M) {
    const VadWeights& W = *ctx->vad;
    char* base = static_cast<char*>(ws);
    int64_t off = 0;
    auto take = [&](int64_t nbytes) {
        char* p = base + off;
        off += (nbytes + 255) / 256 * 256;
        return p;
    };
    VadBuffers b;
    b.a0 = reinterpret_cast<uint16_t*>(take(M * W.in_pad * 2));
    b.h0 = reinterpret_cast<uint16_t*>(take(M * W.w_pad * 2));
    b.h1 = reinterpret_cast<uint16_t*>(take(M * W.w_pad * 2));
    b.p = reinterpret_cast<uint16_t*>(take(M * W.w_pad * 2));
    b.q = reinterpret_cast<uint16_t*>(take(M * W.proj_pad * 2));
    b.logits = reinterpret_cast<float*>(take(M * W.w_pad * 4));
    b.bytes = off;
    return b;
}

int32_t gemm(asrfe_ctx* ctx, const uint16_t* A, int lda, const uint16_t* Wt, int ldw, const float* bias, int M, int N, int K, void* out,
             int ldo, int epi, hipStream_t s) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.bias = bias; g.M = M; g.N = N; g.K = K; g.out = out; g.ldo = ldo;
    return launch_gemm(ctx, g, epi, s);
}
}  // namespace

extern "C" int64_t asrfe_vad_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t t) {
    if (!ctx || !ctx->vad || batch < 0 || t < 0) return -1;
    return vad_carve(ctx, nullptr, static_cast<int64_t>(batch) * t).bytes;
}

extern "C" int64_t asrfe_vad_cache_floats(const asrfe_ctx* ctx) {
    if (!ctx || !ctx->vad) return -1;
    const asrfe_vad_config& c = ctx->vad->cfg;
    return static_cast<int64_t>(c.fsmn_layers) * (c.lorder - 1) * c.proj_dim;
}

extern "C" int32_t asrfe_vad_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch, int32_t t, float* cache,
                                     float* scores, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!ctx->vad) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_vad_forward: weights not loaded");
    if (batch < 0 || batch > 65535 || t < 0 || static_cast<int64_t>(batch) * t > (1 << 24))
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_vad_forward: batch in 0..65535, batch * t up to 2^24 rows");
    if (batch == 0 || t == 0) return ASRFE_OK;
    if (!feats || !scores) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_vad_forward: null argument");
    if (reinterpret_cast<uintptr_t>(feats) % 16 != 0) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_vad_forward: feats must be 16-byte aligned");
    if (!workspace || reinterpret_cast<uintptr_t>(workspace) % 256 != 0)
        return set_err(ctx, ASRFE_ERR_WORKSPACE, "asrfe_vad_forward: workspace must be 256-byte aligned");
    const int M = batch * t;
    if (workspace_bytes < vad_carve(ctx, nullptr, M).bytes) return set_err(ctx, ASRFE_ERR_WORKSPACE, "asrfe_vad_forward: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const VadWeights& W = *ctx->vad;
    const asrfe_vad_config& c = W.cfg;
    const VadBuffers b = vad_carve(ctx, workspace, M);
    const int wp = W.w_pad, hist = c.lorder - 1;
    const int64_t cache_stride = static_cast<int64_t>(c.fsmn_layers) * hist * c.proj_dim;
    int32_t rc;
    if ((rc = launch_vad_cast(ctx, feats, M, c.input_dim, b.a0, W.in_pad, s))) return rc;
    if ((rc = gemm(ctx, b.a0, W.in_pad, W.w_in1, W.in_pad, W.b_in1, M, wp, W.in_pad, b.h0, wp, EPI_BF16, s))) return rc;
    if ((rc = gemm(ctx, b.h0, wp, W.w_in2, wp, W.b_in2, M, wp, wp, b.h1, wp, EPI_BF16_RELU, s))) return rc;
    for (int i = 0; i < c.fsmn_layers; ++i) {
        const VadWeights::Block& L = W.blocks[static_cast<size_t>(i)];
        if ((rc = gemm(ctx, b.h1, wp, L.w_lin, wp, W.zero_bias, M, wp, wp, b.p, wp, EPI_BF16, s))) return rc;
        float* layer_cache = cache ? cache + static_cast<int64_t>(i) * hist * c.proj_dim : nullptr;
        VadFsmnParams f{};
        f.p = b.p; f.ldp = wp; f.cache = layer_cache; f.cache_stride = cache_stride; f.taps = L.taps;
        f.T = t; f.C = c.proj_dim; f.lorder = c.lorder; f.q = b.q; f.ldq = W.proj_pad;
        if ((rc = launch_vad_fsmn(ctx, f, batch, s))) return rc;
        if (layer_cache && (rc = launch_vad_cache_update(ctx, b.p, wp, lens, batch, t, c.proj_dim, hist, layer_cache, cache_stride, s)))
            return rc;
        if ((rc = gemm(ctx, b.q, W.proj_pad, L.w_aff, W.proj_pad, L.b_aff, M, wp, W.proj_pad, b.h1, wp, EPI_BF16_RELU, s))) return rc;
    }
    if ((rc = gemm(ctx, b.h1, wp, W.w_out1, wp, W.b_out1, M, wp, wp, b.h0, wp, EPI_BF16, s))) return rc;
    if ((rc = gemm(ctx, b.h0, wp, W.w_out2, wp, W.b_out2, M, wp, wp, b.logits, wp, EPI_F32_ADD, s))) return rc;
    return launch_vad_softmax(ctx, b.logits, wp, M, c.output_dim, scores, s);
}

extern "C" int32_t asrfe_frame_decibel(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride, const int32_t* pcm_lens,
                                       int32_t batch, int32_t t_max, float* decibel, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (batch < 0 || batch > 65535 || t_max < 0) return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_frame_decibel: batch in 0..65535, t_max >= 0");
    if (batch == 0 || t_max == 0) return ASRFE_OK;
    if (!pcm || !pcm_lens || !decibel) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_frame_decibel: null argument");
    if (ctx->cfg.frame_length != kFrameLen || ctx->cfg.frame_shift != kFrameShift)
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_frame_decibel: frames of 400 samples at a hop of 160 only");
    return launch_frame_decibel(ctx, pcm, pcm_dtype, pcm_stride, pcm_lens, batch, t_max, decibel, static_cast<hipStream_t>(stream));
}
