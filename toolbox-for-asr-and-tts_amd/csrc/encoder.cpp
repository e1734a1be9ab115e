// SAN-M encoder forward: carves the caller's workspace and enqueues the per-layer kernels.
//
// Default plan, two launches per layer (d = 512; 16-bit = the context's compute dtype, fp16 unless bf16 is chosen):
//     layer 0 only:  h = n(feats sqrt(512) + PE)            ln kernel       16-bit [M][640]
//                    qkv = h Wqkv^T + b                      MFMA GEMM       16-bit [M][1536]
//     every layer:   att = softmax(q k^T / sqrt(128)) v,  fsmn = FSMN(v)     attention kernel   16-bit [M][512] each
//                    x = att Wo^T + b + fsmn (+ x);  x += W2 relu(W1 n(x) + b1) + b2;  qkv of the NEXT layer
//                                                             post-attention kernel   f32 [M][512], 16-bit [M][1536]
//                    (x stays in that kernel's register order between its launches; the last layer writes it row-major)
//     at the end:    out = after_norm(x), rows past lens zero  ln kernel
// ASRFE_SEPARATE_QKV=1 / ASRFE_UNFUSED_FFN=1 (read in asrfe_create) select the plans with more launches per layer
// (a projection launch per layer; LayerNorm + four GEMM launches).  The residual stream x stays fp32; MFMA operands
// are 16-bit with fp32 accumulation.
#include <cmath>

#include "common.h"
#include "encoder.h"
#include "gemm.h"

namespace asrfe {

namespace {
struct Carve {
    char* base;
    int64_t off = 0;
    template <typename T>
    T* take(int64_t n) {
        T* p = reinterpret_cast<T*>(base + off);
        off += (n * static_cast<int64_t>(sizeof(T)) + 255) / 256 * 256;
        return p;
    }
};

struct Buffers {
    float* x;        // [M][512]
    uint16_t* h;     // [M][max(kin_pad, 512)]
    uint16_t* qkv;   // [M][1536]
    uint16_t* fsmn;  // [M][512]
    uint16_t* att;   // [M][512]
    uint16_t* ffn;   // [M][ff]
    float* pe;       // [Tp][din]
    int64_t bytes;
};

Buffers carve(const asrfe_ctx* ctx, void* ws, int B, int Tp) {
    const int64_t M = static_cast<int64_t>(B) * Tp;
    const int64_t hk = ctx->kin_pad > 512 ? ctx->kin_pad : 512;
    Carve c{static_cast<char*>(ws)};
    Buffers b;
    b.x = c.take<float>((M + 127) / 128 * 128 * 512);  // whole 128-row tiles: see FfnParams::tiled
    b.h = c.take<uint16_t>(M * hk);
    b.qkv = c.take<uint16_t>(M * 1536);
    b.fsmn = c.take<uint16_t>(M * 512);
    b.att = c.take<uint16_t>(M * 512);
    // the hidden layer exists in memory only for the small-batch plan and under ASRFE_UNFUSED_FFN=1
    b.ffn = (ctx->unfused_ffn || M <= ctx->small_rows) ? c.take<uint16_t>(M * ctx->cfg.enc_linear_units) : nullptr;
    b.pe = c.take<float>(static_cast<int64_t>(Tp) * ctx->cfg.enc_input_size);
    b.bytes = c.off;
    return b;
}
}  // namespace

int64_t encoder_workspace_bytes(const asrfe_ctx* ctx, int batch, int tp) {
    return carve(ctx, nullptr, batch, tp).bytes;
}

int32_t encoder_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int B, int Tp, void* out, int out_dtype,
                        void* ws, int64_t ws_bytes, hipStream_t s) {
    (void)ws_bytes;
    const asrfe_config& c = ctx->cfg;
    const int M = B * Tp;
    const int ff = c.enc_linear_units;
    const Buffers bf = carve(ctx, ws, B, Tp);
    int32_t rc;

    {
        ProfScope ps(ctx, PK_PE, s);
        if ((rc = launch_pe_table(ctx, ctx->pe_inv, Tp, c.enc_input_size, bf.pe, s))) return rc;
    }

    // Small-batch plan (a few hundred rows: one short utterance): six launches per layer, each cut into 32 x 32 output
    // tiles so that the layer's weights are spread over the whole chip instead of streaming through a handful of CUs
    //     att, fsmn = attention(qkv);   x = (x +) fsmn + att Wo^T + b;   h = n(x);   f = relu(h W1^T + b1);
    //     x += f W2^T + b2;   h = n(x);   qkv = h Wqkv^T + b  (next layer)
    const bool small = ctx->small_rows > 0 && M <= ctx->small_rows;
    auto sgemm = [&](const uint16_t* A, int lda, const uint16_t* Wt, int ldw, const float* bias, int N, int K, void* out_,
                     int ldo, const uint16_t* add16, const float* add32, int epi, int kind) -> int32_t {
        GemmParams g{};
        g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.bias = bias; g.M = M; g.N = N; g.K = K; g.out = out_; g.ldo = ldo;
        g.add16 = add16; g.ld16 = 512; g.add32 = add32; g.ld32 = 512;
        ProfScope ps(ctx, kind, s);
        return launch_skinny_gemm(ctx, g, epi, s);
    };

    bool qkv_ready = false;  // the previous layer's post-attention kernel already projected this layer's q, k, v
    // When every layer but the last hands q, k, v to the next one, the residual stream only ever passes from one
    // post-attention launch to the next: it is then kept in that kernel's register order (FfnParams::tiled) and the
    // last layer writes the row-major x that after_norm reads.
    bool tiled = !small && !ctx->unfused_ffn && !ctx->separate_qkv && c.enc_num_blocks > 1 && !ctx->row_major_x;
    for (int li = 0; tiled && li + 1 < c.enc_num_blocks; ++li) tiled = ctx->layers[li].stream_has_qkv;
    for (int li = 0; li < c.enc_num_blocks; ++li) {
        const LayerWeights& L = ctx->layers[li];
        const int kin = li == 0 ? ctx->kin_pad : 512;
        // norm1: layer 0 normalises the scaled, position-coded input; later layers get h from the previous
        // layer's fused feed-forward kernel
        if (li == 0) {
            ProfScope ps(ctx, PK_LN, s);
            rc = launch_ln_input(ctx, feats, bf.pe, std::sqrt(static_cast<float>(c.enc_output_size)), nullptr, nullptr,
                                 c.enc_ln_eps, M, Tp, c.enc_input_size, bf.h, kin, s);
        } else if (ctx->unfused_ffn) {  // (the fused kernel of the previous layer left h or q, k, v)
            ProfScope ps(ctx, PK_LN, s);
            rc = launch_ln512(ctx, bf.x, nullptr, nullptr, c.enc_ln_eps, M, bf.h, s);
        }
        if (rc) return rc;

        GemmParams g{};
        if (!qkv_ready) {
            g.A = bf.h; g.lda = kin; g.W = L.w_qkv; g.ldw = kin; g.bias = L.b_qkv;
            g.M = M; g.N = 1536; g.K = kin; g.out = bf.qkv; g.ldo = 1536;
            { ProfScope ps(ctx, PK_GEMM_QKV, s); rc = launch_gemm(ctx, g, EPI_BF16, s); }
            if (rc) return rc;
        }
        qkv_ready = false;

        AttnParams a = attn_self(bf.qkv, 1536, B, Tp, lens, bf.att, 512);
        a.fsmn = bf.fsmn; a.w_fsmn = L.w_fsmn;
        a.scale_log2 = 1.4426950408889634f / std::sqrt(128.0f);
        { ProfScope ps(ctx, PK_ATTN, s); rc = launch_attention(ctx, a, s); }
        if (rc) return rc;

        if (small) {
            const LayerWeights* Ln = li + 1 < c.enc_num_blocks ? &ctx->layers[li + 1] : nullptr;
            if ((rc = sgemm(bf.att, 512, L.w_out, 512, L.b_out, 512, 512, bf.x, 512, bf.fsmn, li == 0 ? nullptr : bf.x,
                            EPI_F32_ADD, PK_GEMM_OUT)))
                return rc;
            { ProfScope ps(ctx, PK_LN, s); rc = launch_ln512(ctx, bf.x, nullptr, nullptr, c.enc_ln_eps, M, bf.h, s); }
            if (rc) return rc;
            if ((rc = sgemm(bf.h, 512, L.w_ffn1, 512, L.b_ffn1, ff, 512, bf.ffn, ff, nullptr, nullptr, EPI_BF16_RELU, PK_GEMM_FFN1)))
                return rc;
            if ((rc = sgemm(bf.ffn, ff, L.w_ffn2, ff, L.b_ffn2, 512, ff, bf.x, 512, nullptr, bf.x, EPI_F32_ADD, PK_GEMM_FFN2)))
                return rc;
            if (Ln) {
                { ProfScope ps(ctx, PK_LN, s); rc = launch_ln512(ctx, bf.x, nullptr, nullptr, c.enc_ln_eps, M, bf.h, s); }
                if (rc) return rc;
                if ((rc = sgemm(bf.h, 512, Ln->w_qkv, 512, Ln->b_qkv, 1536, 512, bf.qkv, 1536, nullptr, nullptr, EPI_BF16, PK_GEMM_QKV)))
                    return rc;
                qkv_ready = true;
            }
            continue;
        }

        if (!ctx->unfused_ffn) {
            // output projection + FSMN + residual, feed-forward, and the next layer's q, k, v (or, when the weight
            // stream does not carry Wqkv, h = the next layer's normalised input): one kernel
            const bool has_next = li + 1 < c.enc_num_blocks;
            const bool with_qkv = has_next && L.stream_has_qkv && !ctx->separate_qkv;
            FfnParams f{};
            f.att = bf.att; f.fsmn = bf.fsmn;
            f.x = li == 0 ? nullptr : bf.x; f.x_out = bf.x; f.tiled = tiled;
            f.wstream = L.w_stream; f.bo = L.b_out; f.b1 = L.b_ffn1; f.b2 = L.b_ffn2;
            f.y = has_next && !with_qkv ? bf.h : nullptr; f.ldy = 512;
            if (with_qkv) {
                f.bqkv = ctx->layers[li + 1].b_qkv; f.qkv = bf.qkv; f.ldq = 1536;
                qkv_ready = true;
            }
            f.eps = c.enc_ln_eps; f.M = M; f.ff = ff;
            { ProfScope ps(ctx, PK_FFN_FUSED, s); rc = ctx->mfma16 ? launch_post_attn16(ctx, f, s) : launch_post_attn(ctx, f, s); }
            if (rc) return rc;
            continue;
        }

        g = GemmParams{};
        g.A = bf.att; g.lda = 512; g.W = L.w_out; g.ldw = 512; g.bias = L.b_out;
        g.M = M; g.N = 512; g.K = 512; g.out = bf.x; g.ldo = 512;
        g.add16 = bf.fsmn; g.ld16 = 512;
        g.add32 = li == 0 ? nullptr : bf.x; g.ld32 = 512;
        { ProfScope ps(ctx, PK_GEMM_OUT, s); rc = launch_gemm(ctx, g, EPI_F32_ADD, s); }
        if (rc) return rc;

        { ProfScope ps(ctx, PK_LN, s); rc = launch_ln512(ctx, bf.x, nullptr, nullptr, c.enc_ln_eps, M, bf.h, s); }
        if (rc) return rc;

        g = GemmParams{};
        g.A = bf.h; g.lda = 512; g.W = L.w_ffn1; g.ldw = 512; g.bias = L.b_ffn1;
        g.M = M; g.N = ff; g.K = 512; g.out = bf.ffn; g.ldo = ff;
        { ProfScope ps(ctx, PK_GEMM_FFN1, s); rc = launch_gemm(ctx, g, EPI_BF16_RELU, s); }
        if (rc) return rc;

        g = GemmParams{};
        g.A = bf.ffn; g.lda = ff; g.W = L.w_ffn2; g.ldw = ff; g.bias = L.b_ffn2;
        g.M = M; g.N = 512; g.K = ff; g.out = bf.x; g.ldo = 512;
        g.add32 = bf.x; g.ld32 = 512;
        { ProfScope ps(ctx, PK_GEMM_FFN2, s); rc = launch_gemm(ctx, g, EPI_F32_ADD, s); }
        if (rc) return rc;
    }
    ProfScope ps(ctx, PK_LN, s);
    return launch_ln_final(ctx, bf.x, ctx->after_g, ctx->after_b, c.enc_ln_eps, B, Tp, lens, out, out_dtype, s);
}

}  // namespace asrfe
