// CPU restatement of the frontend + encoder part of the C ABI declared in include/asrfe.h, in plain C++ (OpenMP).
//
// TEST INFRASTRUCTURE ONLY.  It is built into oracle/_build/libasrfe_cpu.so and is loaded only by tests/ (as a second,
// independent checker beside the numpy / torch oracles) and by the cpu_baseline leg of bench.py (as the thing timed on
// the host cores).  Nothing in the shipped package loads it, and the product library has no CPU path.
//
// PARITY UNPINNED, like the other oracles: the reference reaches this arithmetic through funasr / torchaudio, which it
// does not vendor (/root/reference/voice-service/Dockerfile:25-36), at
// /root/reference/voice-service/app/services/voice_interface.py:2049-2053 (ASR), :1585-1590 (VAD), :1370-1374 (KWS),
// and holds no fixture for it.  Restated here from the published algorithms: Kaldi compute-fbank-feats as wrapped by
// torchaudio.compliance.kaldi.fbank (snip_edges framing, per-frame DC removal, pre-emphasis with the first sample as its
// own predecessor, symmetric Hamming window, 512-point power spectrum, triangular filters built in the mel domain, log with
// floor FLT_EPSILON), FunASR apply_lfr / apply_cmvn, and FunASR SANMEncoder (SinusoidalPositionEncoder, EncoderLayerSANM,
// MultiHeadedAttentionSANM with its FSMN memory block, PositionwiseFeedForward, LayerNorm eps 1e-12).
//
// Same entry points, argument meaning and status codes as the device library; every pointer is a HOST pointer, the
// stream argument is ignored, outputs are float32 only, and asrfe_encoder_workspace_bytes is 0 (scratch is allocated
// inside the call: a checker may).  One extra symbol, asrfe_cpu_set_threads, sets the OpenMP thread count.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <immintrin.h>
#include <omp.h>

#include "asrfe.h"

struct asrfe_ctx {
    asrfe_config cfg;
    std::vector<double> window;                 // [frame_length]
    std::vector<int> mel_lo, mel_n;             // first bin and bin count of every filter
    std::vector<std::vector<float>> mel_w;      // weights of those bins
    std::vector<float> shift, scale;            // CMVN (empty: identity)
    struct Layer {
        int din;
        std::vector<float> n1g, n1b, wqkv_t, bqkv, wfsmn, wo_t, bo, n2g, n2b, w1_t, b1, w2_t, b2;  // *_t: column panels
    };
    std::vector<Layer> layers;
    std::vector<float> ang, anb;
    bool loaded = false;
};

namespace {
thread_local std::string g_err;

int32_t fail(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}

double mel_scale(double f) { return 1127.0 * std::log(1.0 + f / 700.0); }

// ---- fbank ------------------------------------------------------------------------------------------------------------
// in-place iterative radix-2 FFT of n = 512 complex points (double)
void fft512(double* re, double* im, const double* cs, const double* sn) {
    const int n = 512;
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < half; ++k) {
                const double wr = cs[k * step], wi = -sn[k * step];
                const double xr = re[i + k + half] * wr - im[i + k + half] * wi;
                const double xi = re[i + k + half] * wi + im[i + k + half] * wr;
                re[i + k + half] = re[i + k] - xr; im[i + k + half] = im[i + k] - xi;
                re[i + k] += xr; im[i + k] += xi;
            }
    }
}

double pcm_at(const void* pcm, int32_t dt, int64_t i) {
    if (dt == ASRFE_PCM_I16) return static_cast<const int16_t*>(pcm)[i];
    const float v = static_cast<const float*>(pcm)[i];
    if (dt == ASRFE_PCM_F32) return static_cast<double>(v) * 32768.0;
    // the streaming session's hand-off (voice_interface.py:1997-2010): clip, x 32767, truncate to int16
    const float c = std::min(1.0f, std::max(-1.0f, v));
    return static_cast<double>(static_cast<int16_t>(c * 32767.0f));
}

int64_t frames_of(const asrfe_config& c, int64_t n) { return n < c.frame_length ? 0 : 1 + (n - c.frame_length) / c.frame_shift; }

// log-mel of frame t of one clip into out[n_mels]
void fbank_frame(const asrfe_ctx* ctx, const void* pcm, int32_t dt, int64_t t, const double* cs, const double* sn, float* out) {
    const asrfe_config& c = ctx->cfg;
    const int L = c.frame_length;
    double re[512] = {0.0}, im[512];
    double mean = 0.0;
    for (int i = 0; i < L; ++i) { re[i] = pcm_at(pcm, dt, t * c.frame_shift + i); mean += re[i]; }
    mean /= L;
    for (int i = 0; i < L; ++i) re[i] -= mean;
    for (int i = L - 1; i > 0; --i) re[i] -= c.preemph * re[i - 1];
    re[0] -= c.preemph * re[0];
    for (int i = 0; i < L; ++i) re[i] *= ctx->window[i];
    for (int i = L; i < 512; ++i) re[i] = 0.0;
    std::memset(im, 0, sizeof(im));
    fft512(re, im, cs, sn);
    for (int m = 0; m < c.n_mels; ++m) {
        double acc = 0.0;
        const int lo = ctx->mel_lo[m];
        for (int j = 0; j < ctx->mel_n[m]; ++j) acc += static_cast<double>(ctx->mel_w[m][j]) * (re[lo + j] * re[lo + j] + im[lo + j] * im[lo + j]);
        out[m] = static_cast<float>(std::log(std::max(acc, 1.1920928955078125e-07)));
    }
}

struct Twiddles {
    double cs[256], sn[256];
    Twiddles() {
        for (int k = 0; k < 256; ++k) { cs[k] = std::cos(2.0 * M_PI * k / 512.0); sn[k] = std::sin(2.0 * M_PI * k / 512.0); }
    }
};
const Twiddles& twiddles() {
    static const Twiddles t;
    return t;
}

// ---- encoder pieces -----------------------------------------------------------------------------------------------------
// Weights are kept as column panels: panel p holds columns 16 p .. 16 p + 15 as [K][16] (zero-padded past N), so the
// inner loop of the product streams one panel from memory in order.
constexpr int kPanel = 16;
std::vector<float> pack_panels(const float* w, int64_t n, int64_t k) {  // torch weight [N][K] -> [ceil(N / 16)][K][16]
    const int64_t np = (n + kPanel - 1) / kPanel;
    std::vector<float> p(static_cast<size_t>(np * k * kPanel), 0.0f);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < k; ++j) p[static_cast<size_t>(((i / kPanel) * k + j) * kPanel + i % kPanel)] = w[i * k + j];
    return p;
}

// C[m][n] = bias[n] + sum_k A[m][k] W[n][k]  (+ relu), W given as panels: 6 rows x 16 columns of accumulators stay in
// registers (AVX2: 12 vectors) while k runs
void gemm_t(const float* A, int lda, const float* Wp, const float* bias, int M, int N, int K, float* C, int ldc, bool relu) {
    constexpr int MB = 6;
    const int mblocks = (M + MB - 1) / MB, npanels = (N + kPanel - 1) / kPanel;
#pragma omp parallel for schedule(static)
    for (int mb = 0; mb < mblocks; ++mb) {
        const int m0 = mb * MB, mr = std::min(MB, M - m0);
        const float* ar[MB];
        for (int r = 0; r < MB; ++r) ar[r] = A + static_cast<int64_t>(m0 + std::min(r, mr - 1)) * lda;
        for (int pn = 0; pn < npanels; ++pn) {
            const float* w = Wp + static_cast<int64_t>(pn) * K * kPanel;
            const int n0 = pn * kPanel, nr = std::min(kPanel, N - n0);
            __m256 acc[MB][2];
            float bb[kPanel];
            for (int n = 0; n < kPanel; ++n) bb[n] = (bias && n < nr) ? bias[n0 + n] : 0.0f;
            for (int r = 0; r < MB; ++r) { acc[r][0] = _mm256_loadu_ps(bb); acc[r][1] = _mm256_loadu_ps(bb + 8); }
            for (int k = 0; k < K; ++k) {
                const __m256 w0 = _mm256_loadu_ps(w + k * kPanel), w1 = _mm256_loadu_ps(w + k * kPanel + 8);
#pragma GCC unroll 6
                for (int r = 0; r < MB; ++r) {
                    const __m256 a = _mm256_broadcast_ss(ar[r] + k);
                    acc[r][0] = _mm256_fmadd_ps(a, w0, acc[r][0]);
                    acc[r][1] = _mm256_fmadd_ps(a, w1, acc[r][1]);
                }
            }
            for (int r = 0; r < mr; ++r) {
                float t[kPanel];
                _mm256_storeu_ps(t, acc[r][0]);
                _mm256_storeu_ps(t + 8, acc[r][1]);
                float* c = C + static_cast<int64_t>(m0 + r) * ldc + n0;
                for (int n = 0; n < nr; ++n) c[n] = relu ? std::max(t[n], 0.0f) : t[n];
            }
        }
    }
}

void layer_norm(const float* x, int ldx, const float* g, const float* b, int M, int D, float eps, float* y, int ldy) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* xr = x + static_cast<int64_t>(m) * ldx;
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += xr[d];
        const double mean = s / D;
        double v = 0.0;
        for (int d = 0; d < D; ++d) v += (xr[d] - mean) * (xr[d] - mean);
        const double rstd = 1.0 / std::sqrt(v / D + eps);
        float* yr = y + static_cast<int64_t>(m) * ldy;
        for (int d = 0; d < D; ++d) yr[d] = static_cast<float>((xr[d] - mean) * rstd * g[d] + b[d]);
    }
}

}  // namespace

extern "C" {

int32_t asrfe_default_config(asrfe_config* c) {
    if (!c) return ASRFE_ERR_BAD_ARG;
    std::memset(c, 0, sizeof(*c));
    c->sample_rate = 16000; c->frame_length = 400; c->frame_shift = 160; c->n_fft = 512; c->n_mels = 80;
    c->preemph = 0.97f; c->low_freq = 20.0f; c->high_freq = 0.0f; c->window_type = 0; c->lfr_m = 7; c->lfr_n = 6;
    c->enc_input_size = 560; c->enc_output_size = 512; c->enc_heads = 4; c->enc_linear_units = 2048; c->enc_num_blocks = 50;
    c->enc_kernel_size = 11; c->enc_sanm_shift = 0; c->enc_ln_eps = 1e-12f; c->enc_compute_dtype = ASRFE_DT_F32;
    c->enc_small_batch_rows = 0;
    return ASRFE_OK;
}

int32_t asrfe_create(const asrfe_config* cfg, asrfe_ctx** out) {
    if (!cfg || !out) return fail(ASRFE_ERR_BAD_ARG, "asrfe_create: null argument");
    const asrfe_config& c = *cfg;
    if (c.n_fft != 512 || c.frame_length < 2 || c.frame_length > 512 || c.frame_shift < 1 || c.n_mels < 1 || c.lfr_m < 1 || c.lfr_n < 1 ||
        c.enc_heads < 1 || c.enc_output_size % c.enc_heads != 0 || c.enc_kernel_size < 1)
        return fail(ASRFE_ERR_BAD_ARG, "asrfe_create: unsupported configuration");
    asrfe_ctx* x = new asrfe_ctx();
    x->cfg = c;
    const int L = c.frame_length;
    x->window.resize(L);
    for (int i = 0; i < L; ++i) {
        const double a = 2.0 * M_PI * i / (L - 1);
        switch (c.window_type) {
            case 0: x->window[i] = 0.54 - 0.46 * std::cos(a); break;
            case 1: x->window[i] = 0.5 - 0.5 * std::cos(a); break;
            case 2: x->window[i] = std::pow(0.5 - 0.5 * std::cos(a), 0.85); break;
            default: x->window[i] = 1.0; break;
        }
    }
    const double nyq = 0.5 * c.sample_rate;
    const double hi_f = c.high_freq > 0.0f ? c.high_freq : nyq + c.high_freq;
    const double mlo = mel_scale(c.low_freq), mhi = mel_scale(hi_f), delta = (mhi - mlo) / (c.n_mels + 1);
    const double bin_w = static_cast<double>(c.sample_rate) / c.n_fft;
    x->mel_lo.resize(c.n_mels); x->mel_n.resize(c.n_mels); x->mel_w.resize(c.n_mels);
    for (int m = 0; m < c.n_mels; ++m) {
        const double left = mlo + m * delta, center = mlo + (m + 1.0) * delta, right = mlo + (m + 2.0) * delta;
        int lo = -1, hi = -1;
        std::vector<float> w(c.n_fft / 2, 0.0f);
        for (int k = 0; k < c.n_fft / 2; ++k) {
            const double mel = mel_scale(bin_w * k);
            const double v = std::max(0.0, std::min((mel - left) / (center - left), (right - mel) / (right - center)));
            w[k] = static_cast<float>(v);
            if (v > 0.0) { if (lo < 0) lo = k; hi = k; }
        }
        if (lo < 0) { lo = 0; hi = -1; }
        x->mel_lo[m] = lo; x->mel_n[m] = hi - lo + 1;
        x->mel_w[m].assign(w.begin() + lo, w.begin() + hi + 1);
    }
    *out = x;
    return ASRFE_OK;
}

int32_t asrfe_destroy(asrfe_ctx* ctx) {
    delete ctx;
    return ASRFE_OK;
}

int32_t asrfe_load_cmvn(asrfe_ctx* ctx, const float* shift, const float* scale, int32_t dim) {
    if (!ctx || !shift || !scale) return fail(ASRFE_ERR_BAD_ARG, "asrfe_load_cmvn: null argument");
    if (dim != ctx->cfg.lfr_m * ctx->cfg.n_mels) return fail(ASRFE_ERR_SHAPE, "asrfe_load_cmvn: dim must be lfr_m * n_mels");
    ctx->shift.assign(shift, shift + dim);
    ctx->scale.assign(scale, scale + dim);
    return ASRFE_OK;
}

int32_t asrfe_load_encoder_weights(asrfe_ctx* ctx, const asrfe_tensor_desc* descs, int32_t n) {
    if (!ctx || !descs || n <= 0) return fail(ASRFE_ERR_BAD_ARG, "asrfe_load_encoder_weights: null argument");
    std::map<std::string, const asrfe_tensor_desc*> m;
    for (int i = 0; i < n; ++i) {
        std::string name = descs[i].name ? descs[i].name : "";
        if (name.rfind("encoder.", 0) == 0) name = name.substr(8);
        m[name] = &descs[i];
    }
    const asrfe_config& c = ctx->cfg;
    const int d = c.enc_output_size, ff = c.enc_linear_units, ks = c.enc_kernel_size;
    bool ok = true;
    std::string missing;
    auto get = [&](const std::string& name, std::initializer_list<int64_t> shape) -> const float* {
        auto it = m.find(name);
        if (it == m.end()) { ok = false; missing = "missing tensor " + name; return nullptr; }
        const asrfe_tensor_desc& t = *it->second;
        if (t.ndim != static_cast<int32_t>(shape.size()) || !std::equal(shape.begin(), shape.end(), t.shape)) {
            ok = false; missing = "wrong shape for " + name; return nullptr;
        }
        return t.data_host;
    };
    std::vector<asrfe_ctx::Layer> layers(c.enc_num_blocks);
    for (int li = 0; li < c.enc_num_blocks && ok; ++li) {
        const std::string p = li == 0 ? "encoders0.0" : "encoders." + std::to_string(li - 1);
        const int din = li == 0 ? c.enc_input_size : d;
        asrfe_ctx::Layer& L = layers[li];
        L.din = din;
        const float *n1g = get(p + ".norm1.weight", {din}), *n1b = get(p + ".norm1.bias", {din});
        const float *wqkv = get(p + ".self_attn.linear_q_k_v.weight", {3 * d, din}), *bqkv = get(p + ".self_attn.linear_q_k_v.bias", {3 * d});
        const float* wf = get(p + ".self_attn.fsmn_block.weight", {d, 1, ks});
        const float *wo = get(p + ".self_attn.linear_out.weight", {d, d}), *bo = get(p + ".self_attn.linear_out.bias", {d});
        const float *n2g = get(p + ".norm2.weight", {d}), *n2b = get(p + ".norm2.bias", {d});
        const float *w1 = get(p + ".feed_forward.w_1.weight", {ff, d}), *b1 = get(p + ".feed_forward.w_1.bias", {ff});
        const float *w2 = get(p + ".feed_forward.w_2.weight", {d, ff}), *b2 = get(p + ".feed_forward.w_2.bias", {d});
        if (!ok) break;
        L.n1g.assign(n1g, n1g + din); L.n1b.assign(n1b, n1b + din);
        L.wqkv_t = pack_panels(wqkv, 3 * d, din); L.bqkv.assign(bqkv, bqkv + 3 * d);
        L.wfsmn.assign(wf, wf + static_cast<size_t>(d) * ks);
        L.wo_t = pack_panels(wo, d, d); L.bo.assign(bo, bo + d);
        L.n2g.assign(n2g, n2g + d); L.n2b.assign(n2b, n2b + d);
        L.w1_t = pack_panels(w1, ff, d); L.b1.assign(b1, b1 + ff);
        L.w2_t = pack_panels(w2, d, ff); L.b2.assign(b2, b2 + d);
    }
    const float *ang = ok ? get("after_norm.weight", {d}) : nullptr, *anb = ok ? get("after_norm.bias", {d}) : nullptr;
    if (!ok) return fail(ASRFE_ERR_SHAPE, "asrfe_load_encoder_weights: " + missing);
    ctx->layers.swap(layers);
    ctx->ang.assign(ang, ang + d);
    ctx->anb.assign(anb, anb + d);
    ctx->loaded = true;
    return ASRFE_OK;
}

int32_t asrfe_num_frames(const asrfe_ctx* ctx, int64_t n_samples) {
    return ctx ? static_cast<int32_t>(frames_of(ctx->cfg, n_samples)) : -1;
}
int32_t asrfe_num_lfr_frames(const asrfe_ctx* ctx, int32_t n_frames) {
    return ctx ? (n_frames + ctx->cfg.lfr_n - 1) / ctx->cfg.lfr_n : -1;
}

int32_t asrfe_fbank(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride, const int32_t* pcm_lens, int32_t batch,
                    int32_t t_max, float* logmel, int32_t* n_frames_out, void*) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (batch < 0 || t_max < 0) return fail(ASRFE_ERR_SHAPE, "asrfe_fbank: negative size");
    if (batch == 0) return ASRFE_OK;
    if (!pcm || !pcm_lens || !logmel) return fail(ASRFE_ERR_BAD_ARG, "asrfe_fbank: null argument");
    if (pcm_dtype < 0 || pcm_dtype > 2) return fail(ASRFE_ERR_BAD_ARG, "asrfe_fbank: pcm_dtype");
    const asrfe_config& c = ctx->cfg;
    const Twiddles& tw = twiddles();
    const int64_t esz = pcm_dtype == ASRFE_PCM_I16 ? 2 : 4;
    for (int b = 0; b < batch; ++b) {
        const int64_t T = std::min<int64_t>(frames_of(c, pcm_lens[b]), t_max);
        if (n_frames_out) n_frames_out[b] = static_cast<int32_t>(T);
        const char* row = static_cast<const char*>(pcm) + b * pcm_stride * esz;
        float* o = logmel + static_cast<int64_t>(b) * t_max * c.n_mels;
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < t_max; ++t) {
            if (t < T) fbank_frame(ctx, row, pcm_dtype, t, tw.cs, tw.sn, o + t * c.n_mels);
            else std::memset(o + t * c.n_mels, 0, sizeof(float) * c.n_mels);
        }
    }
    return ASRFE_OK;
}

int32_t asrfe_lfr_cmvn(asrfe_ctx* ctx, const float* logmel, const int32_t* n_frames, int32_t batch, int32_t t_max, int32_t tp_max,
                       void* feats, int32_t out_dtype, int32_t* lens_out, void*) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (batch < 0 || t_max < 0 || tp_max < 0) return fail(ASRFE_ERR_SHAPE, "asrfe_lfr_cmvn: negative size");
    if (batch == 0) return ASRFE_OK;
    if (!logmel || !n_frames || !feats) return fail(ASRFE_ERR_BAD_ARG, "asrfe_lfr_cmvn: null argument");
    if (out_dtype != ASRFE_DT_F32) return fail(ASRFE_ERR_BAD_ARG, "asrfe_lfr_cmvn: the CPU restatement writes float32 only");
    const asrfe_config& c = ctx->cfg;
    const int dim = c.lfr_m * c.n_mels, left = (c.lfr_m - 1) / 2;
    float* out = static_cast<float*>(feats);
    for (int b = 0; b < batch; ++b) {
        const int T = std::min(n_frames[b], t_max);
        const int Tp = std::min((T + c.lfr_n - 1) / c.lfr_n, tp_max);
        if (lens_out) lens_out[b] = Tp;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < tp_max; ++i) {
            float* o = out + (static_cast<int64_t>(b) * tp_max + i) * dim;
            if (i >= Tp) { std::memset(o, 0, sizeof(float) * dim); continue; }
            for (int j = 0; j < c.lfr_m; ++j) {
                const int src = std::min(std::max(c.lfr_n * i + j - left, 0), T - 1);
                const float* f = logmel + (static_cast<int64_t>(b) * t_max + src) * c.n_mels;
                for (int m = 0; m < c.n_mels; ++m) {
                    const int k = j * c.n_mels + m;
                    o[k] = ctx->shift.empty() ? f[m] : (f[m] + ctx->shift[k]) * ctx->scale[k];
                }
            }
        }
    }
    return ASRFE_OK;
}

int32_t asrfe_frontend(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride, const int32_t* pcm_lens, int32_t batch,
                       int32_t tp_max, void* feats, int32_t out_dtype, int32_t* lens_out, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (batch <= 0 || tp_max < 0) return batch == 0 ? ASRFE_OK : fail(ASRFE_ERR_SHAPE, "asrfe_frontend: negative size");
    const int t_max = tp_max * ctx->cfg.lfr_n;
    std::vector<float> logmel(static_cast<size_t>(batch) * t_max * ctx->cfg.n_mels);
    std::vector<int32_t> nf(batch);
    int32_t rc = asrfe_fbank(ctx, pcm, pcm_dtype, pcm_stride, pcm_lens, batch, t_max, logmel.data(), nf.data(), stream);
    if (rc) return rc;
    return asrfe_lfr_cmvn(ctx, logmel.data(), nf.data(), batch, t_max, tp_max, feats, out_dtype, lens_out, stream);
}

int64_t asrfe_encoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t tp) {
    return (!ctx || batch < 0 || tp < 0) ? -1 : 0;
}

int32_t asrfe_encoder_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch, int32_t tp, void* out,
                              int32_t out_dtype, void*, int64_t, void*) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!ctx->loaded) return fail(ASRFE_ERR_NOT_LOADED, "asrfe_encoder_forward: encoder weights not loaded");
    if (batch < 0 || tp < 0) return fail(ASRFE_ERR_SHAPE, "asrfe_encoder_forward: negative size");
    if (batch == 0 || tp == 0) return ASRFE_OK;
    if (!feats || !lens || !out) return fail(ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward: null argument");
    if (out_dtype != ASRFE_DT_F32) return fail(ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward: the CPU restatement writes float32 only");
    const asrfe_config& c = ctx->cfg;
    const int d = c.enc_output_size, din0 = c.enc_input_size, H = c.enc_heads, dk = d / H, ff = c.enc_linear_units, ks = c.enc_kernel_size;
    const int M = batch * tp;
    const int left = (ks - 1) / 2 + std::max(c.enc_sanm_shift, 0);
    std::vector<int> len(batch);
    for (int b = 0; b < batch; ++b) len[b] = std::min(std::max(lens[b], 0), tp);
    // x = feats * sqrt(d) + PE(positions 1..tp over din0 channels: [sin | cos], increment ln(1e4) / (din0 / 2 - 1))
    std::vector<float> x(static_cast<size_t>(M) * std::max(d, din0)), xn(static_cast<size_t>(M) * std::max(d, din0));
    std::vector<float> qkv(static_cast<size_t>(M) * 3 * d), att(static_cast<size_t>(M) * d), fsmn(static_cast<size_t>(M) * d);
    std::vector<float> hid(static_cast<size_t>(M) * ff), tmp(static_cast<size_t>(M) * d);
    {
        const int half = din0 / 2;
        const double inc = std::log(10000.0) / (half - 1);
        const float sc = std::sqrt(static_cast<float>(d));
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m) {
            const int t = m % tp;
            for (int j = 0; j < half; ++j) {
                const double a = (t + 1.0) * std::exp(-inc * j);
                x[static_cast<size_t>(m) * din0 + j] = feats[static_cast<size_t>(m) * din0 + j] * sc + static_cast<float>(std::sin(a));
                x[static_cast<size_t>(m) * din0 + half + j] = feats[static_cast<size_t>(m) * din0 + half + j] * sc + static_cast<float>(std::cos(a));
            }
        }
    }
    int ldx = din0;
    const float qscale = 1.0f / std::sqrt(static_cast<float>(dk));
    for (const asrfe_ctx::Layer& L : ctx->layers) {
        layer_norm(x.data(), ldx, L.n1g.data(), L.n1b.data(), M, L.din, c.enc_ln_eps, xn.data(), L.din);
        gemm_t(xn.data(), L.din, L.wqkv_t.data(), L.bqkv.data(), M, 3 * d, L.din, qkv.data(), 3 * d, false);
        // FSMN memory block on the masked V: depth-wise convolution over time (zero padding), plus its input, masked
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m) {
            const int b = m / tp, t = m % tp;
            float* o = fsmn.data() + static_cast<size_t>(m) * d;
            if (t >= len[b]) { std::memset(o, 0, sizeof(float) * d); continue; }
            const float* v0 = qkv.data() + static_cast<size_t>(m) * 3 * d + 2 * d;
            for (int ch = 0; ch < d; ++ch) o[ch] = v0[ch];
            for (int j = 0; j < ks; ++j) {
                const int tt = t + j - left;
                if (tt < 0 || tt >= len[b]) continue;
                const float* v = qkv.data() + (static_cast<size_t>(b) * tp + tt) * 3 * d + 2 * d;
                for (int ch = 0; ch < d; ++ch) o[ch] += L.wfsmn[static_cast<size_t>(ch) * ks + j] * v[ch];
            }
        }
        // softmax(q k^T / sqrt(dk)) v over the valid keys, per (utterance, head)
#pragma omp parallel for collapse(2) schedule(dynamic, 16)
        for (int bh = 0; bh < batch * H; ++bh)
            for (int t = 0; t < tp; ++t) {
                const int b = bh / H, h = bh % H, n = len[b];
                float* o = att.data() + (static_cast<size_t>(b) * tp + t) * d + h * dk;
                if (n == 0) { std::memset(o, 0, sizeof(float) * dk); continue; }
                const float* q = qkv.data() + (static_cast<size_t>(b) * tp + t) * 3 * d + h * dk;
                std::vector<float> s(n);
                float mx = -INFINITY;
                for (int j = 0; j < n; ++j) {
                    const float* k = qkv.data() + (static_cast<size_t>(b) * tp + j) * 3 * d + d + h * dk;
                    float acc = 0.0f;
#pragma omp simd reduction(+ : acc)
                    for (int e = 0; e < dk; ++e) acc += q[e] * k[e];
                    s[j] = acc * qscale;
                    mx = std::max(mx, s[j]);
                }
                double sum = 0.0;
                for (int j = 0; j < n; ++j) { s[j] = std::exp(s[j] - mx); sum += s[j]; }
                const float inv = static_cast<float>(1.0 / sum);
                float accv[512];
                for (int e = 0; e < dk; ++e) accv[e] = 0.0f;
                for (int j = 0; j < n; ++j) {
                    const float* v = qkv.data() + (static_cast<size_t>(b) * tp + j) * 3 * d + 2 * d + h * dk;
                    const float pj = s[j] * inv;
#pragma omp simd
                    for (int e = 0; e < dk; ++e) accv[e] += pj * v[e];
                }
                for (int e = 0; e < dk; ++e) o[e] = accv[e];
            }
        gemm_t(att.data(), d, L.wo_t.data(), L.bo.data(), M, d, d, tmp.data(), d, false);
        const bool res = L.din == d;
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m)
            for (int ch = 0; ch < d; ++ch) {
                const float r = res ? x[static_cast<size_t>(m) * ldx + ch] : 0.0f;
                tmp[static_cast<size_t>(m) * d + ch] += r + fsmn[static_cast<size_t>(m) * d + ch];
            }
        // (x now has d channels)
        std::memcpy(x.data(), tmp.data(), sizeof(float) * static_cast<size_t>(M) * d);
        ldx = d;
        layer_norm(x.data(), d, L.n2g.data(), L.n2b.data(), M, d, c.enc_ln_eps, xn.data(), d);
        gemm_t(xn.data(), d, L.w1_t.data(), L.b1.data(), M, ff, d, hid.data(), ff, true);
        gemm_t(hid.data(), ff, L.w2_t.data(), L.b2.data(), M, d, ff, tmp.data(), d, false);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < static_cast<int64_t>(M) * d; ++i) x[i] += tmp[i];
    }
    float* o = static_cast<float*>(out);
    layer_norm(x.data(), d, ctx->ang.data(), ctx->anb.data(), M, d, c.enc_ln_eps, o, d);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        if (m % tp >= len[m / tp]) std::memset(o + static_cast<size_t>(m) * d, 0, sizeof(float) * d);
    return ASRFE_OK;
}

const char* asrfe_last_error(const asrfe_ctx*) { return g_err.c_str(); }
const char* asrfe_version(void) { return "asrfe-cpu-restatement (test infrastructure; frontend + encoder subset, host pointers)"; }

// not part of include/asrfe.h: OpenMP threads the calls above may use (<= 0: the runtime's default)
void asrfe_cpu_set_threads(int32_t n) {
    if (n > 0) omp_set_num_threads(n);
}

}  // extern "C"
