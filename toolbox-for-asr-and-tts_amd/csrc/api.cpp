// libasrfe C ABI: context, device tables, weight loading and argument checking.
// The compute calls only validate, fill a parameter block and launch; they never allocate.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "common.h"
#include "encoder.h"
#include "gemm.h"
#include "weights.h"

using namespace asrfe;

// The error text is kept per thread (and tagged with the context it belongs to): compute calls may run on several
// threads of one context at once, and each reads back the text of its own failure.
static thread_local std::string g_create_err;
static thread_local const asrfe_ctx* g_err_ctx = nullptr;
static thread_local std::string g_err;

namespace asrfe {
int32_t set_err(asrfe_ctx* ctx, int32_t code, const std::string& msg) {
    if (ctx) {
        g_err_ctx = ctx;
        g_err = msg;
    } else {
        g_create_err = msg;
    }
    return code;
}
}  // namespace asrfe

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return set_err(ctx, ASRFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

extern "C" const char* asrfe_version(void) { return "asrfe 0.1.0 (gfx950)"; }

extern "C" const char* asrfe_last_error(const asrfe_ctx* ctx) {
    if (!ctx) return g_create_err.c_str();
    return g_err_ctx == ctx ? g_err.c_str() : "";
}

extern "C" int32_t asrfe_default_config(asrfe_config* c) {
    if (!c) return ASRFE_ERR_BAD_ARG;
    c->sample_rate = 16000;
    c->frame_length = 400;
    c->frame_shift = 160;
    c->n_fft = 512;
    c->n_mels = 80;
    c->preemph = 0.97f;
    c->low_freq = 20.0f;
    c->high_freq = 0.0f;
    c->window_type = 0;
    c->lfr_m = 7;
    c->lfr_n = 6;
    c->enc_input_size = 560;
    c->enc_output_size = 512;
    c->enc_heads = 4;
    c->enc_linear_units = 2048;
    c->enc_num_blocks = 50;
    c->enc_kernel_size = 11;
    c->enc_sanm_shift = 0;
    c->enc_ln_eps = 1e-12f;
    c->enc_compute_dtype = ASRFE_DT_F16;
    c->enc_small_batch_rows = 2048;
    return ASRFE_OK;
}

// ---- host-side table builders (double precision) -------------------------------------------------
static double mel_of(double f) { return 1127.0 * std::log(1.0 + f / 700.0); }

// Kaldi mel banks over bins 0..n_fft/2-1; out is [n_mels][n_fft/2] row-major.
static void build_mel_banks(const asrfe_config& c, std::vector<double>& out) {
    const int nb = c.n_fft / 2;
    const double nyq = 0.5 * c.sample_rate;
    const double hi = c.high_freq > 0.0f ? c.high_freq : nyq + c.high_freq;
    const double bw = static_cast<double>(c.sample_rate) / c.n_fft;
    const double mlo = mel_of(c.low_freq), mhi = mel_of(hi);
    const double delta = (mhi - mlo) / (c.n_mels + 1);
    out.assign(static_cast<size_t>(c.n_mels) * nb, 0.0);
    for (int m = 0; m < c.n_mels; ++m) {
        const double left = mlo + m * delta, center = mlo + (m + 1.0) * delta, right = mlo + (m + 2.0) * delta;
        for (int k = 0; k < nb; ++k) {
            const double mel = mel_of(bw * k);
            const double up = (mel - left) / (center - left);
            const double down = (right - mel) / (right - center);
            const double w = std::fmax(0.0, std::fmin(up, down));
            out[static_cast<size_t>(m) * nb + k] = w;
        }
    }
}

static void build_window(const asrfe_config& c, std::vector<double>& w) {
    const int n = c.frame_length;
    w.assign(n, 1.0);
    const double a = 2.0 * M_PI / (n - 1);
    for (int i = 0; i < n; ++i) {
        switch (c.window_type) {
            case 0: w[i] = 0.54 - 0.46 * std::cos(a * i); break;
            case 1: w[i] = 0.5 - 0.5 * std::cos(a * i); break;
            case 2: w[i] = std::pow(0.5 - 0.5 * std::cos(a * i), 0.85); break;
            default: w[i] = 1.0; break;
        }
    }
}

// Host-only helpers exported for the CPU tests (no device needed).
extern "C" int32_t asrfe_host_mel_banks(const asrfe_config* cfg, double* out /* [n_mels][n_fft/2] */) {
    if (!cfg || !out || cfg->n_mels <= 0 || cfg->n_fft <= 0) return ASRFE_ERR_BAD_ARG;
    std::vector<double> v;
    build_mel_banks(*cfg, v);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
    return ASRFE_OK;
}
extern "C" int32_t asrfe_host_window(const asrfe_config* cfg, double* out /* [frame_length] */) {
    if (!cfg || !out || cfg->frame_length <= 1) return ASRFE_ERR_BAD_ARG;
    std::vector<double> v;
    build_window(*cfg, v);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
    return ASRFE_OK;
}

static int32_t check_config(const asrfe_config& c, std::string& why) {
    if (c.frame_length != kFrameLen || c.frame_shift != kFrameShift || c.n_fft != kNfft) {
        why = "only frame_length 400 / frame_shift 160 / n_fft 512 kernels are built";
        return ASRFE_ERR_BAD_ARG;
    }
    if (c.n_mels < 1 || c.n_mels > kMaxMels) { why = "n_mels must be in 1..80"; return ASRFE_ERR_BAD_ARG; }
    if (c.lfr_m < 1 || c.lfr_n < 1 || c.lfr_m > 16) { why = "bad lfr_m / lfr_n"; return ASRFE_ERR_BAD_ARG; }
    if (c.lfr_m * c.n_mels > asrfe::kMaxFeatDim) { why = "lfr_m * n_mels must be <= 640"; return ASRFE_ERR_BAD_ARG; }
    if (c.preemph != 0.97f) { why = "only pre-emphasis 0.97 is built"; return ASRFE_ERR_BAD_ARG; }
    if (c.enc_output_size != 512 || c.enc_heads != 4) {
        why = "only enc_output_size 512 with 4 heads (d_k 128) kernels are built";
        return ASRFE_ERR_BAD_ARG;
    }
    if (c.enc_linear_units % 256 != 0 || c.enc_linear_units <= 0) { why = "enc_linear_units must be a multiple of 256"; return ASRFE_ERR_BAD_ARG; }
    if (c.enc_kernel_size != 11 || c.enc_sanm_shift != 0) { why = "only FSMN kernel 11 / shift 0 is built"; return ASRFE_ERR_BAD_ARG; }
    if (c.enc_input_size < 8 || c.enc_input_size > 640 || c.enc_input_size % 8 != 0) { why = "enc_input_size must be a multiple of 8 in 8..640"; return ASRFE_ERR_BAD_ARG; }
    if (c.enc_num_blocks < 1) { why = "enc_num_blocks must be >= 1"; return ASRFE_ERR_BAD_ARG; }
    if (c.enc_small_batch_rows < 0 || c.enc_small_batch_rows > 8192) { why = "enc_small_batch_rows must be in 0..8192"; return ASRFE_ERR_BAD_ARG; }
    if (c.enc_compute_dtype != ASRFE_DT_F16 && c.enc_compute_dtype != ASRFE_DT_BF16) { why = "enc_compute_dtype must be ASRFE_DT_F16 or ASRFE_DT_BF16"; return ASRFE_ERR_BAD_ARG; }
    return ASRFE_OK;
}

extern "C" int32_t asrfe_create(const asrfe_config* cfg, asrfe_ctx** out) {
    if (!cfg || !out) return set_err(nullptr, ASRFE_ERR_BAD_ARG, "asrfe_create: null argument");
    *out = nullptr;
    std::string why;
    if (check_config(*cfg, why) != ASRFE_OK) return set_err(nullptr, ASRFE_ERR_BAD_ARG, "asrfe_create: " + why);
    int dev = 0;
    HIP_TRY(nullptr, hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(nullptr, ASRFE_ERR_UNSUPPORTED_ARCH, std::string("asrfe_create: built for gfx950, device is ") + prop.gcnArchName);

    asrfe_ctx* ctx = new asrfe_ctx();
    ctx->cfg = *cfg;
    ctx->device = dev;
    ctx->cmvn_loaded = false;
    ctx->enc_loaded = false;
    ctx->kin_pad = (cfg->enc_input_size + 127) / 128 * 128;
    {
        const char* e = std::getenv("ASRFE_UNFUSED_FFN");
        ctx->unfused_ffn = e && e[0] == '1';
        e = std::getenv("ASRFE_SEPARATE_QKV");
        ctx->separate_qkv = e && e[0] == '1';
        e = std::getenv("ASRFE_ROW_MAJOR_X");
        ctx->row_major_x = e && e[0] == '1';
        e = std::getenv("ASRFE_MFMA16");
        ctx->mfma16 = !(e && e[0] == '0') && !ctx->unfused_ffn && !ctx->separate_qkv && !ctx->row_major_x &&
                      cfg->enc_num_blocks > 1 && cfg->enc_linear_units <= kPostAttnQkvMaxFf;
    }
    // the latency plan's second feed-forward product takes linear_units in 512-deep steps up to 3072
    ctx->small_rows = (cfg->enc_linear_units % 512 == 0 && cfg->enc_linear_units <= 3072 && !ctx->unfused_ffn) ? cfg->enc_small_batch_rows : 0;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->after_g = ctx->after_b = ctx->pe_inv = nullptr;
    std::memset(&ctx->tab, 0, sizeof(ctx->tab));

    auto fail = [&](int32_t rc) {
        g_create_err = g_err_ctx == ctx ? g_err : std::string();
        g_err_ctx = nullptr;
        asrfe_destroy(ctx);
        return rc;
    };

    // window, zero-padded
    std::vector<double> wd;
    build_window(*cfg, wd);
    std::vector<float> wf(kWinPad, 0.0f);
    for (int i = 0; i < kFrameLen; ++i) wf[i] = static_cast<float>(wd[i]);
    int32_t rc = upload(ctx, wf, &ctx->tab.window);
    if (rc) return fail(rc);

    // twiddles
    std::vector<float2> t256(256), t512(129);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double a = 2.0 * M_PI * n2 * k1 / 256.0;
            t256[n2 * 16 + k1] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a)));
        }
    for (int k = 0; k <= 128; ++k) {
        const double a = 2.0 * M_PI * k / 512.0;
        t512[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a)));
    }
    if ((rc = upload(ctx, t256, &ctx->tab.tw256))) return fail(rc);
    if ((rc = upload(ctx, t512, &ctx->tab.tw512))) return fail(rc);

    // mel filters: per-lane CSR, zero-padded to the widest filter of each group of 16 rounded up to 4 terms
    std::vector<double> banks;
    build_mel_banks(*cfg, banks);
    const int nb = kNc;
    std::vector<int> start(kMelGroups * 16, 1), width(kMelGroups * 16, 0);
    for (int m = 0; m < cfg->n_mels; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < nb; ++k)
            if (banks[static_cast<size_t>(m) * nb + k] != 0.0) {
                if (lo < 0) lo = k;
                hi = k;
            }
        if (lo == 0) lo = 1;  // bin 0 is never produced by the kernel (its Kaldi weight is 0 for low_freq > 0)
        if (lo >= 1 && hi >= lo) {
            start[(m / 16) * 16 + (m % 16)] = lo;
            width[(m / 16) * 16 + (m % 16)] = hi - lo + 1;
        }
    }
    if (cfg->low_freq <= 0.0f) {
        return fail(set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_create: low_freq must be > 0 (the DC bin is not computed)"));
    }
    // Where a filter's window starts is free as long as its non-zero terms stay inside: a group's windows are as wide
    // as its widest filter rounded up to 4 terms, and a term in front of a filter's first bin has weight zero.  The
    // kernel's gather reads one term of 32 lanes (the 16 filters of a group in two frames) per LDS pass over 32 banks,
    // with the power row of every other frame 16 floats further on: the pass is free of bank conflicts exactly when
    // the different start bins of a group differ modulo 16.  The starts are moved down (by at most the slack of the
    // window, never below bin 1) to make them so; when a group's own width leaves too little room, one more chunk
    // of 4 terms does (20 conflict-free passes beat 16 two-way ones).
    std::vector<int> shift(kMelGroups * 16, 0);
    auto spread = [&](int g, int gw) -> bool {
        int lo[16], hi[16], order[16];
        for (int l = 0; l < 16; ++l) {
            const int st = start[g * 16 + l], w = width[g * 16 + l];
            if (w == 0) { lo[l] = 1; hi[l] = 16; }  // an unused lane reads zero weights: any window will do
            else { hi[l] = st; lo[l] = std::max(1, st - (gw - w)); }
            order[l] = l;
        }
        std::sort(order, order + 16, [&](int a, int b) { return hi[a] - lo[a] < hi[b] - lo[b]; });
        int owner[16], pick[16];
        std::fill(owner, owner + 16, -1);
        long budget = 2000000;  // (search steps; the layouts in use need a few hundred)
        auto place = [&](auto&& self, int i) -> bool {
            if (i == 16) return true;
            if (--budget < 0) return false;
            const int l = order[i];
            for (int st = hi[l]; st >= lo[l]; --st) {
                const int r = st & 15;
                if (owner[r] != -1 && owner[r] != st) continue;
                const int before = owner[r];
                owner[r] = st;
                pick[l] = st;
                if (self(self, i + 1)) return true;
                owner[r] = before;
            }
            return false;
        };
        if (!place(place, 0)) return false;
        for (int l = 0; l < 16; ++l) {
            shift[g * 16 + l] = width[g * 16 + l] ? start[g * 16 + l] - pick[l] : 0;
            if (!width[g * 16 + l]) start[g * 16 + l] = pick[l];
        }
        return true;
    };
    int total = 0;
    for (int g = 0; g < kMelGroups; ++g) {
        int gw = 0;
        for (int l = 0; l < 16; ++l) gw = std::max(gw, width[g * 16 + l]);
        gw = (gw + 3) / 4 * 4;
        int room = kMelMaxTerms - total - gw;  // terms the later groups need at least are not known yet: checked below
        if (!spread(g, gw) && room >= 4 && spread(g, gw + 4)) gw += 4;
        ctx->tab.mel_gw[g] = gw;
        ctx->tab.mel_go[g] = total;
        total += gw;
    }
    if (total > kMelMaxTerms) {
        return fail(set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_create: mel filters too wide for the kernel's table"));
    }
    std::vector<float> mw(kMelMaxTerms * 16, 0.0f);
    std::vector<int32_t> msb(kMelGroups * 16, 1);
    for (int g = 0; g < kMelGroups; ++g)
        for (int l = 0; l < 16; ++l) {
            const int m = g * 16 + l;
            msb[g * 16 + l] = start[g * 16 + l] - shift[g * 16 + l];
            if (m >= cfg->n_mels) continue;
            for (int j = 0; j < width[g * 16 + l]; ++j) {
                const int term = ctx->tab.mel_go[g] + shift[g * 16 + l] + j;  // chunk term / 4, lane l, element term % 4
                mw[((term / 4) * 16 + l) * 4 + term % 4] =
                    static_cast<float>(0.25 * banks[static_cast<size_t>(m) * nb + start[g * 16 + l] + j]);
            }
        }
    if ((rc = upload(ctx, mw, &ctx->tab.mel_w))) return fail(rc);
    if ((rc = upload(ctx, msb, &ctx->tab.mel_sb))) return fail(rc);

    // identity CMVN until asrfe_load_cmvn is called
    const int dim = cfg->lfr_m * cfg->n_mels;
    std::vector<float> z(dim, 0.0f), o(dim, 1.0f);
    if ((rc = upload(ctx, z, &ctx->tab.cmvn_shift))) return fail(rc);
    if ((rc = upload(ctx, o, &ctx->tab.cmvn_scale))) return fail(rc);

    // inverse timescales of the sinusoidal position code over enc_input_size dims
    const int half = cfg->enc_input_size / 2;
    std::vector<float> inv(half);
    const double inc = std::log(10000.0) / (half - 1);
    for (int i = 0; i < half; ++i) inv[i] = static_cast<float>(std::exp(-inc * i));
    if ((rc = upload(ctx, inv, &ctx->pe_inv))) return fail(rc);

    // per-device kernel attributes (dynamic LDS above 64 KiB), set here so that no compute call ever has to
    if ((rc = prepare_gemm(ctx)) || (rc = prepare_post_attn(ctx)) || (rc = prepare_post_attn16(ctx)) || (rc = prepare_static_scratch(ctx)))
        return fail(rc);

    *out = ctx;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_profile_begin(asrfe_ctx* ctx, int32_t max_launches) {
    if (!ctx || max_launches < 1 || max_launches > (1 << 20)) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_profile_begin: bad argument");
    while (ctx->prof_events.size() < static_cast<size_t>(max_launches) * 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->prof_events.push_back(e);
    }
    ctx->prof_kind.assign(ctx->prof_events.size() / 2, 0);
    ctx->prof_used = 0;
    ctx->prof_on = true;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_profile_end(asrfe_ctx* ctx, asrfe_profile* out) {
    if (!ctx || !out) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_profile_end: null argument");
    ctx->prof_on = false;
    std::memset(out, 0, sizeof(*out));
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->prof_events[2 * i + 1]));
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
        const int k = ctx->prof_kind[i];
        if (k >= 0 && k < ASRFE_PROF_KINDS) {
            out->launches[k] += 1;
            out->total_ms[k] += ms;
        }
    }
    ctx->prof_used = 0;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_destroy(asrfe_ctx* ctx) {
    if (!ctx) return ASRFE_OK;
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    free_decoder(ctx);
    free_vocoder(ctx);
    free_vad(ctx);
    for (void* p : ctx->owned) (void)hipFree(p);
    delete ctx;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_load_cmvn(asrfe_ctx* ctx, const float* shift, const float* scale, int32_t dim) {
    if (!ctx || !shift || !scale) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_cmvn: null argument");
    if (dim != ctx->cfg.lfr_m * ctx->cfg.n_mels)
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_load_cmvn: dim must be lfr_m * n_mels");
    // kernels already enqueued on any stream (non-blocking ones included) read the old table to the end
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(ctx->tab.cmvn_shift, shift, dim * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->tab.cmvn_scale, scale, dim * sizeof(float), hipMemcpyHostToDevice));
    ctx->cmvn_loaded = true;
    return ASRFE_OK;
}

// ---- encoder weights ---------------------------------------------------------------------------
// Wo [512][512], W1 [ff][512] and W2 [512][ff] as the piece stream of the fused post-attention kernel (16 KiB
// LDS images in consumption order).  Four chunks of Wo come first, one per quarter of the output columns (the kernel
// finishes a quarter before it starts the next, so the residual tiles of the later quarters can arrive meanwhile):
// piece r of chunk c is Wo[128 c .. +128][64 r .. +64] as 128 rows of 128 bytes, 16-byte slot cs of row r holding
// source chunk cs ^ ((r >> 1) & 7).  Then, for every chunk j of 64 hidden units, four images of
// W1[64 j .. +64][128 ks .. +128] (64 rows of 256 bytes, slot cs of row r holding source chunk cs ^ (r & 15))
// followed by four images of W2[128 nq .. +128][64 j .. +64] (128-byte rows as for Wo).  Inside every group of 16
// columns of W1 and W2 the k order is p -> swap of bits 2 and 3 of p, the order in which accumulator registers
// feed the next MFMA; Wo keeps the natural order (its operand is loaded from memory).
// With wqkv (the NEXT layer's folded [1536][512] projection) twelve more chunks follow, three passes of 512 output
// rows in images like Wo's but k-major (chunk c of a pass holds k-slices 2 c and 2 c + 1 of all four column quarters:
// piece r is rows 128 (r & 3) .. +128, k-slice 2 c + (r >> 2)) and in the swapped k order (their operand is fed from
// accumulator registers too), and one chunk of zeros that the kernel's last pass prefetches without reading.
static int32_t up_layer_stream(asrfe_ctx* ctx, const HostTensor* wo, const HostTensor* w1, const HostTensor* w2,
                               const HostTensor* wqkv, uint16_t** dev) {
    const int64_t ff = w1->shape[0], d = w1->shape[1];
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    auto cvt = [&](float w) { return f16 ? f16_rne(w) : bf16_rne(w); };
    // k order of an operand that is fed from accumulator registers.  32x32x16 kernel: bits 2 and 3 swapped inside every
    // 16.  16x16x32 kernel: stored position 8 q + i of every block of 32 holds column 16 (i >> 2) + 4 q + (i & 3).
    const bool m16 = ctx->mfma16;
    auto swap23 = [m16](int64_t c) {
        if (m16) return (c & ~int64_t{31}) | (((c >> 2) & 1) << 4) | (((c >> 3) & 3) << 2) | (c & 3);
        return (c & ~int64_t{12}) | ((c & 4) << 1) | ((c & 8) >> 1);
    };
    const int64_t nch = ff / 64;
    std::vector<uint16_t> v(static_cast<size_t>((4 + nch + (wqkv ? 13 : 0)) * 8 * 8192), 0);
    for (int64_t c = 0; c < 4; ++c)
        for (int64_t r8 = 0; r8 < 8; ++r8) {
            uint16_t* img = v.data() + (8 * c + r8) * 8192;
            const int64_t ks = r8, nq = c;
            for (int64_t r = 0; r < 128; ++r)
                for (int64_t cs = 0; cs < 8; ++cs)
                    for (int64_t e = 0; e < 8; ++e)
                        img[(r * 8 + cs) * 8 + e] = cvt(wo->data[(128 * nq + r) * d + 64 * ks + 8 * (cs ^ ((r >> 1) & 7)) + e]);
        }
    for (int64_t j = 0; j < nch; ++j) {
        for (int64_t ks = 0; ks < 4; ++ks) {
            uint16_t* img = v.data() + (8 * (4 + j) + ks) * 8192;
            for (int64_t r = 0; r < 64; ++r)
                for (int64_t cs = 0; cs < 16; ++cs)
                    for (int64_t e = 0; e < 8; ++e) {
                        const int64_t kp = 128 * ks + 8 * (cs ^ (r & 15)) + e;
                        img[(r * 16 + cs) * 8 + e] = cvt(w1->data[(64 * j + r) * d + swap23(kp)]);
                    }
        }
        for (int64_t nq = 0; nq < 4; ++nq) {
            uint16_t* img = v.data() + (8 * (4 + j) + 4 + nq) * 8192;
            for (int64_t r = 0; r < 128; ++r)
                for (int64_t cs = 0; cs < 8; ++cs)
                    for (int64_t e = 0; e < 8; ++e) {
                        const int64_t cp = 64 * j + 8 * (cs ^ ((r >> 1) & 7)) + e;
                        img[(r * 8 + cs) * 8 + e] = cvt(w2->data[(128 * nq + r) * ff + swap23(cp)]);
                    }
        }
    }
    if (wqkv)
        for (int64_t c = 0; c < 12; ++c)
            for (int64_t r8 = 0; r8 < 8; ++r8) {
                uint16_t* img = v.data() + (8 * (4 + nch + c) + r8) * 8192;
                const int64_t ks = 2 * (c & 3) + (r8 >> 2), n0 = 512 * (c >> 2) + 128 * (r8 & 3);
                for (int64_t r = 0; r < 128; ++r)
                    for (int64_t cs = 0; cs < 8; ++cs)
                        for (int64_t e = 0; e < 8; ++e) {
                            const int64_t kp = 64 * ks + 8 * (cs ^ ((r >> 1) & 7)) + e;
                            img[(r * 8 + cs) * 8 + e] = cvt(wqkv->data[(n0 + r) * d + swap23(kp)]);
                        }
            }
    return upload(ctx, v, dev);
}

// ---- fp16 range proof ---------------------------------------------------------------------------
// Every activation the kernels round to 16 bits is either bounded by construction (a normalised row has Euclidean
// norm sqrt(K), so its entries stay below 23; probabilities stay below 1) or is a linear image of a normalised row:
// |w . n + c| <= ||w||_2 sqrt(K) + |c| (Cauchy-Schwarz).  With fp16 operands the load call evaluates these bounds for
// q, k, v, the attention output (a convex combination of v rows), the FSMN memory (v plus 11 taps of v) and the
// feed-forward hidden layer, and refuses weights whose bound passes the largest fp16 number: such a model needs
// ASRFE_DT_BF16.  (The kernels additionally convert with saturation, so even then no infinity would arise.)
//
// Two of these bounds can be lowered without changing the function the layer computes, and the load call does so
// before it judges (only for rows whose bound passes the fp16 range, so weights that pass as they are keep their bits):
//   * a score is sum_j q_j k_j: multiplying dimension j of q by 2^-e and the same dimension of k by 2^e leaves every
//     score as it was.  rebalance_qk moves the excess of one side to the other, dimension by dimension;
//   * ReLU is positively homogeneous: W2 relu(W1 n + b1) = (W2 diag(s)) relu(diag(1 / s) (W1 n + b1)) for s > 0.
//     rescale_hidden divides a hidden unit's row of W1 and its bias by 2^e and multiplies its column of W2 by 2^e.
// Powers of two, so nothing is rounded.  v, the attention output and the FSMN memory have no such partner (the FSMN
// memory reaches the residual stream directly): their refusal stands.
static void rebalance_qk(std::vector<float>& w, std::vector<float>& c, int64_t d, int64_t din, int heads) {
    (void)heads;  // (a dimension of one head pairs q row r with k row d + r whatever the head count)
    std::vector<double> bq(static_cast<size_t>(d)), bk(static_cast<size_t>(d));
    linear_bound(w, c, 3 * d, din, 0, d, &bq);
    linear_bound(w, c, 3 * d, din, d, 2 * d, &bk);
    for (int64_t r = 0; r < d; ++r) {
        if (bq[r] <= kF16Max && bk[r] <= kF16Max) continue;
        if (!(bq[r] > 0.0) || !(bk[r] > 0.0) || !std::isfinite(bq[r]) || !std::isfinite(bk[r])) continue;
        // equal bounds on both sides: q row / 2^e, k row * 2^e with 2^e = sqrt(bq / bk) rounded to a power of two
        const int e = static_cast<int>(std::lround(0.5 * std::log2(bq[r] / bk[r])));
        const float down = std::ldexp(1.0f, -e), up = std::ldexp(1.0f, e);
        for (int64_t i = 0; i < din; ++i) {
            w[r * din + i] *= down;
            w[(d + r) * din + i] *= up;
        }
        c[r] *= down;
        c[d + r] *= up;
    }
}

// -> false when a column of W2 would leave the fp16 range itself after the rescale (worst: the offending bound)
static bool rescale_hidden(std::vector<float>& w1, std::vector<float>& b1, const HostTensor* w2, std::vector<float>& w2s,
                           int64_t ff, int64_t d, double* worst) {
    std::vector<double> bound(static_cast<size_t>(ff));
    linear_bound(w1, b1, ff, d, 0, ff, &bound);
    bool any = false;
    for (int64_t r = 0; r < ff; ++r) {
        if (bound[r] <= kF16Max || !std::isfinite(bound[r])) continue;
        if (!any) w2s.assign(w2->data, w2->data + w2->numel());
        any = true;
        int e = static_cast<int>(std::ceil(std::log2(bound[r] / 16384.0)));
        // A unit this large usually comes with a tiny column of W2 (their product is what the model learnt): lift the
        // column towards 2^-8, out of fp16's subnormals, while the unit's bound stays above 256.
        double colmax = 0.0;
        for (int64_t o = 0; o < d; ++o) colmax = std::fmax(colmax, std::fabs(static_cast<double>(w2s[o * ff + r])));
        colmax = std::ldexp(colmax, e);
        if (colmax > 0.0 && colmax < 0x1p-8) {
            const int lift = static_cast<int>(std::floor(std::log2(0x1p-8 / colmax)));
            const int room = static_cast<int>(std::floor(std::log2(std::ldexp(bound[r], -e) / 256.0)));
            e += std::max(0, std::min(lift, room));
        }
        const float down = std::ldexp(1.0f, -e), up = std::ldexp(1.0f, e);
        for (int64_t i = 0; i < d; ++i) w1[r * d + i] *= down;
        b1[r] *= down;
        for (int64_t o = 0; o < d; ++o) {
            float& v = w2s[o * ff + r];
            v *= up;
            if (std::fabs(static_cast<double>(v)) > kF16Max) {
                *worst = std::fabs(static_cast<double>(v));
                return false;
            }
        }
    }
    return true;
}

static int32_t range_refusal(asrfe_ctx* ctx, int layer, const char* what, double bound) {
    char buf[256];
    std::snprintf(buf, sizeof(buf),
                  "asrfe_load_encoder_weights: layer %d %s can reach %.4g, beyond the fp16 range (65504): load this model "
                  "with enc_compute_dtype = ASRFE_DT_BF16",
                  layer, what, bound);
    return set_err(ctx, ASRFE_ERR_RANGE, buf);
}

extern "C" int32_t asrfe_load_encoder_weights(asrfe_ctx* ctx, const asrfe_tensor_desc* descs, int32_t n) {
    if (!ctx || !descs || n <= 0) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_encoder_weights: null argument");
    if (ctx->enc_loaded) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_load_encoder_weights: weights already loaded");
    TensorMap m;
    if (int32_t rc0 = tensor_map(ctx, descs, n, "encoder.", "asrfe_load_encoder_weights", m)) return rc0;
    const asrfe_config& c = ctx->cfg;
    const int64_t d = c.enc_output_size, ff = c.enc_linear_units, ks = c.enc_kernel_size;
    ctx->layers.assign(c.enc_num_blocks, LayerWeights{});
    int32_t rc;
    // norm1 folded into linear_q_k_v of layer li
    auto fold_qkv = [&](int li, std::vector<float>& wfold, std::vector<float>& cfold, const HostTensor** w_shape) -> int32_t {
        const std::string p = li == 0 ? std::string("encoders0.0") : "encoders." + std::to_string(li - 1);
        const int64_t din = li == 0 ? c.enc_input_size : d;
        const HostTensor *tg, *tb, *tw, *tc;
        int32_t r;
        if ((r = need(ctx, m, p + ".norm1.weight", {din}, &tg)) || (r = need(ctx, m, p + ".norm1.bias", {din}, &tb)) ||
            (r = need(ctx, m, p + ".self_attn.linear_q_k_v.weight", {3 * d, din}, &tw)) ||
            (r = need(ctx, m, p + ".self_attn.linear_q_k_v.bias", {3 * d}, &tc)))
            return r;
        fold_layernorm(tw, tc, tg, tb, wfold, cfold);
        *w_shape = tw;
        if (c.enc_compute_dtype == ASRFE_DT_F16) rebalance_qk(wfold, cfold, d, din, c.enc_heads);
        return ASRFE_OK;
    };
    const bool check_range = c.enc_compute_dtype == ASRFE_DT_F16;
    // Row-major copies of a layer's matrices exist only for the launch plans that read them: the layer-0 projection
    // always, the other projections when a layer's stream does not carry the next layer's Wqkv (or ASRFE_SEPARATE_QKV=1),
    // and all of Wqkv / Wo / W1 / W2 for the small-batch plan (enc_small_batch_rows > 0) or under ASRFE_UNFUSED_FFN=1.
    // The throughput plan reads everything else from the piece stream.
    const bool row_major = ctx->unfused_ffn || ctx->small_rows > 0;
    for (int li = 0; li < c.enc_num_blocks; ++li) {
        const std::string p = li == 0 ? std::string("encoders0.0") : "encoders." + std::to_string(li - 1);
        const int kpad = li == 0 ? ctx->kin_pad : static_cast<int>(d);
        LayerWeights& L = ctx->layers[li];
        const HostTensor* t;
        const HostTensor *tg, *tb, *tw, *tc;
        std::vector<float> wfold, cfold;
        if ((rc = fold_qkv(li, wfold, cfold, &tw))) return rc;
        std::vector<double> vbound(static_cast<size_t>(d), 0.0);
        if (check_range) {
            const int64_t din = tw->shape[1];
            const double bq = linear_bound(wfold, cfold, 3 * d, din, 0, 2 * d);
            if (bq > kF16Max) return range_refusal(ctx, li, "q / k", bq);
            const double bv = linear_bound(wfold, cfold, 3 * d, din, 2 * d, 3 * d, &vbound);
            if (bv > kF16Max) return range_refusal(ctx, li, "v (and the attention output)", bv);
        }
        {
            const bool prev_stream_has_it = li > 0 && ctx->layers[li - 1].stream_has_qkv && !ctx->separate_qkv && !row_major;
            const HostTensor view{wfold.data(), tw->shape};
            if (!prev_stream_has_it && (rc = up_bf16(ctx, &view, kpad, &L.w_qkv))) return rc;
            if ((rc = upload(ctx, cfold, &L.b_qkv))) return rc;
        }
        if ((rc = need(ctx, m, p + ".self_attn.fsmn_block.weight", {d, 1, ks}, &t))) return rc;
        {
            std::vector<float> w(static_cast<size_t>(ks) * d);
            for (int64_t ch = 0; ch < d; ++ch) {
                double gain = 1.0;
                for (int64_t j = 0; j < ks; ++j) {
                    w[j * d + ch] = t->data[ch * ks + j];
                    gain += std::fabs(static_cast<double>(t->data[ch * ks + j]));
                }
                if (check_range && vbound[ch] * gain > kF16Max) return range_refusal(ctx, li, "FSMN memory", vbound[ch] * gain);
            }
            if ((rc = upload(ctx, w, &L.w_fsmn))) return rc;
        }
        if ((rc = need(ctx, m, p + ".self_attn.linear_out.weight", {d, d}, &t))) return rc;
        if (row_major && (rc = up_bf16(ctx, t, static_cast<int>(d), &L.w_out))) return rc;
        const HostTensor* t_wo = t;
        if ((rc = need(ctx, m, p + ".self_attn.linear_out.bias", {d}, &t)) || (rc = up_f32(ctx, t, &L.b_out))) return rc;
        if ((rc = need(ctx, m, p + ".norm2.weight", {d}, &tg)) || (rc = need(ctx, m, p + ".norm2.bias", {d}, &tb)) ||
            (rc = need(ctx, m, p + ".feed_forward.w_1.weight", {ff, d}, &tw)) ||
            (rc = need(ctx, m, p + ".feed_forward.w_1.bias", {ff}, &tc)))
            return rc;
        fold_layernorm(tw, tc, tg, tb, wfold, cfold);
        const HostTensor w1view{wfold.data(), tw->shape};
        if ((rc = need(ctx, m, p + ".feed_forward.w_2.weight", {d, ff}, &t))) return rc;
        std::vector<float> w2s;  // W2 with rescaled columns, when a hidden unit had to be brought into the fp16 range
        HostTensor w2view{t->data, t->shape};
        if (check_range) {
            double worst = 0.0;
            if (!rescale_hidden(wfold, cfold, t, w2s, ff, d, &worst))
                return range_refusal(ctx, li, "feed-forward output weight (after moving the hidden layer's scale into it)", worst);
            if (!w2s.empty()) w2view.data = w2s.data();
            const double bh = linear_bound(wfold, cfold, ff, d, 0, ff);
            if (bh > kF16Max) return range_refusal(ctx, li, "feed-forward hidden layer", bh);
        }
        t = &w2view;
        if (row_major && (rc = up_bf16(ctx, &w1view, static_cast<int>(d), &L.w_ffn1))) return rc;
        if ((rc = upload(ctx, cfold, &L.b_ffn1))) return rc;
        if (row_major && (rc = up_bf16(ctx, t, static_cast<int>(ff), &L.w_ffn2))) return rc;
        if (d != 512 || ff % 64 != 0 || ff > 7168)
            return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_load_encoder_weights: the fused post-attention kernel needs d = 512 and linear_units a multiple of 64 up to 7168");
        {
            // the next layer's projection closes this layer's stream when the kernel has the LDS for it
            std::vector<float> qfold, qcfold;
            const HostTensor* qshape = nullptr;
            L.stream_has_qkv = li + 1 < c.enc_num_blocks && ff <= kPostAttnQkvMaxFf;
            if (L.stream_has_qkv && (rc = fold_qkv(li + 1, qfold, qcfold, &qshape))) return rc;
            const HostTensor qview{qfold.data(), {3 * d, d}};
            if ((rc = up_layer_stream(ctx, t_wo, &w1view, t, L.stream_has_qkv ? &qview : nullptr, &L.w_stream))) return rc;
        }
        if ((rc = need(ctx, m, p + ".feed_forward.w_2.bias", {d}, &t)) || (rc = up_f32(ctx, t, &L.b_ffn2))) return rc;
    }
    const HostTensor* t;
    if ((rc = need(ctx, m, "after_norm.weight", {d}, &t)) || (rc = up_f32(ctx, t, &ctx->after_g))) return rc;
    if ((rc = need(ctx, m, "after_norm.bias", {d}, &t)) || (rc = up_f32(ctx, t, &ctx->after_b))) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->enc_loaded = true;
    return ASRFE_OK;
}

// ---- frontend entry points ---------------------------------------------------------------------
extern "C" int32_t asrfe_num_frames(const asrfe_ctx* ctx, int64_t n) {
    const int fl = ctx ? ctx->cfg.frame_length : kFrameLen, fs = ctx ? ctx->cfg.frame_shift : kFrameShift;
    return n < fl ? 0 : static_cast<int32_t>(1 + (n - fl) / fs);
}
extern "C" int32_t asrfe_num_lfr_frames(const asrfe_ctx* ctx, int32_t t) {
    const int n = ctx ? ctx->cfg.lfr_n : 6;
    return t <= 0 ? 0 : (t + n - 1) / n;
}

static int32_t fill_common(asrfe_ctx* ctx, FbankParams& p, const void* pcm, int32_t pcm_dtype, int64_t stride,
                           const int32_t* lens, int32_t batch, const char* who) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (batch < 0 || batch > 65535) return set_err(ctx, ASRFE_ERR_SHAPE, std::string(who) + ": batch must be in 0..65535");
    if (batch == 0) return ASRFE_OK;
    if (!pcm || !lens) return set_err(ctx, ASRFE_ERR_BAD_ARG, std::string(who) + ": null pcm / lens");
    if (pcm_dtype != ASRFE_PCM_I16 && pcm_dtype != ASRFE_PCM_F32 && pcm_dtype != ASRFE_PCM_F32_WIRE) return set_err(ctx, ASRFE_ERR_BAD_ARG, std::string(who) + ": bad pcm_dtype");
    if (stride < 0) return set_err(ctx, ASRFE_ERR_SHAPE, std::string(who) + ": negative pcm_stride");
    const int64_t es = pcm_dtype == ASRFE_PCM_I16 ? 2 : 4;
    p.pcm = pcm;
    p.pcm_stride = stride;
    p.pcm_lens = lens;
    p.batch = batch;
    p.n_mels = ctx->cfg.n_mels;
    p.lfr_m = ctx->cfg.lfr_m;
    p.lfr_n = ctx->cfg.lfr_n;
    p.aligned16 = (reinterpret_cast<uintptr_t>(pcm) % 16 == 0 && (stride * es) % 16 == 0) ? 1 : 0;
    p.tab = ctx->tab;
    return ASRFE_OK;
}

extern "C" int32_t asrfe_fbank(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride,
                               const int32_t* pcm_lens, int32_t batch, int32_t t_max, float* logmel,
                               int32_t* n_frames_out, void* stream) {
    FbankParams p{};
    int32_t rc = fill_common(ctx, p, pcm, pcm_dtype, pcm_stride, pcm_lens, batch, "asrfe_fbank");
    if (rc || batch == 0) return rc;
    if (t_max < 1 || t_max > (1 << 26)) return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_fbank: t_max must be >= 1");
    if (!logmel) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_fbank: null logmel");
    p.t_max = t_max;
    p.logmel = logmel;
    p.n_frames = n_frames_out;
    ProfScope ps(ctx, PK_FBANK, static_cast<hipStream_t>(stream));
    return launch_fbank(ctx, p, pcm_dtype, false, ASRFE_DT_F32, static_cast<hipStream_t>(stream));
}

extern "C" int32_t asrfe_lfr_cmvn(asrfe_ctx* ctx, const float* logmel, const int32_t* n_frames, int32_t batch,
                                  int32_t t_max, int32_t tp_max, void* feats, int32_t out_dtype, int32_t* lens_out,
                                  void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (batch < 0 || batch > 65535 || t_max < 1 || tp_max < 1 || tp_max > 65535)
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_lfr_cmvn: batch in 0..65535, t_max >= 1, tp_max in 1..65535");
    if (batch == 0) return ASRFE_OK;
    if (!logmel || !n_frames || !feats) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_lfr_cmvn: null argument");
    if (out_dtype != ASRFE_DT_F32 && out_dtype != ASRFE_DT_BF16) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_lfr_cmvn: bad out_dtype");
    ProfScope ps(ctx, PK_LFR_CMVN, static_cast<hipStream_t>(stream));
    return launch_lfr_cmvn(ctx, logmel, n_frames, batch, t_max, tp_max, feats, out_dtype, lens_out,
                           static_cast<hipStream_t>(stream));
}

extern "C" int32_t asrfe_frontend(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride,
                                  const int32_t* pcm_lens, int32_t batch, int32_t tp_max, void* feats,
                                  int32_t out_dtype, int32_t* lens_out, void* stream) {
    FbankParams p{};
    int32_t rc = fill_common(ctx, p, pcm, pcm_dtype, pcm_stride, pcm_lens, batch, "asrfe_frontend");
    if (rc || batch == 0) return rc;
    if (tp_max < 1 || tp_max > (1 << 24)) return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_frontend: tp_max must be >= 1");
    if (!feats) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_frontend: null feats");
    if (out_dtype != ASRFE_DT_F32 && out_dtype != ASRFE_DT_BF16) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_frontend: bad out_dtype");
    p.t_max = tp_max * ctx->cfg.lfr_n;
    p.tp_max = tp_max;
    p.feats = feats;
    p.lfr_lens = lens_out;
    ProfScope ps(ctx, PK_FRONTEND, static_cast<hipStream_t>(stream));
    return launch_fbank(ctx, p, pcm_dtype, true, out_dtype, static_cast<hipStream_t>(stream));
}

extern "C" int32_t asrfe_debug_gemm(asrfe_ctx* ctx, const void* a, int32_t lda, const void* w, int32_t ldw,
                                    const float* bias, int32_t m, int32_t n, int32_t k, void* out, int32_t ldo,
                                    const void* add16, const float* add32, int32_t epi, int32_t flags, void* stream) {
    if (!ctx || !a || !w || !bias || !out) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_debug_gemm: null argument");
    GemmParams g{};
    g.A = static_cast<const uint16_t*>(a); g.lda = lda;
    g.W = static_cast<const uint16_t*>(w); g.ldw = ldw;
    g.bias = bias; g.M = m; g.N = n; g.K = k; g.out = out; g.ldo = ldo;
    g.add16 = static_cast<const uint16_t*>(add16); g.ld16 = ldo;
    g.add32 = add32; g.ld32 = ldo;
    g.flags = flags;
    return launch_gemm(ctx, g, epi, static_cast<hipStream_t>(stream));
}

extern "C" int32_t asrfe_debug_post_attn(asrfe_ctx* ctx, int32_t layer, const void* att, const void* fsmn, const float* x,
                                         float* x_out, void* y, void* qkv, int32_t m, int32_t flags, void* stream) {
    if (!ctx || !att || !fsmn || !x_out) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_debug_post_attn: null argument");
    if (!ctx->enc_loaded) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_debug_post_attn: encoder weights not loaded");
    if (layer < 0 || layer >= ctx->cfg.enc_num_blocks) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_debug_post_attn: no such layer");
    const LayerWeights& L = ctx->layers[layer];
    FfnParams f{};
    f.att = static_cast<const uint16_t*>(att); f.fsmn = static_cast<const uint16_t*>(fsmn);
    f.x = x; f.x_out = x_out;
    f.wstream = L.w_stream; f.bo = L.b_out; f.b1 = L.b_ffn1; f.b2 = L.b_ffn2;
    f.y = y; f.ldy = 512;
    if (qkv && (flags & 256)) {  // timing build: y carries the stamp buffer
        f.trace = static_cast<long long*>(y);
        f.y = nullptr;
    }
    if (qkv) {
        if (!L.stream_has_qkv) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_debug_post_attn: this layer's stream has no QKV weights");
        f.qkv = qkv; f.ldq = 1536; f.bqkv = ctx->layers[layer + 1].b_qkv;
    }
    f.eps = ctx->cfg.enc_ln_eps; f.M = m; f.ff = ctx->cfg.enc_linear_units; f.flags = flags & ~512;
    f.tiled = (flags & 512) ? 1 : 0;
    if (ctx->mfma16) return launch_post_attn16(ctx, f, static_cast<hipStream_t>(stream));  // (refuses what is not tiled)
    return launch_post_attn(ctx, f, static_cast<hipStream_t>(stream));
}

extern "C" int32_t asrfe_debug_attention(asrfe_ctx* ctx, int32_t layer, const void* qkv, const int32_t* lens, int32_t batch,
                                         int32_t tp, void* att, void* fsmn, int32_t flags, void* stream) {
    if (!ctx || !qkv || !lens || !att || !fsmn) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_debug_attention: null argument");
    if (!ctx->enc_loaded) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_debug_attention: encoder weights not loaded");
    if (layer < 0 || layer >= ctx->cfg.enc_num_blocks) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_debug_attention: no such layer");
    AttnParams a = attn_self(static_cast<const uint16_t*>(qkv), 1536, batch, tp, lens, static_cast<uint16_t*>(att), 512);
    a.fsmn = static_cast<uint16_t*>(fsmn); a.w_fsmn = ctx->layers[layer].w_fsmn;
    a.scale_log2 = 1.4426950408889634f / std::sqrt(128.0f);
    a.flags = flags;
    return launch_attention(ctx, a, static_cast<hipStream_t>(stream));
}

// ---- encoder entry points ----------------------------------------------------------------------
extern "C" int64_t asrfe_encoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t tp) {
    if (!ctx || batch < 0 || tp < 0) return -1;
    return encoder_workspace_bytes(ctx, batch, tp);
}

extern "C" int32_t asrfe_encoder_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch,
                                         int32_t tp, void* out, int32_t out_dtype, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!ctx->enc_loaded) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_encoder_forward: encoder weights not loaded");
    if (batch < 0 || batch > 16384 || tp < 0 || tp > 8192)
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_encoder_forward: batch in 0..16384, tp in 0..8192");
    if (batch == 0 || tp == 0) return ASRFE_OK;
    if (static_cast<int64_t>(batch) * tp > (1 << 24)) return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_encoder_forward: batch * tp too large");
    if (!feats || !lens || !out || !workspace) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward: null argument");
    if (out_dtype != ASRFE_DT_F32 && out_dtype != ASRFE_DT_BF16 && out_dtype != ASRFE_DT_F16) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward: bad out_dtype");
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return set_err(ctx, ASRFE_ERR_WORKSPACE, "asrfe_encoder_forward: workspace must be 256-byte aligned");
    if (reinterpret_cast<uintptr_t>(feats) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward: feats / out must be 16-byte aligned");
    if (workspace_bytes < encoder_workspace_bytes(ctx, batch, tp))
        return set_err(ctx, ASRFE_ERR_WORKSPACE, "asrfe_encoder_forward: workspace too small");
    return encoder_forward(ctx, feats, lens, batch, tp, out, out_dtype, workspace, workspace_bytes,
                           static_cast<hipStream_t>(stream));
}
